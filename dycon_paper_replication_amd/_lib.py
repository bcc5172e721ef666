"""ctypes binding of libdycon_hip.so (the C ABI declared in include/dycon_hip.h).

There is NO fallback: if the shared library is missing or a symbol cannot be resolved this
module raises, and every op of the package fails loudly.  Build it with
``python -c "import __graft_entry__ as g; g.build()"`` (or ``make -C dycon_paper_replication_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DYCON_LIB", os.path.join(_HERE, "libdycon_hip.so"))   # DYCON_LIB: diagnostic builds only

F32, BF16 = 0, 1
CONV_1X1, CONV_K3, CONV_K2S2 = 0, 1, 2

P, I, L, F, Z, U64 = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_size_t, C.c_uint64

# name -> (restype, argtypes)        -- must list EVERY function declared in include/dycon_hip.h
SIGNATURES = {
    "dycon_version": (I, []),
    "dycon_last_error": (C.c_char_p, []),
    "dycon_bfrag_bytes": (Z, [I, I, I, I]),
    "dycon_pack_bfrag": (I, [P, P, I, I, I, I, I, L, L, L, L, I, P]),
    "dycon_pack_batch": (I, [P, I, I, P]),
    "dycon_pack_tcn": (I, [P, P, I, I, I, I, L, L, L, L, I, P]),
    "dycon_conv_gemm_workspace": (Z, [I, I, I, I, I, I, I, I, I]),
    "dycon_conv_gemm": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, P, Z, P]),
    "dycon_conv_gemm_splits": (I, [I, I, I, I, I, I, I, I, I]),
    "dycon_conv_gemm_ex": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, P, Z, I, P]),
    "dycon_conv_direct": (I, [P, I, P, P, P, I, I, I, I, I, I, I, I, I, P]),
    "dycon_conv_wgrad_workspace": (Z, [I, I, I, I, I, I, I]),
    "dycon_conv_wgrad": (I, [P, I, P, I, P, P, I, I, I, I, I, I, I, L, L, L, P, Z, P]),
    "dycon_colsum_workspace": (Z, [L, I]),
    "dycon_colsum": (I, [P, I, P, L, I, P, Z, P]),
    "dycon_norm_workspace": (Z, [I, L, I]),
    "dycon_norm_stats": (I, [P, I, I, L, I, I, F, P, P, P, F, P, Z, P]),
    "dycon_norm_fwd": (I, [P, P, I, I, L, I, I, F, P, P, P, I, P, P, P, P, F, P, Z, P]),
    "dycon_norm_fwd_is_fused": (I, [I, L, I, I]),
    "dycon_norm_fwd_slab": (I, [P, I, P, P, P, I, I, L, I, I, F, P, P, P, I, P, P, P]),
    "dycon_norm_apply": (I, [P, P, I, I, L, I, I, P, P, P, I, P, P, P]),
    "dycon_norm_bwd": (I, [P, I, P, P, I, I, L, I, I, P, P, P, I, P, P, P, P, Z, P]),
    "dycon_norm_bwd_ex": (I, [P, I, P, P, I, I, L, I, I, P, P, P, I, P, P, P, I, P, Z, P]),
    "dycon_norm_sum_dparams": (I, [P, I, I, P, P, P]),
    "dycon_norm_head_workspace": (Z, [I, L]),
    "dycon_norm_head_fwd": (I, [P, I, I, L, I, P, P, P, I, P, P, P, P, P]),
    "dycon_norm_head_bwd": (I, [P, P, P, I, I, L, I, P, P, P, I, P, P, P, P, P, Z, P]),
    "dycon_norm_head_dparams": (I, [P, I, L, P, P, P]),
    "dycon_conv_stats_chunks": (I, [I, I, I, I, I, I, I, I]),
    "dycon_conv_gemm_stats": (I, [P, P, P, P, I, I, I, I, I, I, I, P, Z, P]),
    "dycon_norm_fwd_parts": (I, [P, P, I, I, L, I, I, F, P, P, P, I, P, P, P, P, F, P, I, P]),
    "dycon_norm_stats_parts": (I, [I, I, L, I, I, F, P, P, P, F, P, I, P]),
    "dycon_norm_bwd_ab_offset": (Z, [I, L, I]),
    "dycon_norm_bwd_stats": (I, [P, P, I, I, L, I, I, P, P, P, I, P, P, P, P, Z, P]),
    "dycon_first_block_bwd_workspace": (Z, [I, I, I, I]),
    "dycon_first_block_bwd": (I, [P, P, P, I, I, I, I, I, I, P, P, P, I, P, P, P, P, P, L, L, L, P, Z, P]),
    "dycon_conv1_wgrad_normbwd_workspace": (Z, [I, I, I, I]),
    "dycon_conv1_wgrad_normbwd": (I, [P, P, P, I, I, I, I, I, I, P, P, P, I, P, P, P, P, L, L, L, P, Z, P]),
    "dycon_norm_acc_doubles": (Z, [I, L, I]),
    "dycon_norm_fwd_acc": (I, [P, P, I, I, L, I, I, F, P, P, P, I, P, P, P, P, F, P, P]),
    "dycon_norm_bwd_acc": (I, [P, P, P, I, I, L, I, I, P, P, P, I, P, P, P, P, P, Z, P]),
    "dycon_maxpool2_fwd": (I, [P, P, P, I, I, I, I, I, I, P]),
    "dycon_maxpool2_bwd": (I, [P, P, P, I, I, I, I, I, I, P]),
    "dycon_trilinear_fwd": (I, [P, P, I, I, I, I, I, I, I, I, I, I, I, I, P]),
    "dycon_trilinear_bwd": (I, [P, P, I, I, I, I, I, I, I, I, I, I, I, I, P]),
    "dycon_copy_channels": (I, [P, I, I, P, I, I, L, I, I, P]),
    "dycon_relu_fwd": (I, [P, P, P, P, I, I, L, I, P]),
    "dycon_relu_bwd": (I, [P, P, P, P, I, I, L, I, P]),
    "dycon_scale_channels": (I, [P, P, P, I, I, L, I, P]),
    "dycon_mul_mask": (I, [P, P, F, P, I, L, P]),
    "dycon_dropout_philox": (I, [P, P, I, L, F, U64, U64, P]),
    "dycon_channel_mask_philox": (I, [P, L, F, U64, U64, P]),
    "dycon_add_noise": (I, [P, P, P, I, L, F, F, U64, U64, P]),
    "dycon_tanh": (I, [P, I, P, L, P]),
    "dycon_cast": (I, [P, I, P, I, L, P]),
    "dycon_add": (I, [P, P, P, I, L, P]),
    "dycon_seg_losses_fwd": (I, [P, P, P, I, I, I, L, F, P, I, P]),
    "dycon_seg_losses_bwd": (I, [P, P, P, I, I, I, L, F, P, P, I, P, I, P]),
    "dycon_seg_losses_finalize": (I, [P, I, I, L, F, P, P]),
    "dycon_step_loss": (I, [P, P, F, F, F, I, I, P, P, P]),
    "dycon_step_losses": (I, [P, P, I, I, L, F, C.c_double, F, I, F, F, F, I, I, P, P, P]),
    "dycon_softmax_mse_fwd": (I, [P, P, P, L, I, L, I, P]),
    "dycon_softmax_mse_bwd": (I, [P, P, P, P, L, I, L, I, P]),
    "dycon_softmax_kl_fwd": (I, [P, P, L, I, L, I, P, P, P]),
    "dycon_softmax_kl_bwd": (I, [P, P, L, I, L, I, I, P, P, P]),
    "dycon_dice_fwd": (I, [P, P, I, I, L, I, L, I, P, F, P, P, P]),
    "dycon_dice_bwd": (I, [P, P, I, I, L, I, L, I, P, F, P, P, P, P]),
    "dycon_l2norm_fwd": (I, [P, P, P, I, L, I, F, P]),
    "dycon_l2norm_bwd": (I, [P, P, P, P, I, L, I, F, P]),
    "dycon_mask_pool": (I, [P, I, P, I, I, I, I, I, I, I, P]),
    "dycon_fecl_workspace": (Z, [I, I, I]),
    "dycon_fecl_fwd": (I, [P, P, P, P, I, I, I, I, F, F, I, F, F, P, P, P, Z, P]),
    "dycon_fecl_finalize": (I, [P, C.c_double, F, I, P, P]),
    "dycon_set_scalars": (I, [P, I, F, F, F, F, F, F, F, F, P]),
    "dycon_fecl_bwd": (I, [P, P, P, P, I, I, I, I, F, F, I, F, F, P, P, P, P, Z, P]),
    "dycon_sumsq": (I, [P, L, P, P]),
    "dycon_sgd_ema": (I, [P, P, P, P, L, L, P, F, F, F, F, F, F, P, P]),
    "dycon_nonfinite_flag": (I, [P, P, P]),
    "dycon_sw_accumulate": (I, [P, I, I, I, I, P, P, P, I, I, I, P]),
    "dycon_sw_finalize": (I, [P, P, L, F, P, P, P]),
    "dycon_binary_overlap": (I, [P, P, I, L, P, P]),
    "dycon_batch_overlap": (I, [P, P, I, I, L, P, P]),
    "dycon_kernel_timing": (I, [I]),
    "dycon_kernel_timing_count": (L, []),
    "dycon_kernel_timing_fetch": (I, [L, L, P, P, P]),
    "dycon_kernel_timing_name": (C.c_char_p, [I]),
}


class View(C.Structure):
    """dycon_view_t: a strided (n, C, V) fp32 view (element strides)."""
    _fields_ = [("p", P), ("sn", L), ("sc", L), ("sv", L)]


class DyconLibraryError(RuntimeError):
    pass


_lib = None


def load():
    """Load (once) and return the ctypes handle with all prototypes set."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DyconLibraryError(
            f"{LIB_PATH} not found: the HIP extension is not built.  Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` from the repo root.  "
            "There is no CPU/PyTorch fallback for the DyCON hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise DyconLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class Recorder:
    """Software command buffer of one training step (trainer.DyconTrainer, `replay`): every C-ABI call is logged as
    [function, argument list, name] and every torch-level stream / event / tensor operation as a thunk, in issue order; tensors whose
    addresses were passed are kept alive, so the same list can be re-issued step after step with only the schedule scalars patched.
    (hipGraph replay of the same step costs more host time than this list: DESIGN.md section 9.)"""

    def __init__(self):
        self.entries, self.keep = [], []

    def replay(self):
        for e in self.entries:
            if type(e) is list:
                rc = e[0](*e[1])
                if rc:
                    raise DyconLibraryError(f"{e[2]} failed ({rc}) in replay: {load().dycon_last_error().decode()}")
            else:
                e()


RECORDER = None

# ---------------------------------------------------------------- HIP runtime: events of the step's fork / join structure
# torch.cuda.Event() is created with hipEventDisableTiming only, so every record carries a SYSTEM-scope release (cache write-back
# and invalidate, hip_runtime_api.h: hipEventDisableSystemFence).  The step's ~50 cross-stream dependencies are device-local
# (producer and consumer kernels on the same GPU; the one host-read value of a step, the NaN flag, keeps torch's event): their
# events are created here without that fence and recorded / waited through ctypes (also cheaper on the host than the torch objects).
# Measured (profiles/r03_ablation_timing.txt): step 5.05 ms with torch's flags, 4.95 ms with hipEventDisableSystemFence, 5.05 ms with
# hipEventReleaseToDevice (the release flags are mutually exclusive).  DYCON_EVENT_FLAGS = default | nofence | device (diagnostic).
HIP_EVENT_DISABLE_TIMING, HIP_EVENT_NO_SYSTEM_FENCE, HIP_EVENT_RELEASE_TO_DEVICE = 0x2, 0x20000000, 0x40000000
_EVENT_FLAGS = {"default": HIP_EVENT_DISABLE_TIMING, "nofence": HIP_EVENT_DISABLE_TIMING | HIP_EVENT_NO_SYSTEM_FENCE,
                "device": HIP_EVENT_DISABLE_TIMING | HIP_EVENT_RELEASE_TO_DEVICE}
EVENT_MODE = os.environ.get("DYCON_EVENT_FLAGS", "nofence")
_hip = None


def hip():
    """libamdhip64 (the runtime torch already loaded) with the four event entry points prototyped"""
    global _hip
    if _hip is None:
        h = C.CDLL("libamdhip64.so")
        h.hipEventCreateWithFlags.restype, h.hipEventCreateWithFlags.argtypes = I, [C.POINTER(P), C.c_uint]
        h.hipEventRecord.restype, h.hipEventRecord.argtypes = I, [P, P]
        h.hipStreamWaitEvent.restype, h.hipStreamWaitEvent.argtypes = I, [P, P, C.c_uint]
        h.hipEventDestroy.restype, h.hipEventDestroy.argtypes = I, [P]
        _hip = h
    return _hip


class HipEvent:
    """A device-local HIP event (no timing, narrowed fence: see above)."""
    __slots__ = ("h",)

    def __init__(self):
        ev = P()
        rc = hip().hipEventCreateWithFlags(C.byref(ev), _EVENT_FLAGS[EVENT_MODE])
        if rc:
            raise DyconLibraryError(f"hipEventCreateWithFlags failed ({rc})")
        self.h = ev.value

    def __del__(self):
        try:
            if self.h and _hip is not None:
                _hip.hipEventDestroy(self.h)
        except Exception:       # interpreter shutdown
            pass


def hip_call(name, *args, keep=None):
    """An event / stream entry point of the HIP runtime, logged like a C-ABI call while a step is being recorded."""
    fn = getattr(hip(), name)
    rc = fn(*args)
    if rc:
        raise DyconLibraryError(f"{name} failed ({rc})")
    if RECORDER is not None:
        RECORDER.entries.append([fn, list(args), name])
        if keep is not None:
            RECORDER.keep.append(keep)


def call(name, *args):
    """Invoke an int-returning entry point; raise with dycon_last_error() on failure."""
    lib = load()
    fn = getattr(lib, name)
    rc = fn(*args)
    if rc != 0:
        raise DyconLibraryError(f"{name} failed ({rc}): {lib.dycon_last_error().decode()}")
    if RECORDER is not None:
        RECORDER.entries.append([fn, list(args), name])


_QUERY_CACHE = {}


def query(name, *args):
    """Pure host-side size / plan queries: memoised (a step repeats the same ~200 shapes)."""
    key = (name,) + args
    try:
        return _QUERY_CACHE[key]
    except KeyError:
        v = _QUERY_CACHE[key] = getattr(load(), name)(*args)
        return v
