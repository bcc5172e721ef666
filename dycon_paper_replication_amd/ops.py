"""Tensor-level wrappers over the C ABI (include/dycon_hip.h).

Every function takes/returns torch CUDA tensors and only enqueues HIP kernels on torch's current
stream (no sync, graph-capturable).  Activations are NDHWC: shape (B, D, H, W, C), contiguous.
PyTorch is used for device memory and streams only -- no torch math runs here.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from ._lib import BF16, CONV_1X1, CONV_K2S2, CONV_K3, F32, call, query

_DT = {torch.float32: F32, torch.bfloat16: BF16}


def dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"unsupported dtype {t.dtype}: the HIP path stores fp32 or bf16") from None


# torch.cuda.current_stream() costs several microseconds of Python per call (device-index and availability lookups); a step makes
# ~500 launches.  Inside `on_stream` contexts (the trainer wraps the whole step, the engine its side-stream sections) the current
# stream is tracked here; outside of them every op still asks torch.
_STREAMS = []          # stack of torch.cuda.Stream objects


def cur_stream():
    return _STREAMS[-1] if _STREAMS else torch.cuda.current_stream()


def _s():
    return _STREAMS[-1].cuda_stream if _STREAMS else torch.cuda.current_stream().cuda_stream


class on_stream:
    """`with on_stream(s)`: make s (a torch.cuda.Stream; None = keep torch's current stream) current for torch AND for the
    launches of this module.  Nestable.  light=True redirects only this module's launches (no torch stream switch, which costs
    ~10 us of host time per enter/exit): for sections that allocate nothing and call no torch op."""

    def __init__(self, stream=None, light=False):
        self.stream = stream
        self.light = light
        self.ctx = None

    def __enter__(self):
        if self.stream is None:
            _STREAMS.append(torch.cuda.current_stream())
        elif self.light:
            _STREAMS.append(self.stream)
        else:
            self.ctx = torch.cuda.stream(self.stream)
            self.ctx.__enter__()
            _STREAMS.append(self.stream)
        return _STREAMS[-1]

    def __exit__(self, *exc):
        _STREAMS.pop()
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not (t.is_cuda and t.is_contiguous()):
        raise ValueError("HIP ops need contiguous CUDA tensors")
    if _lib.RECORDER is not None:
        _lib.RECORDER.keep.append(t)        # the recorded call holds the raw address: keep the storage alive
    return t.data_ptr()


def rec(thunk):
    """Run a torch-level operation of the step (stream wait, event record, tensor op) and, while a step is being recorded, log
    it so that the replay re-issues it at the same position.  The thunk must name its streams explicitly."""
    thunk()
    if _lib.RECORDER is not None:
        _lib.RECORDER.entries.append(thunk)


# ------------------------------------------------------------------ cross-stream dependencies of the step
class Event:
    """A device-local dependency marker (HIP event without timing and with the fence narrowed to the device, _lib.HipEvent):
    record(stream) / wait(stream) enqueue through ctypes and are part of the recorded step."""
    __slots__ = ("ev",)

    def __init__(self):
        self.ev = _lib.HipEvent()

    def record(self, stream):
        _lib.hip_call("hipEventRecord", self.ev.h, stream.cuda_stream, keep=self.ev)

    def wait(self, stream):
        """`stream` waits for the work captured by the last record()"""
        _lib.hip_call("hipStreamWaitEvent", stream.cuda_stream, self.ev.h, 0, keep=self.ev)


def fork(src, dst):
    """`dst` waits for everything enqueued on `src` so far (torch: dst.wait_stream(src)); returns the event"""
    ev = Event()
    ev.record(src)
    ev.wait(dst)
    return ev


# ------------------------------------------------------------------ optional per-kernel timing (bench.py)
class KernelProfiler:
    """Per-KERNEL timing of the step (bench.py's roofline object): the library brackets every launch it makes with two HIP timing
    events on the launch stream (csrc/ktimer.cpp, dycon_kernel_timing); the wrappers of this module note, per entry-point call, which
    records it produced and the ALGORITHMIC bytes / flops of the call (inputs + outputs read / written once, true K / N without
    padding).  Those are attributed to the call's kernels by name (`parts`: the statistics and the apply pass of a normalisation
    each move their own bytes) or, by default, to its longest kernel; the finalize / reduce / finish launches of a call are
    records of their own with no algorithmic bytes -- every figure is a kernel's, as in `rocprofv3 --kernel-trace --stats`."""

    def __init__(self):
        call("dycon_kernel_timing", 1)
        self.regions = []          # (region name, first record, one past the last record, bytes, flops, parts)

    def count(self):
        return _lib.load().dycon_kernel_timing_count()

    def add(self, name, i0, i1, nbytes, flops, parts):
        if i1 > i0:
            self.regions.append((name, i0, i1, nbytes, flops, parts))

    def close(self):
        call("dycon_kernel_timing", 0)

    def summary(self):
        """kernel name -> {launches, ms, bytes, flops, ms_by_stream, regions}; waits for the recorded launches"""
        import ctypes as C
        lib = _lib.load()
        n = lib.dycon_kernel_timing_count()
        ms, st, nm = (C.c_float * n)(), (C.c_ulonglong * n)(), (C.c_int * n)()
        rc = lib.dycon_kernel_timing_fetch(0, n, ms, st, nm)
        if rc:
            raise _lib.DyconLibraryError(f"dycon_kernel_timing_fetch failed ({rc}): {lib.dycon_last_error().decode()}")
        names = {}
        kname = lambda i: names.setdefault(nm[i], lib.dycon_kernel_timing_name(nm[i]).decode())   # noqa: E731
        out = {}

        def row(i):
            r = out.setdefault(kname(i), {"launches": 0, "ms": 0.0, "bytes": 0, "flops": 0, "ms_by_stream": {}, "regions": set()})
            return r
        owned = [None] * n
        for name, i0, i1, nbytes, flops, parts in self.regions:
            for i in range(i0, i1):
                owned[i] = name
            matched = False
            if parts:
                for sub, b, f in parts:
                    hit = [i for i in range(i0, i1) if sub in kname(i)]
                    if hit:
                        matched = True
                        r = row(max(hit, key=lambda i: ms[i]))
                        r["bytes"] += b
                        r["flops"] += f
            if not matched:
                r = row(max(range(i0, i1), key=lambda i: ms[i]))
                r["bytes"] += nbytes
                r["flops"] += flops
        for i in range(n):
            r = row(i)
            r["launches"] += 1
            r["ms"] += ms[i]
            r["ms_by_stream"][st[i]] = r["ms_by_stream"].get(st[i], 0.0) + ms[i]
            r["regions"].add(owned[i] or "-")
        for r in out.values():
            r["regions"] = sorted(r["regions"])
        return out


PROFILER: Optional[KernelProfiler] = None


class _Timed:
    __slots__ = ("name", "nbytes", "flops", "parts", "i0")

    def __init__(self, name, nbytes, flops, parts):
        self.name, self.nbytes, self.flops, self.parts = name, nbytes, flops, parts

    def __enter__(self):
        self.i0 = PROFILER.count()

    def __exit__(self, *a):
        PROFILER.add(self.name, self.i0, PROFILER.count(), self.nbytes, self.flops, self.parts)


class _Untimed:
    __slots__ = ()

    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_UNTIMED = _Untimed()


def _Region(name, nbytes, flops, parts=None):
    """Accounting bracket around one entry point while bench.py profiles (see KernelProfiler); a shared no-op object otherwise.
    parts: [(kernel-name substring, bytes, flops), ...] where the call's bytes belong to more than one of its kernels."""
    return _UNTIMED if PROFILER is None else _Timed(name, nbytes, flops, parts)


def _es(t):
    return t.element_size()


def _ws(nbytes: int, like: torch.Tensor) -> torch.Tensor:
    return torch.empty(max(int(nbytes) // 4, 1), dtype=torch.float32, device=like.device)


# ------------------------------------------------------------------ weight packing
def pack_bfrag(w: torch.Tensor, dtype: torch.dtype, T, Cin, N, N0, s_t, s_c, s_n1, s_n0, flip=False, out=None):
    d = _DT[dtype]
    nbytes = query("dycon_bfrag_bytes", d, T, Cin, N)
    if out is None:
        out = torch.empty(nbytes // (2 if d == BF16 else 4), dtype=dtype, device=w.device)
    call("dycon_pack_bfrag", _p(w), _p(out), d, T, Cin, N, N0, s_t, s_c, s_n1, s_n0, int(flip), _s())
    return out


def pack_tcn(w: torch.Tensor, T, Cin, N, N0, s_t, s_c, s_n1, s_n0, flip=False, out=None):
    if out is None:
        out = torch.empty(T * Cin * N, dtype=torch.float32, device=w.device)
    call("dycon_pack_tcn", _p(w), _p(out), T, Cin, N, N0, s_t, s_c, s_n1, s_n0, int(flip), _s())
    return out


PACK_JOB_DTYPE = [("w", "<u8"), ("out", "<u8"), ("s_t", "<i8"), ("s_c", "<i8"), ("s_n1", "<i8"), ("s_n0", "<i8"), ("total", "<i8"),
                  ("kind", "<i4"), ("T", "<i4"), ("Cin", "<i4"), ("N", "<i4"), ("N0", "<i4"), ("flip", "<i4"), ("NT", "<i4"), ("pad_", "<i4")]


def pack_job(w, out, kind, T, Cin, N, N0, s_t, s_c, s_n1, s_n0, flip):
    """One dycon_pack_job_t as a tuple (kind: 0 bf16 fragments, 1 fp32 fragments, 2 plain fp32 [T][Cin][N])."""
    return (w.data_ptr(), out.data_ptr(), s_t, s_c, s_n1, s_n0, out.numel(), kind, T, Cin, N, N0, int(flip), (N + 15) // 16, 0)


def upload_pack_jobs(jobs, device):
    import numpy as np
    arr = np.array(jobs, dtype=np.dtype(PACK_JOB_DTYPE))
    assert arr.dtype.itemsize == 88
    return torch.from_numpy(arr.view(np.uint8).copy()).to(device)


def pack_batch(jobs_dev, njobs, blocks_per_job=8):
    call("dycon_pack_batch", _p(jobs_dev), njobs, blocks_per_job, _s())


def conv_uses_lds(x, Cin, Cout):
    """True when dycon_conv_gemm routes a k=3 conv of this shape to the LDS-halo kernel (bf16, >= 24^3 voxels)."""
    _, D, H, W, _ = x.shape
    if x.dtype != torch.bfloat16 or D * H * W < 13824:
        return False
    if Cin == 1:
        return Cout in (16, 32, 64)
    return (Cin in (16, 48) or Cin % 32 == 0) and (Cout in (16, 32) or Cout % 64 == 0 or Cout % 48 == 0)


# ------------------------------------------------------------------ conv family
class DeferredConv:
    """A split-K convolution whose finish was left to the norm that follows (conv_gemm(defer_finish=True))."""
    __slots__ = ("ws", "splits", "bias")

    def __init__(self, ws, splits, bias):
        self.ws, self.splits, self.bias = ws, splits, bias


def conv_gemm(x, wfrag, bias, mode, N, Cout, scatter=False, out=None, accumulate=False, defer_finish=False):
    """defer_finish: on split-K shapes return (out, DeferredConv): `out` is allocated but NOT written; norm_fwd_slab completes it."""
    B, D, H, W, Cin = x.shape
    if defer_finish:
        assert out is None and not accumulate and not scatter
        splits = query("dycon_conv_gemm_splits", dt(x), mode, 0, B, D, H, W, Cin, N)
        if splits <= 1:
            return conv_gemm(x, wfrag, bias, mode, N, Cout), None
        shape = (B, D // 2, H // 2, W // 2, N) if mode == CONV_K2S2 else (B, D, H, W, N)
        out = torch.empty(shape, dtype=x.dtype, device=x.device)
        nws = query("dycon_conv_gemm_workspace", dt(x), mode, 0, B, D, H, W, Cin, N)
        ws = _ws(nws, x)
        rn = "conv_k3_tile" if (mode == CONV_K3 and Cin % 32 == 0 and Cin >= 64 and N % 128 == 0) else "conv_gemm_splitk"
        taps = {CONV_K3: 27, CONV_K2S2: 8, CONV_1X1: 1}[mode]
        with _Region(rn, (x.numel() + out.numel()) * _es(x) + taps * Cin * N * _es(x), 2 * (out.numel() // Cout) * taps * Cin * N):
            call("dycon_conv_gemm_ex", _p(x), _p(wfrag), _p(bias), _p(out), dt(x), mode, 0, 0, B, D, H, W, Cin, N, Cout, _p(ws), nws, 1, _s())
        return out, DeferredConv(ws, splits, bias)
    if out is None:
        assert not accumulate
        if scatter:
            shape = (B, 2 * D, 2 * H, 2 * W, Cout)
        elif mode == CONV_K2S2:
            shape = (B, D // 2, H // 2, W // 2, N)
        else:
            shape = (B, D, H, W, N)
        out = torch.empty(shape, dtype=x.dtype, device=x.device)
    nws = query("dycon_conv_gemm_workspace", dt(x), mode, int(scatter), B, D, H, W, Cin, N)
    ws = _ws(nws, x) if nws else None
    if PROFILER is None:      # hot path: no region bookkeeping (a step makes ~90 of these calls)
        call("dycon_conv_gemm", _p(x), _p(wfrag), _p(bias), _p(out), dt(x), mode, int(scatter), int(accumulate),
             B, D, H, W, Cin, N, Cout, _p(ws), nws, _s())
        return out
    taps = {CONV_K3: 27, CONV_K2S2: 8, CONV_1X1: 1}[mode]
    rows = out.numel() // Cout if not scatter else x.numel() // Cin
    lds_path = mode == CONV_K3 and not scatter and conv_uses_lds(x, Cin, Cout)
    if lds_path:      # the persistent kernels of the 16-channel level, or the generic LDS-halo kernel
        rname = "conv_k3_p16" if (Cin, Cout) == (16, 16) else ("conv_k3_c1" if Cin == 1 else "conv_k3_lds")
    elif (mode == CONV_K3 and not scatter and x.dtype == torch.bfloat16 and Cin % 32 == 0 and Cin >= 64 and N % 128 == 0
          and D * H * W < 13824):
        rname = "conv_k3_tile"
    else:
        rname = "conv_gemm_splitk" if nws else "conv_gemm"
    with _Region(rname, (x.numel() + out.numel() * (2 if accumulate else 1)) * _es(x) + taps * Cin * N * _es(x),
                 2 * rows * taps * Cin * N):
        call("dycon_conv_gemm", _p(x), _p(wfrag), _p(bias), _p(out), dt(x), mode, int(scatter), int(accumulate),
             B, D, H, W, Cin, N, Cout, _p(ws), nws, _s())
    return out


def conv_stats_chunks(x, Cin, Cout):
    """rows per sample of the statistics partials conv_gemm_stats writes for this k=3 shape (0: not served)"""
    B, D, H, W, _ = x.shape
    return query("dycon_conv_stats_chunks", dt(x), CONV_K3, B, D, H, W, Cin, Cout)


def conv_gemm_stats(x, wfrag, bias, Cout, chunks):
    """k=3 convolution that also leaves {sum, sum of squares} partials of its stored outputs: returns (y, part[B][chunks][Cout][2])"""
    B, D, H, W, Cin = x.shape
    y = torch.empty((B, D, H, W, Cout), dtype=x.dtype, device=x.device)
    part = torch.empty(B * chunks * Cout * 2, dtype=torch.float32, device=x.device)
    rname = "conv_k3_p16" if (Cin, Cout) == (16, 16) else ("conv_k3_c1" if Cin == 1 else "conv_k3_lds")
    with _Region(rname, (x.numel() + y.numel()) * _es(x) + 27 * Cin * Cout * _es(x), 2 * (y.numel() // Cout) * 27 * Cin * Cout):
        call("dycon_conv_gemm_stats", _p(x), _p(wfrag), _p(bias), _p(y), dt(x), B, D, H, W, Cin, Cout, _p(part), part.numel() * 4, _s())
    return y, part


def conv_direct(x, w_tcn, bias, mode, N, out_dtype, out=None, accumulate=False):
    B, D, H, W, Cin = x.shape
    if out is None:
        assert not accumulate
        shape = (B, D // 2, H // 2, W // 2, N) if mode == CONV_K2S2 else (B, D, H, W, N)
        out = torch.empty(shape, dtype=out_dtype, device=x.device)
    taps = {CONV_K3: 27, CONV_K2S2: 8, CONV_1X1: 1}[mode]
    with _Region("conv_direct", x.numel() * _es(x) + out.numel() * _es(out), 2 * (out.numel() // N) * taps * Cin * N):
        call("dycon_conv_direct", _p(x), dt(x), _p(w_tcn), _p(bias), _p(out), dt(out), mode, int(accumulate),
             B, D, H, W, Cin, N, _s())
    return out


def conv_wgrad_workspace(x, gy, mode):
    """A workspace tensor for conv_wgrad of these shapes (callers that run on a side stream keep one per layer)."""
    B, D, H, W, Cin = x.shape
    return _ws(query("dycon_conv_wgrad_workspace", mode, B, D, H, W, Cin, gy.shape[-1]), x)


def conv_wgrad(x, gy, dw, mode, s_t, s_c, s_n, dbias=None, ws=None):
    """dw[t*s_t + c*s_c + n*s_n] = sum_rows x[src(row,t), c] * gy[row, n]; dw is an fp32 tensor (any shape).
    dbias (optional, fp32 (Cout,)): column sums of gy, fused into the same pass where the kernel allows."""
    B, D, H, W, Cin = x.shape
    Cout = gy.shape[-1]
    if ws is None:
        ws = _ws(query("dycon_conv_wgrad_workspace", mode, B, D, H, W, Cin, Cout), x)
    taps = {CONV_K3: 27, CONV_K2S2: 8, CONV_1X1: 1}[mode]
    k3bf = (x.dtype == torch.bfloat16 and gy.dtype == torch.bfloat16 and mode == CONV_K3 and (Cin % 16 == 0 or Cin == 1)
            and (Cout in (16, 32) or Cout % 64 == 0))
    with _Region("wgrad_k3_bf16" if k3bf else "conv_wgrad", x.numel() * _es(x) + gy.numel() * _es(gy) + taps * Cin * Cout * 4,
                 2 * (gy.numel() // Cout) * taps * Cin * Cout):
        call("dycon_conv_wgrad", _p(x), dt(x), _p(gy), dt(gy), _p(dw), _p(dbias), mode, B, D, H, W, Cin, Cout, s_t, s_c, s_n,
             _p(ws), ws.numel() * 4, _s())
    return dw


def colsum(x2d_like, out, ws=None):
    """out[c] = sum over all leading dims of x[..., c]."""
    C = x2d_like.shape[-1]
    rows = x2d_like.numel() // C
    if ws is None:
        ws = _ws(query("dycon_colsum_workspace", rows, C), x2d_like)
    with _Region("colsum", x2d_like.numel() * _es(x2d_like), x2d_like.numel()):
        call("dycon_colsum", _p(x2d_like), dt(x2d_like), _p(out), rows, C, _p(ws), ws.numel() * 4, _s())
    return out


# ------------------------------------------------------------------ normalisation
def norm_stats(x, Nb, V, C, G, eps=1e-5, running_mean=None, running_var=None, momentum=0.1):
    stats = torch.empty(Nb * G * 2, dtype=torch.float32, device=x.device)
    ws = _ws(query("dycon_norm_workspace", Nb, V, C), x)
    with _Region("norm_stats", x.numel() * _es(x), 3 * x.numel()):
        call("dycon_norm_stats", _p(x), dt(x), Nb, V, C, G, eps, _p(stats), _p(running_mean), _p(running_var), momentum,
             _p(ws), ws.numel() * 4, _s())
    return stats


def norm_fwd(x, Nb, V, C, G, gamma=None, beta=None, relu=True, skip=None, chan_scale=None, eps=1e-5, running_mean=None,
             running_var=None, momentum=0.1, acc=None):
    """statistics + apply (one launch on the small levels).  Returns (y, stats).
    acc: Nb*C*2 ZEROED doubles (a slice of the caller's per-step arena): two launches instead of three on the large levels."""
    y = torch.empty_like(x)
    stats = torch.empty(Nb * G * 2, dtype=torch.float32, device=x.device)
    if acc is not None:
        with _Region("norm_fwd", x.numel() * _es(x) * (4 if skip is not None else 3), 6 * x.numel()):
            call("dycon_norm_fwd_acc", _p(x), _p(y), dt(x), Nb, V, C, G, eps, _p(stats), _p(gamma), _p(beta), int(relu), _p(skip),
                 _p(chan_scale), _p(running_mean), _p(running_var), momentum, _p(acc), _s())
        return y, stats
    ws = _ws(query("dycon_norm_workspace", Nb, V, C), x)
    nb = x.numel() * _es(x)
    with _Region("norm_fwd", nb * (4 if skip is not None else 3), 6 * x.numel(),
                 parts=[("partial", nb, 3 * x.numel()), ("apply", nb * (3 if skip is not None else 2), 3 * x.numel())]):
        call("dycon_norm_fwd", _p(x), _p(y), dt(x), Nb, V, C, G, eps, _p(stats), _p(gamma), _p(beta), int(relu), _p(skip),
             _p(chan_scale), _p(running_mean), _p(running_var), momentum, _p(ws), ws.numel() * 4, _s())
    return y, stats


def norm_fwd_parts(x, part, chunks, Nb, V, C, G, gamma=None, beta=None, relu=True, skip=None, chan_scale=None, eps=1e-5,
                   running_mean=None, running_var=None, momentum=0.1):
    """norm_fwd without its statistics pass: finalize the producer's partials (conv_gemm_stats) + apply.  Returns (y, stats)."""
    y = torch.empty_like(x)
    stats = torch.empty(Nb * G * 2, dtype=torch.float32, device=x.device)
    with _Region("norm_fwd", x.numel() * _es(x) * (3 if skip is not None else 2), 3 * x.numel()):
        call("dycon_norm_fwd_parts", _p(x), _p(y), dt(x), Nb, V, C, G, eps, _p(stats), _p(gamma), _p(beta), int(relu), _p(skip),
             _p(chan_scale), _p(running_mean), _p(running_var), momentum, _p(part), chunks, _s())
    return y, stats


def norm_stats_parts(x, part, chunks, Nb, V, C, G, eps=1e-5, running_mean=None, running_var=None, momentum=0.1):
    """norm_stats without its pass over the tensor: finalize the producing convolution's partials (conv_gemm_stats)"""
    stats = torch.empty(Nb * G * 2, dtype=torch.float32, device=x.device)
    with _Region("norm_stats", 0, 0):
        call("dycon_norm_stats_parts", dt(x), Nb, V, C, G, eps, _p(stats), _p(running_mean), _p(running_var), momentum, _p(part), chunks, _s())
    return stats


def norm_fwd_is_fused(x, V, C, G):
    return bool(query("dycon_norm_fwd_is_fused", dt(x), V, C, G))


def norm_fwd_slab(x_out, dc: DeferredConv, Nb, V, C, G, gamma=None, beta=None, relu=True, skip=None, chan_scale=None, eps=1e-5):
    """Complete a deferred split-K convolution (writes x_out) and normalise it in the same launch.  Returns (y, stats)."""
    y = torch.empty_like(x_out)
    stats = torch.empty(Nb * G * 2, dtype=torch.float32, device=x_out.device)
    with _Region("norm_fwd", x_out.numel() * _es(x_out) * (4 if skip is not None else 3), 6 * x_out.numel()):
        call("dycon_norm_fwd_slab", _p(dc.ws), dc.splits, _p(dc.bias), _p(x_out), _p(y), dt(x_out), Nb, V, C, G, eps, _p(stats),
             _p(gamma), _p(beta), int(relu), _p(skip), _p(chan_scale), _s())
    return y, stats


def norm_apply(x, stats, Nb, V, C, G, gamma=None, beta=None, relu=True, skip=None, out=None, chan_scale=None):
    if out is None:
        out = torch.empty_like(x)
    with _Region("norm_apply", x.numel() * _es(x) * (3 if skip is not None else 2), 3 * x.numel()):
        call("dycon_norm_apply", _p(x), _p(out), dt(x), Nb, V, C, G, _p(stats), _p(gamma), _p(beta), int(relu), _p(skip),
             _p(chan_scale), _s())
    return out


def norm_bwd(src, from_y, gy, stats, Nb, V, C, G, gamma=None, beta=None, relu=True, dgamma=None, dbeta=None, out=None,
             chan_scale=None, acc=None, defer_dparams=False):
    if out is None:
        out = torch.empty_like(gy)
    if acc is not None and not from_y:
        fused = norm_fwd_is_fused(gy, V, C, G)
        ws = _ws(query("dycon_norm_workspace", Nb, V, C), gy) if fused else None      # (the one-launch kernels keep their workspace)
        with _Region("norm_bwd", gy.numel() * _es(gy) * 5, 12 * gy.numel()):
            call("dycon_norm_bwd_acc", _p(src), _p(gy), _p(out), dt(gy), Nb, V, C, G, _p(stats), _p(gamma), _p(beta), int(relu),
                 _p(chan_scale), _p(dgamma), _p(dbeta), _p(acc), _p(ws), ws.numel() * 4 if fused else 0, _s())
        return out
    ws = _ws(query("dycon_norm_workspace", Nb, V, C), gy)
    nb = gy.numel() * _es(gy)
    deferred = bool(defer_dparams) and not from_y and (dgamma is not None or dbeta is not None) and norm_fwd_is_fused(gy, V, C, G)
    with _Region("norm_bwd", nb * 5, 12 * gy.numel(), parts=[("partial", 2 * nb, 6 * gy.numel()), ("apply", 3 * nb, 6 * gy.numel())]):
        call("dycon_norm_bwd_ex", _p(src), int(from_y), _p(gy), _p(out), dt(gy), Nb, V, C, G, _p(stats), _p(gamma), _p(beta),
             int(relu), _p(chan_scale), _p(dgamma), _p(dbeta), int(deferred), _p(ws), ws.numel() * 4, _s())
    if defer_dparams:
        return out, ((ws, Nb, C, dgamma, dbeta) if deferred else None)
    return out


def norm_head_fwd(x, stats, Nb, V, G, head_w, head_b, gamma=None, beta=None, relu=True, chan_scale=None):
    """the 2-class head fused into the last normalisation (16 channels): logits (Nb, D, H, W, 2) fp32 straight from the pre-norm tensor"""
    logits = torch.empty(x.shape[:-1] + (2,), dtype=torch.float32, device=x.device)
    with _Region("norm_fwd", x.numel() * _es(x) + logits.numel() * 4, 8 * x.numel()):
        call("dycon_norm_head_fwd", _p(x), dt(x), Nb, V, G, _p(stats), _p(gamma), _p(beta), int(relu), _p(chan_scale), _p(head_w),
             _p(head_b), _p(logits), _s())
    return logits


def norm_head_bwd(x, g_logits, stats, Nb, V, G, head_w, gamma=None, beta=None, relu=True, dgamma=None, dbeta=None, chan_scale=None):
    """returns (gx, pending): pending -> norm_head_dparams(pending, d_head_w, d_head_b) on any stream"""
    gx = torch.empty_like(x)
    ws = _ws(query("dycon_norm_head_workspace", Nb, V), x)
    nb = x.numel() * _es(x)
    with _Region("norm_bwd", nb * 3 + g_logits.numel() * 8, 14 * x.numel(),
                 parts=[("partial", nb + g_logits.numel() * 4, 7 * x.numel()), ("apply", 2 * nb + g_logits.numel() * 4, 7 * x.numel())]):
        call("dycon_norm_head_bwd", _p(x), _p(g_logits), _p(gx), dt(x), Nb, V, G, _p(stats), _p(gamma), _p(beta), int(relu),
             _p(chan_scale), _p(head_w), _p(dgamma), _p(dbeta), _p(ws), ws.numel() * 4, _s())
    return gx, (ws, Nb, V)


def norm_head_dparams(pending, d_head_w, d_head_b):
    ws, Nb, V = pending
    call("dycon_norm_head_dparams", _p(ws), Nb, V, _p(d_head_w), _p(d_head_b), _s())


def norm_bwd_stats(src, gy, stats, Nb, V, C, G, gamma=None, beta=None, relu=True, dgamma=None, dbeta=None, chan_scale=None):
    """statistics pass + finalize of the norm backward only; returns (workspace, ab) -- ab = the per-group {A, B} sums (a view)"""
    ws = _ws(query("dycon_norm_workspace", Nb, V, C), gy)
    with _Region("norm_bwd", gy.numel() * _es(gy) * 2, 6 * gy.numel()):
        call("dycon_norm_bwd_stats", _p(src), _p(gy), dt(gy), Nb, V, C, G, _p(stats), _p(gamma), _p(beta), int(relu), _p(chan_scale),
             _p(dgamma), _p(dbeta), _p(ws), ws.numel() * 4, _s())
    off = query("dycon_norm_bwd_ab_offset", Nb, V, C)
    return ws, ws[off:off + Nb * G * 2]


def first_block_bwd(x, z, gy, stats, Nb, G, dw, dbias, gamma=None, beta=None, relu=True, dgamma=None, dbeta=None, chan_scale=None, ws=None):
    """backward of (conv 1 -> 16, k=3) + norm + ReLU in one pass over (x, z, gy): writes dw, dbias, dgamma, dbeta"""
    B, D, H, W, _ = x.shape
    if ws is None:
        ws = _ws(query("dycon_first_block_bwd_workspace", B, D, H, W), x)
    with _Region("wgrad_k3_bf16", (x.numel() + z.numel() + gy.numel()) * 2 + 27 * 16 * 4, 6 * (gy.numel() // 16) * 27 * 16):
        call("dycon_first_block_bwd", _p(x), _p(z), _p(gy), B, D, H, W, Nb, G, _p(stats), _p(gamma), _p(beta), int(relu),
             _p(chan_scale), _p(dgamma), _p(dbeta), _p(dw), _p(dbias), 1, 27, 27, _p(ws), ws.numel() * 4, _s())
    return dw


def conv1_wgrad_normbwd(x, z, gy, stats, ab, Nb, G, dw, dbias, gamma=None, beta=None, relu=True, chan_scale=None, ws=None):
    """weight + bias gradient of the first (1 -> 16, k=3, bf16) convolution with the following normalisation's data gradient formed on load"""
    B, D, H, W, _ = x.shape
    if ws is None:
        ws = _ws(query("dycon_conv1_wgrad_normbwd_workspace", B, D, H, W), x)
    with _Region("wgrad_k3_bf16", (x.numel() + z.numel() + gy.numel()) * 2 + 27 * 16 * 4, 2 * (gy.numel() // 16) * 27 * 16):
        call("dycon_conv1_wgrad_normbwd", _p(x), _p(z), _p(gy), B, D, H, W, Nb, G, _p(stats), _p(gamma), _p(beta), int(relu),
             _p(chan_scale), _p(ab), _p(dw), _p(dbias), 1, 27, 27, _p(ws), ws.numel() * 4, _s())
    return dw


def norm_sum_dparams(pending):
    """finish a deferred norm_bwd (ops.norm_bwd(..., defer_dparams=True)) on the CURRENT launch stream"""
    ws, Nb, C, dgamma, dbeta = pending
    call("dycon_norm_sum_dparams", _p(ws), Nb, C, _p(dgamma), _p(dbeta), _s())


# ------------------------------------------------------------------ data movement / pointwise
def maxpool2_fwd(x):
    B, D, H, W, C = x.shape
    y = torch.empty((B, D // 2, H // 2, W // 2, C), dtype=x.dtype, device=x.device)
    idx = torch.empty(y.shape, dtype=torch.uint8, device=x.device)
    call("dycon_maxpool2_fwd", _p(x), _p(y), _p(idx), dt(x), B, D, H, W, C, _s())
    return y, idx


def maxpool2_bwd(gy, idx, in_shape):
    B, D, H, W, C = in_shape
    gx = torch.empty(in_shape, dtype=gy.dtype, device=gy.device)
    call("dycon_maxpool2_bwd", _p(gy), _p(idx), _p(gx), dt(gy), B, D, H, W, C, _s())
    return gx


def trilinear_fwd(x, out_dhw, align_corners, out=None, coff=0):
    B, D, H, W, C = x.shape
    Do, Ho, Wo = out_dhw
    if out is None:
        out = torch.empty((B, Do, Ho, Wo, C), dtype=x.dtype, device=x.device)
    call("dycon_trilinear_fwd", _p(x), _p(out), dt(x), B, D, H, W, Do, Ho, Wo, C, out.shape[-1], coff, int(align_corners), _s())
    return out


def trilinear_bwd(gy, in_shape, align_corners, coff=0):
    B, D, H, W, C = in_shape
    _, Do, Ho, Wo, ld = gy.shape
    gx = torch.empty(in_shape, dtype=gy.dtype, device=gy.device)
    call("dycon_trilinear_bwd", _p(gy), _p(gx), dt(gy), B, D, H, W, Do, Ho, Wo, C, ld, coff, int(align_corners), _s())
    return gx


def copy_channels(src, soff, dst, doff, C):
    rows = src.numel() // src.shape[-1]
    call("dycon_copy_channels", _p(src), src.shape[-1], soff, _p(dst), dst.shape[-1], doff, rows, C, dt(src), _s())
    return dst


def scale_channels(x, scale):
    B, C = x.shape[0], x.shape[-1]
    V = x.numel() // (B * C)
    y = torch.empty_like(x)
    call("dycon_scale_channels", _p(x), _p(scale), _p(y), dt(x), B, V, C, _s())
    return y


def relu_fwd(z, skip=None, chan_scale=None):
    B, C = z.shape[0], z.shape[-1]
    y = torch.empty_like(z)
    with _Region("relu", z.numel() * _es(z) * (3 if skip is not None else 2), z.numel()):
        call("dycon_relu_fwd", _p(z), _p(skip), _p(chan_scale), _p(y), dt(z), B, z.numel() // (B * C), C, _s())
    return y


def relu_bwd(z, gy, chan_scale=None):
    B, C = z.shape[0], z.shape[-1]
    gz = torch.empty_like(gy)
    with _Region("relu", z.numel() * _es(z) * 3, z.numel()):
        call("dycon_relu_bwd", _p(z), _p(gy), _p(chan_scale), _p(gz), dt(z), B, z.numel() // (B * C), C, _s())
    return gz


def mul_mask(x, mask, inv_keep):
    y = torch.empty_like(x)
    call("dycon_mul_mask", _p(x), _p(mask), inv_keep, _p(y), dt(x), x.numel(), _s())
    return y


def dropout_philox(x, p, seed, offset):
    y = torch.empty_like(x)
    call("dycon_dropout_philox", _p(x), _p(y), dt(x), x.numel(), p, seed, offset, _s())
    return y


def channel_mask_philox(n, p, seed, offset, device):
    s = torch.empty(n, dtype=torch.float32, device=device)
    call("dycon_channel_mask_philox", _p(s), n, p, seed, offset, _s())
    return s


def add_noise(x, noise=None, sigma=0.1, clip=0.2, seed=0, offset=0):
    y = torch.empty_like(x)
    call("dycon_add_noise", _p(x), _p(noise), _p(y), dt(x), x.numel(), sigma, clip, seed, offset, _s())
    return y


def tanh(x):
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    call("dycon_tanh", _p(x), dt(x), _p(y), x.numel(), _s())
    return y


def cast(x, dtype):
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    call("dycon_cast", _p(x), dt(x), _p(y), _DT[dtype], x.numel(), _s())
    return y


def add(a, b=None, out=None):
    if out is None:
        out = torch.empty_like(a)
    call("dycon_add", _p(a), _p(b), _p(out), dt(a), a.numel(), _s())
    return out


# ------------------------------------------------------------------ losses
def _label_bytes(labels):
    if labels.dtype == torch.int64:
        return 8
    if labels.dtype == torch.uint8:
        return 1
    raise TypeError("labels must be int64 or uint8")


def seg_losses_fwd(s_logits, t_logits, labels, LB, beta, fast=False, out=None):
    """s/t_logits: (B, D, H, W, 2) fp32; returns sums (16 doubles on device; `out`: the caller's buffer).  fast: hardware exp/log
    sequences (bf16 step)."""
    B = s_logits.shape[0]
    V = s_logits.numel() // (2 * B)
    sums = out if out is not None else torch.empty(16, dtype=torch.float64, device=s_logits.device)
    call("dycon_seg_losses_fwd", _p(s_logits), _p(t_logits), _p(labels), _label_bytes(labels), B, LB, V, float(beta), _p(sums), int(fast), _s())
    return sums


def seg_losses_bwd(s_logits, t_logits, labels, LB, beta, sums, coef, cons_kind=0, fast=False):
    B = s_logits.shape[0]
    V = s_logits.numel() // (2 * B)
    g = torch.empty_like(s_logits)
    call("dycon_seg_losses_bwd", _p(s_logits), _p(t_logits), _p(labels), _label_bytes(labels), B, LB, V, float(beta), _p(sums),
         _p(coef), cons_kind, _p(g), int(fast), _s())
    return g


def seg_losses_finalize(sums, B, LB, V, beta):
    vals = torch.empty(8, dtype=torch.float32, device=sums.device)
    call("dycon_seg_losses_finalize", _p(sums), B, LB, V, float(beta), _p(vals), _s())
    return vals


def step_losses(sums, fecl_out, B, LB, V, beta, fecl_rows, lambda_cross, has_teacher, l_w, cons_w, u_w, dice_kind, cons_kind,
                nonfinite=None):
    """seg_losses_finalize + fecl_finalize + step_loss as one launch, from the raw accumulators (out[6]: total, ce, dice, cons, fecl, uncl)"""
    out = torch.empty(8, dtype=torch.float32, device=sums.device)
    call("dycon_step_losses", _p(sums), _p(fecl_out), B, LB, V, float(beta), float(fecl_rows), float(lambda_cross), int(bool(has_teacher)),
         float(l_w), float(cons_w), float(u_w), dice_kind, cons_kind, _p(out), _p(nonfinite), _s())
    return out


def step_loss(vals, fecl, l_w, cons_w, u_w, dice_kind, cons_kind, nonfinite=None):
    out = torch.empty(8, dtype=torch.float32, device=vals.device)
    call("dycon_step_loss", _p(vals), _p(fecl), float(l_w), float(cons_w), float(u_w), dice_kind, cons_kind, _p(out),
         _p(nonfinite), _s())
    return out


def l2norm_fwd(x, eps=1e-12):
    C = x.shape[-1]
    R = x.numel() // C
    y = torch.empty_like(x)
    norms = torch.empty(R, dtype=torch.float32, device=x.device)
    call("dycon_l2norm_fwd", _p(x), _p(y), _p(norms), dt(x), R, C, eps, _s())
    return y, norms


def l2norm_bwd(y, norms, gy, eps=1e-12):
    C = y.shape[-1]
    gx = torch.empty_like(gy)
    call("dycon_l2norm_bwd", _p(y), _p(norms), _p(gy), _p(gx), dt(y), y.numel() // C, C, eps, _s())
    return gx


def mask_pool(labels, k):
    """labels (B, D, H, W) int64/uint8 -> (B, N) float mask = avg_pool3d(label, k) > 0.5."""
    B, D, H, W = labels.shape
    kd, kh, kw = (k, k, k) if isinstance(k, int) else k
    m = torch.empty((B, (D // kd) * (H // kh) * (W // kw)), dtype=torch.float32, device=labels.device)
    call("dycon_mask_pool", _p(labels), _label_bytes(labels), _p(m), B, D, H, W, kd, kh, kw, _s())
    return m


class FeclState:
    __slots__ = ("out", "ws", "loss")


def fecl_fwd(feat, teacher, mask, gambling, temperature, gamma, use_focal, cross_thresh, lambda_cross, out=None):
    """feat/teacher (B, N, Dm) normalised rows; mask (B, N) float.  Returns (loss[1] fp32, state); `out`: the caller's 4 doubles."""
    B, N, Dm = feat.shape
    st = FeclState()
    st.out = out if out is not None else torch.empty(4, dtype=torch.float64, device=feat.device)
    st.loss = torch.empty(1, dtype=torch.float32, device=feat.device)
    st.ws = _ws(query("dycon_fecl_workspace", B, N, Dm), feat)
    call("dycon_fecl_fwd", _p(feat), _p(teacher), _p(mask), _p(gambling), dt(feat), B, N, Dm, temperature, gamma,
         int(use_focal), cross_thresh, lambda_cross, _p(st.out), _p(st.loss), _p(st.ws), st.ws.numel() * 4, _s())
    return st.loss, st


def fecl_finalize(st, rows, lambda_cross, has_teacher):
    call("dycon_fecl_finalize", _p(st.out), float(rows), float(lambda_cross), int(bool(has_teacher)), _p(st.loss), _s())
    return st.loss


def set_scalars(dst, values):
    v = [float(x) for x in values] + [0.0] * (8 - len(values))
    call("dycon_set_scalars", _p(dst), len(values), *v, _s())
    return dst


def fecl_bwd(feat, teacher, mask, gambling, temperature, gamma, use_focal, cross_thresh, lambda_cross, st, coef):
    B, N, Dm = feat.shape
    g = torch.empty_like(feat)
    call("dycon_fecl_bwd", _p(feat), _p(teacher), _p(mask), _p(gambling), dt(feat), B, N, Dm, temperature, gamma,
         int(use_focal), cross_thresh, lambda_cross, _p(st.out), _p(coef), _p(g), _p(st.ws), st.ws.numel() * 4, _s())
    return g


# ------------------------------------------------------------------ optimiser
def sumsq(g, out):
    call("dycon_sumsq", _p(g), g.numel(), _p(out), _s())
    return out


def sgd_ema(p, g, mom, teacher, n_sgd, sumsq_t, max_norm, grad_scale, lr, momentum, weight_decay, ema_alpha, skip_flag=None):
    call("dycon_sgd_ema", _p(p), _p(g), _p(mom), _p(teacher), n_sgd, p.numel(), _p(sumsq_t), max_norm, grad_scale, lr,
         momentum, weight_decay, ema_alpha, _p(skip_flag), _s())


def nonfinite_flag(x, flag):
    call("dycon_nonfinite_flag", _p(x), _p(flag), _s())
    return flag
