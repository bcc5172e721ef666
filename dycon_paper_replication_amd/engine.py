"""Explicit forward/backward executor for the two DyCON segmentation nets on the HIP kernels.

The reference builds its nets from torch.nn modules and lets autograd walk them
(code/networks/VNet.py:145-239, code/networks/UNet3D_contrastive.py:207-316).  Here the topology is
a fixed launch sequence over libdycon_hip.so: the forward pushes one backward closure per op on a
tape, the backward replays the tape in reverse.  No torch math, no autograd graph, no host sync:
one step is a pure stream of kernel launches and can be captured into a hipGraph.

Activations are NDHWC tensors (B, D, H, W, C) in ``dtype`` (fp32 parity mode or bf16); logits are
always fp32.  Parameters stay in the reference's torch layouts/names (state_dict contract).
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Dict, Optional

import torch

from . import ops
from ._lib import CONV_1X1, CONV_K2S2, CONV_K3

UNET_FILTERS = (16, 32, 64, 128, 256)
VNET_NORMS = ("groupnorm", "instancenorm", "batchnorm", "none")      # VNet.py:17-24
# tools/ablate.py only (timing experiments with parts of the step switched off, each case in a fresh process): read ONCE at import,
# immutable afterwards -- nothing in a training or test process can switch a part of the step off
ABLATE = frozenset(a for a in os.environ.get("DYCON_ABLATE", "").split(",") if a)
ABLATE_N = int(os.environ.get("DYCON_ABLATE_N", "0"))


# --------------------------------------------------------------------------------------
# parameter specifications (names / shapes / order == the reference modules' state_dict)
# --------------------------------------------------------------------------------------
def _projection_spec(spec, c_in):
    spec["projection.0.weight"] = (512, c_in, 1, 1, 1)
    spec["projection.0.bias"] = (512,)
    spec["projection.1.weight"] = (512,)
    spec["projection.1.bias"] = (512,)
    spec["projection.3.weight"] = (256, 512, 1, 1, 1)
    spec["projection.3.bias"] = (256,)
    spec["projection.4.weight"] = (256,)
    spec["projection.4.bias"] = (256,)


def projection_buffers():
    return OrderedDict([("projection.1.running_mean", (512,)), ("projection.1.running_var", (512,)),
                        ("projection.1.num_batches_tracked", ()), ("projection.4.running_mean", (256,)),
                        ("projection.4.running_var", (256,)), ("projection.4.num_batches_tracked", ())])


def vnet_param_spec(in_ch=1, n_classes=2, normalization="groupnorm", nf=16):
    """VNet.__init__ registration order (VNet.py:150-175) + the DyCON projection head."""
    if normalization not in VNET_NORMS:
        raise ValueError(f"normalization must be one of {VNET_NORMS}")
    spec = OrderedDict()
    affine = normalization in ("groupnorm", "batchnorm")     # InstanceNorm3d: affine=False (VNet.py:21-22)
    step = 2 if normalization == "none" else 3              # nn.Sequential indices: conv, [norm,] relu (VNet.py:16-24)

    def block(name, n, cin, cout):
        for i in range(n):
            ci = cin if i == 0 else cout
            spec[f"{name}.conv.{step * i}.weight"] = (cout, ci, 3, 3, 3)
            spec[f"{name}.conv.{step * i}.bias"] = (cout,)
            if affine:
                spec[f"{name}.conv.{step * i + 1}.weight"] = (cout,)
                spec[f"{name}.conv.{step * i + 1}.bias"] = (cout,)

    def down(name, cin, cout):
        spec[f"{name}.conv.0.weight"] = (cout, cin, 2, 2, 2)
        spec[f"{name}.conv.0.bias"] = (cout,)
        if affine:
            spec[f"{name}.conv.1.weight"] = (cout,)
            spec[f"{name}.conv.1.bias"] = (cout,)

    def up(name, cin, cout):
        spec[f"{name}.conv.0.weight"] = (cin, cout, 2, 2, 2)
        spec[f"{name}.conv.0.bias"] = (cout,)
        if affine:
            spec[f"{name}.conv.1.weight"] = (cout,)
            spec[f"{name}.conv.1.bias"] = (cout,)

    block("block_one", 1, in_ch, nf); down("block_one_dw", nf, 2 * nf)
    block("block_two", 2, 2 * nf, 2 * nf); down("block_two_dw", 2 * nf, 4 * nf)
    block("block_three", 3, 4 * nf, 4 * nf); down("block_three_dw", 4 * nf, 8 * nf)
    block("block_four", 3, 8 * nf, 8 * nf); down("block_four_dw", 8 * nf, 16 * nf)
    block("block_five", 3, 16 * nf, 16 * nf); up("block_five_up", 16 * nf, 8 * nf)
    block("block_six", 3, 8 * nf, 8 * nf); up("block_six_up", 8 * nf, 4 * nf)
    block("block_seven", 3, 4 * nf, 4 * nf); up("block_seven_up", 4 * nf, 2 * nf)
    block("block_eight", 2, 2 * nf, 2 * nf); up("block_eight_up", 2 * nf, nf)
    block("block_nine", 1, nf, nf)
    spec["out_conv.weight"] = (n_classes, nf, 1, 1, 1)
    spec["out_conv.bias"] = (n_classes,)
    _projection_spec(spec, 16 * nf)
    return spec


def vnet_norm_sites(normalization="groupnorm"):
    """state_dict prefixes of the V-Net's norm modules, in registration order (VNet.py:150-174)."""
    if normalization == "none":
        return []
    sites = []
    for blk, n, nxt in (("block_one", 1, "block_one_dw"), ("block_two", 2, "block_two_dw"), ("block_three", 3, "block_three_dw"),
                        ("block_four", 3, "block_four_dw"), ("block_five", 3, "block_five_up"), ("block_six", 3, "block_six_up"),
                        ("block_seven", 3, "block_seven_up"), ("block_eight", 2, "block_eight_up"), ("block_nine", 1, None)):
        sites += [f"{blk}.conv.{3 * i + 1}" for i in range(n)]
        if nxt:
            sites.append(f"{nxt}.conv.1")
    return sites


def net_buffers(net_type, normalization="groupnorm", params=None):
    """BatchNorm buffers of the net in state_dict order: the V-Net's own (normalization='batchnorm', VNet.py:17-18) + the head's."""
    bufs = OrderedDict()
    if net_type == "vnet" and normalization == "batchnorm":
        spec = params if params is not None else vnet_param_spec(normalization=normalization)
        for site in vnet_norm_sites(normalization):
            c = tuple(spec[site + ".weight"])
            bufs[site + ".running_mean"], bufs[site + ".running_var"], bufs[site + ".num_batches_tracked"] = c, c, ()
    bufs.update(projection_buffers())
    return bufs


def unet_param_spec(in_ch=1, n_classes=2):
    """UNet3D.__init__ registration order (UNet3D_contrastive.py:222-267), feature_scale=4."""
    spec = OrderedDict()
    f = UNET_FILTERS

    def uc(prefix, cin, cout):
        spec[f"{prefix}.conv1.0.weight"] = (cout, cin, 3, 3, 3)
        spec[f"{prefix}.conv1.0.bias"] = (cout,)
        spec[f"{prefix}.conv2.0.weight"] = (cout, cout, 3, 3, 3)
        spec[f"{prefix}.conv2.0.bias"] = (cout,)

    uc("conv1", in_ch, f[0]); uc("conv2", f[0], f[1]); uc("conv3", f[1], f[2]); uc("conv4", f[2], f[3])
    uc("center", f[3], f[4])
    uc("up_concat4.conv", f[4] + f[3], f[3]); uc("up_concat3.conv", f[3] + f[2], f[2])
    uc("up_concat2.conv", f[2] + f[1], f[1]); uc("up_concat1.conv", f[1] + f[0], f[0])
    spec["final.weight"] = (n_classes, f[0], 1, 1, 1)
    spec["final.bias"] = (n_classes,)
    spec["out_conv2.weight"] = (n_classes, f[0], 1, 1, 1)
    spec["out_conv2.bias"] = (n_classes,)
    _projection_spec(spec, f[4])
    return spec


def param_spec(net_type, in_ch=1, n_classes=2, normalization="groupnorm"):
    if net_type == "vnet":
        return vnet_param_spec(in_ch, n_classes, normalization)
    if net_type == "unet_3D":
        return unet_param_spec(in_ch, n_classes)
    raise ValueError(f"unknown net_type {net_type!r}")


# --------------------------------------------------------------------------------------
class DropoutSpec:
    """How the dropout sites draw their masks.

    mode "off"    : identity (eval / parity runs with p = 0)
    mode "mask"   : explicit keep-masks in ``masks`` (parity tests feed the oracle's masks)
                    V-Net keys "drop5", "drop9": (B, C) ; U-Net keys "drop_center", "drop_up1": NDHWC-shaped
    mode "philox" : on-device counter RNG, (seed, offset); offset must change every step
    """

    def __init__(self, mode="off", masks=None, seed=0, offset=0):
        self.mode, self.masks, self.seed, self.offset = mode, masks or {}, seed, offset


class Engine:
    def __init__(self, net_type: str, params: Dict[str, torch.Tensor], grads: Optional[Dict[str, torch.Tensor]] = None,
                 buffers: Optional[Dict[str, torch.Tensor]] = None, dtype=torch.float32, scale_factor=2,
                 normalization="groupnorm"):
        if net_type == "vnet" and normalization not in VNET_NORMS:
            raise ValueError(f"normalization must be one of {VNET_NORMS} (VNet.py:17-24)")
        self.net_type, self.p, self.g, self.buf = net_type, params, grads, buffers or {}
        self.dtype, self.scale_factor, self.normalization = dtype, scale_factor, normalization
        self.gen = 0                 # bump whenever parameter values change (invalidates packed weights)
        self._packed = {}
        self._jobs, self._jobs_dev, self._packed_gen = [], None, -1
        self._jobs_split, self._pack_event = False, None
        self.on_param_grads = None   # optional callback(name): called in backward once a layer's parameter gradients are enqueued
        self.wgrad_stream = None     # optional side stream for the weight-gradient launches of the backward (set by the trainer)
        self.wgrad_stream2 = None    # optional second one: the convolutions' weight gradients then alternate between the two
        self.wgrad_stream3 = None    # (diagnostic) a third
        self._wg_flip = 0
        self.feat_stream = None      # optional stream for the feature branch (projection head forward + backward), see forward()
        self.mark = None             # optional callback(tag, stream): timeline marks (trainer._mark)
        self.stage_hook = None       # optional callback(name): called in the V-Net forward when the encoder tensor `name` (x1..x5) is enqueued
        # per-step arena of zeroed doubles for the accumulator form of the norms (ops.norm_fwd / norm_bwd `acc`): slices are handed
        # out in call order; whoever owns the arena clears it once per step BEFORE the first norm (the trainer: one launch for both
        # nets; a stand-alone engine: at the start of forward()).
        # Measured SLOWER than partials -> finalize -> apply (5.8 -> 6.4 ms/step with one accumulator copy per sample: <= 512 adds
        # per address; 6.6 ms with 32 copies: the prologue of every apply workgroup then reads 64 doubles per channel) -- DESIGN.md
        # section 9.  Kept (tested: test_norm_accumulator_form_equals_three_launch_form), off by default.
        self.use_acc = False
        self.acc_arena = None
        self.acc_external = False
        self._acc_off = 0
        self.Gs = {}
        self._head_range = (0, 0)
        self._wws = {}               # per-layer workspaces of the weight-gradient launches
        # split-K finish of a conv fused into the one-launch norm that follows (small levels, bf16): bit-identical and one launch
        # less, but measured SLOWER (5.8 -> 6.1 ms/step): the norm's workgroups own 8-16 channels of one sample, so they read the
        # fp32 slabs in 32-byte pieces at a C*4-byte stride where splitk_finish streams them fully coalesced.  Off by default.
        self.fuse_finish = False
        self.fuse_head = True            # V-Net: the 2-class head inside the last normalisation's passes (_norm_head)
        self.conv_stats = False          # 48^3 level: the persistent convolution takes the statistics of its output (ops.conv_gemm_stats);
                                         # measured neutral for the step (+5 us per convolution against a 7 us statistics launch): off
        self.conv_stats96 = False        # 96^3 level (conv_k3_c1 / conv_k3_p16): the same; measured neutral as well (TrainConfig.conv_stats96)
        self._stat_parts = {}
        self.fuse_first = True           # V-Net: block_one's norm backward formed on load by the first layer's weight gradient (_first_block)
        self.one_pass_first = True       # ... as ONE pass (dycon_first_block_bwd) instead of statistics + weight gradient
        self._deferred = {}
        self._pending_dparams = []
        self.tape = []
        self.G = {}

    # ---------------------------------------------------------------- packed-weight cache
    def params_changed(self):
        self.gen += 1

    def _pk(self, key, kind, w, T, Cin, N, N0, s_t, s_c, s_n1, s_n0, flip=False):
        """Packed operand for ``key`` (kind "frag": MFMA B fragments in the compute dtype; "tcn": plain fp32).
        The first request packs on the spot and registers a job; later generations are refreshed by repack()
        (one launch for every registered job) or, if that was not called, lazily here."""
        hit = self._packed.get(key)
        if hit is not None and hit[0] == self.gen:
            return hit[1]
        buf = hit[1] if hit is not None else None
        if kind == "frag":
            buf = ops.pack_bfrag(w, self.dtype, T, Cin, N, N0, s_t, s_c, s_n1, s_n0, flip, out=buf)
            code = 0 if self.dtype == torch.bfloat16 else 1
        else:
            buf = ops.pack_tcn(w, T, Cin, N, N0, s_t, s_c, s_n1, s_n0, flip, out=buf)
            code = 2
        if hit is None:
            self._jobs.append((key, ops.pack_job(w, buf, code, T, Cin, N, N0, s_t, s_c, s_n1, s_n0, flip)))
            self._jobs_dev = None
        self._packed[key] = (self.gen, buf)
        return buf

    def _pk_chunked16(self, key, w, nchunks, N, s_t, s_c, s_n0, flip=False):
        """bf16 k=3 weights for the LDS-halo kernel with 48 input channels: chunk-major, one (T=27, Cin=16) pack per 16-channel
        chunk, back to back in one buffer (include/dycon_hip.h, dycon_conv_gemm)."""
        hit = self._packed.get(key)
        if hit is not None and hit[0] == self.gen:
            return hit[1]
        per = ops.query("dycon_bfrag_bytes", ops.BF16, 27, 16, N) // 2
        buf = hit[1] if hit is not None else torch.empty(per * nchunks, dtype=torch.bfloat16, device=w.device)
        wflat = w.reshape(-1)
        for ch in range(nchunks):
            src, dst = wflat[ch * 16 * s_c:], buf[ch * per:(ch + 1) * per]
            ops.pack_bfrag(src, self.dtype, 27, 16, N, N, s_t, s_c, 0, s_n0, flip, out=dst)
            if hit is None:
                self._jobs.append(((key, ch), ops.pack_job(src, dst, 0, 27, 16, N, N, s_t, s_c, 0, s_n0, flip)))
                self._jobs_dev = None
        self._packed[key] = (self.gen, buf)
        return buf

    def repack(self, early=None, helper=None):
        """Refresh every registered packed operand (call after the parameters changed): one launch -- or, with `early` (a layer-name
        prefix) and `helper` (an idle HIP stream), two: the packs of the `early` layers on the launch stream, everything else on
        `helper`, whose completion event the forward waits for right before the first layer that is not `early` (pack_ready).
        The 55 us pack of a V-Net then no longer stands in front of the step's first convolution."""
        self._pack_event = None
        if not self._jobs or self._packed_gen == self.gen:
            return
        split = early is not None and helper is not None
        if self._jobs_dev is None or self._jobs_split != split:
            dev = next(iter(self._packed.values()))[1].device
            name = lambda key: (key[0][0] if isinstance(key[0], tuple) else key[0])     # noqa: E731  (chunked packs: ((name, tag), ch))
            first = [j for k, j in self._jobs if split and name(k).startswith(early)]
            rest = [j for k, j in self._jobs if not (split and name(k).startswith(early))]
            self._jobs_dev = [(ops.upload_pack_jobs(js, dev), len(js), max(1, min(512, (max(j[6] for j in js) + 2047) // 2048)))
                              for js in (first, rest) if js]
            self._jobs_split = split
        if split and len(self._jobs_dev) == 2:
            (d0, n0, b0), (d1, n1, b1) = self._jobs_dev
            ops.pack_batch(d0, n0, b0)
            ops.fork(ops.cur_stream(), helper)       # the update that changed the parameters precedes this point
            with ops.on_stream(helper, light=True):
                ops.pack_batch(d1, n1, b1)
                ev1 = ops.Event()
                ev1.record(helper)
            self._pack_event = ev1
        else:
            for d, n, b in self._jobs_dev:
                ops.pack_batch(d, n, b)
        for key, _ in self._jobs:
            key = key[0] if isinstance(key[0], tuple) else key        # chunked packs register one job per chunk
            self._packed[key] = (self.gen, self._packed[key][1])
        self._packed_gen = self.gen

    def pack_ready(self):
        """the packs launched on the helper stream by repack(early=..., helper=...) are needed from here on"""
        ev = getattr(self, "_pack_event", None)
        if ev is not None:
            ev.wait(ops.cur_stream())
            self._pack_event = None

    # ---------------------------------------------------------------- gradient bookkeeping
    # With the feature branch on its own stream (feat_stream) gradients cross streams at the tensor both branches consume
    # (x5 / center): every entry of G remembers the stream that produced it, and whoever reads or accumulates it from another
    # stream waits for that stream first (and tells the caching allocator about the second user).
    def _sync_to(self, key, g):
        if self.feat_stream is None:
            return
        cur = ops.cur_stream()
        src = self.Gs.get(key)
        if src is not None and src != cur:
            ops.fork(src, cur)
            g.record_stream(cur)

    def _take(self, t):
        g = self.G.pop(id(t))
        self._sync_to(id(t), g)
        self.Gs.pop(id(t), None)
        return g

    def _peek(self, t):
        """the gradient accumulated so far for t (or None), safe to read / accumulate into on the current stream"""
        g = self.G.get(id(t))
        if g is not None:
            self._sync_to(id(t), g)
        return g

    def _put(self, t, g):
        self.G[id(t)] = g
        if self.feat_stream is not None:
            self.Gs[id(t)] = ops.cur_stream()

    def _give(self, t, g):
        cur = self._peek(t)
        self._put(t, g if cur is None else ops.add(cur, g))

    # ---------------------------------------------------------------- conv layers
    def _conv(self, name, x, kind, need_gx=True, out_dtype=None, norm_groups=0):
        """kind: k3 | k2s2 | deconv | 1x1.  Weight layouts as in torch (Conv3d: (Co,Ci,k..); ConvTranspose3d: (Ci,Co,k..))."""
        w, b = self.p[name + ".weight"], self.p[name + ".bias"]
        dtype = self.dtype
        if kind == "deconv":
            Cin, Cout = w.shape[0], w.shape[1]
        else:
            Cout, Cin = w.shape[0], w.shape[1]
        T = {"k3": 27, "k2s2": 8, "deconv": 8, "1x1": 1}[kind]
        mode = {"k3": CONV_K3, "k2s2": CONV_K2S2, "deconv": CONV_1X1, "1x1": CONV_1X1}[kind]
        gq = 8 if dtype == torch.bfloat16 else 4
        skinny = (Cin % gq != 0) or (Cout % 16 != 0)
        out_dtype = out_dtype or dtype
        first_lds = (kind == "k3" and Cin == 1 and dtype == torch.bfloat16 and Cout in (16, 32, 64) and out_dtype == dtype
                     and x.shape[1] * x.shape[2] * x.shape[3] >= 13824)
        if first_lds:   # first layer on the matrix cores: K = 27 taps in one 32-wide k-step (conv_k3_c1_kernel), see dycon_hip.h
            wf = self._pk((name, "f1"), "frag", w, 27, 1, Cout, Cout, 1, 27, 0, 27)
            chunks = ops.conv_stats_chunks(x, Cin, Cout) if (norm_groups and self.conv_stats96 and Cout == 16) else 0
            if chunks:      # the persistent kernel also takes the statistics of its output (see below)
                y, part = ops.conv_gemm_stats(x, wf, b, Cout, chunks)
                self._stat_parts[id(y)] = (part, chunks)
            else:
                y = ops.conv_gemm(x, wf, b, CONV_K3, Cout, Cout)
        elif skinny:
            assert kind in ("k3", "1x1"), "skinny path only for the first conv and the 1x1 heads"
            wt = self._pk((name, "tcn"), "tcn", w, T, Cin, Cout, Cout, 1, T, 0, Cin * T)
            y = ops.conv_direct(x, wt, b, mode, Cout, out_dtype)
        elif kind == "deconv":
            wf = self._pk((name, "f"), "frag", w, 1, Cin, 8 * Cout, Cout, 0, Cout * 8, 1, 8)
            y = ops.conv_gemm(x, wf, b, CONV_1X1, 8 * Cout, Cout, scatter=True)
        elif kind == "k3" and Cin == 48 and ops.conv_uses_lds(x, Cin, Cout):
            wf = self._pk_chunked16((name, "f48"), w, 3, Cout, 1, 27, Cin * 27)
            y = ops.conv_gemm(x, wf, b, mode, Cout, Cout)
        else:
            wf = self._pk((name, "f"), "frag", w, T, Cin, Cout, Cout, 1, T, 0, Cin * T)
            # a GroupNorm / InstanceNorm of the one-launch kind follows (norm_groups > 0): leave a split-K finish to it
            Vo = (x.shape[1] * x.shape[2] * x.shape[3]) // (8 if kind == "k2s2" else 1)
            want_stats = self.conv_stats96 if (Cin, Cout) == (16, 16) else self.conv_stats       # 96^3 level / 48^3 level
            chunks = (ops.conv_stats_chunks(x, Cin, Cout)
                      if (norm_groups and want_stats and kind == "k3" and out_dtype == dtype and dtype == torch.bfloat16) else 0)
            if chunks:      # the persistent kernel also takes the statistics of its output: the norm that follows skips its statistics pass
                y, part = ops.conv_gemm_stats(x, wf, b, Cout, chunks)
                self._stat_parts[id(y)] = (part, chunks)
            elif (norm_groups and self.fuse_finish and dtype == torch.bfloat16 and out_dtype == dtype
                    and ops.norm_fwd_is_fused(x, Vo, Cout, norm_groups)):
                y, dc = ops.conv_gemm(x, wf, b, mode, Cout, Cout, defer_finish=True)
                if dc is not None:
                    self._deferred[id(y)] = dc
            else:
                y = ops.conv_gemm(x, wf, b, mode, Cout, Cout)

        if self.recording:
            def wgrad(gy):
                gw, gb = self.g[name + ".weight"], self.g[name + ".bias"]
                # workspaces of the weight-gradient launches are kept per layer: on the side stream nothing may be allocated
                # (the section only redirects this module's launches, torch's current stream -- and allocator pool -- stays put)
                wkey = (name, tuple(x.shape[:4]))       # (one Engine serves training patches and sliding-window batches)
                ws = self._wws.get(wkey)
                if kind == "deconv":   # dW[ci][co][t] = sum_m x[m,ci] * gy[2m+t,co]   (roles of x and gy swapped)
                    if ws is None:
                        ws = self._wws[wkey] = (ops._ws(ops.query("dycon_colsum_workspace", gy.numel() // gy.shape[-1], gy.shape[-1]), gy),
                                                ops.conv_wgrad_workspace(gy, x, CONV_K2S2))
                    ops.colsum(gy, gb, ws=ws[0])
                    ops.conv_wgrad(gy, x, gw, CONV_K2S2, 1, 8, Cout * 8, ws=ws[1])
                else:                  # bias gradient = column sums of gy, fused into the wgrad pass
                    if ws is None:
                        ws = self._wws[wkey] = ops.conv_wgrad_workspace(x, gy, mode)
                    ops.conv_wgrad(x, gy, gw, mode, 1 if T > 1 else 0, T, Cin * T, dbias=gb, ws=ws)
                if self.on_param_grads is not None:
                    self.on_param_grads(name + ".weight")      # this layer's gradients are enqueued (DDP bucket trigger)

            def bwd():
                gy = self._take(y)
                if "wgrad" in ABLATE:
                    pass
                elif "wgrad_events_only" in ABLATE and self.wgrad_stream is not None:      # the fork, without the kernels
                    ops.fork(ops.cur_stream(), self.wgrad_stream)
                elif "wgrad_no_events" in ABLATE and self.wgrad_stream is not None:        # the kernels, without the fork (a race: timing only)
                    with ops.on_stream(self.wgrad_stream, light=True):
                        wgrad(gy)
                elif self.wgrad_stream is not None:
                    # The weight gradient only feeds the optimiser; the data gradient is the critical chain.  Enqueue the former on a
                    # second HIP stream (behind an event that marks gy ready) so the small-level wgrad / reduce launches fill the
                    # CUs the latency-bound dgrad / norm-backward kernels leave idle.  backward() joins the streams at the end.
                    wst = self.wgrad_stream
                    if self.wgrad_stream2 is not None:
                        ring = [self.wgrad_stream, self.wgrad_stream2] + ([self.wgrad_stream3] if self.wgrad_stream3 is not None else [])
                        self._wg_flip = (self._wg_flip + 1) % len(ring)
                        wst = ring[self._wg_flip]
                    ops.fork(ops.cur_stream(), wst)
                    with ops.on_stream(wst, light=True):
                        self._flush_dparams()
                        wgrad(gy)
                    gy.record_stream(wst)
                else:
                    wgrad(gy)
                if not need_gx:
                    return
                cur = self._peek(x)
                if "dgrad" in ABLATE:
                    self._put(x, cur if cur is not None else torch.empty_like(x))
                    return
                if skinny:             # 1x1 head: gx[m,ci] = sum_co gy[m,co] W[co][ci]
                    wd = self._pk((name, "tcn_d"), "tcn", w, 1, Cout, Cin, Cin, 0, Cin, 0, 1)
                    gx = ops.conv_direct(gy, wd, None, CONV_1X1, Cin, x.dtype, out=cur, accumulate=cur is not None)
                elif kind == "k3":     # conv with flipped taps and transposed channels
                    if Cout == 48 and ops.conv_uses_lds(gy, Cout, Cin):
                        wd = self._pk_chunked16((name, "d48"), w, 3, Cin, 1, Cin * 27, 27, flip=True)
                    else:
                        wd = self._pk((name, "d"), "frag", w, 27, Cout, Cin, Cin, 1, Cin * 27, 0, 27, flip=True)
                    gx = ops.conv_gemm(gy, wd, None, CONV_K3, Cin, Cin, out=cur, accumulate=cur is not None)
                elif kind == "k2s2":   # scatter: gx[2m+t, ci] = sum_co gy[m,co] W[co][ci][t]
                    wd = self._pk((name, "d"), "frag", w, 1, Cout, 8 * Cin, Cin, 0, Cin * 8, 1, 8)
                    gx = ops.conv_gemm(gy, wd, None, CONV_1X1, 8 * Cin, Cin, scatter=True, out=cur, accumulate=cur is not None)
                elif kind == "deconv":  # gather: gx[m, ci] = sum_{t,co} gy[2m+t,co] W[ci][co][t]
                    wd = self._pk((name, "d"), "frag", w, 8, Cout, Cin, Cin, 1, 8, 0, Cout * 8)
                    gx = ops.conv_gemm(gy, wd, None, CONV_K2S2, Cin, Cin, out=cur, accumulate=cur is not None)
                else:                  # 1x1
                    wd = self._pk((name, "d"), "frag", w, 1, Cout, Cin, Cin, 0, Cin, 0, 1)
                    gx = ops.conv_gemm(gy, wd, None, CONV_1X1, Cin, Cin, out=cur, accumulate=cur is not None)
                self._put(x, gx)
            self.tape.append(bwd)
        return y

    ACC_DOUBLES = 1 << 20      # 8 MB: ~40 slices of Nb x 32 slots x C x 2 doubles

    def _acc(self, n):
        """n zeroed doubles of this step's arena (None when the arena is exhausted: the caller then takes the three-launch path)"""
        if not self.use_acc or self.acc_arena is None or self._acc_off + n > self.acc_arena.numel():
            return None
        a = self.acc_arena[self._acc_off:self._acc_off + n]
        self._acc_off += n
        return a

    def _flush_dparams(self):
        """deferred dgamma / dbeta sums of the norms whose backward has been enqueued (call inside the weight-gradient section)"""
        for pend in self._pending_dparams:
            ops.norm_sum_dparams(pend)
            pend[0].record_stream(ops.cur_stream())
        self._pending_dparams = []

    # ---------------------------------------------------------------- norm (+ReLU, +skip)
    def _norm(self, prefix, z, kind, relu=True, skip=None, training=True, chan_scale=None):
        B, C = z.shape[0], z.shape[-1]
        V = z.numel() // (B * C)
        gamma = beta = None
        rm = rv = None
        if kind == "bn" and chan_scale is not None:
            # BatchNorm runs as ONE sample of B*V voxels, while nn.Dropout3d draws a mask per (sample, channel) (VNet.py:195-196,
            # 225-226): the norm kernels index their dropout factor per norm-sample, so here the factor is applied by its own pass
            y = self._norm(prefix, z, kind, relu=relu, skip=skip, training=training)
            y2 = ops.scale_channels(y, chan_scale)
            if self.recording:
                def bwd_scale():
                    self._give(y, ops.scale_channels(self._take(y2), chan_scale))
                self.tape.append(bwd_scale)
            return y2
        if kind == "gn":
            Nb, G = B, 16
            gamma, beta = self.p[prefix + ".weight"], self.p[prefix + ".bias"]
        elif kind == "in":
            Nb, G = B, C
        elif kind == "bn":
            Nb, G, V = 1, C, B * V
            gamma, beta = self.p[prefix + ".weight"], self.p[prefix + ".bias"]
            rm, rv = self.buf.get(prefix + ".running_mean"), self.buf.get(prefix + ".running_var")
        elif kind == "none":           # normalization='none': conv -> ReLU (VNet.py:23-24); same fusions (dropout factor, skip add)
            assert relu
            y = ops.relu_fwd(z, skip, chan_scale)
            if self.recording:
                def bwd_relu():
                    gy = self._take(y)
                    if skip is not None:
                        self._give(skip, gy)
                    self._give(z, ops.relu_bwd(z, gy, chan_scale))
                self.tape.append(bwd_relu)
            return y
        else:
            raise ValueError(kind)
        # z is kept: the ReLU is not invertible, so the backward needs the pre-norm tensor (xhat of the
        # clamped voxels still enters the group means)
        if kind == "bn" and not training:
            # eval-mode BatchNorm (ISLES teacher, train_DyCON_ISLES22.py:114): running statistics
            stats = torch.stack([rm, torch.rsqrt(rv + 1e-5)], 1).reshape(-1).contiguous()
            y = ops.norm_apply(z, stats, Nb, V, C, G, gamma, beta, relu, skip, chan_scale=chan_scale)
        elif id(z) in self._stat_parts and kind in ("gn", "in"):     # the producing convolution left the statistics partials behind
            part, chunks = self._stat_parts.pop(id(z))
            y, stats = ops.norm_fwd_parts(z, part, chunks, Nb, V, C, G, gamma, beta, relu, skip, chan_scale)
        elif id(z) in self._deferred:     # z is still split-K slabs: bias + ordered sum + rounding + norm in ONE launch
            y, stats = ops.norm_fwd_slab(z, self._deferred.pop(id(z)), Nb, V, C, G, gamma, beta, relu, skip, chan_scale)
        else:
            upd = kind == "bn" and training and self.update_bn
            acc = None if ops.norm_fwd_is_fused(z, V, C, G) else self._acc(ops.query("dycon_norm_acc_doubles", Nb, V, C))
            y, stats = ops.norm_fwd(z, Nb, V, C, G, gamma, beta, relu, skip, chan_scale, 1e-5, rm if upd else None,
                                    rv if upd else None, 0.1, acc=acc)
            if upd and prefix + ".num_batches_tracked" in self.buf:
                nbt = self.buf[prefix + ".num_batches_tracked"]
                ops.rec(lambda: nbt.add_(1))
        if self.recording:
            def bwd():
                gy = self._take(y)
                if skip is not None:
                    self._give(skip, gy)
                dg = self.g[prefix + ".weight"] if gamma is not None else None
                db = self.g[prefix + ".bias"] if beta is not None else None
                acc = None if ops.norm_fwd_is_fused(z, V, C, G) else self._acc(ops.query("dycon_norm_acc_doubles", Nb, V, C))
                if "norm_bwd" in ABLATE:
                    gz = gy
                elif acc is None and self.wgrad_stream is not None and dg is not None:
                    # one-launch shapes: the tiny sum of the per-sample dgamma / dbeta contributions leaves the dependent chain -- it
                    # is enqueued on the weight-gradient stream by the convolution's backward that follows (same fork event)
                    gz, pend = ops.norm_bwd(z, False, gy, stats, Nb, V, C, G, gamma, beta, relu, dg, db, chan_scale=chan_scale,
                                            defer_dparams=True)
                    if pend is not None:
                        self._pending_dparams.append(pend)
                else:
                    gz = ops.norm_bwd(z, False, gy, stats, Nb, V, C, G, gamma, beta, relu, dg, db, chan_scale=chan_scale, acc=acc)
                if self.on_param_grads is not None and gamma is not None:
                    self.on_param_grads(prefix + ".weight")
                self._give(z, gz)
            self.tape.append(bwd)
        return y

    def _first_block(self, name, x, kind, training):
        """block_one (conv 1 -> 16, norm, ReLU; VNet.py:176): the normalisation's data gradient has ONE consumer, the convolution's weight
        gradient (the image needs no gradient), so the backward never stores it: norm statistics pass + finalize on the chain, then the
        weight gradient forms it on load (csrc/conv.hip, wgrad_k3_c1_kernel<true>) -- the backward-apply pass over the step's largest
        tensor and the weight gradient's read of its result are gone from the exposed tail of the backward."""
        cname, nname = f"{name}.conv.0", f"{name}.conv.1"
        rec, self.recording = self.recording, False
        try:
            z = self._conv(cname, x, "k3", need_gx=False, norm_groups=(16 if kind == "gn" else (16 if kind == "in" else 0)))
        finally:
            self.recording = rec
        B, C = z.shape[0], z.shape[-1]
        V = z.numel() // (B * C)
        gamma = beta = rm = rv = None
        if kind == "gn":
            Nb, G = B, 16
            gamma, beta = self.p[nname + ".weight"], self.p[nname + ".bias"]
        elif kind == "in":
            Nb, G = B, C
        else:
            Nb, G, V = 1, C, B * V
            gamma, beta = self.p[nname + ".weight"], self.p[nname + ".bias"]
            rm, rv = self.buf.get(nname + ".running_mean"), self.buf.get(nname + ".running_var")
        upd = kind == "bn" and training and self.update_bn
        if id(z) in self._stat_parts and kind in ("gn", "in"):     # the convolution left the statistics partials behind
            part, chunks = self._stat_parts.pop(id(z))
            y, stats = ops.norm_fwd_parts(z, part, chunks, Nb, V, C, G, gamma, beta, True, None, None)
        else:
            self._stat_parts.pop(id(z), None)
            y, stats = ops.norm_fwd(z, Nb, V, C, G, gamma, beta, True, None, None, 1e-5, rm if upd else None, rv if upd else None, 0.1)
        if upd and nname + ".num_batches_tracked" in self.buf:
            nbt = self.buf[nname + ".num_batches_tracked"]
            ops.rec(lambda: nbt.add_(1))
        if rec:
            def bwd():
                gy = self._take(y)
                dg = self.g[nname + ".weight"] if gamma is not None else None
                db = self.g[nname + ".bias"] if beta is not None else None
                gw, gb = self.g[cname + ".weight"], self.g[cname + ".bias"]
                if self.one_pass_first:       # everything in ONE pass over (x, z, gy), on the chain (it is the exposed tail of the backward)
                    wws = self._wws.get((cname, "#fb", tuple(x.shape[:4])))
                    if wws is None:
                        wws = self._wws[(cname, "#fb", tuple(x.shape[:4]))] = ops._ws(ops.query("dycon_first_block_bwd_workspace", *x.shape[:4]), x)
                    ops.first_block_bwd(x, z, gy, stats, Nb, G, gw, gb, gamma, beta, True, dg, db, None, ws=wws)
                else:
                    nws, ab = ops.norm_bwd_stats(z, gy, stats, Nb, V, C, G, gamma, beta, True, dg, db)
                    wws = self._wws.get((cname, "#nb", tuple(x.shape[:4])))
                    if wws is None:
                        wws = self._wws[(cname, "#nb", tuple(x.shape[:4]))] = ops._ws(ops.query("dycon_conv1_wgrad_normbwd_workspace", *x.shape[:4]), x)
                    if self.wgrad_stream is not None:
                        ops.fork(ops.cur_stream(), self.wgrad_stream)
                        with ops.on_stream(self.wgrad_stream, light=True):
                            self._flush_dparams()
                            ops.conv1_wgrad_normbwd(x, z, gy, stats, ab, Nb, G, gw, gb, gamma, beta, True, None, ws=wws)
                        for t in (gy, nws):
                            t.record_stream(self.wgrad_stream)
                    else:
                        ops.conv1_wgrad_normbwd(x, z, gy, stats, ab, Nb, G, gw, gb, gamma, beta, True, None, ws=wws)
                if self.on_param_grads is not None:
                    if gamma is not None:
                        self.on_param_grads(nname + ".weight")
                    self.on_param_grads(cname + ".weight")
            self.tape.append(bwd)
        return y

    def _norm_head(self, prefix, z, kind, head, training=True, chan_scale=None):
        """block_nine's norm -> ReLU [-> Dropout3d] -> out_conv (VNet.py:225-227) as ONE pass over the pre-norm tensor, forward and
        backward: the normalised 16-channel tensor (the largest activation of the step) and its gradient are never written
        (csrc/norm.hip, dycon_norm_head_*).  Same logits as _norm + _conv('1x1') bit for bit, same data gradient to fp32 round-off."""
        B, C = z.shape[0], z.shape[-1]
        V = z.numel() // (B * C)
        gamma = beta = rm = rv = None
        if kind == "gn":
            Nb, G = B, 16
            gamma, beta = self.p[prefix + ".weight"], self.p[prefix + ".bias"]
        elif kind == "in":
            Nb, G = B, C
        else:
            Nb, G, V = 1, C, B * V
            gamma, beta = self.p[prefix + ".weight"], self.p[prefix + ".bias"]
            rm, rv = self.buf.get(prefix + ".running_mean"), self.buf.get(prefix + ".running_var")
        hw, hb = self.p[head + ".weight"], self.p[head + ".bias"]
        if kind == "bn" and not training:
            self._stat_parts.pop(id(z), None)
            stats = torch.stack([rm, torch.rsqrt(rv + 1e-5)], 1).reshape(-1).contiguous()
        elif id(z) in self._stat_parts and kind in ("gn", "in"):   # block_nine's convolution left the statistics partials behind
            part, chunks = self._stat_parts.pop(id(z))
            stats = ops.norm_stats_parts(z, part, chunks, Nb, V, C, G)
        else:
            self._stat_parts.pop(id(z), None)
            upd = kind == "bn" and training and self.update_bn
            stats = ops.norm_stats(z, Nb, V, C, G, 1e-5, rm if upd else None, rv if upd else None, 0.1)
            if upd and prefix + ".num_batches_tracked" in self.buf:
                nbt = self.buf[prefix + ".num_batches_tracked"]
                ops.rec(lambda: nbt.add_(1))
        logits = ops.norm_head_fwd(z, stats, Nb, V, G, hw, hb, gamma, beta, True, chan_scale)
        if self.recording:
            def bwd():
                gl = self._take(logits)
                dg = self.g[prefix + ".weight"] if gamma is not None else None
                db = self.g[prefix + ".bias"] if beta is not None else None
                gz, pend = ops.norm_head_bwd(z, gl, stats, Nb, V, G, hw, gamma, beta, True, dg, db, chan_scale)
                gw, gb = self.g[head + ".weight"], self.g[head + ".bias"]
                if self.wgrad_stream is not None:      # the head's 2 x 16 weight gradient: a sum of per-chunk partials, off the chain
                    ops.fork(ops.cur_stream(), self.wgrad_stream)
                    with ops.on_stream(self.wgrad_stream, light=True):
                        ops.norm_head_dparams(pend, gw, gb)
                    pend[0].record_stream(self.wgrad_stream)
                else:
                    ops.norm_head_dparams(pend, gw, gb)
                if self.on_param_grads is not None:
                    self.on_param_grads(head + ".weight")
                    if gamma is not None:
                        self.on_param_grads(prefix + ".weight")
                self._give(z, gz)
            self.tape.append(bwd)
        return logits

    # ---------------------------------------------------------------- dropout sites
    def _channel_scale(self, B, C, key, p, site, device):
        """nn.Dropout3d(p) as a per-(sample, channel) factor keep/(1-p); applied INSIDE the preceding norm kernel."""
        d = self.dropout
        if d.mode == "off":
            return None
        if d.mode == "mask":
            m = d.masks.get(key)
            if m is None:
                return None
            return (m.to(torch.float32) / (1.0 - p)).contiguous().reshape(-1)   # tiny (B*C) host-provided mask
        return ops.channel_mask_philox(B * C, p, d.seed, d.offset + site * (1 << 20), device)

    def _drop_elements(self, x, key, p, site):
        d = self.dropout
        if d.mode == "off":
            return x
        if d.mode == "mask":
            m = d.masks.get(key)
            if m is None:
                return x
            f = lambda t: ops.mul_mask(t, m, 1.0 / (1.0 - p))   # noqa: E731
        else:
            off = d.offset + site * (1 << 40)
            f = lambda t: ops.dropout_philox(t, p, d.seed, off)  # noqa: E731
        y = f(x)
        if self.recording:
            def bwd():
                self._give(x, f(self._take(y)))
            self.tape.append(bwd)
        return y

    # ---------------------------------------------------------------- feature head
    def _head(self, center, training):
        """UNet3D_contrastive.py:261-267, 308-310: trilinear x scale (align_corners=True) -> 1x1 -> BN -> ReLU -> 1x1 -> BN."""
        B, d, h, w, C = center.shape
        s = self.scale_factor
        c = ops.trilinear_fwd(center, (d * s, h * s, w * s), True)
        if self.recording:
            def bwd():
                self._give(center, ops.trilinear_bwd(self._take(c), center.shape, True))
            self.tape.append(bwd)
        hdn = self._conv("projection.0", c, "1x1")
        hdn = self._norm("projection.1", hdn, "bn", relu=True, training=training)
        out = self._conv("projection.3", hdn, "1x1")
        return self._norm("projection.4", out, "bn", relu=False, training=training)

    def _mark_ready(self):
        if self.feat_stream is None:
            return None
        ev = ops.Event()
        ev.record(ops.cur_stream())
        return ev

    def _head_branch(self, center, ready, training):
        """The projection head depends on the bottleneck only: with feat_stream set it is enqueued there, behind the event
        recorded when the bottleneck was complete, and overlaps the decoder; its tape entries are replayed on that stream too."""
        h0 = len(self.tape)
        if self.feat_stream is None:
            feats = self._head(center, training)
        else:
            with ops.on_stream(self.feat_stream):
                ready.wait(self.feat_stream)
                feats = self._head(center, training)
            center.record_stream(self.feat_stream)
        self._head_range = (h0, len(self.tape))
        return feats

    # ---------------------------------------------------------------- V-Net
    def _vnet(self, x, training):
        nk = {"groupnorm": "gn", "instancenorm": "in", "batchnorm": "bn", "none": "none"}[self.normalization]
        st = 2 if nk == "none" else 3       # nn.Sequential index step: conv, [norm,] relu

        def ngroups(t, conv):       # groups of the one-launch norm that may take over this conv's split-K finish (GN / IN only)
            return 16 if nk == "gn" else (self.p[conv + ".weight"].shape[0] if nk == "in" else 0)

        def block(name, t, n, first=False, drop=None):
            for i in range(n):
                t = self._conv(f"{name}.conv.{st * i}", t, "k3", need_gx=not (first and i == 0), norm_groups=ngroups(t, f"{name}.conv.{st * i}"))
                cs = None
                if drop is not None and i == n - 1:
                    cs = self._channel_scale(t.shape[0], t.shape[-1], drop[0], 0.5, drop[1], t.device)
                t = self._norm(f"{name}.conv.{st * i + 1}", t, nk, chan_scale=cs, training=training)
            return t

        def down(name, t):
            return self._norm(f"{name}.conv.1", self._conv(f"{name}.conv.0", t, "k2s2", norm_groups=ngroups(t, f"{name}.conv.0")), nk,
                              training=training)

        def up(name, t, skip):
            return self._norm(f"{name}.conv.1", self._conv(f"{name}.conv.0", t, "deconv"), nk, skip=skip, training=training)

        w1 = self.p["block_one.conv.0.weight"]
        if (self.fuse_first and self.dtype == torch.bfloat16 and nk != "none" and not (nk == "bn" and not training)
                and w1.shape[0] == 16 and w1.shape[1] == 1 and x.shape[0] <= 16):
            x1 = self._first_block("block_one", x, nk, training)
        else:
            x1 = block("block_one", x, 1, first=True)
        hook = self.stage_hook if self.stage_hook is not None else (lambda name: None)
        hook("x1")
        self.pack_ready()            # (repack with early="block_one.": every other layer's operands were packed on a helper stream)
        x2 = block("block_two", down("block_one_dw", x1), 2)
        hook("x2")
        x3 = block("block_three", down("block_two_dw", x2), 3)
        hook("x3")
        x4 = block("block_four", down("block_three_dw", x3), 3)
        hook("x4")
        x5 = block("block_five", down("block_four_dw", x4), 3, drop=("drop5", 0))   # + Dropout3d, VNet.py:195-196
        hook("x5")
        x5_ready = self._mark_ready()
        u = up("block_five_up", x5, x4)
        u = up("block_six_up", block("block_six", u, 3), x3)
        u = up("block_seven_up", block("block_seven", u, 3), x2)
        u = up("block_eight_up", block("block_eight", u, 2), x1)
        w9, wo = self.p["block_nine.conv.0.weight"], self.p["out_conv.weight"]
        head_ok = w9.shape[0] == 16 and wo.shape[0] == 2           # the fused kernels are written for 16 channels -> 2 classes
        bn_drop = nk == "bn" and self.dropout.mode != "off"        # per-sample Dropout3d masks under BatchNorm: unfused (see _norm)
        if self.fuse_head and nk != "none" and head_ok and not bn_drop:   # block_nine's norm + ReLU + Dropout3d + out_conv in one pass
            z9 = self._conv("block_nine.conv.0", u, "k3", norm_groups=ngroups(u, "block_nine.conv.0"))
            cs9 = self._channel_scale(z9.shape[0], z9.shape[-1], "drop9", 0.5, 1, z9.device)
            logits = self._norm_head("block_nine.conv.1", z9, nk, "out_conv", training=training, chan_scale=cs9)
        else:
            x9 = block("block_nine", u, 1, drop=("drop9", 1))                        # + Dropout3d, VNet.py:225-226
            logits = self._conv("out_conv", x9, "1x1", out_dtype=torch.float32)
        feats = self._head_branch(x5, x5_ready, training)
        return logits, feats, None

    # ---------------------------------------------------------------- U-Net
    def _unet(self, x, training, want_sdf):
        def uconv(prefix, t, first=False):
            c1, c2 = self.p[prefix + ".conv1.0.weight"].shape[0], self.p[prefix + ".conv2.0.weight"].shape[0]
            t = self._norm(None, self._conv(prefix + ".conv1.0", t, "k3", need_gx=not first, norm_groups=c1), "in")
            return self._norm(None, self._conv(prefix + ".conv2.0", t, "k3", norm_groups=c2), "in")

        def pool(t):
            y, idx = ops.maxpool2_fwd(t)
            if self.recording:
                def bwd():
                    self._give(t, ops.maxpool2_bwd(self._take(y), idx, t.shape))
                self.tape.append(bwd)
            return y

        def upcat(prefix, skip, low):
            B, D, H, W, C1 = skip.shape
            C2 = low.shape[-1]
            cat = torch.empty((B, D, H, W, C1 + C2), dtype=skip.dtype, device=skip.device)
            ops.copy_channels(skip, 0, cat, 0, C1)                       # torch.cat([skip, up], 1)  (networks/utils.py:276)
            ops.trilinear_fwd(low, (D, H, W), False, out=cat, coff=C1)   # nn.Upsample(2, 'trilinear')  (networks/utils.py:264)
            if self.recording:
                def bwd():
                    gc = self._take(cat)
                    gs = torch.empty(skip.shape, dtype=gc.dtype, device=gc.device)
                    ops.copy_channels(gc, 0, gs, 0, C1)
                    self._give(skip, gs)
                    self._give(low, ops.trilinear_bwd(gc, low.shape, False, coff=C1))
                self.tape.append(bwd)
            return uconv(prefix + ".conv", cat)

        c1 = uconv("conv1", x, first=True)
        c2 = uconv("conv2", pool(c1))
        c3 = uconv("conv3", pool(c2))
        c4 = uconv("conv4", pool(c3))
        center = self._drop_elements(uconv("center", pool(c4)), "drop_center", 0.3, 0)
        center_ready = self._mark_ready()
        u4 = upcat("up_concat4", c4, center)
        u3 = upcat("up_concat3", c3, u4)
        u2 = upcat("up_concat2", c2, u3)
        u1 = self._drop_elements(upcat("up_concat1", c1, u2), "drop_up1", 0.3, 1)
        logits = self._conv("out_conv2", u1, "1x1", out_dtype=torch.float32)
        feats = self._head_branch(center, center_ready, training)
        sdf = None
        if want_sdf:   # tanh(final(up1)) -- returned for API parity, discarded by the training step
            rec, self.recording = self.recording, False
            sdf = ops.tanh(self._conv("final", u1, "1x1", out_dtype=torch.float32))
            self.recording = rec
        return logits, feats, sdf

    # ---------------------------------------------------------------- public
    def forward(self, x, training=True, record=True, dropout: Optional[DropoutSpec] = None, update_bn=True,
                want_sdf=False):
        """x: (B, D, H, W, Cin) fp32 or ``dtype``.  Returns (logits fp32 (B,D,H,W,2), feats (B,d,h,w,256), sdf|None)."""
        assert x.dim() == 5 and x.is_contiguous()
        self.recording = bool(record)
        self.update_bn = update_bn
        self.dropout = dropout or DropoutSpec("off")
        self.tape, self.G = [], {}
        self._deferred = {}
        self._stat_parts = {}
        if self.use_acc and not self.acc_external:        # stand-alone engine (module route, tests): own arena, cleared here
            if self.acc_arena is None:
                self.acc_arena = torch.empty(self.ACC_DOUBLES, dtype=torch.float64, device=x.device)
            arena = self.acc_arena
            ops.rec(lambda: arena.zero_())
        self._acc_off = 0
        if x.dtype != self.dtype:
            x = ops.cast(x, self.dtype)
        if self.net_type == "vnet":
            logits, feats, sdf = self._vnet(x, training)
            if want_sdf:
                sdf = ops.tanh(logits)
        else:
            logits, feats, sdf = self._unet(x, training, want_sdf)
        self._out = (logits, feats)
        return logits, feats, sdf

    def backward(self, g_logits: Optional[torch.Tensor], g_feats: Optional[torch.Tensor]):
        """Seeds the two outputs' gradients and replays the tape.  Parameter gradients are WRITTEN (not
        accumulated) into ``grads``; parameters that receive no gradient (UNet3D ``final.*``) are left untouched."""
        logits, feats = self._out
        if g_logits is None:
            g_logits = torch.zeros_like(logits)
        if g_feats is None:
            g_feats = torch.zeros_like(feats)
        self.Gs = {}
        self._put(logits, g_logits)
        if self.feat_stream is not None:
            with ops.on_stream(self.feat_stream):
                self._put(feats, g_feats)                    # produced by the caller on the feature stream
        else:
            self._put(feats, g_feats)
        h0, h1 = self._head_range
        for i in range(len(self.tape) - 1, -1, -1):
            if self.feat_stream is not None and h0 <= i < h1:
                with ops.on_stream(self.feat_stream):   # feature branch: concurrent with the decoder's backward
                    self.tape[i]()
            else:
                self.tape[i]()
        cs = ops.cur_stream()
        if self._pending_dparams:                             # (a norm whose convolution does not run its weight gradient on the side stream)
            ops.fork(cs, self.wgrad_stream)
            with ops.on_stream(self.wgrad_stream, light=True):
                self._flush_dparams()
        if self.mark is not None:                             # tools/timeline.py
            self.mark("bwd_chain_end", cs)
            if self.feat_stream is not None:
                self.mark("feat_bwd_end", self.feat_stream)
            if self.wgrad_stream is not None:
                self.mark("wgrad_end", self.wgrad_stream)
        if self.feat_stream is not None:
            ops.fork(self.feat_stream, cs)
        if self.wgrad_stream is not None:
            ops.fork(self.wgrad_stream, cs)                   # all parameter gradients are complete behind this point
            if self.wgrad_stream2 is not None:
                ops.fork(self.wgrad_stream2, cs)
            if self.wgrad_stream3 is not None and self.wgrad_stream3 is not self.feat_stream:
                ops.fork(self.wgrad_stream3, cs)
        self.tape, self.G = [], {}
