"""The DyCON training step on the MI355X -- code/train_DyCON_BraTS19.py:298-372 as one launch sequence.

    noise -> student fwd -> teacher fwd -> fused voxel losses + FeCL -> backward -> (RCCL all-reduce)
          -> clip + SGD + EMA (one fused pass over flat arenas) -> weight repack

Everything between the input tensors and the updated arenas is HIP kernels from libdycon_hip.so
enqueued on the current stream; the host only computes the reference's scalar schedules
(adaptive beta, consistency ramp, FeCL threshold ramp, EMA alpha, poly LR).  The only device->host
read is the reference's own NaN/Inf guard (:360-362), one int per step (``strict_nan_check``), copied
out right after the loss and waited for at the end of the step (the backward is already queued).

Multi-GPU (one process per GPU, ``torch.distributed`` backend "nccl" == RCCL over xGMI): the batch
is sharded [labelled | unlabelled] per rank; one all-reduce of the flat gradient arena after the
backward, plus two tiny all-reduces (16 + 4 doubles) that keep the reference's batch-global
semantics of the Dice ratio and of the FeCL cross-branch ratio.
"""
from __future__ import annotations

import contextlib
import math
import os
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Optional

import torch

from . import _lib, ops
from .engine import ABLATE, ABLATE_N, DropoutSpec, Engine, param_spec, projection_buffers
from .networks.net_factory_3d import net_factory_3d
from .utils import ramps
from .utils.dycon_losses import adaptive_beta, sigmoid_rampup


@dataclass
class TrainConfig:
    """Flag names and defaults of code/train_DyCON_BraTS19.py:26-67."""
    model: str = "vnet"                 # "vnet" | "unet_3D"
    normalization: str = "groupnorm"    # V-Net only
    max_iterations: int = 20000
    batch_size: int = 8                 # per process
    labeled_bs: int = 4                 # per process
    base_lr: float = 0.01
    labelnum: int = 25
    ema_decay: float = 0.99
    consistency: float = 0.1
    consistency_type: str = "mse"       # "mse" | "kl"
    consistency_rampup: float = 200.0
    gamma: float = 2.0
    beta_min: float = 0.5
    beta_max: float = 5.0
    s_beta: Optional[float] = None
    temp: float = 0.6
    l_weight: float = 1.0
    u_weight: float = 0.5
    use_focal: int = 1
    use_teacher_loss: int = 1
    feature_scaler: int = 2
    seed: int = 1337
    rampup_epochs: int = 1500           # FeCLoss(rampup_epochs=1500), train_DyCON_BraTS19.py:287-288
    momentum: float = 0.9
    weight_decay: float = 1e-4
    max_grad_norm: float = 1.0
    dice_variant: str = "fg"            # "fg": losses.dice_loss (BraTS/Pancreas) | "multiclass": losses.DiceLoss (ISLES)
    teacher_mode: str = "train"         # "train" (BraTS/Pancreas, :264) | "eval" (ISLES, train_DyCON_ISLES22.py:114)
    poly_lr: bool = False               # ISLES: lr = base*(1 - it/max)^0.9 after each step (train_DyCON_ISLES22.py:322-324)
    dtype: torch.dtype = torch.bfloat16  # activation storage of the HIP kernels (fp32 = parity mode)
    strict_nan_check: bool = True       # read the NaN/Inf flag every step, as the reference does
    global_batch_losses: bool = True    # DDP: all-reduce the Dice / FeCL-cross sums (exact global-batch semantics)
    overlap_teacher: bool = True        # teacher forward on a second HIP stream, concurrent with the student forward
    teacher_after: Optional[str] = None # V-Net: start the teacher forward only when the student's encoder tensor x1..x5 is enqueued (phase shift
                                        # of the two forwards: the teacher's HBM-bound top level beside the student's latency-bound deep levels)
    teacher_priority: int = -1          # HIP priority of that stream (-1 high, 0 normal, 1 low): high measured 0.06 ms/step faster
    overlap_wgrad: bool = True          # weight-gradient launches of the backward on a second HIP stream (off the dgrad chain)
    wgrad_two_streams: bool = True      # ... alternating with the teacher's stream, which is idle during the backward
    overlap_features: bool = True       # feature branch (projection head, embeddings, FeCL forward + backward) on its own stream
    split_repack: bool = True           # the student's weight repack off the head of the dependent chain (Engine.repack)
    conv_stats: bool = False            # 48^3 level: norm statistics taken by the persistent convolution (measured neutral: off)
    conv_stats96: bool = False          # 96^3 level (block_one, block_nine): the same.  Saves a 113 MB statistics pass per site, but the
                                        # epilogue costs the convolutions as much (conv_k3_c1 38 -> 78 us, conv_k3_p16 74 -> 90 us per launch
                                        # against 4 x 20 us of statistics launches): step unchanged over three A/B pairs -- off
    one_pass_first: bool = True         # block_one's backward as one pass over (x, z, gy) (Engine.one_pass_first)
    fuse_first: bool = True             # V-Net: block_one's norm backward inside the first layer's weight gradient (Engine._first_block)
    fuse_head: bool = True              # V-Net: out_conv fused into block_nine's normalisation passes (Engine._norm_head)
    fuse_finish: bool = False           # small levels: split-K finish of a convolution done by the one-launch norm that follows (Engine.fuse_finish)
    norm_accumulators: bool = False     # two-launch norms through double-atomic accumulators (measured slower: DESIGN.md section 9)
    ddp_force: bool = False             # run the data-parallel exchange with a ONE-rank process group as well (RCCL test on one GPU)
    replay: bool = True                 # after two eager steps of a given input signature, record the step's launch list once and
                                        # re-issue it with patched scalars (the step is host-enqueue-bound: see DyconTrainer.step);
                                        # single-process and data-parallel runs alike


def bucket_cuts(head_offsets, n_sgd, nb=4, tail_frac=0.06):
    """Arena offsets at which the flat gradient buffer [0, n_sgd) is cut into at most `nb` all-reduce buckets; every cut is one of
    `head_offsets` (ascending offsets of the convolution weights, whose gradients the backward announces).

    The backward produces gradients in REVERSE registration order, so a bucket [lo, hi) is complete when the parameter at `lo` is,
    and the bucket starting at 0 completes last, at the very end of the backward, with nothing left to hide its transfer behind.
    That bucket is kept small -- the leading parameters up to `tail_frac` of the arena (the V-Net's three top encoder levels: 2 MB
    of 39 MB, where an equal-size cut gives it 15 MB) -- and the rest is cut into equal parts."""
    tail = [o for o in head_offsets if 0 < o <= tail_frac * n_sgd]
    cuts = [0] + ([max(tail)] if tail and nb > 2 else [])
    base, first = cuts[-1], len(cuts)
    target = (n_sgd - base) / (nb - first + 1)
    for o in head_offsets:
        if len(cuts) < nb and o > cuts[-1] and o >= base + target * (len(cuts) - first + 1):
            cuts.append(o)
    return cuts + [n_sgd]


class DyconTrainer:
    def __init__(self, cfg: TrainConfig, device="cuda:0", student_init: Optional[Dict[str, torch.Tensor]] = None,
                 teacher_init: Optional[Dict[str, torch.Tensor]] = None, process_group=None):
        self.cfg, self.device = cfg, torch.device(device)
        self.pg = process_group
        self.world = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        self.rank = torch.distributed.get_rank(process_group) if process_group is not None else 0
        # the data-parallel exchange (bucketed gradient all-reduce, 16 + 4-double loss exchange); ddp_force runs it with ONE rank too,
        # where every collective is the identity (tests: the RCCL calls inside the recorded step, on the single GPU of a test box)
        self.ddp = self.world > 1 or (cfg.ddp_force and process_group is not None)
        spec = param_spec(cfg.model, 1, 2, cfg.normalization)
        # parameters the loss never reaches keep grad=None in the reference and are skipped by
        # clip_grad_norm_/SGD (weight decay included): put them behind the SGD range of the arena
        nograd = [k for k in spec if k.startswith("final.")]
        order = [k for k in spec if k not in nograd] + nograd
        sizes = {k: int(math.prod(spec[k])) for k in spec}
        pad = lambda n: (n + 3) // 4 * 4      # keep every view 16-byte aligned  # noqa: E731
        offs, off = {}, 0
        for k in order:
            if k == (nograd[0] if nograd else None):
                self.n_sgd = off
            offs[k] = off
            off += pad(sizes[k])
        if not nograd:
            self.n_sgd = off
        self.n_all = off
        f32 = dict(dtype=torch.float32, device=self.device)
        self.flat_p = torch.zeros(self.n_all, **f32)
        self.flat_t = torch.zeros(self.n_all, **f32)
        self.flat_g = torch.zeros(self.n_all, **f32)
        self.flat_m = torch.zeros(self.n_all, **f32)
        view = lambda flat, k: flat[offs[k]:offs[k] + sizes[k]].view(spec[k])  # noqa: E731
        self.names = list(spec)
        self.p = OrderedDict((k, view(self.flat_p, k)) for k in spec)
        self.t = OrderedDict((k, view(self.flat_t, k)) for k in spec)
        self.g = OrderedDict((k, view(self.flat_g, k)) for k in spec)

        # nn.Module shells (state_dict / eval-mode inference); their parameters alias the arenas
        self.model = net_factory_3d(cfg.model, 1, 2, cfg.feature_scaler, dtype=cfg.dtype, normalization=cfg.normalization)
        self.ema_model = net_factory_3d(cfg.model, 1, 2, cfg.feature_scaler, dtype=cfg.dtype, normalization=cfg.normalization)
        g0 = torch.Generator().manual_seed(cfg.seed)
        for mod, init, arena in ((self.model, student_init, self.p), (self.ema_model, teacher_init, self.t)):
            if init is None:     # two independently initialised nets, as create_model() x2 (:212-230); same on every rank
                init = {k: mod._init_tensor(k, spec[k], g0) for k in spec}
            mod.to(self.device)
            for k, prm in mod.named_parameters():
                arena[k].copy_(init[k].to(self.device, torch.float32))
                prm.data = arena[k]
                if mod is self.ema_model:
                    prm.requires_grad_(False)   # `param.detach_()` in create_model(ema=True), :220-224
            for k, b in mod.named_buffers():
                if init is not None and k in init:
                    b.copy_(init[k].to(self.device))
        self.s_buf = dict(self.model.named_buffers())
        self.t_buf = dict(self.ema_model.named_buffers())
        self._fast_math = cfg.dtype == torch.bfloat16     # voxel-loss exp / log by the hardware sequences; fp32 = parity mode, libm
        self.s_eng = Engine(cfg.model, self.p, self.g, self.s_buf, cfg.dtype, cfg.feature_scaler, cfg.normalization)
        self.t_eng = Engine(cfg.model, self.t, None, self.t_buf, cfg.dtype, cfg.feature_scaler, cfg.normalization)
        self.s_eng.fuse_finish = self.t_eng.fuse_finish = cfg.fuse_finish
        self.s_eng.fuse_head = self.t_eng.fuse_head = cfg.fuse_head
        self.s_eng.fuse_first = self.t_eng.fuse_first = cfg.fuse_first
        self.s_eng.one_pass_first = self.t_eng.one_pass_first = cfg.one_pass_first
        self.s_eng.conv_stats = self.t_eng.conv_stats = cfg.conv_stats
        self.s_eng.conv_stats96 = self.t_eng.conv_stats96 = cfg.conv_stats96
        # accumulator form of the norms (engine.use_acc; measured slower, off by default): one arena of zeroed doubles per step,
        # shared by both nets and cleared by ONE launch at the start of the step, before the teacher stream forks
        self.acc_arena = None
        if cfg.norm_accumulators:
            self.acc_arena = torch.zeros(2 * Engine.ACC_DOUBLES, dtype=torch.float64, device=self.device)
            for eng, half in ((self.s_eng, 0), (self.t_eng, 1)):
                eng.acc_arena = self.acc_arena[half * Engine.ACC_DOUBLES:(half + 1) * Engine.ACC_DOUBLES]
                eng.acc_external, eng.use_acc = True, True
        self.iter_num = 0
        self.lr = cfg.base_lr * (self.world if self.world > 1 else 1)   # LR x n_gpu, train_DyCON_BraTS19.py:108-110
        self.base_lr = self.lr
        self.iters_per_epoch = max(cfg.labelnum // max(cfg.labeled_bs * self.world, 1), 1)
        self.max_epoch = cfg.max_iterations // self.iters_per_epoch + 1
        self.sumsq = torch.zeros(1, dtype=torch.float64, device=self.device)
        self.flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.coef = torch.zeros(8, dtype=torch.float32, device=self.device)
        self.acc20 = torch.zeros(20, dtype=torch.float64, device=self.device)      # [0:16] voxel-loss sums, [16:20] FeCL accumulators
        self.skipped_steps = 0
        # the NaN/Inf flag is final once step_loss has run: it is copied to pinned host memory right there and read at the end
        # of the step, when the copy has long completed -- the host never waits for the backward, the GPU never runs dry
        self.flag_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.flag_evt = torch.cuda.Event()
        # HIP priorities of the side streams (teacher, weight gradients, features).  Measured (profiles/r03_stream_scheduling.txt, pairs on
        # one box): the TEACHER's stream at high priority takes 0.06 ms off the step (4.90 -> 4.84 ms) -- its forward ends earlier and
        # leaves the student's large levels alone sooner; high priority for the weight-gradient or feature stream returns nothing, low
        # priority for the side streams costs 0.03 ms (and round 2 had the MAIN stream at high priority at 8.2 ms against 5.8).
        # DYCON_SIDE_PRIORITY="t,w,f" overrides (diagnostic).
        prios = [cfg.teacher_priority, 0, 0]
        if os.environ.get("DYCON_SIDE_PRIORITY"):
            prios = [int(v) for v in os.environ["DYCON_SIDE_PRIORITY"].split(",")]
            prios = (prios * 3)[:3] if len(prios) == 1 else (prios + [0, 0, 0])[:3]

        def side_stream(prio):
            if prio == 0:
                return torch.cuda.Stream(device=self.device)
            import ctypes
            h = _lib.hip()
            sp = ctypes.c_void_p()
            h.hipStreamCreateWithPriority.restype = ctypes.c_int
            h.hipStreamCreateWithPriority.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint, ctypes.c_int]
            with torch.cuda.device(self.device):
                rc = h.hipStreamCreateWithPriority(ctypes.byref(sp), 1, prio)      # hipStreamNonBlocking
            if rc:
                raise RuntimeError(f"hipStreamCreateWithPriority({prio}) failed: {rc}")
            return torch.cuda.ExternalStream(sp.value, device=self.device)
        self.side = side_stream(prios[0])
        if cfg.overlap_wgrad:
            self.s_eng.wgrad_stream = side_stream(prios[1])
        # The feature branch -- projection head -> normalised embeddings -> FeCL, forward and backward -- meets the segmentation
        # branch only at the bottleneck tensor and in the scalar loss: it runs on a third stream, beside the decoder.
        # the teacher's stream is idle during the backward: the convolutions' weight gradients alternate between it and the weight-gradient
        # stream (4.781 -> 4.755 ms/step over three pairs, profiles/r03_stream_scheduling.txt); still four streams in all
        if cfg.overlap_wgrad and cfg.overlap_teacher and cfg.wgrad_two_streams and os.environ.get("DYCON_WGRAD_TWO_STREAMS", "1") == "1":
            self.s_eng.wgrad_stream2 = self.side
        # the teacher's projection head depends on its bottleneck only: it runs beside the teacher's decoder on the weight-gradient
        # stream (idle during the forwards), so the teacher's chain ends one head earlier
        if (cfg.overlap_teacher and cfg.overlap_features and self.s_eng.wgrad_stream is not None
                and os.environ.get("DYCON_TEACHER_HEAD_STREAM", "1") == "1"):
            self.t_eng.feat_stream = self.s_eng.wgrad_stream
        self.feat = side_stream(prios[2]) if cfg.overlap_features else None
        # HIP multiplexes a process's streams onto 4 hardware queues.  The data-parallel run adds torch's collective stream: with
        # five or more busy streams two of them SHARE a queue and serialise (profiles/r03_ddp_one_rank_trace.txt: the teacher forward and
        # the weight-gradient launches, +0.28 ms/step), and raising GPU_MAX_HW_QUEUES to 8 oversubscribes the queues (7.4 ms/step).
        # So a data-parallel rank runs the feature branch on the teacher's stream -- it needs the teacher's features anyway and that
        # stream is idle from the end of the teacher forward on -- and issues its collectives from the weight-gradient stream:
        # main, teacher + features, weight gradients, RCCL = four.
        if self.ddp and self.feat is not None and cfg.overlap_teacher:
            self.feat = self.side
        self.s_eng.feat_stream = self.feat
        if self.s_eng.wgrad_stream2 is not None and self.feat is not None and os.environ.get("DYCON_WGRAD_THREE_STREAMS") == "1":
            self.s_eng.wgrad_stream3 = self.feat        # (diagnostic)
        self.marks = None            # see _mark
        self._rp = None              # recorded step: dict(sig, rec, by_name, it, out, vol, lab)
        self._eager_seen = {}
        # DDP gradient buckets: contiguous arena ranges cut at parameter boundaries.  The backward writes gradients in reverse
        # registration order, so a bucket is complete when its FIRST parameter's gradient has been enqueued; its all-reduce is
        # issued right then and overlaps the rest of the backward (xGMI ring; `bucket_cuts`: V-Net 9.4 + 14.2 + 13.5 MB hidden behind the
        # backward and 1.9 MB exposed after it, instead of one 39 MB transfer at the end).
        self.buckets = []
        if self.ddp:
            heads = [k for k in order if offs[k] < self.n_sgd and len(spec[k]) == 5]   # conv weights: notified by the backward
            cuts = bucket_cuts([offs[k] for k in heads], self.n_sgd, nb=4,
                               tail_frac=float(os.environ.get("DYCON_DDP_TAIL_FRAC", "0.06")))
            by_off = {offs[k]: k for k in heads}
            self.buckets = [(by_off[lo], lo, hi) for lo, hi in zip(cuts[:-1], cuts[1:])]
            self._bucket_of = {name: (lo, hi) for name, lo, hi in self.buckets}
            self._pending = []
            self.s_eng.on_param_grads = self._on_param_grads

    def _mark(self, tag, stream=None):
        """tools/timeline.py: with `self.marks` set to a dict, record a HIP event at this point of the step on `stream` (default: the
        launch stream); the marks are part of the recorded step, so replayed steps are timed undisturbed."""
        if self.marks is None:
            return
        ev = self.marks.setdefault(tag, torch.cuda.Event(enable_timing=True))
        st = stream if stream is not None else ops.cur_stream()
        ops.rec(lambda: ev.record(st))

    def _on_param_grads(self, name):
        rng = self._bucket_of.get(name)
        if rng is not None:
            # The bucket holds gradients written on up to three streams: conv weights / biases (weight-gradient stream), norm affine
            # parameters (main) and the projection head's parameters (feature stream, whose backward was enqueued first).  The
            # collective is issued from the WEIGHT-GRADIENT stream, which first waits for the other two: that stream trails the chain
            # it is forked from, so the waits are free, and the data-gradient chain on main never waits for anything in mid-backward.
            issuer = self.s_eng.wgrad_stream if self.s_eng.wgrad_stream is not None else self._main
            others = []
            for o in (self._main, self.feat, self.s_eng.wgrad_stream, self.s_eng.wgrad_stream2):
                if o is not None and o != issuer and o not in others:
                    others.append(o)
            evs = [torch.cuda.Event() for _ in others]
            view = self.flat_g[rng[0]:rng[1]]

            def issue():
                for o, ev in zip(others, evs):
                    ev.record(o)
                    issuer.wait_event(ev)
                with torch.cuda.stream(issuer):       # torch.distributed orders the collective after the current stream
                    self._pending.append(torch.distributed.all_reduce(view, group=self.pg, async_op=True))
            ops.rec(issue)

    # ------------------------------------------------------------------ schedules (host scalars)
    def epoch_of(self, it):
        return it // self.iters_per_epoch

    def beta_at(self, epoch):
        c = self.cfg
        return c.s_beta if c.s_beta is not None else adaptive_beta(epoch, self.max_epoch, c.beta_max, c.beta_min)

    def consistency_weight(self, it):
        return self.cfg.consistency * ramps.sigmoid_rampup(it // 150, self.cfg.consistency_rampup)

    # ------------------------------------------------------------------ the step
    def step(self, volume, label, noise=None, s_drop: Optional[DropoutSpec] = None, t_drop: Optional[DropoutSpec] = None,
             epoch: Optional[int] = None, beta: Optional[float] = None):
        """One DyCON iteration (see _step); the enclosing on_stream context caches the current-stream handle for all launches.

        Replay (cfg.replay): the step is a fixed launch sequence and, at ~500 launches, bound by the host's enqueue rate.  After
        two eager steps with the same input signature the third is RECORDED (every C-ABI call with its arguments, every stream /
        event operation, in order; all tensors of the step stay allocated) and later steps re-issue that list with only the
        schedule scalars (Philox offsets, beta, consistency weight, FeCL threshold, LR, EMA alpha) patched and the batch copied into
        the recorded input buffers.  Explicit randomness (parity tests), a different shape or kernel timing fall back to eager.
        The tensors in the returned dict are then the recorded step's buffers: valid until the next step."""
        c = self.cfg
        sig = None
        # With a process group the recorded list contains the step's torch.distributed calls as closures (the bucketed gradient
        # all-reduces issued from the backward, the two accumulator exchanges, the join): re-issuing them is the same sequence of
        # torch.distributed calls an eager step makes, on the same tensors (tests/test_ddp_gpu.py: replay == eager over 2 ranks).
        if c.replay and noise is None and s_drop is None and t_drop is None and ops.PROFILER is None:
            sig = (tuple(volume.shape), volume.dtype, tuple(label.shape), label.dtype, torch.cuda.current_stream().cuda_stream)
        with ops.on_stream(None):
            if sig is None:
                return self._step(volume, label, noise, s_drop, t_drop, epoch, beta)
            if self._rp is not None and self._rp["sig"] == sig:
                return self._replay_step(volume, label, epoch, beta)
            n = self._eager_seen.get(sig, 0)
            self._eager_seen[sig] = n + 1
            if n < 2:
                return self._step(volume, label, None, None, None, epoch, beta)
            # record: the batch goes through static input buffers so that the recorded addresses stay valid
            vol = torch.empty_like(volume, memory_format=torch.contiguous_format)
            lab = torch.empty_like(label, memory_format=torch.contiguous_format)
            vol.copy_(volume)
            lab.copy_(label)
            rec = _lib.Recorder()
            it0 = self.iter_num
            _lib.RECORDER = rec
            try:
                out = self._step(vol, lab, None, None, None, epoch, beta)
            finally:
                _lib.RECORDER = None
            by_name = {}
            for e in rec.entries:
                if type(e) is list:
                    by_name.setdefault(e[2], []).append(e)
            base = {id(e): list(e[1]) for es in by_name.values() for e in es}
            self._rp = dict(sig=sig, rec=rec, by_name=by_name, base=base, it=it0, out=out, vol=vol, lab=lab)
            return out

    # Philox offsets are affine in the iteration counter: argument index and stride per entry point
    _PHILOX = {"dycon_add_noise": (8, 1 << 32), "dycon_channel_mask_philox": (4, 2 << 42), "dycon_dropout_philox": (6, 2 << 42)}

    def _replay_step(self, volume, label, epoch, beta):
        c, rp = self.cfg, self._rp
        it = self.iter_num
        epoch = self.epoch_of(it) if epoch is None else epoch
        beta = self.beta_at(epoch) if beta is None else beta
        cw = self.consistency_weight(it)
        thr = sigmoid_rampup(epoch, c.rampup_epochs, 0.3, 0.5)
        alpha = min(1 - 1 / (it + 1), c.ema_decay)
        by, base = rp["by_name"], rp["base"]
        for name, (idx, stride) in self._PHILOX.items():
            for e in by.get(name, ()):
                e[1][idx] = base[id(e)][idx] + (it - rp["it"]) * stride
        for name, idx in (("dycon_seg_losses_fwd", 7), ("dycon_seg_losses_bwd", 7), ("dycon_step_losses", 5)):
            for e in by.get(name, ()):
                e[1][idx] = float(beta)
        for name in ("dycon_fecl_fwd", "dycon_fecl_bwd"):
            for e in by.get(name, ()):
                e[1][11] = thr
        for e in by.get("dycon_step_losses", ()):
            e[1][10] = float(cw)
        for e in by.get("dycon_set_scalars", ()):           # coef = (l_w, l_w*(1-dk)*gw, l_w*dk*gw, cw, u_w, u_w): only cw moves
            e[1][5] = float(cw)
        for e in by.get("dycon_sgd_ema", ()):
            e[1][9], e[1][12] = float(self.lr), float(alpha)
        rp["vol"].copy_(volume)
        rp["lab"].copy_(label)
        rp["rec"].replay()
        for e in (self.s_eng, self.t_eng, self.model, self.ema_model):
            e.params_changed()
        skipped = False
        if c.strict_nan_check:
            self.flag_evt.synchronize()
            skipped = bool(self.flag_host[0])
        if skipped:
            self.skipped_steps += 1
        else:
            if c.poly_lr:
                self.lr = self.base_lr * (1.0 - it / c.max_iterations) ** 0.9
            self.iter_num += 1
        out = dict(rp["out"])
        out.update(cons_weight=cw, beta=beta, skipped=skipped)
        return out

    def _step(self, volume, label, noise=None, s_drop: Optional[DropoutSpec] = None, t_drop: Optional[DropoutSpec] = None,
              epoch: Optional[int] = None, beta: Optional[float] = None):
        """volume (B,1,D,H,W) fp32, label (B,D,H,W) int64|uint8, both on the device.  The first
        ``labeled_bs`` samples are the labelled ones (TwoStreamBatchSampler order, dataloaders/brats19.py:309-314).

        noise / s_drop / t_drop: explicit randomness for parity tests; default = on-device Philox."""
        c = self.cfg
        B, LB = volume.shape[0], c.labeled_bs
        D, H, W = volume.shape[2:]
        V = D * H * W
        it = self.iter_num
        epoch = self.epoch_of(it) if epoch is None else epoch
        beta = self.beta_at(epoch) if beta is None else beta
        cw = self.consistency_weight(it)
        thr = sigmoid_rampup(epoch, c.rampup_epochs, 0.3, 0.5)
        seed = (c.seed * 1000003 + self.rank) & 0x7FFFFFFFFFFFFFFF
        if s_drop is None:
            s_drop = DropoutSpec("philox", seed=seed, offset=(2 * it + 0) << 42)
        if t_drop is None:
            t_drop = DropoutSpec("philox", seed=seed, offset=(2 * it + 1) << 42) if c.teacher_mode == "train" else DropoutSpec("off")

        self._mark("step_begin")
        # coef = d total / d (ce, dice_fg, dice_mc, cons, uncl | fecl): host-known, so it is written at the head of the step -- the
        # feature branch's backward (FeCL gradient, embeddings, projection head) then depends on nothing but its own forward and MAY
        # start as soon as that is done, beside the student's decoder, instead of behind the scalar end of the loss on the main stream
        # (DYCON_FEAT_BWD_EARLY=1).  Measured: 15.07 -> 15.00 ms at 112 x 112 x 80, where that branch is the step's critical path, but
        # 4.45 -> 4.48 ms on the headline step (it then shares the CUs with the student's HBM-bound top level): off by default.
        # Means over equal shards (CE, cons, UnCL, FeCL student part) become global through the 1/world arena average;
        # the two global-ratio terms are differentiated w.r.t. LOCAL voxels and must be SUMMED over ranks -> x world.
        glob = self.ddp and c.global_batch_losses
        gw = self.world if glob else 1
        dice_kind = 0 if c.dice_variant == "fg" else 1
        ops.set_scalars(self.coef, [c.l_weight, c.l_weight * (1 - dice_kind) * gw, c.l_weight * dice_kind * gw, cw,
                                    c.u_weight, c.u_weight])
        ops.rec(lambda: self.sumsq.zero_())          # (its first use is mid-backward at the earliest)
        if self.acc_arena is not None:
            arena = self.acc_arena
            ops.rec(lambda: arena.zero_())
        # all weight packs of the step (student fwd + dgrad, teacher fwd): one launch per net; the student's is split so that only
        # block_one's operands are packed in front of the first convolution, the rest on the (idle) weight-gradient stream
        if c.split_repack and self.s_eng.wgrad_stream is not None:
            self.s_eng.repack(early="block_one.", helper=self.s_eng.wgrad_stream)
        else:
            self.s_eng.repack()
        if not c.overlap_teacher:
            self.t_eng.repack()
        x = volume.reshape(B, D, H, W, 1) if volume.is_contiguous() else volume.contiguous().reshape(B, D, H, W, 1)
        x_t = ops.add_noise(x, None if noise is None else noise.contiguous(), 0.1, 0.2, seed ^ 0x5DEECE66D, it << 32)  # :301-302

        # The teacher forward (:305-306) is independent of the student forward (:304): it runs on a second HIP stream so the
        # small, launch-latency-bound kernels of the deep levels (6^3, 12^3: 50-100 workgroups on 256 CUs) of the two nets overlap.
        t_train = c.teacher_mode == "train"
        main = self._main = ops.cur_stream()
        if "teacher" in ABLATE:      # tools/ablate.py (timing experiment only)
            s_logits, s_feat, _ = self.s_eng.forward(x, training=True, record=True, dropout=s_drop, update_bn=True)
            t_logits, t_feat = s_logits, s_feat
        elif c.overlap_teacher and c.teacher_after and c.model == "vnet":
            # phase-shifted forwards: the student's is enqueued first and records an event when the chosen encoder tensor is enqueued;
            # the teacher's weight packs start at once, its forward behind that event
            side = self.side
            ops.fork(main, side)
            with ops.on_stream(self.side):
                self.t_eng.repack()
            gate = ops.Event()
            self.s_eng.stage_hook = lambda name: gate.record(main) if name == c.teacher_after else None
            s_logits, s_feat, _ = self.s_eng.forward(x, training=True, record=True, dropout=s_drop, update_bn=True)   # :304
            self.s_eng.stage_hook = None
            with ops.on_stream(self.side):
                gate.wait(side)
                t_logits, t_feat, _ = self.t_eng.forward(x_t, training=t_train, record=False, dropout=t_drop, update_bn=t_train)
                self._mark("teacher_fwd_end")
            x_t.record_stream(self.side)
        elif c.overlap_teacher:
            side = self.side
            ops.fork(main, side)
            head_start = os.environ.get("DYCON_STUDENT_AFTER")      # diagnostic: the student's forward waits for the teacher's x1..x5
            gate = ops.Event() if head_start else None
            with ops.on_stream(self.side):
                # the teacher's packs belong to its stream (the EMA update that changed them precedes the fork).  Splitting them like the
                # student's (block_one's operands first, the rest on an idle stream) measured neutral to +0.01 ms: off (DYCON_TEACHER_SPLIT_PACK)
                if c.split_repack and self.feat is not None and self.feat is not self.side and os.environ.get("DYCON_TEACHER_SPLIT_PACK", "0") == "1":
                    self.t_eng.repack(early="block_one.", helper=self.feat)
                else:
                    self.t_eng.repack()
                if gate is not None:
                    self.t_eng.stage_hook = lambda name: gate.record(side) if name == head_start else None
                t_logits, t_feat, _ = self.t_eng.forward(x_t, training=t_train, record=False, dropout=t_drop, update_bn=t_train)
                self.t_eng.stage_hook = None
                self._mark("teacher_fwd_end")
            if gate is not None:
                gate.wait(main)
            x_t.record_stream(self.side)
        if "teacher" not in ABLATE and not (c.overlap_teacher and c.teacher_after and c.model == "vnet"):
            s_logits, s_feat, _ = self.s_eng.forward(x, training=True, record=True, dropout=s_drop, update_bn=True)   # :304
        if "teacher" in ABLATE:
            pass
        elif c.overlap_teacher:
            ops.fork(side, main)
            t_logits.record_stream(main)
            t_feat.record_stream(main)
        else:
            t_logits, t_feat, _ = self.t_eng.forward(x_t, training=t_train, record=False, dropout=t_drop, update_bn=t_train)

        if "extra_launches" in ABLATE:     # tools/ablate.py: is the step bound by the dispatch rate?  N trivial launches on an idle stream
            if not hasattr(self, "_xs"):
                self._xs, self._xbuf = torch.cuda.Stream(device=self.device), torch.zeros(8, device=self.device)
            with ops.on_stream(self._xs, light=True):
                for _ in range(ABLATE_N):
                    ops.set_scalars(self._xbuf, [0.0])
        self._mark("student_fwd_end")
        # ---- losses (:308-357)
        world = self.world
        # the step's 16 + 4 loss accumulators live in ONE buffer: a data-parallel run exchanges them with one all-reduce
        sums, fo = self.acc20[:16], self.acc20[16:]
        ops.seg_losses_fwd(s_logits, t_logits, label, LB, beta, fast=self._fast_math, out=sums)
        fctx = (lambda: ops.on_stream(self.feat)) if self.feat is not None else contextlib.nullcontext
        with fctx():
            if self.feat is not None:                     # teacher features (the student's head was enqueued on self.feat)
                feat, src = self.feat, (self.side if c.overlap_teacher else main)
                if self.t_eng.feat_stream is not None and self.t_eng.feat_stream is not src:
                    # the teacher's projection head ran beside its decoder on another stream.  The embeddings and FeCL still wait for the
                    # END of the teacher's forward as well: starting them as soon as the head is done (DYCON_FECL_EARLY=1, i.e. while both
                    # decoders are in their latency-bound deep levels) measured 0.04 ms SLOWER than beside the student's HBM-bound top level
                    ops.fork(self.t_eng.feat_stream, feat)
                    if os.environ.get("DYCON_FECL_EARLY", "0") != "1":
                        ops.fork(src, feat)
                else:
                    ops.fork(src, feat)
                t_feat.record_stream(self.feat)
            s_emb, s_nrm = ops.l2norm_fwd(s_feat.reshape(B, -1, s_feat.shape[-1]))          # :316-319
            t_emb, _ = ops.l2norm_fwd(t_feat.reshape(B, -1, t_feat.shape[-1]))              # :321-323
            k = (D // s_feat.shape[1], H // s_feat.shape[2], W // s_feat.shape[3])
            mask = ops.mask_pool(label, k)                                                   # :326-330
            teacher_emb = t_emb if c.use_teacher_loss else None
            fargs = (s_emb, teacher_emb, mask, None, c.temp, c.gamma, bool(c.use_focal), thr)
            f_loss, fst = ops.fecl_fwd(*fargs, 1.0, out=fo)
            self._mark("fecl_fwd_end")
        if self.feat is not None:
            ops.fork(feat, main)       # the scalar loss (and the DDP exchange below) needs the FeCL sums
            for t in (t_feat, mask):
                t.record_stream(main)
        if glob:
            # Dice is a ratio of batch-GLOBAL sums (losses.py:11-14) and the FeCL cross branch a global sum over a
            # global count (dycon_losses.py:229): exchange the 16 + 4 accumulators (one collective), then finalise on every rank
            acc20 = self.acc20
            ops.rec(lambda: torch.distributed.all_reduce(acc20, group=self.pg))
        cons_kind = 0 if c.consistency_type == "mse" else 1
        # the scalar end of the loss forward (:355-362) in one launch: voxel-loss ratios, FeCL finalize, weighted total, NaN/Inf flag
        out = ops.step_losses(sums, fo, B * gw, LB * gw, V, beta, B * gw * s_emb.shape[1], 1.0, teacher_emb is not None,
                              c.l_weight, cw, c.u_weight, dice_kind, cons_kind, self.flag)
        if c.strict_nan_check:
            ops.rec(lambda: (self.flag_host.copy_(self.flag, non_blocking=True), self.flag_evt.record(main)))

        # ---- backward (:364-365); coef was written at the head of the step
        self._mark("loss_end")
        g_logits = ops.seg_losses_bwd(s_logits, t_logits, label, LB, beta, sums, self.coef, cons_kind, fast=self._fast_math)
        if self.feat is not None and (glob or os.environ.get("DYCON_FEAT_BWD_EARLY", "0") != "1"):
            ops.fork(main, feat)       # DDP: the all-reduced FeCL sums (cross-branch count) are exchanged on main
        with fctx():
            if "feat_bwd" in ABLATE:     # tools/ablate.py (timing only): the feature branch's loss backward switched off
                g_feat = torch.zeros_like(s_feat)
            else:
                g_emb = ops.fecl_bwd(*fargs, float(gw), fst, self.coef[5:6])
                g_feat = ops.l2norm_bwd(s_emb, s_nrm, g_emb).reshape(s_feat.shape)
        self.s_eng.backward(g_logits, g_feat)            # head entries replay on self.feat, joins at the bottleneck gradient

        # ---- all-reduce, clip, SGD, EMA (:368-372)
        if self.ddp:       # bucketed all-reduces were issued during the backward (see __init__); wait for them here
            def join():
                assert len(self._pending) == len(self.buckets), "a gradient bucket was never triggered"
                for h in self._pending:
                    h.wait()
                self._pending = []
            ops.rec(join)
        self._mark("bwd_joined")
        ops.sumsq(self.flat_g[: self.n_sgd], self.sumsq)
        alpha = min(1 - 1 / (it + 1), c.ema_decay)
        ops.sgd_ema(self.flat_p, self.flat_g, self.flat_m, self.flat_t, self.n_sgd, self.sumsq, c.max_grad_norm, 1.0 / world,
                    self.lr, c.momentum, c.weight_decay, alpha, self.flag)
        self._mark("step_end")
        self.s_eng.params_changed()
        self.t_eng.params_changed()
        self.model.params_changed()
        self.ema_model.params_changed()

        skipped = False
        if c.strict_nan_check:
            self.flag_evt.synchronize()          # the reference's own per-step check (:360); see __init__
            skipped = bool(self.flag_host[0])
        if skipped:
            self.skipped_steps += 1              # `continue`: no update happened, iter_num unchanged
        else:
            if c.poly_lr:
                self.lr = self.base_lr * (1.0 - it / c.max_iterations) ** 0.9
            self.iter_num += 1
        return {"loss": out[0], "ce": out[1], "dice": out[2], "cons": out[3], "fecl": out[4], "uncl": out[5],
                "cons_weight": cw, "beta": beta, "grad_sumsq": self.sumsq, "skipped": skipped,
                "s_logits": s_logits, "t_logits": t_logits, "s_feat": s_feat, "t_feat": t_feat, "mask": mask}

    # ------------------------------------------------------------------ checkpoints (reference contract)
    def state_dict(self):
        """Student weights under the reference's keys (what train_DyCON_BraTS19.py:411-418 saves)."""
        return OrderedDict((k, v.detach().clone()) for k, v in self.model.state_dict().items())

    def teacher_state_dict(self):
        return OrderedDict((k, v.detach().clone()) for k, v in self.ema_model.state_dict().items())

    def full_state(self):
        """Everything needed to resume (the reference cannot resume: SURVEY.md section 5)."""
        return {"student": self.state_dict(), "teacher": self.teacher_state_dict(), "momentum": self.flat_m.clone(),
                "iter_num": self.iter_num, "lr": self.lr, "skipped_steps": self.skipped_steps}

    def save_checkpoint(self, path, resume_sidecar=True):
        """`torch.save(student.state_dict(), path)` exactly as the reference does (train_DyCON_BraTS19.py:411-430: student only,
        reference keys -- loadable by code/test_BraTS19.py:62-63), plus `<path>.resume` with everything the reference drops
        (teacher, momentum, iteration counter, LR): SURVEY section 8f-3."""
        torch.save(OrderedDict((k, v.cpu()) for k, v in self.state_dict().items()), path)
        if resume_sidecar:
            st = self.full_state()
            st["student"] = None                                 # lives in `path`
            st["teacher"] = OrderedDict((k, v.cpu()) for k, v in st["teacher"].items())
            st["momentum"] = st["momentum"].cpu()
            torch.save(st, str(path) + ".resume")

    def load_checkpoint(self, path):
        """Load a reference-format student checkpoint; with a `.resume` sidecar next to it, continue the run bit for bit."""
        import os
        student = torch.load(path, map_location="cpu", weights_only=True)
        side = str(path) + ".resume"
        if os.path.exists(side):
            st = torch.load(side, map_location="cpu", weights_only=True)
            st["student"] = student
            self.load_full_state(st)
        else:                                                   # evaluation-style load: student weights only.  The teacher keeps
            self.model.load_state_dict(student)                 # its weights (as after create_model(ema=True), :229-230); with
            #                                                     iter_num = 0 the first EMA update has alpha = 0: teacher := student
            for e in (self.s_eng, self.t_eng):
                e.params_changed()
            self.model.params_changed()

    def load_full_state(self, st):
        self.model.load_state_dict(st["student"])
        self.ema_model.load_state_dict(st["teacher"])
        self.flat_m.copy_(st["momentum"].to(self.device))
        self.iter_num, self.lr, self.skipped_steps = st["iter_num"], st["lr"], st["skipped_steps"]
        for e in (self.s_eng, self.t_eng):
            e.params_changed()
        self.model.params_changed()
        self.ema_model.params_changed()
