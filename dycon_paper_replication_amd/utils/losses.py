"""Supervised / consistency losses of the step with the reference's names (code/utils/losses.py).

Two families:

* the reference's OWN callables with the reference's argument meaning -- ``dice_loss(score, target)``,
  ``softmax_mse_loss(a, b, sigmoid=False)`` (element-wise result), ``softmax_kl_loss``,
  ``DiceLoss(n)(inputs, target, weight=None, softmax=False)`` -- on small streaming HIP kernels
  (csrc/reflosses.hip), so the reference's training-loop body (train_DyCON_BraTS19.py:298-372) runs on this
  package with only its imports changed;
* the fast forms the fused trainer uses: ONE pass over the logits for every voxel loss
  (``fused_voxel_losses`` and the ``*_from_logits`` / ``*_mean`` wrappers; the 2-class softmax is part of the kernel).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from .._lib import View, call
from .dycon_losses import _ndhwc_logits

CE, DICE_FG, DICE_MC, CONS_MSE, CONS_KL, UNCL = range(6)


class _VoxelLossFunction(torch.autograd.Function):
    """vals[6] = ce, dice(class 1), dice(multi-class), consistency mse, consistency kl, uncl."""

    @staticmethod
    def forward(ctx, s_logits, t_logits, labels, labeled_bs, beta):
        s, t = _ndhwc_logits(s_logits), _ndhwc_logits(t_logits)
        B = s.shape[0]
        V = s.numel() // (2 * B)
        lab = labels.contiguous()
        sums = ops.seg_losses_fwd(s, t, lab, labeled_bs, beta)
        vals = ops.seg_losses_finalize(sums, B, labeled_bs, V, beta)
        ctx.save_for_backward(s, t, sums, lab)
        ctx.meta = (labeled_bs, beta)
        return vals[:6]

    @staticmethod
    def backward(ctx, g):
        s, t, sums, lab = ctx.saved_tensors
        LB, beta = ctx.meta
        g = g.float()
        # kernel coefficient slots: ce, dice_fg, dice_mc, consistency, uncl ; mse and kl need separate passes
        out = None
        for kind, cons_g in ((0, g[CONS_MSE:CONS_MSE + 1]), (1, g[CONS_KL:CONS_KL + 1])):
            coef = torch.zeros(5, dtype=torch.float32, device=s.device)
            if kind == 0:
                coef[0:3] = g[0:3]
                coef[4:5] = g[UNCL:UNCL + 1]
            coef[3:4] = cons_g
            part = ops.seg_losses_bwd(s, t, lab, LB, beta, sums, coef, kind)
            out = part if out is None else ops.add(out, part)
        return out.permute(0, 4, 1, 2, 3), None, None, None, None


def fused_voxel_losses(s_logits, t_logits, labels, labeled_bs, beta=1.0):
    """One pass over student/teacher logits (B,2,D,H,W) and labels (B,D,H,W): returns the 6 scalars
    (ce, dice_fg, dice_multiclass, cons_mse, cons_kl, uncl) as a differentiable tensor (train_DyCON_BraTS19.py:308-352)."""
    return _VoxelLossFunction.apply(s_logits, t_logits.detach(), labels, int(labeled_bs), float(beta))


def dice_loss_from_logits(logits, target):
    """losses.dice_loss(softmax(logits)[:,1], target) (losses.py:8-16) -- fused: takes the LOGITS."""
    lab = target.to(torch.uint8) if target.dtype == torch.bool else target
    return fused_voxel_losses(logits, logits, lab, logits.shape[0])[DICE_FG]


def cross_entropy_from_logits(logits, target):
    """F.cross_entropy(logits, target) for 2 classes (train_DyCON_BraTS19.py:313)."""
    return fused_voxel_losses(logits, logits, target, logits.shape[0])[CE]


def softmax_mse_loss_mean(input_probs_logits, target_probs_logits):
    """mean of losses.softmax_mse_loss(softmax(a), softmax(b)) as the step uses it (train_DyCON_BraTS19.py:352):
    pass the LOGITS; both softmaxes of the reference's double application are inside the kernel."""
    dummy = torch.empty(1, dtype=torch.uint8, device=input_probs_logits.device)
    return fused_voxel_losses(input_probs_logits, target_probs_logits, dummy, 0)[CONS_MSE]


def softmax_kl_loss_mean(input_probs_logits, target_probs_logits):
    """losses.softmax_kl_loss(softmax(a), softmax(b)) (losses.py:85-104) from LOGITS."""
    dummy = torch.empty(1, dtype=torch.uint8, device=input_probs_logits.device)
    return fused_voxel_losses(input_probs_logits, target_probs_logits, dummy, 0)[CONS_KL]


# ------------------------------------------------------------------ the reference's callables, reference semantics
_TKIND = {torch.float32: 0, torch.uint8: 1, torch.bool: 1, torch.int64: 2}


def _f32(t):
    if not t.is_cuda:
        raise RuntimeError("the loss kernels run on the MI355X only (no CPU fallback): pass CUDA tensors")
    return t if t.dtype == torch.float32 else t.float()


def _collapse(shape, strides):
    """one (count, element stride) for a run of dims visited in natural order, or None if they do not collapse"""
    dims = [(n, s) for n, s in zip(shape, strides) if n != 1]
    if not dims:
        return 1, 1
    for (_, s0), (n1, s1) in zip(dims[:-1], dims[1:]):
        if s0 != s1 * n1:
            return None
    cnt = 1
    for n, _ in dims:
        cnt *= n
    return cnt, dims[-1][1]


def _view_ncv(t):
    """(tensor, View, n, C, V) of a (n, C, *spatial) tensor; copies only when the spatial dims do not collapse to one stride."""
    sp = _collapse(t.shape[2:], t.stride()[2:])
    if sp is None:
        t = t.contiguous()
        sp = _collapse(t.shape[2:], t.stride()[2:])
    V, sv = sp
    return t, View(t.data_ptr(), t.stride(0), t.stride(1), sv), t.shape[0], t.shape[1], V


def _view_flat(t):
    """(tensor, View, numel) of any tensor visited in natural order as (1, 1, numel)."""
    fl = _collapse(t.shape, t.stride())
    if fl is None:
        t = t.contiguous()
        fl = _collapse(t.shape, t.stride())
    return t, View(t.data_ptr(), 0, 0, fl[1]), fl[0]


def _ref(v):
    import ctypes
    return ctypes.byref(v)


class _SoftmaxMSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, sigmoid):
        a, b = _f32(a), _f32(b)
        a, va, n, C, V = _view_ncv(a)
        b, vb, _, _, _ = _view_ncv(b)
        out = torch.empty_like(a)
        out, vo, _, _, _ = _view_ncv(out)
        call("dycon_softmax_mse_fwd", _ref(va), _ref(vb), _ref(vo), n, C, V, int(sigmoid), ops._s())
        ctx.save_for_backward(a, b)
        ctx.sigmoid = sigmoid
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g, vg, n, C, V = _view_ncv(_f32(g))
        grads = []
        for need, (x, y) in zip(ctx.needs_input_grad[:2], ((a, b), (b, a))):     # symmetric loss: swap for the second argument
            if not need:
                grads.append(None)
                continue
            _, vx, _, _, _ = _view_ncv(x)
            _, vy, _, _, _ = _view_ncv(y)
            gx = torch.empty_like(x)
            gx, vgx, _, _, _ = _view_ncv(gx)
            call("dycon_softmax_mse_bwd", _ref(vx), _ref(vy), _ref(vg), _ref(vgx), n, C, V, int(ctx.sigmoid), ops._s())
            grads.append(gx)
        return grads[0], grads[1], None


def softmax_mse_loss(input_logits, target_logits, sigmoid=False):
    """losses.py:65-82: (softmax(input, 1) - softmax(target, 1))**2, ELEMENT-WISE (the caller takes .mean(),
    train_DyCON_BraTS19.py:352).  The reference passes probabilities here, so its softmax is applied twice -- so is this one."""
    assert input_logits.size() == target_logits.size()
    return _SoftmaxMSE.apply(input_logits, target_logits, bool(sigmoid))


class _SoftmaxKL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, sigmoid):
        a, b = _f32(a), _f32(b)
        a, va, n, C, V = _view_ncv(a)
        b, vb, _, _, _ = _view_ncv(b)
        scratch = torch.empty(1, dtype=torch.float64, device=a.device)
        out = torch.empty(1, dtype=torch.float32, device=a.device)
        call("dycon_softmax_kl_fwd", _ref(va), _ref(vb), n, C, V, int(sigmoid), scratch.data_ptr(), out.data_ptr(), ops._s())
        ctx.save_for_backward(a, b)
        ctx.sigmoid = sigmoid
        return out[0]

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        _, va, n, C, V = _view_ncv(a)
        _, vb, _, _, _ = _view_ncv(b)
        gu = _f32(g).reshape(1).contiguous()
        grads = []
        for which, (need, x) in enumerate(zip(ctx.needs_input_grad[:2], (a, b))):
            if not need:
                grads.append(None)
                continue
            gx = torch.empty_like(x)
            gx, vgx, _, _, _ = _view_ncv(gx)
            call("dycon_softmax_kl_bwd", _ref(va), _ref(vb), n, C, V, int(ctx.sigmoid), which, gu.data_ptr(), _ref(vgx), ops._s())
            grads.append(gx)
        return grads[0], grads[1], None


def softmax_kl_loss(input_logits, target_logits, sigmoid=False):
    """losses.py:85-104: F.kl_div(log_softmax(input, 1), softmax(target, 1), reduction='mean') -> 0-dim tensor."""
    assert input_logits.size() == target_logits.size()
    return _SoftmaxKL.apply(input_logits, target_logits, bool(sigmoid))


class _Dice(torch.autograd.Function):
    @staticmethod
    def forward(ctx, score, target, onehot, softmax, weights, n_div):
        score = _f32(score)
        if target.dtype not in _TKIND:
            target = target.float()
        if onehot:
            score, vs, n, C, V = _view_ncv(score)
            target, vt_, tn, tC, tV = _view_ncv(target)          # (n, 1, *spatial) label map
            if (tn, tC, tV) != (n, 1, V):
                raise ValueError(f"DiceLoss: target {tuple(target.shape)} does not match inputs {tuple(score.shape)}")
            vt = View(target.data_ptr(), target.stride(0), 0, vt_.sv)
        else:
            if score.shape != target.shape:
                raise ValueError(f"dice_loss: score {tuple(score.shape)} vs target {tuple(target.shape)}")
            score, vs, V = _view_flat(score)
            target, vt, _ = _view_flat(target)
            n, C = 1, 1
        import ctypes
        w = (ctypes.c_float * C)(*[float(x) for x in weights]) if weights is not None else None
        sums = torch.empty(24, dtype=torch.float64, device=score.device)
        out = torch.empty(1, dtype=torch.float32, device=score.device)
        args = (_ref(vs), _ref(vt), _TKIND[target.dtype], int(onehot), n, C, V, int(softmax), w, float(n_div))
        call("dycon_dice_fwd", *args, sums.data_ptr(), out.data_ptr(), ops._s())
        ctx.save_for_backward(score, target, sums)
        ctx.args = args
        return out[0]

    @staticmethod
    def backward(ctx, g):
        score, target, sums = ctx.saved_tensors
        gu = _f32(g).reshape(1).contiguous()
        gs = torch.empty(score.shape, dtype=torch.float32, device=score.device)     # contiguous, natural order
        if ctx.args[3]:
            _, vg, _, _, _ = _view_ncv(gs)
        else:
            _, vg, _ = _view_flat(gs)
        call("dycon_dice_bwd", *ctx.args, sums.data_ptr(), gu.data_ptr(), _ref(vg), ops._s())
        return gs, None, None, None, None, None


def dice_loss(score, target):
    """losses.py:8-16: 1 - (2 sum(score*target) + 1e-5) / (sum(score^2) + sum(target^2) + 1e-5), sums over the whole tensors
    (call site train_DyCON_BraTS19.py:314: score = probs[:LB, 1], target = label[:LB] == 1)."""
    return _Dice.apply(score, target, False, False, None, 1.0)


class DiceLoss(nn.Module):
    """losses.DiceLoss(n_classes) (losses.py:156-192): forward(inputs, target, weight=None, softmax=False) with the reference's
    argument meaning -- ``inputs`` (B, n, ...) probabilities (or logits with softmax=True), ``target`` (B, 1, ...) label map
    (train_DyCON_ISLES22.py:194,247).  (The reference also reads every class's dice back to the host -- ``.item()`` per class --
    into a list it never uses; that synchronisation is not reproduced.)"""

    def __init__(self, n_classes):
        super().__init__()
        if not 1 <= n_classes <= 8:
            raise NotImplementedError("1..8 classes")
        self.n_classes = n_classes

    def forward(self, inputs, target, weight=None, softmax=False):
        if target.dim() == inputs.dim() - 1:
            target = target.unsqueeze(1)
        assert inputs.shape[1] == self.n_classes and inputs.shape[0] == target.shape[0] and inputs.shape[2:] == target.shape[2:], \
            "predict & target shape do not match"
        return _Dice.apply(inputs, target, True, bool(softmax), weight, float(self.n_classes))
