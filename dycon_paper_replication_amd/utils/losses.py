"""Supervised / consistency losses of the step with the reference's names (code/utils/losses.py).

The reference composes them from probabilities it computed with torch (softmax -> dice_loss, ...).
On the HIP path all voxel losses come out of ONE fused pass over the logits
(``fused_voxel_losses``); the reference-named wrappers below route to that same pass so that a
script written against the reference API still runs entirely on the HIP kernels.  They therefore
take LOGITS where noted -- the 2-class softmax is part of the kernel.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
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


class DiceLoss(nn.Module):
    """losses.DiceLoss(n_classes) (losses.py:156-192), weight=None: forward(logits, target, softmax=True)."""

    def __init__(self, n_classes=2):
        super().__init__()
        if n_classes != 2:
            raise NotImplementedError("2 classes only (train_DyCON_ISLES22.py:194)")

    def forward(self, inputs, target, weight=None, softmax=True):
        if weight is not None or not softmax:
            raise NotImplementedError("fused path: pass logits with softmax=True and weight=None")
        if target.dim() == 5:
            target = target[:, 0]
        return fused_voxel_losses(inputs, inputs, target.long(), inputs.shape[0])[DICE_MC]
