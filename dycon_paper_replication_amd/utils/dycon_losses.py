"""UnCL / FeCL with the reference's call signatures (code/utils/dycon_losses.py), on HIP kernels.

Both are torch.autograd.Functions over the C ABI: the forward launches the fused reduction(s), the
backward one more pass.  Nothing of size (B,N,N) or (B,B,V) is ever materialised (the reference
builds ~12 N x N tensors for FeCL and a (B,B,H,W,D) broadcast for UnCL).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import ops


def adaptive_beta(epoch, total_epochs, max_beta=5.0, min_beta=0.5):
    """dycon_losses.py:8-12."""
    return max_beta * ((min_beta / max_beta) ** (epoch / total_epochs))


def sigmoid_rampup(current_epoch, total_rampup_epochs, min_threshold, max_threshold, steepness=5.0):
    """dycon_losses.py:28-47 (threshold schedule of FeCL)."""
    if total_rampup_epochs == 0:
        return max_threshold
    e = max(0.0, min(float(current_epoch), total_rampup_epochs))
    phase = 1.0 - e / total_rampup_epochs
    return min_threshold + (max_threshold - min_threshold) * math.exp(-steepness * phase * phase)


def _ndhwc_logits(x):
    """(B,2,D,H,W) any strides -> contiguous fp32 (B,D,H,W,2); free for channels_last_3d fp32 inputs."""
    if x.dim() != 5 or x.shape[1] != 2:
        raise ValueError(f"expected 2-class logits (B,2,D,H,W), got {tuple(x.shape)}")
    return x.permute(0, 2, 3, 4, 1).contiguous().float()


class _UnCLFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s_logits, t_logits, beta):
        s, t = _ndhwc_logits(s_logits), _ndhwc_logits(t_logits)
        B = s.shape[0]
        V = s.numel() // (2 * B)
        dummy = torch.empty(1, dtype=torch.uint8, device=s.device)
        sums = ops.seg_losses_fwd(s, t, dummy, 0, beta)
        vals = ops.seg_losses_finalize(sums, B, 0, V, beta)
        ctx.save_for_backward(s, t, sums, dummy)
        ctx.beta = beta
        return vals[5]

    @staticmethod
    def backward(ctx, g):
        s, t, sums, dummy = ctx.saved_tensors
        coef = torch.zeros(5, dtype=torch.float32, device=s.device)
        coef[4:5] = g.reshape(1).float()
        gs = ops.seg_losses_bwd(s, t, dummy, 0, ctx.beta, sums, coef)
        return gs.permute(0, 4, 1, 2, 3), None, None


class UnCLoss(nn.Module):
    """Uncertainty-weighted consistency (dycon_losses.py:50-118): forward(s_logits, t_logits, beta) -> 0-dim."""

    def forward(self, s_logits, t_logits, beta):
        return _UnCLFunction.apply(s_logits, t_logits.detach(), float(beta))


class _FeCLFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, teacher, mask, gambling, temperature, gamma, use_focal, thr, lambda_cross):
        f = feat.contiguous()
        if f.dtype not in (torch.float32, torch.bfloat16):
            f = f.float()
        t = teacher.contiguous().to(f.dtype) if teacher is not None else None
        m = mask.reshape(f.shape[0], f.shape[1]).contiguous().float()
        gmb = gambling.reshape(f.shape[0], f.shape[1]).contiguous().float() if gambling is not None else None
        args = (f, t, m, gmb, float(temperature), float(gamma), bool(use_focal), float(thr), float(lambda_cross))
        loss, st = ops.fecl_fwd(*args)
        ctx.args, ctx.st, ctx.in_dtype = args, st, feat.dtype
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        gf = ops.fecl_bwd(*ctx.args, ctx.st, g.reshape(1).float().contiguous())
        return (gf.to(ctx.in_dtype),) + (None,) * 8


class FeCLoss(nn.Module):
    """Focal patch-contrastive loss (dycon_losses.py:120-235); same constructor and call signature."""

    def __init__(self, device=None, temperature=0.6, gamma=2.0, use_focal=False, rampup_epochs=2000, lambda_cross=1.0):
        super().__init__()
        self.device, self.temperature, self.gamma = device, temperature, gamma
        self.use_focal, self.rampup_epochs, self.lambda_cross = use_focal, rampup_epochs, lambda_cross

    def forward(self, feat, mask, teacher_feat=None, gambling_uncertainty=None, epoch=0):
        thr = sigmoid_rampup(epoch, self.rampup_epochs, min_threshold=0.3, max_threshold=0.5)
        return _FeCLFunction.apply(feat, teacher_feat.detach() if teacher_feat is not None else None, mask,
                                   gambling_uncertainty, self.temperature, self.gamma, self.use_focal, thr, self.lambda_cross)
