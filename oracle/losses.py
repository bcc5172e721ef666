"""Oracle losses and host schedules for the DyCON step.  Test infrastructure only.

Each function restates one reference function; the citation gives the lines followed.
All tensor functions are differentiable plain PyTorch and dtype-generic (fp32/fp64).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------ host scalars
def adaptive_beta(epoch, total_epochs, max_beta=5.0, min_beta=0.5):
    """utils/dycon_losses.py:8-12 -- geometric decay from max_beta to min_beta."""
    return max_beta * ((min_beta / max_beta) ** (epoch / total_epochs))


def threshold_rampup(current_epoch, total_rampup_epochs, min_threshold, max_threshold, steepness=5.0):
    """utils/dycon_losses.py:28-47 (the 5-argument sigmoid_rampup used for FeCL thresholds)."""
    if total_rampup_epochs == 0:
        return max_threshold
    e = max(0.0, min(float(current_epoch), total_rampup_epochs))
    phase = 1.0 - e / total_rampup_epochs
    return min_threshold + (max_threshold - min_threshold) * math.exp(-steepness * phase * phase)


def consistency_rampup(current, rampup_length):
    """utils/ramps.py:19-26 (the 2-argument sigmoid_rampup used for the consistency weight)."""
    if rampup_length == 0:
        return 1.0
    c = min(max(float(current), 0.0), float(rampup_length))
    phase = 1.0 - c / rampup_length
    return float(math.exp(-5.0 * phase * phase))


def consistency_weight(iter_num, consistency=0.1, rampup=200.0):
    """train_DyCON_BraTS19.py:150-152, 310: weight = consistency * rampup(iter // 150, length)."""
    return consistency * consistency_rampup(iter_num // 150, rampup)


# ------------------------------------------------------------------ voxel losses
def uncl(s_logits, t_logits, beta: float):
    """UnCLoss.forward, utils/dycon_losses.py:94-118, in closed form.

    The reference adds a (B,H,W,D) tensor to a (B,1,H,W,D) one (line 116), which broadcasts to
    (B,B,H,W,D); its mean equals mean_v[sum_c d^2/(e^{bHs}+e^{bHt})] + beta*mean_v[Hs+Ht]
    (re-checked against the imported reference in make_golden.py)."""
    eps = 1e-6
    ps = F.softmax(s_logits, dim=1)
    pt = F.softmax(t_logits, dim=1)
    hs = -(ps * torch.log(ps + eps)).sum(1)
    ht = -(pt * torch.log(pt + eps)).sum(1)
    w = torch.exp(beta * hs) + torch.exp(beta * ht)
    return (((ps - pt) ** 2).sum(1) / w).mean() + beta * (hs + ht).mean()


def dice_loss(score, target):
    """utils/losses.py:8-16 -- batch-global soft Dice, smooth 1e-5."""
    target = target.to(score.dtype)
    inter = (score * target).sum()
    return 1 - (2 * inter + 1e-5) / ((score * score).sum() + (target * target).sum() + 1e-5)


def dice_loss_multiclass(probs, target, n_classes: int = 2):
    """utils/losses.py:156-192 (DiceLoss.forward, weight=None, softmax=False): mean over classes of
    dice_loss(probs[:, c], target == c)."""
    tot = 0.0
    for c in range(n_classes):
        tot = tot + dice_loss(probs[:, c], (target == c))
    return tot / n_classes


def softmax_mse(in_logits, tgt_logits):
    """utils/losses.py:65-82 -- elementwise (softmax(a)-softmax(b))^2 (caller takes .mean())."""
    return (F.softmax(in_logits, 1) - F.softmax(tgt_logits, 1)) ** 2


def softmax_kl(in_logits, tgt_logits):
    """utils/losses.py:85-104 -- F.kl_div(log_softmax(a), softmax(b), reduction='mean')."""
    lp = F.log_softmax(in_logits, 1)
    q = F.softmax(tgt_logits, 1)
    return (torch.xlogy(q, q) - q * lp).mean()


# ------------------------------------------------------------------ FeCL
def _fecl_samples(feat, mask, teacher_feat, gambling_uncertainty, epoch, temperature, gamma, use_focal, rampup_epochs):
    """per sample: (per-patch loss vector (N,), cross-branch numerator, cross-branch count) -- dycon_losses.py:172-231"""
    B, N, _ = feat.shape
    m = mask.reshape(B, N)
    eye = torch.eye(N, dtype=feat.dtype)
    off = 1 - eye
    thr = threshold_rampup(epoch, rampup_epochs, 0.3, 0.5)
    for b in range(B):
        same = (m[b][:, None] == m[b][None, :]).to(feat.dtype)
        diff = 1 - same
        L = (feat[b] @ feat[b].t()) / temperature * off
        L = L - L.max(dim=0, keepdim=True)[0].detach()          # column max (dycon_losses.py:180-181)
        E = torch.exp(L)
        neg = (E * diff).sum(-1, keepdim=True)
        P = E / (E + neg + 1e-18)
        ell = -torch.log(P + 1e-18) * same * off
        denom = same.sum(-1) - 1 + 1e-18
        if gambling_uncertainty is not None:                       # :209-211 (overrides focal)
            per_patch = ell.sum(-1) / denom * gambling_uncertainty[b]
        elif use_focal:                                            # :196-206
            w = torch.where(same.bool(), (1 - P) ** gamma, torch.ones_like(P))
            per_patch = (ell * w).sum(-1) / denom
        else:
            per_patch = ell.sum(-1) / denom
        num = cnt = feat.new_zeros(())
        if teacher_feat is not None:                               # :214-231
            S = feat[b] @ teacher_feat[b].t()
            hard = (diff.bool() & (S > thr)).to(feat.dtype)
            num, cnt = (-torch.log(1 - S + 1e-18) * hard).sum(), hard.sum()
        yield per_patch, num, cnt


def fecl(feat, mask, teacher_feat: Optional[torch.Tensor] = None,
         gambling_uncertainty: Optional[torch.Tensor] = None, epoch=0, temperature=0.6, gamma=2.0,
         use_focal=False, rampup_epochs=2000, lambda_cross=1.0):
    """FeCLoss.forward, utils/dycon_losses.py:150-235, written per sample so that only one (N,N)
    block set is alive at a time.

    Kept exactly: the *column* max stabiliser (detached), the
    always-on positive focal weight (threshold >= 1.3 > any probability) which stays in the
    autograd graph, the hard-negative focal weights that multiply zeros (omitted: value and
    gradient are identical), the literal 1e-18 epsilons, the batch-global cross-branch ratio."""
    per_patch = []
    cross_num = feat.new_zeros(())
    cross_cnt = feat.new_zeros(())
    for pp, num, cnt in _fecl_samples(feat, mask, teacher_feat, gambling_uncertainty, epoch, temperature, gamma, use_focal, rampup_epochs):
        per_patch.append(pp)
        cross_num = cross_num + num
        cross_cnt = cross_cnt + cnt
    loss = torch.stack(per_patch).mean()
    if teacher_feat is not None and cross_cnt.item() > 0:
        loss = loss + lambda_cross * cross_num / (cross_cnt + 1e-18)
    return loss


def fecl_parts(feat, mask, teacher_feat=None, epoch=0, temperature=0.6, gamma=2.0, use_focal=False, rampup_epochs=2000):
    """The accumulators a data-parallel rank contributes: (sum of per-patch losses, cross numerator, cross
    count, rows = B*N).  fecl == student_sum / rows + cross_num / (cross_cnt + 1e-18) over the sums of all ranks."""
    tot = num_t = cnt_t = feat.new_zeros(())
    for pp, num, cnt in _fecl_samples(feat, mask, teacher_feat, None, epoch, temperature, gamma, use_focal, rampup_epochs):
        tot, num_t, cnt_t = tot + pp.sum(), num_t + num, cnt_t + cnt
    return tot, num_t, cnt_t, feat.shape[0] * feat.shape[1]


def fecl_rowblocks(feat, mask, teacher_feat, epoch=0, temperature=0.6, gamma=2.0, use_focal=True, rampup_epochs=2000,
                   lambda_cross=1.0, block=1024):
    """Memory-light evaluation of the SAME function as ``fecl`` (no gambling branch) for large N: rows are processed in blocks
    against all N columns, the backward runs block by block.  Returns (loss value, d loss / d feat) as detached tensors.
    Used to pin the HIP kernel at N = 15 680 (ISLES, feature_scaler 4), where the (N, N) formulation needs ~12 GB/sample.
    Checked against ``fecl`` itself in tests/test_oracle_golden.py."""
    B, N, _ = feat.shape
    m = mask.reshape(B, N)
    thr = threshold_rampup(epoch, rampup_epochs, 0.3, 0.5)
    f0 = feat.detach()
    # pass A (no grad): column max and the global cross-branch count
    colmax = torch.zeros(B, N, dtype=feat.dtype)
    cnt = 0.0
    for b in range(B):
        for r0 in range(0, N, block):
            rows = slice(r0, min(N, r0 + block))
            L = f0[b, rows] @ f0[b].t() / temperature
            idx = torch.arange(rows.start, rows.stop)
            L[torch.arange(len(idx)), idx] = 0.0
            colmax[b] = torch.maximum(colmax[b], L.max(0)[0])
            if teacher_feat is not None:
                S = f0[b, rows] @ teacher_feat[b].t()
                cnt += float(((m[b, rows][:, None] != m[b][None, :]) & (S > thr)).sum())
    # pass B: loss and gradient block by block
    leaf = f0.clone().requires_grad_(True)
    total = 0.0
    for b in range(B):
        for r0 in range(0, N, block):
            rows = slice(r0, min(N, r0 + block))
            nr = rows.stop - rows.start
            idx = torch.arange(rows.start, rows.stop)
            same = (m[b, rows][:, None] == m[b][None, :]).to(feat.dtype)
            off = torch.ones(nr, N, dtype=feat.dtype)
            off[torch.arange(nr), idx] = 0.0
            L = (leaf[b, rows] @ leaf[b].t()) / temperature * off - colmax[b][None, :]
            E = torch.exp(L)
            neg = (E * (1 - same)).sum(-1, keepdim=True)
            P = E / (E + neg + 1e-18)
            ell = -torch.log(P + 1e-18) * same * off
            if use_focal:
                ell = ell * torch.where(same.bool(), (1 - P) ** gamma, torch.ones_like(P))
            part = (ell.sum(-1) / (same.sum(-1) - 1 + 1e-18)).sum() / (B * N)
            if teacher_feat is not None and cnt > 0:
                S = leaf[b, rows] @ teacher_feat[b].t()
                hard = ((1 - same).bool() & (S > thr)).to(feat.dtype)
                part = part + lambda_cross * (-torch.log(1 - S + 1e-18) * hard).sum() / (cnt + 1e-18)
            part.backward()
            total += float(part.detach())
    return torch.tensor(total, dtype=feat.dtype), leaf.grad.detach()


def embed(features):
    """train_DyCON_BraTS19.py:316-323: (B,C,d,h,w) -> (B,N,C) rows L2-normalised (eps 1e-12)."""
    B, C = features.shape[:2]
    return F.normalize(features.reshape(B, C, -1).transpose(1, 2), dim=-1)


def contrast_mask(label, kernel):
    """train_DyCON_BraTS19.py:326-330: avg_pool3d(label.float(), k, k) > 0.5 -> (B,1,N) float."""
    m = F.avg_pool3d(label.float().unsqueeze(1) if label.dim() == 4 else label.float(), kernel, kernel)
    return (m > 0.5).float().reshape(label.shape[0], 1, -1)
