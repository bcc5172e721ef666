"""Oracle training step: restatement of code/train_DyCON_BraTS19.py:298-372 (and the ISLES
variants at code/train_DyCON_ISLES22.py:228-326) as one function over explicit state.

Test infrastructure only.  Randomness (teacher input noise, dropout masks) is passed in, so
the HIP path can be fed the very same tensors.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import losses as L
from . import nets


@dataclass
class StepConfig:
    net_type: str = "vnet"            # "vnet" | "unet_3D"
    normalization: str = "groupnorm"  # V-Net only
    labeled_bs: int = 2
    feature_scaler: int = 2
    base_lr: float = 0.01
    momentum: float = 0.9
    weight_decay: float = 1e-4
    ema_decay: float = 0.99
    consistency: float = 0.1
    consistency_rampup: float = 200.0
    consistency_type: str = "mse"     # "mse" | "kl"
    temp: float = 0.6
    gamma: float = 2.0
    use_focal: bool = True
    use_teacher_loss: bool = True
    rampup_epochs: int = 1500
    l_weight: float = 1.0
    u_weight: float = 0.5
    max_grad_norm: float = 1.0
    dice_variant: str = "fg"          # "fg" (BraTS/Pancreas: losses.dice_loss on class 1) | "multiclass" (ISLES DiceLoss)
    teacher_bn_training: bool = True   # BraTS/Pancreas: teacher in .train(); ISLES: .eval()
    poly_lr_max_iter: int = 0          # >0: ISLES poly schedule lr = base*(1-it/max)^0.9, applied AFTER the step


@dataclass
class StepState:
    student: Dict[str, torch.Tensor]
    teacher: Dict[str, torch.Tensor]
    momentum: Dict[str, torch.Tensor] = field(default_factory=dict)
    iter_num: int = 0
    lr: Optional[float] = None


def clip_grad_norm(grads: Dict[str, torch.Tensor], max_norm: float):
    """torch.nn.utils.clip_grad_norm_ semantics: total L2 norm; coef = min(1, max/(norm+1e-6))."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).to(torch.float32)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return total, {k: g * coef for k, g in grads.items()}


def sgd_step(params, grads, momentum_buf, lr, momentum, weight_decay):
    """torch.optim.SGD (train_DyCON_BraTS19.py:268): g += wd*p; buf = g (first) | mu*buf + g; p -= lr*buf."""
    for k, p in params.items():
        g = grads[k] + weight_decay * p
        if k not in momentum_buf:
            momentum_buf[k] = g.clone()
        else:
            momentum_buf[k] = momentum * momentum_buf[k] + g
        params[k] = p - lr * momentum_buf[k]


def ema_update(teacher, student, decay, global_step):
    """update_ema_variables, train_DyCON_BraTS19.py:155-164 -- parameters only, buffers untouched."""
    alpha = min(1 - 1 / (global_step + 1), decay)
    for k in student:
        teacher[k] = teacher[k] * alpha + student[k] * (1 - alpha)


def losses_from_outputs(cfg: StepConfig, s_logits, s_feat, t_logits, t_feat, label, beta, epoch, iter_num):
    """train_DyCON_BraTS19.py:308-357.  Returns (total, dict of the five components)."""
    LB = cfg.labeled_bs
    s_prob = F.softmax(s_logits, 1)
    t_prob = F.softmax(t_logits, 1)
    cw = L.consistency_weight(iter_num, cfg.consistency, cfg.consistency_rampup)
    ce = F.cross_entropy(s_logits[:LB], label[:LB])
    if cfg.dice_variant == "fg":
        dice = L.dice_loss(s_prob[:LB, 1], label[:LB] == 1)
    else:
        dice = L.dice_loss_multiclass(s_prob[:LB], label[:LB], s_logits.shape[1])
    k = label.shape[1] // s_feat.shape[2]
    mask = L.contrast_mask(label, k)
    f = L.fecl(L.embed(s_feat), mask, L.embed(t_feat) if cfg.use_teacher_loss else None, None, epoch,
               cfg.temp, cfg.gamma, cfg.use_focal, cfg.rampup_epochs, 1.0)
    u = L.uncl(s_logits, t_logits, beta)
    if cfg.consistency_type == "mse":
        cons = L.softmax_mse(s_prob[LB:], t_prob[LB:]).mean()   # softmax applied twice, as the reference does
    else:
        cons = L.softmax_kl(s_prob[LB:], t_prob[LB:])
    total = cfg.l_weight * (ce + dice) + cw * cons + cfg.u_weight * (f + u)
    return total, {"ce": ce, "dice": dice, "cons": cons, "fecl": f, "uncl": u, "cons_weight": cw}


def train_step(cfg: StepConfig, st: StepState, volume, label, noise, beta: float, epoch: int,
               s_drop: Optional[dict] = None, t_drop: Optional[dict] = None):
    """One DyCON iteration.  Mutates ``st``; returns a dict with every observable quantity."""
    s_drop = s_drop or {}
    t_drop = t_drop or {}
    names = list(nets.trainable(st.student).keys())
    sp = {k: (v.detach().clone().requires_grad_(True) if k in names else v) for k, v in st.student.items()}
    kw = dict(scale_factor=cfg.feature_scaler)
    if cfg.net_type == "vnet":
        kw["normalization"] = cfg.normalization
    _, s_logits, s_feat = nets.forward(cfg.net_type, volume, sp, update_buffers=True, **kw, **s_drop)
    with torch.no_grad():
        _, t_logits, t_feat = nets.forward(cfg.net_type, volume + noise, st.teacher,
                                           bn_training=cfg.teacher_bn_training,
                                           update_buffers=cfg.teacher_bn_training, **kw, **t_drop)
    total, comps = losses_from_outputs(cfg, s_logits, s_feat, t_logits, t_feat, label, beta, epoch, st.iter_num)
    out = {"loss": total.detach(), **{k: (v.detach() if torch.is_tensor(v) else v) for k, v in comps.items()},
           "s_logits": s_logits.detach(), "t_logits": t_logits, "s_feat": s_feat.detach(), "t_feat": t_feat}
    if not math.isfinite(float(total.detach())):
        out["skipped"] = True   # train_DyCON_BraTS19.py:360-362 -- `continue`: no update, iter_num unchanged
        return out
    gl = torch.autograd.grad(total, [sp[k] for k in names], allow_unused=True)
    # parameters the loss never touches (UNet3D.final: its tanh output is discarded by the step) keep
    # grad=None in the reference, so clip_grad_norm_ and SGD (weight decay included) skip them.
    grads = {k: g for k, g in zip(names, gl) if g is not None}
    out["grads"] = grads
    gnorm, grads = clip_grad_norm(grads, cfg.max_grad_norm)
    out["grad_norm"] = gnorm
    lr = st.lr if st.lr is not None else cfg.base_lr
    params = {k: st.student[k] for k in grads}
    sgd_step(params, grads, st.momentum, lr, cfg.momentum, cfg.weight_decay)
    st.student.update(params)
    t_params = {k: st.teacher[k] for k in names}
    ema_update(t_params, {k: st.student[k] for k in names}, cfg.ema_decay, st.iter_num)
    st.teacher.update(t_params)
    if cfg.poly_lr_max_iter > 0:
        st.lr = cfg.base_lr * (1.0 - st.iter_num / cfg.poly_lr_max_iter) ** 0.9
    st.iter_num += 1
    out["skipped"] = False
    return out


def ddp_train_step(cfg: StepConfig, states, shards, beta: float, epoch: int, student_kw: Optional[dict] = None):
    """The data-parallel iteration as the trainer runs it over W ranks, emulated in one process: every rank
    forwards its own shard [labelled | unlabelled] (so the projection head's BatchNorm sees PER-RANK batch statistics, as in the
    reference's DataParallel replicas), the Dice and FeCL-cross accumulators are summed over ranks (the 16 + 4-double
    all-reduces), each rank differentiates its local objective with the two global-ratio terms pre-scaled by W, the gradients
    are averaged (the arena all-reduce with 1/W folded into SGD), and every rank applies the same clip + SGD + EMA.
    states: one StepState per rank (identical parameters, own BatchNorm buffers); shards: [(volume, label, noise)] per rank.
    cfg.base_lr is the already scaled LR (x W, train_DyCON_BraTS19.py:108-110).  dice_variant 'fg', consistency 'mse'."""
    assert cfg.dice_variant == "fg" and cfg.consistency_type == "mse"
    W, LB = len(states), cfg.labeled_bs
    names = list(nets.trainable(states[0].student).keys())
    kw = dict(scale_factor=cfg.feature_scaler)
    if cfg.net_type == "vnet":
        kw["normalization"] = cfg.normalization
    ranks = []
    for st, (vol, lab, noise) in zip(states, shards):
        sp = {k: (v.detach().clone().requires_grad_(True) if k in names else v) for k, v in st.student.items()}
        _, s_logits, s_feat = nets.forward(cfg.net_type, vol, sp, update_buffers=True, **kw, **(student_kw or {}))
        with torch.no_grad():
            _, t_logits, t_feat = nets.forward(cfg.net_type, vol + noise, st.teacher, bn_training=cfg.teacher_bn_training,
                                               update_buffers=cfg.teacher_bn_training, **kw)
        s_prob, t_prob = F.softmax(s_logits, 1), F.softmax(t_logits, 1)
        t1 = (lab[:LB] == 1).to(s_prob.dtype)
        acc = dict(I=(s_prob[:LB, 1] * t1).sum(), Z=(s_prob[:LB, 1] ** 2).sum(), Y=(t1 * t1).sum())
        k = lab.shape[1] // s_feat.shape[2]
        stud, num, cnt, rows = L.fecl_parts(L.embed(s_feat), L.contrast_mask(lab, k), L.embed(t_feat) if cfg.use_teacher_loss else None,
                                            epoch, cfg.temp, cfg.gamma, cfg.use_focal, cfg.rampup_epochs)
        ranks.append(dict(sp=sp, acc=acc, stud=stud, num=num, cnt=cnt, rows=rows, ce=F.cross_entropy(s_logits[:LB], lab[:LB]),
                          cons=L.softmax_mse(s_prob[LB:], t_prob[LB:]).mean(), uncl=L.uncl(s_logits, t_logits, beta)))
    g = {k: sum(r["acc"][k].detach() for r in ranks) for k in ("I", "Z", "Y")}
    numg, cntg = sum(r["num"].detach() for r in ranks), sum(r["cnt"].detach() for r in ranks)
    rows_g = sum(r["rows"] for r in ranks)
    cw = L.consistency_weight(states[0].iter_num, cfg.consistency, cfg.consistency_rampup)
    avg = None
    for r in ranks:
        a = r["acc"]
        dice_loc = 1 - (2 * (a["I"] + (g["I"] - a["I"].detach())) + 1e-5) / ((a["Z"] + (g["Z"] - a["Z"].detach())) + g["Y"] + 1e-5)
        cross_loc = (r["num"] + (numg - r["num"].detach())) / (cntg + 1e-18) if (cfg.use_teacher_loss and float(cntg) > 0) else 0.0
        local = cfg.l_weight * (r["ce"] + W * dice_loc) + cw * r["cons"] + cfg.u_weight * (W * r["stud"] / rows_g + W * cross_loc + r["uncl"])
        gl = torch.autograd.grad(local, [r["sp"][k] for k in names], allow_unused=True)
        gl = {k: v for k, v in zip(names, gl) if v is not None}
        r["local_grads"] = gl
        avg = gl if avg is None else {k: avg[k] + gl[k] for k in avg}
    avg = {k: v / W for k, v in avg.items()}
    dice = 1 - (2 * g["I"] + 1e-5) / (g["Z"] + g["Y"] + 1e-5)
    fecl = sum(r["stud"].detach() for r in ranks) / rows_g + (numg / (cntg + 1e-18) if float(cntg) > 0 else 0.0)
    ce, cons, uncl = (sum(r[k].detach() for r in ranks) / W for k in ("ce", "cons", "uncl"))
    total = cfg.l_weight * (ce + dice) + cw * cons + cfg.u_weight * (fecl + uncl)
    gnorm, clipped = clip_grad_norm(avg, cfg.max_grad_norm)
    for st in states:
        params = {k: st.student[k] for k in clipped}
        sgd_step(params, clipped, st.momentum, st.lr if st.lr is not None else cfg.base_lr, cfg.momentum, cfg.weight_decay)
        st.student.update(params)
        t_params = {k: st.teacher[k] for k in names}
        ema_update(t_params, {k: st.student[k] for k in names}, cfg.ema_decay, st.iter_num)
        st.teacher.update(t_params)
        st.iter_num += 1
    return {"loss": total, "ce": ce, "dice": dice, "cons": cons, "fecl": fecl, "uncl": uncl, "grad_norm": gnorm, "grads": avg,
            "local_grads": [r["local_grads"] for r in ranks]}
