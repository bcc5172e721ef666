#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (needs /root/reference): is oracle/step.py a fair stand-in for "the reference CPU path"?

Times one DyCON iteration at 96^3, B = 2 (1+1) and B = 4 (2+2), 1 warm-up + 3 timed steps (median), (a) with the IMPORTED reference
modules / losses / torch.optim.SGD / clip_grad_norm_ / EMA loop driven in the order of train_DyCON_BraTS19.py:298-372 (the
stub-package import recipe of tests/golden/make_golden.py) and (b) with oracle/step.py -- same weights, same inputs, same thread
count.  SURVEY.md section 8d asks for the two to agree within +-10 %; bench.py then times (b) on the GPU box's host cores.
    python tools/oracle_vs_reference.py [unet_3D|vnet ...]   ->  prints a table (committed as profiles/r02_oracle_vs_reference.txt)
"""
import os
import statistics
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as MG  # noqa: E402  (imports the reference through the stub packages)
from dycon_paper_replication_amd.synthetic import make_batch  # noqa: E402
from oracle import nets as ON  # noqa: E402
from oracle import step as OS  # noqa: E402

torch.set_num_threads(int(os.environ.get("THREADS", "8")))


def reference_step_fn(kind):
    if kind == "unet_3D":
        model, ema = MG.ref_factory.net_factory_3d("unet_3D", 1, 2, 2), MG.ref_factory.net_factory_3d("unet_3D", 1, 2, 2)
        model.load_state_dict(ON.make_unet_params(1))
        ema.load_state_dict(ON.make_unet_params(2))
        for m_ in (model, ema):
            m_.dropout1.p = m_.dropout2.p = 0.0
        sp, tp = list(model.parameters()), list(ema.parameters())
    else:
        model, ema = MG.VNetWithHead(2), MG.VNetWithHead(2)
        model.load_flat(ON.make_vnet_params(1))
        ema.load_flat(ON.make_vnet_params(2))
        sp, tp = [p for _, p in model.flat_named_parameters()], [p for _, p in ema.flat_named_parameters()]
    for p_ in tp:
        p_.detach_()
    model.train(); ema.train()
    opt = torch.optim.SGD(sp, lr=0.01, momentum=0.9, weight_decay=0.0001)
    uncl, fecl = MG.ref_dycon.UnCLoss(), MG.ref_dycon.FeCLoss(device="cpu", temperature=0.6, gamma=2.0, use_focal=True, rampup_epochs=1500)
    it = [0]

    def step(vol, lab, noise, LB):
        _, s_logits, s_feat = model(vol)
        with torch.no_grad():
            _, t_logits, t_feat = ema(vol + noise)
        s_prob, t_prob = F.softmax(s_logits, 1), F.softmax(t_logits, 1)
        cw = 0.1 * MG.ref_ramps.sigmoid_rampup(it[0] // 150, 200.0)
        ce = F.cross_entropy(s_logits[:LB], lab[:LB])
        dice = MG.ref_losses.dice_loss(s_prob[:LB, 1], lab[:LB] == 1)
        B, C = s_feat.shape[:2]
        s_emb = F.normalize(s_feat.view(B, C, -1).transpose(1, 2), dim=-1)
        t_emb = F.normalize(t_feat.view(B, C, -1).transpose(1, 2), dim=-1)
        mask = (F.avg_pool3d(lab.float(), kernel_size=8, stride=8) > 0.5).float().reshape(B, -1).unsqueeze(1)
        f_loss = fecl(feat=s_emb, mask=mask, teacher_feat=t_emb, gambling_uncertainty=None, epoch=0)
        u_loss = uncl(s_logits, t_logits, 5.0)
        cons = MG.ref_losses.softmax_mse_loss(s_prob[LB:], t_prob[LB:]).mean()
        loss = 1.0 * (ce + dice) + cw * cons + 0.5 * (f_loss + u_loss)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(sp, max_norm=1.0)
        opt.step()
        alpha = min(1 - 1 / (it[0] + 1), 0.99)
        for e_, s_ in zip(tp, sp):
            e_.data.mul_(alpha).add_(s_.data, alpha=1 - alpha)
        it[0] += 1
        return float(loss)
    return step


def oracle_step_fn(kind):
    mk = ON.make_unet_params if kind == "unet_3D" else ON.make_vnet_params
    st = OS.StepState(student=mk(1), teacher=mk(2))

    def step(vol, lab, noise, LB):
        return float(OS.train_step(OS.StepConfig(net_type=kind, labeled_bs=LB, feature_scaler=2), st, vol, lab, noise, 5.0, 0)["loss"])
    return step


def timed(fn, *a):
    fn(*a)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        last = fn(*a)
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), last


print(f"torch {torch.__version__}, {torch.get_num_threads()} threads, 96^3 patches, 1 warm-up + 3 timed steps, median")
print(f"{'model':9s} {'batch':8s} {'reference s/step':>17s} {'oracle s/step':>14s} {'ratio':>6s}   4th-step loss (reference / oracle)")
for kind in (sys.argv[1:] or ["vnet", "unet_3D"]):
    for B in (2, 4):
        vol, lab, noise = make_batch(1337, B, (96, 96, 96))
        tr, lr_ = timed(reference_step_fn(kind), vol, lab, noise, B // 2)
        to, lo_ = timed(oracle_step_fn(kind), vol, lab, noise, B // 2)
        print(f"{kind:9s} {B // 2}+{B // 2:<6d} {tr:17.2f} {to:14.2f} {to / tr:6.2f}   {lr_:.6f} / {lo_:.6f}", flush=True)
