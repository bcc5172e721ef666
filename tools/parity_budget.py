#!/usr/bin/env python3
"""fp32 parity on an fp64 footing: for every tensor the fp32 GPU tests compare, print the HIP path's error against the fp64 twin
of the reference (tests/golden/*.f64 keys, or the oracle run in double) next to the reference's OWN fp32-vs-fp64 error.  The
tests then hold  |hip - ref64| <= max(1e-4-style north-star bound, 2 |ref32 - ref64| + 1e-6).   Run on the GPU box."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden  # noqa: E402
from dycon_paper_replication_amd.engine import DropoutSpec  # noqa: E402
from dycon_paper_replication_amd.trainer import DyconTrainer, TrainConfig  # noqa: E402
from oracle import nets as ON  # noqa: E402
from test_engine_gpu import build  # noqa: E402

DEV = "cuda:0"
T = torch.from_numpy


def row(name, hip, r32, r64):
    hip, r32, r64 = (np.asarray(a, dtype=np.float64) for a in (hip, r32, r64))
    scale = np.abs(r64).max()
    print(f"{name:46s} |hip-ref64| {np.abs(hip - r64).max():.3e}   |ref32-ref64| {np.abs(r32 - r64).max():.3e}   scale {scale:.3e}   "
          f"|hip-ref32| {np.abs(hip - r32).max():.3e}")


def full_nets():
    g = load_golden("full_nets")
    rng = np.random.default_rng(int(g["x_seed"]))
    draw = lambda *s: T(rng.standard_normal(s).astype(np.float32))  # noqa: E731
    x = draw(2, 1, 32, 32, 32)
    rs = {"vnet": (draw(2, 2, 32, 32, 32), draw(2, 256, 4, 4, 4))}
    rs["unet"] = (draw(2, 2, 32, 32, 32), draw(2, 256, 4, 4, 4))
    for kind in ("vnet", "unet"):
        eng, _ = build(kind, int(g[f"{kind}.param_seed"]))
        logits, feats, _ = eng.forward(x.permute(0, 2, 3, 4, 1).contiguous().to(DEV), training=True, record=True)
        lo = logits.cpu().permute(0, 4, 1, 2, 3)[..., ::2, ::2, ::2].numpy()
        fe = feats.cpu().permute(0, 4, 1, 2, 3).numpy()
        row(f"full_nets {kind} logits", lo, g[f"{kind}.logits_sub"], g[f"{kind}.logits_sub.f64"])
        row(f"full_nets {kind} feats", fe, g[f"{kind}.feats"], g[f"{kind}.feats.f64"])
        r1, r2 = rs[kind]
        eng.backward(r1.permute(0, 2, 3, 4, 1).contiguous().to(DEV), r2.permute(0, 2, 3, 4, 1).contiguous().to(DEV))
        torch.cuda.synchronize()
        names = list(g[f"{kind}.grad_names"])
        worst = (0, None)
        worst_ref = (0, None)
        for k, s32, s64 in zip(names, g[f"{kind}.grad_stats"], g[f"{kind}.grad_stats.f64"]):
            if k.startswith("final."):
                continue
            t = eng.g[k].double().cpu()
            got = np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])
            e_h = abs(got[1] - s64[1]) / (s64[1] + 1e-12)
            e_r = abs(s32[1] - s64[1]) / (s64[1] + 1e-12)
            if e_h > worst[0]:
                worst = (e_h, k)
            if e_r > worst_ref[0]:
                worst_ref = (e_r, k)
        print(f"full_nets {kind} grad sum|.| rel err: hip worst {worst[0]:.3e} ({worst[1]}), ref32 worst {worst_ref[0]:.3e} ({worst_ref[1]})")


def steps():
    for kind in ("unet", "vnet"):
        g = load_golden(f"step_{kind}")
        net_type = "unet_3D" if kind == "unet" else "vnet"
        mk = ON.make_unet_params if kind == "unet" else ON.make_vnet_params
        s0, s1 = [int(v) for v in g["seeds"]]
        tr = DyconTrainer(TrainConfig(model=net_type, labeled_bs=int(g["LB"]), batch_size=int(g["B"]), dtype=torch.float32), DEV,
                          student_init=mk(s0), teacher_init=mk(s1))
        off = DropoutSpec("off")
        for step in range(2):
            out = tr.step(T(g[f"s{step}.vol"]).to(DEV), T(g[f"s{step}.label"]).to(DEV), noise=T(g[f"s{step}.noise"]).to(DEV),
                          s_drop=off, t_drop=off, epoch=int(g[f"s{step}.epoch"]), beta=float(g[f"s{step}.beta"]))
            got = [float(out[k]) for k in ("loss", "ce", "dice", "cons", "fecl", "uncl")] + [out["cons_weight"], float(out["grad_sumsq"].sqrt())]
            r32, r64 = g[f"s{step}.scalars"], g[f"s{step}.scalars.f64"]
            rel = lambda a, b: np.abs(np.asarray(a) - b) / (np.abs(b) + 1e-12)  # noqa: E731
            print(f"step_{kind} s{step} scalars rel: hip {rel(got, r64).round(7)}\n{'':22s}ref32 {rel(r32, r64).round(7)}")
            lo = out["s_logits"].cpu().permute(0, 4, 1, 2, 3)[..., ::2, ::2, ::2].numpy()
            row(f"step_{kind} s{step} s_logits", lo, g[f"s{step}.logits_sub"], g[f"s{step}.logits_sub.f64"])
            tl = out["t_logits"].cpu().permute(0, 4, 1, 2, 3)[..., ::2, ::2, ::2].numpy()
            row(f"step_{kind} s{step} t_logits", tl, g[f"s{step}.t_logits_sub"], g[f"s{step}.t_logits_sub.f64"])
            wh = wr = 0.0
            for k, a32, a64 in zip(tr.names, g[f"s{step}.student_stats"], g[f"s{step}.student_stats.f64"]):
                t = tr.p[k].double().cpu()
                got_s = np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])
                wh = max(wh, float((np.abs(got_s - a64) / (np.abs(a64) + 1e-4)).max()))
                wr = max(wr, float((np.abs(a32 - a64) / (np.abs(a64) + 1e-4)).max()))
            print(f"step_{kind} s{step} student stats rel (atol 1e-4): hip {wh:.3e} ref32 {wr:.3e}")


def geometry():
    for name, kind, shape, sf in (("isles 112x112x80 sf4", "vnet", (112, 112, 80), 4), ("pancreas 112x112x96 vnet", "vnet", (112, 112, 96), 2),
                                  ("pancreas 112x112x96 unet", "unet", (112, 112, 96), 2)):
        eng, p_all = build(kind, 3 if sf == 4 else 5)
        eng.scale_factor = sf
        torch.manual_seed(1 if sf == 4 else 2)
        x = torch.randn(1, 1, *shape)
        fwd = ON.vnet_forward if kind == "vnet" else ON.unet_forward
        with torch.no_grad():
            _, l32, f32 = fwd(x, p_all, scale_factor=sf)
            p64 = {k: (v.double() if v.is_floating_point() else v) for k, v in p_all.items()}
            _, l64, f64 = fwd(x.double(), p64, scale_factor=sf)
        logits, feats, _ = eng.forward(x.permute(0, 2, 3, 4, 1).contiguous().to(DEV), record=False)
        row(f"{name} logits", logits.cpu().permute(0, 4, 1, 2, 3).numpy(), l32.numpy(), l64.numpy())
        row(f"{name} feats", feats.cpu().permute(0, 4, 1, 2, 3).numpy(), f32.numpy(), f64.numpy())


if __name__ == "__main__":
    full_nets()
    steps()
    geometry()
