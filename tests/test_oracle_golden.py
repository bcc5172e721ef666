"""CPU: pin the oracle (oracle/) against the golden fixtures generated from the imported
reference (tests/golden/make_golden.py).  Tolerances: fp32 round-off only (1e-5..1e-6);
the GPU parity tests then compare the HIP path against this oracle at 1e-4."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, load_golden
from oracle import losses as OL
from oracle import nets as ON
from oracle import step as OS

T = torch.from_numpy


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), rtol=rtol, atol=atol)


def test_schedules():
    tab = json.load(open(os.path.join(GOLDEN, "schedules.json")))
    for e, tot, mx, mn, v in tab["adaptive_beta"]:
        assert OL.adaptive_beta(e, tot, mx, mn) == pytest.approx(v, rel=1e-12)
    for e, R, lo, hi, v in tab["threshold_rampup"]:
        assert OL.threshold_rampup(e, R, lo, hi) == pytest.approx(v, rel=1e-12)
    for c, R, v in tab["consistency_rampup"]:
        assert OL.consistency_rampup(c, R) == pytest.approx(v, rel=1e-12)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_uncl(tag):
    g = load_golden(f"uncl_{tag}")
    for i, beta in enumerate(g["betas"]):
        s = T(g["s"]).requires_grad_(True)
        l = OL.uncl(s, T(g["t"]), float(beta))
        close(l, g[f"loss{i}"], 2e-6, 1e-7)
        close(torch.autograd.grad(l, s)[0], g[f"grad{i}"], 1e-4, 1e-9)
        close(OL.uncl(T(g["s"]).double(), T(g["t"]).double(), float(beta)), g[f"loss64_{i}"], 1e-12, 0)


@pytest.mark.parametrize("tag", ["small", "mid", "oneclass", "singleton", "ragged"])
def test_fecl(tag):
    g = load_golden(f"fecl_{tag}")
    for i in range(int(g["n_cfg"])):
        epoch, focal, use_t, use_g = [int(v) for v in g[f"cfg{i}"]]
        f = T(g["feat"]).requires_grad_(True)
        kw = dict(epoch=epoch, temperature=0.6, gamma=2.0, use_focal=bool(focal), rampup_epochs=1500)
        l = OL.fecl(f, T(g["mask"]), T(g["teacher"]) if use_t else None, T(g["gambling"]) if use_g else None, **kw)
        close(l, g[f"loss{i}"], 1e-5, 1e-6)
        close(torch.autograd.grad(l, f)[0], g[f"grad{i}"], 1e-4, 1e-7)
        l64 = OL.fecl(T(g["feat"]).double(), T(g["mask"]).double(), T(g["teacher"]).double() if use_t else None,
                      T(g["gambling"]).double() if use_g else None, **kw)
        close(l64, g[f"loss64_{i}"], 1e-11, 0)


def test_fecl_rowblocks_equals_fecl():
    """the memory-light evaluation used at N = 15 680 is the same function as the pinned oracle"""
    g = load_golden("fecl_mid")
    f, t, m = T(g["feat"]), T(g["teacher"]), T(g["mask"])
    for epoch, focal in ((0, True), (900, False), (3000, True)):
        fr = f.clone().requires_grad_(True)
        ref = OL.fecl(fr, m, t, None, epoch, 0.6, 2.0, focal, 1500, 1.0)
        (gr,) = torch.autograd.grad(ref, fr)
        val, grad = OL.fecl_rowblocks(f, m, t, epoch, 0.6, 2.0, focal, 1500, 1.0, block=50)
        close(val, ref.detach(), 1e-5, 1e-7)
        close(grad, gr, 1e-4, 1e-8)


def test_voxel_losses():
    g = load_golden("voxel_losses")
    a, b, lab = T(g["a"]), T(g["b"]), T(g["label"])

    def chk(fn, key):
        x = a.clone().requires_grad_(True)
        l = fn(x)
        close(l, g[key], 1e-6, 1e-7)
        close(torch.autograd.grad(l, x)[0], g[key + "_grad"], 1e-4, 1e-9)

    chk(lambda x: OL.dice_loss(F.softmax(x, 1)[:, 1], lab == 1), "dice")
    chk(lambda x: F.cross_entropy(x, lab), "ce")
    chk(lambda x: OL.dice_loss_multiclass(F.softmax(x, 1), lab, 2), "dice_mc")
    chk(lambda x: OL.softmax_mse(F.softmax(x, 1), F.softmax(b, 1)).mean(), "cons_mse")
    chk(lambda x: OL.softmax_kl(F.softmax(x, 1), F.softmax(b, 1)), "cons_kl")
    close(OL.softmax_mse(a, b), g["mse_elem"], 1e-6, 1e-7)


def _sub_params(g, prefix):
    return {k[len(prefix) + 3:]: T(g[k]) for k in g.files if k.startswith(prefix + ".p.")}


def test_vnet_layers():
    g = load_golden("vnet_layers")
    for norm in ("groupnorm", "none", "instancenorm", "batchnorm"):
        pre = f"convblock_{norm}"
        p = {"blk." + k: v for k, v in _sub_params(g, pre).items()}
        # keys in the fixture are "conv.N.weight"; the oracle wants "<name>.conv.N.weight"
        names = [k for k in p if p[k].is_floating_point() and "running" not in k]
        leaves = {k: p[k].clone().requires_grad_(True) for k in names}
        x = T(g[pre + ".x"]).requires_grad_(True)
        y = ON.vnet_conv_block(x, {**p, **leaves}, "blk", norm)
        close(y, g[pre + ".y"], 2e-5, 2e-6)
        grads = torch.autograd.grad((y * T(g[pre + ".r"])).sum(), [x] + [leaves[k] for k in names])
        close(grads[0], g[pre + ".gx"], 1e-4, 2e-5)
        for k, gr in zip(names, grads[1:]):
            close(gr, g[f"{pre}.g.{k[4:]}"], 1e-4, 5e-5)
    for pre, fn in (("down", ON.vnet_down_block), ("up", ON.vnet_up_block), ("first", ON.vnet_conv_block)):
        p = {"blk." + k: v for k, v in _sub_params(g, pre).items()}
        names = list(p)
        leaves = {k: p[k].clone().requires_grad_(True) for k in names}
        x = T(g[pre + ".x"]).requires_grad_(True)
        y = fn(x, leaves, "blk", "groupnorm")
        close(y, g[pre + ".y"], 2e-5, 2e-6)
        grads = torch.autograd.grad((y * T(g[pre + ".r"])).sum(), [x] + [leaves[k] for k in names])
        close(grads[0], g[pre + ".gx"], 1e-4, 2e-5)
        for k, gr in zip(names, grads[1:]):
            close(gr, g[f"{pre}.g.{k[4:]}"], 1e-4, 5e-5)


def test_unet_layers():
    g = load_golden("unet_layers")
    p = {"blk." + k: v for k, v in _sub_params(g, "unetconv").items()}
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    x = T(g["unetconv.x"]).requires_grad_(True)
    y = ON._unet_conv3(x, leaves, "blk")
    close(y, g["unetconv.y"], 2e-5, 2e-6)
    grads = torch.autograd.grad((y * T(g["unetconv.r"])).sum(), [x] + list(leaves.values()))
    close(grads[0], g["unetconv.gx"], 1e-4, 2e-5)
    for k, gr in zip(leaves, grads[1:]):
        close(gr, g[f"unetconv.g.{k[4:]}"], 1e-4, 5e-5)
    # up + concat
    p = {"blk." + k: v for k, v in _sub_params(g, "upcat").items()}
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    skip = T(g["upcat.skip"]).requires_grad_(True)
    low = T(g["upcat.low"]).requires_grad_(True)
    y = ON._unet_up(skip, low, leaves, "blk")
    close(y, g["upcat.y"], 2e-5, 2e-6)
    grads = torch.autograd.grad((y * T(g["upcat.r"])).sum(), [skip, low] + list(leaves.values()))
    close(grads[0], g["upcat.gskip"], 1e-4, 2e-5)
    close(grads[1], g["upcat.glow"], 1e-4, 2e-5)
    for k, gr in zip(leaves, grads[2:]):
        close(gr, g[f"upcat.g.{k[4:]}"], 1e-4, 5e-5)
    # projection head (train-mode BN), weights rebuilt from the seed
    pu = ON.make_unet_params(int(g["proj.param_seed"]))
    names = [k for k in ON.trainable(pu) if k.startswith("projection.")]
    leaves = {k: pu[k].clone().requires_grad_(True) for k in names}
    pp = {**pu, **leaves}
    x = T(g["proj.x"]).requires_grad_(True)
    # projection_head interpolates first; scale_factor=1 is the identity for align_corners=True
    y = ON.projection_head(x, pp, 1, True, True)
    close(y, g["proj.y"], 2e-5, 2e-6)
    grads = torch.autograd.grad((y * T(g["proj.r"])).sum(), [x] + [leaves[k] for k in names])
    close(grads[0], g["proj.gx"], 1e-4, 2e-5)
    for k, gr in zip(names, grads[1:]):
        short = k[len("projection."):]
        if f"proj.g.{short}" in g.files:
            close(gr, g[f"proj.g.{short}"], 1e-4, 5e-5)
        else:
            close(gr.reshape(-1)[::97], g[f"proj.gsub.{short}"], 1e-4, 5e-5)
    close(pp["projection.1.running_mean"], g["proj.p.1.running_mean"], 1e-5, 1e-7)
    close(pp["projection.4.running_var"], g["proj.p.4.running_var"], 1e-5, 1e-7)


def _stats(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


def test_full_nets():
    g = load_golden("full_nets")
    rng = np.random.default_rng(int(g["x_seed"]))
    draw = lambda *s: T(rng.standard_normal(s).astype(np.float32))  # noqa: E731
    x = draw(2, 1, 32, 32, 32)
    for kind, mk in (("vnet", ON.make_vnet_params), ("unet", ON.make_unet_params)):
        p = mk(int(g[f"{kind}.param_seed"]))
        names = list(ON.trainable(p))
        leaves = {k: p[k].clone().requires_grad_(True) for k in names}
        sdf, logits, feats = ON.forward("vnet" if kind == "vnet" else "unet_3D", x, {**p, **leaves})
        close(logits[..., ::2, ::2, ::2], g[f"{kind}.logits_sub"], 1e-4, 2e-5)
        close(feats, g[f"{kind}.feats"], 1e-4, 2e-5)
        r1, r2 = draw(*logits.shape), draw(*feats.shape)
        obj = (logits * r1).sum() + (feats * r2).sum()
        grads = torch.autograd.grad(obj, [leaves[k] for k in names], allow_unused=True)
        assert list(g[f"{kind}.grad_names"]) == names
        for k, gr, ref in zip(names, grads, g[f"{kind}.grad_stats"]):
            if gr is None:
                assert k.startswith("final.") and not ref.any()
                continue
            np.testing.assert_allclose(_stats(gr), ref, rtol=2e-3, atol=5e-3, err_msg=k)


@pytest.mark.parametrize("kind", ["unet", "vnet"])
def test_step_trace(kind):
    g = load_golden(f"step_{kind}")
    net_type = "unet_3D" if kind == "unet" else "vnet"
    mk = ON.make_unet_params if kind == "unet" else ON.make_vnet_params
    s0, s1 = [int(v) for v in g["seeds"]]
    cfg = OS.StepConfig(net_type=net_type, labeled_bs=int(g["LB"]), feature_scaler=2)
    st = OS.StepState(student=mk(s0), teacher=mk(s1))
    names = list(g["param_names"])
    assert names == list(ON.trainable(st.student))
    for step in range(2):
        out = OS.train_step(cfg, st, T(g[f"s{step}.vol"]), T(g[f"s{step}.label"]).long(), T(g[f"s{step}.noise"]),
                            float(g[f"s{step}.beta"]), int(g[f"s{step}.epoch"]))
        ref = g[f"s{step}.scalars"]
        got = [out["loss"], out["ce"], out["dice"], out["cons"], out["fecl"], out["uncl"], out["cons_weight"], out["grad_norm"]]
        np.testing.assert_allclose([float(v) for v in got], ref, rtol=2e-4, atol=1e-6)
        close(out["s_logits"][..., ::2, ::2, ::2], g[f"s{step}.logits_sub"], 2e-4, 5e-5)
        close(out["t_logits"][..., ::2, ::2, ::2], g[f"s{step}.t_logits_sub"], 2e-4, 5e-5)
        for k, ref_s, ref_t in zip(names, g[f"s{step}.student_stats"], g[f"s{step}.teacher_stats"]):
            np.testing.assert_allclose(_stats(st.student[k]), ref_s, rtol=1e-4, atol=1e-4, err_msg=k)
            np.testing.assert_allclose(_stats(st.teacher[k]), ref_t, rtol=1e-4, atol=1e-4, err_msg=k)
