"""GPU parity, step level: DyconTrainer (HIP path, fp32 storage) against the 2-step traces recorded
from the REFERENCE modules/losses/optimizer (tests/golden/step_{unet,vnet}.npz)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from dycon_paper_replication_amd.engine import DropoutSpec
    from dycon_paper_replication_amd.trainer import DyconTrainer, TrainConfig
from oracle import nets as ON
from oracle import step as OS
from test_engine_gpu import _double, within_budget

DEV = "cuda:0"
T = torch.from_numpy


def _stats(t):
    t = t.detach().double().cpu()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


@pytest.mark.parametrize("kind", ["unet", "vnet"])
def test_two_step_trace_vs_reference(kind):
    g = load_golden(f"step_{kind}")
    net_type = "unet_3D" if kind == "unet" else "vnet"
    mk = ON.make_unet_params if kind == "unet" else ON.make_vnet_params
    s0, s1 = [int(v) for v in g["seeds"]]
    cfg = TrainConfig(model=net_type, labeled_bs=int(g["LB"]), batch_size=int(g["B"]), dtype=torch.float32, feature_scaler=2)
    tr = DyconTrainer(cfg, DEV, student_init=mk(s0), teacher_init=mk(s1))
    names = list(g["param_names"])
    assert names == tr.names
    off = DropoutSpec("off")
    for step in range(2):
        vol = T(g[f"s{step}.vol"]).to(DEV)
        lab = T(g[f"s{step}.label"]).to(DEV)            # uint8 labels
        noise = T(g[f"s{step}.noise"]).to(DEV)
        out = tr.step(vol, lab, noise=noise, s_drop=off, t_drop=off, epoch=int(g[f"s{step}.epoch"]), beta=float(g[f"s{step}.beta"]))
        # fp64 footing: the ".f64" keys are the same trace with every reference module and loss run in double.  The HIP path is
        # held to the north-star's 1e-4 against THAT (the reference's own fp32 trace is up to 3e-4 away from it in step 2:
        # tools/parity_budget.py) and to twice the reference's own fp32 error where that is the larger budget.
        ref, ref32 = g[f"s{step}.scalars.f64"], g[f"s{step}.scalars"]   # loss, ce, dice, cons, fecl, uncl, cons_weight, grad_norm
        got = [float(out[k]) for k in ("loss", "ce", "dice", "cons", "fecl", "uncl")] + [out["cons_weight"], float(out["grad_sumsq"].sqrt())]
        np.testing.assert_allclose(got[:7], ref[:7], rtol=1e-4, atol=1e-6, err_msg=f"loss scalars step {step}")
        np.testing.assert_allclose(got[7], ref[7], rtol=1e-4, err_msg=f"grad norm step {step}")
        lo = out["s_logits"].cpu().permute(0, 4, 1, 2, 3)[..., ::2, ::2, ::2]
        within_budget(lo.numpy(), g[f"s{step}.logits_sub"], g[f"s{step}.logits_sub.f64"], f"student logits step {step}")
        tl = out["t_logits"].cpu().permute(0, 4, 1, 2, 3)[..., ::2, ::2, ::2]
        within_budget(tl.numpy(), g[f"s{step}.t_logits_sub"], g[f"s{step}.t_logits_sub.f64"], f"teacher logits step {step}")
        np.testing.assert_array_equal(out["mask"].cpu().numpy().reshape(g[f"s{step}.mask"].shape), g[f"s{step}.mask"])
        for k, ref_s, ref_t in zip(names, g[f"s{step}.student_stats.f64"], g[f"s{step}.teacher_stats.f64"]):
            np.testing.assert_allclose(_stats(tr.p[k]), ref_s, rtol=1e-4, atol=1e-4, err_msg=f"student {k} step {step}")
            np.testing.assert_allclose(_stats(tr.t[k]), ref_t, rtol=1e-4, atol=1e-4, err_msg=f"teacher {k} step {step}")
    assert tr.iter_num == 2 and tr.skipped_steps == 0
    # state_dict contract: reference keys, shapes; loadable into the oracle's functional net
    sd = tr.state_dict()
    ref_sd = mk(s0)
    assert list(sd) == list(ref_sd)
    assert all(tuple(sd[k].shape) == tuple(ref_sd[k].shape) for k in sd)


def test_nan_guard_skips_update():
    cfg = TrainConfig(model="vnet", labeled_bs=1, batch_size=2, dtype=torch.float32)
    tr = DyconTrainer(cfg, DEV)
    vol = torch.randn(2, 1, 16, 16, 16, device=DEV)
    vol[0, 0, 0, 0, 0] = float("nan")
    lab = torch.zeros(2, 16, 16, 16, dtype=torch.int64, device=DEV)
    before = tr.flat_p.clone()
    tb = tr.flat_t.clone()
    out = tr.step(vol, lab)
    assert out["skipped"] and tr.iter_num == 0 and tr.skipped_steps == 1
    assert torch.equal(before, tr.flat_p) and torch.equal(tb, tr.flat_t)


def test_bf16_step_tracks_fp32():
    """bf16 storage mode stays close to the fp32 parity mode over a few steps (loss components)."""
    torch.manual_seed(0)
    outs = {}
    for dt in (torch.float32, torch.bfloat16):
        cfg = TrainConfig(model="vnet", labeled_bs=1, batch_size=2, dtype=dt, seed=7)
        tr = DyconTrainer(cfg, DEV)
        gen = torch.Generator(device=DEV).manual_seed(5)
        rows = []
        for i in range(3):
            vol = torch.randn(2, 1, 32, 32, 32, device=DEV, generator=gen)
            lab = (torch.rand(2, 32, 32, 32, device=DEV, generator=gen) > 0.7).long()
            noise = torch.clamp(torch.randn(2, 1, 32, 32, 32, device=DEV, generator=gen) * 0.1, -0.2, 0.2)
            o = tr.step(vol, lab, noise=noise, s_drop=DropoutSpec("off"), t_drop=DropoutSpec("off"))
            rows.append([float(o[k]) for k in ("loss", "ce", "dice", "cons", "fecl", "uncl")])
        outs[dt] = np.array(rows)
    np.testing.assert_allclose(outs[torch.bfloat16], outs[torch.float32], rtol=5e-2, atol=5e-3)


def test_philox_step_runs_and_is_reproducible():
    res = []
    for _ in range(2):
        cfg = TrainConfig(model="vnet", labeled_bs=1, batch_size=2, dtype=torch.bfloat16, seed=3)
        tr = DyconTrainer(cfg, DEV)
        gen = torch.Generator(device=DEV).manual_seed(1)
        vol = torch.randn(2, 1, 32, 32, 32, device=DEV, generator=gen)
        lab = (torch.rand(2, 32, 32, 32, device=DEV, generator=gen) > 0.7).long()
        o = tr.step(vol, lab)
        o = tr.step(vol, lab)
        res.append((float(o["loss"]), tr.flat_p.clone()))
    assert np.isfinite(res[0][0])
    assert res[0][0] == pytest.approx(res[1][0], rel=1e-5)


def test_checkpoint_resume_continues_the_run(tmp_path):
    """save (reference-format student .pth + resume sidecar) after 2 steps, load into a fresh trainer, run 2 more:
    the same as 4 uninterrupted steps (on-device Philox streams are functions of the iteration counter); the .pth
    alone loads into the oracle's functional net (reference key contract)."""
    from dycon_paper_replication_amd.synthetic import make_batch
    vol, lab, _ = make_batch(11, 2, (32, 32, 32))
    vol, lab = vol.cuda(), lab.cuda()
    cfg = TrainConfig(model="vnet", labeled_bs=1, batch_size=2, dtype=torch.bfloat16, seed=5)
    a = DyconTrainer(cfg, "cuda:0")
    for _ in range(2):
        a.step(vol, lab)
    path = tmp_path / "iter_2.pth"
    a.save_checkpoint(path)
    for _ in range(2):
        a.step(vol, lab)
    b = DyconTrainer(cfg, "cuda:0")
    b.load_checkpoint(path)
    assert b.iter_num == 2
    for _ in range(2):
        out = b.step(vol, lab)
    # the kernels are deterministic up to the double-precision atomic loss sums (their last bits can differ between runs)
    for x, y in ((a.flat_p, b.flat_p), (a.flat_t, b.flat_t), (a.flat_m, b.flat_m)):
        assert float((x - y).abs().max()) <= 1e-6 * float(x.abs().max())
    assert np.isfinite(float(out["loss"]))
    sd = torch.load(path, map_location="cpu", weights_only=True)
    assert list(sd) == list(a.state_dict())


def test_self_consistency_dice_after_k_steps():
    """BASELINE.json's metric asks for "Dice vs ref"; without the datasets the measurable form is SURVEY 8d's self-consistency
    Dice: train the HIP path (fp32 storage, and bf16) and the CPU oracle for K identical steps (same init, batches, noise; dropout
    off) and compare the segmentations both students then predict for a held-out synthetic volume.  The reference tolerance on
    real data is +-0.2 Dice; here the two label maps must agree to Dice >= 0.99 (fp32) / 0.95 (bf16) and the class-1
    probabilities (max over the volume; an untrained net sits near 0.5 everywhere, the worst case for label flips) to 5e-3 / 1.5e-1 (mean 1e-3 / 1e-2)."""
    from dycon_paper_replication_amd.synthetic import make_batch
    from dycon_paper_replication_amd.utils.test_3d_patch import overlap_counts
    K, shape = 4, (32, 32, 32)
    batches = [make_batch(100 + i, 2, shape) for i in range(K)]
    held, _, _ = make_batch(999, 1, shape)
    cfg_o = OS.StepConfig(net_type="vnet", labeled_bs=1)
    st = OS.StepState(student=ON.make_vnet_params(21), teacher=ON.make_vnet_params(22))
    for vol, lab, noise in batches:
        OS.train_step(cfg_o, st, vol, lab, noise, 5.0, 0)
    with torch.no_grad():
        ref_prob = torch.softmax(ON.vnet_forward(held, st.student, bn_training=False)[1], 1)[0, 1]
    ref_lab = (ref_prob > 0.5)
    off = DropoutSpec("off")
    for dt, dice_min, ptol in ((torch.float32, 0.99, 5e-3), (torch.bfloat16, 0.95, 1.5e-1)):
        tr = DyconTrainer(TrainConfig(model="vnet", labeled_bs=1, batch_size=2, dtype=dt), DEV,
                          student_init=ON.make_vnet_params(21), teacher_init=ON.make_vnet_params(22))
        for vol, lab, noise in batches:
            tr.step(vol.to(DEV), lab.to(DEV), noise=noise.to(DEV), s_drop=off, t_drop=off, epoch=0, beta=5.0)
        tr.model.eval()
        with torch.no_grad():
            prob = torch.softmax(tr.model(held.to(DEV))[1].float(), 1)[0, 1]
        tr.model.train()
        err = float((prob.cpu() - ref_prob).abs().max())
        mean_err = float((prob.cpu() - ref_prob).abs().mean())
        print(f"self-consistency {dt}: max |dp| = {err:.2e}, mean |dp| = {mean_err:.2e}")
        assert mean_err < (1e-3 if dt == torch.float32 else 1e-2)
        assert err < ptol, f"class-1 probabilities drift {err:.2e} ({dt})"
        n_p, n_g, n_i = overlap_counts((prob > 0.5).to(torch.uint8).contiguous(), ref_lab.to(torch.uint8).to(DEV))
        dice = 2.0 * n_i / (n_p + n_g) if n_p + n_g else 1.0
        print(f"self-consistency {dt}: Dice = {dice:.4f}")
        assert dice >= dice_min, f"self-consistency Dice {dice:.4f} ({dt})"


@pytest.mark.parametrize("net", ["vnet", "unet_3D"])
def test_isles_variants_vs_oracle(net):
    """The ISLES / kl variants of the step (train_DyCON_ISLES22.py:114,247,322-324; --consistency_type kl): multi-class DiceLoss,
    eval-mode teacher (BatchNorm running statistics, no dropout), KL consistency, poly learning rate, feature_scaler 4 -- HIP path
    (fp32 storage) against the oracle step run in DOUBLE over 3 steps: loss terms, parameters 1e-4; learning rate schedule exact."""
    from dycon_paper_replication_amd.synthetic import make_batch
    shape = (32, 32, 32)
    mk = ON.make_vnet_params if net == "vnet" else ON.make_unet_params
    cfg_o = OS.StepConfig(net_type=net, labeled_bs=1, feature_scaler=4, consistency_type="kl", dice_variant="multiclass",
                          teacher_bn_training=False, poly_lr_max_iter=50)
    st = OS.StepState(student=_double(mk(31)), teacher=_double(mk(32)))      # the oracle step in DOUBLE: the true values
    tr = DyconTrainer(TrainConfig(model=net, labeled_bs=1, batch_size=2, dtype=torch.float32, feature_scaler=4, consistency_type="kl",
                                  dice_variant="multiclass", teacher_mode="eval", poly_lr=True, max_iterations=50), DEV,
                      student_init=mk(31), teacher_init=mk(32))
    off = DropoutSpec("off")
    for i in range(3):
        vol, lab, noise = make_batch(300 + i, 2, shape)
        ref = OS.train_step(cfg_o, st, vol.double(), lab, noise.double(), 2.5, 7)
        out = tr.step(vol.to(DEV), lab.to(DEV), noise=noise.to(DEV), s_drop=off, t_drop=off, epoch=7, beta=2.5)
        got = np.array([float(out[k]) for k in ("loss", "ce", "dice", "cons", "fecl", "uncl")])
        exp = np.array([float(ref[k]) for k in ("loss", "ce", "dice", "cons", "fecl", "uncl")])
        np.testing.assert_allclose(got, exp, rtol=1e-4, atol=1e-6, err_msg=f"step {i}")
        assert tr.lr == pytest.approx(st.lr, rel=1e-12)
    for k in ("block_one.conv.0.weight", "block_five.conv.0.weight", "out_conv.weight") if net == "vnet" else \
            ("conv1.conv1.0.weight", "center.conv1.0.weight", "out_conv2.weight"):
        np.testing.assert_allclose(tr.p[k].cpu().numpy(), st.student[k].numpy(), rtol=1e-4, atol=2e-6, err_msg=k)
        np.testing.assert_allclose(tr.t[k].cpu().numpy(), st.teacher[k].numpy(), rtol=1e-4, atol=2e-6, err_msg="teacher " + k)


@pytest.mark.parametrize("net", ["vnet", "unet_3D"])
def test_replay_equals_eager(net):
    """cfg.replay: steps 1-2 run eagerly, step 3 is recorded, steps 4+ re-issue the recorded launch list with patched scalars
    (Philox offsets, schedules) and the batch copied into the recorded input buffers.  Seven steps with changing batches must
    give the same parameters, teacher and loss sequence as seven eager steps (on-device Philox dropout and noise active)."""
    from dycon_paper_replication_amd.synthetic import make_batch
    batches = [make_batch(500 + i, 2, (32, 32, 32)) for i in range(7)]
    runs = {}
    for replay in (False, True):
        tr = DyconTrainer(TrainConfig(model=net, labeled_bs=1, batch_size=2, dtype=torch.bfloat16, seed=9, replay=replay), DEV)
        losses = []
        for vol, lab, _ in batches:
            out = tr.step(vol.to(DEV), lab.to(DEV))
            losses.append([float(out[k]) for k in ("loss", "ce", "dice", "cons", "fecl", "uncl")])
        assert (tr._rp is not None) == replay and tr.iter_num == 7
        runs[replay] = (np.array(losses), tr.flat_p.clone(), tr.flat_t.clone(), tr.flat_m.clone())
    np.testing.assert_allclose(runs[True][0], runs[False][0], rtol=1e-5, atol=1e-6)
    for a, b in zip(runs[True][1:], runs[False][1:]):
        assert float((a - b).abs().max()) <= 1e-6 * float(b.abs().max())
