"""GPU, the drop-in boundary: the reference's loss callables with the reference's argument meaning, and the
reference's training-loop body (code/train_DyCON_BraTS19.py:298-372) restated here against THIS package's imports only --
``net_factory_3d``, ``dycon_losses.*``, ``losses.*``, ``ramps.*`` -- plus the torch calls the script itself makes
(F.softmax, F.cross_entropy, F.normalize, F.avg_pool3d, torch.optim.SGD, clip_grad_norm_, the EMA loop).  Checked against the
2-step traces recorded from the imported reference (tests/golden/step_{unet,vnet}.npz).  This is the autograd route of
INTEGRATION.md section 1: _NetFunction.backward, _UnCLFunction, _FeCLFunction, the loss Functions of utils/losses.py."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from dycon_paper_replication_amd.networks.net_factory_3d import net_factory_3d
    from dycon_paper_replication_amd.utils import dycon_losses, losses, ramps
from oracle import nets as ON
from test_engine_gpu import within_budget

DEV = "cuda:0"
T = torch.from_numpy


def close(a, b, rtol=1e-4, atol=1e-6, msg=""):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().float().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=msg)


@pytest.mark.parametrize("layout", ["ncdhw", "channels_last"])
def test_reference_loss_callables_golden(layout):
    """dice_loss / DiceLoss / softmax_mse_loss / softmax_kl_loss called exactly as the reference's step calls them (on
    probabilities the SCRIPT computed with torch), values and gradients w.r.t. the logits against the reference's own outputs."""
    g = load_golden("voxel_losses")
    a, b, lab = T(g["a"]).to(DEV), T(g["b"]).to(DEV), T(g["label"]).to(DEV)
    if layout == "channels_last":
        a = a.contiguous(memory_format=torch.channels_last_3d)
        b = b.contiguous(memory_format=torch.channels_last_3d)

    def grad_of(fn):
        x = a.clone().requires_grad_(True)
        v = fn(x)
        return v, torch.autograd.grad(v, x)[0]

    v, gx = grad_of(lambda x: losses.dice_loss(F.softmax(x, 1)[:, 1], lab == 1))                  # train_DyCON_BraTS19.py:314
    close(v, g["dice"], 1e-5); close(gx, g["dice_grad"], 1e-4, 1e-8)
    v, gx = grad_of(lambda x: losses.DiceLoss(2)(F.softmax(x, 1), lab.unsqueeze(1)))              # train_DyCON_ISLES22.py:247
    close(v, g["dice_mc"], 1e-5); close(gx, g["dice_mc_grad"], 1e-4, 1e-8)
    v, gx = grad_of(lambda x: losses.DiceLoss(2)(x, lab.unsqueeze(1), softmax=True))              # softmax inside the kernel
    close(v, g["dice_mc"], 1e-5); close(gx, g["dice_mc_grad"], 1e-4, 1e-8)
    v, gx = grad_of(lambda x: losses.DiceLoss(2)(F.softmax(x, 1), lab.unsqueeze(1), weight=[0.0, 2.0]))   # class 1 only == dice_loss
    close(v, g["dice"], 1e-5); close(gx, g["dice_grad"], 1e-4, 1e-8)
    pb = F.softmax(b, 1)
    v, gx = grad_of(lambda x: losses.softmax_mse_loss(F.softmax(x, 1), pb).mean())                # :352
    close(v, g["cons_mse"], 1e-5, 1e-8); close(gx, g["cons_mse_grad"], 1e-4, 1e-9)
    v, gx = grad_of(lambda x: losses.softmax_kl_loss(F.softmax(x, 1), pb))
    close(v, g["cons_kl"], 1e-4, 1e-8); close(gx, g["cons_kl_grad"], 1e-4, 1e-9)
    elem = losses.softmax_mse_loss(a, b)
    assert elem.shape == a.shape
    close(elem, g["mse_elem"], 1e-5, 1e-7)
    # slices, as the step passes them (probs[LB:]), and the gradient of the SECOND argument / the sigmoid variants vs torch
    x = a.clone().requires_grad_(True)
    y = b.clone().requires_grad_(True)
    for sig in (False, True):
        act = (lambda t: torch.sigmoid(t)) if sig else (lambda t: F.softmax(t, 1))
        r = torch.randn_like(a[1:])
        got = torch.autograd.grad((losses.softmax_mse_loss(x[1:], y[1:], sigmoid=sig) * r).sum(), [x, y])
        ref = torch.autograd.grad((((act(x[1:]) - act(y[1:])) ** 2) * r).sum(), [x, y])
        for gg, rr in zip(got, ref):
            close(gg, rr, 1e-4, 1e-6 * float(rr.abs().max()))
        got = torch.autograd.grad(losses.softmax_kl_loss(x, y, sigmoid=sig), [x, y])
        lp = torch.log(torch.sigmoid(x)) if sig else F.log_softmax(x, 1)
        ref = torch.autograd.grad(F.kl_div(lp, act(y), reduction="mean"), [x, y])
        for gg, rr in zip(got, ref):
            close(gg, rr, 1e-4, 1e-6 * float(rr.abs().max()))


def _ref_step_body(model, ema_model, optimizer, uncl_criterion, fecl_criterion, volume_batch, label_batch, noise, labeled_bs,
                   iter_num, epoch_num, beta, feature_scaler=2, consistency=0.1, consistency_rampup=200.0, l_weight=1.0,
                   u_weight=0.5, ema_decay=0.99):
    """train_DyCON_BraTS19.py:298-372, line for line; only the noise is an argument (the fixture's draw) instead of randn_like."""
    consistency_criterion = losses.softmax_mse_loss
    ema_inputs = volume_batch + noise
    _, stud_logits, stud_features = model(volume_batch)
    with torch.no_grad():
        _, ema_logits, ema_features = ema_model(ema_inputs)
    stud_probs = F.softmax(stud_logits, dim=1)
    ema_probs = F.softmax(ema_logits, dim=1)
    consistency_weight = consistency * ramps.sigmoid_rampup(iter_num // 150, consistency_rampup)
    loss_seg = F.cross_entropy(stud_logits[:labeled_bs], label_batch[:labeled_bs])
    loss_seg_dice = losses.dice_loss(stud_probs[:labeled_bs, 1, :, :, :], label_batch[:labeled_bs] == 1)
    B, C, _, _, _ = stud_features.shape
    stud_embedding = stud_features.view(B, C, -1)
    stud_embedding = torch.transpose(stud_embedding, 1, 2)
    stud_embedding = F.normalize(stud_embedding, dim=-1)
    ema_embedding = ema_features.view(B, C, -1)
    ema_embedding = torch.transpose(ema_embedding, 1, 2)
    ema_embedding = F.normalize(ema_embedding, dim=-1)
    mask_con = F.avg_pool3d(label_batch.float(), kernel_size=feature_scaler * 4, stride=feature_scaler * 4)
    mask_con = (mask_con > 0.5).float()
    mask_con = mask_con.reshape(B, -1)
    mask_con = mask_con.unsqueeze(1)
    f_loss = fecl_criterion(feat=stud_embedding, mask=mask_con, teacher_feat=ema_embedding, gambling_uncertainty=None, epoch=epoch_num)
    u_loss = uncl_criterion(stud_logits, ema_logits, beta)
    consistency_loss = consistency_criterion(stud_probs[labeled_bs:], ema_probs[labeled_bs:]).mean()
    loss = l_weight * (loss_seg + loss_seg_dice) + consistency_weight * consistency_loss + u_weight * (f_loss + u_loss)
    assert not (torch.isnan(loss) or torch.isinf(loss))
    optimizer.zero_grad()
    loss.backward()
    gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
    optimizer.step()
    alpha = min(1 - 1 / (iter_num + 1), ema_decay)                                         # update_ema_variables, :155-164
    for ema_param, param in zip(ema_model.parameters(), model.parameters()):
        ema_param.data.mul_(alpha).add_(param.data, alpha=1 - alpha)
    return dict(loss=loss, ce=loss_seg, dice=loss_seg_dice, cons=consistency_loss, fecl=f_loss, uncl=u_loss, cw=consistency_weight,
                gnorm=gnorm, s_logits=stud_logits, t_logits=ema_logits, mask=mask_con)


def _stats(t):
    t = t.detach().double().cpu()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


@pytest.mark.parametrize("kind", ["unet", "vnet"])
def test_reference_step_body_runs_on_this_package(kind):
    g = load_golden(f"step_{kind}")
    net_type = "unet_3D" if kind == "unet" else "vnet"
    mk = ON.make_unet_params if kind == "unet" else ON.make_vnet_params
    s0, s1 = [int(v) for v in g["seeds"]]
    LB = int(g["LB"])

    def create_model(seed, ema=False):                                                     # train_DyCON_BraTS19.py:212-230
        net = net_factory_3d(net_type=net_type, in_chns=1, class_num=2, scaler=2)
        net.load_state_dict(mk(seed))
        net = net.to(DEV)
        net.has_dropout = False               # the fixture was drawn with dropout p = 0 (no shared RNG stream with torch's CPU generator)
        if ema:
            for p in net.parameters():
                p.detach_()
        return net

    model, ema_model = create_model(s0), create_model(s1, ema=True)
    model.train(); ema_model.train()
    optimizer = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=0.0001)
    uncl_criterion = dycon_losses.UnCLoss()
    fecl_criterion = dycon_losses.FeCLoss(device=DEV, temperature=0.6, gamma=2.0, use_focal=True, rampup_epochs=1500)
    names = [k for k, _ in model.named_parameters()]
    assert names == list(g["param_names"])
    for step in range(2):
        vol, lab = T(g[f"s{step}.vol"]).to(DEV), T(g[f"s{step}.label"]).long().to(DEV)
        noise = T(g[f"s{step}.noise"]).to(DEV)
        out = _ref_step_body(model, ema_model, optimizer, uncl_criterion, fecl_criterion, vol, lab, noise, LB, step,
                             int(g[f"s{step}.epoch"]), float(g[f"s{step}.beta"]))
        ref = g[f"s{step}.scalars.f64"]   # loss, ce, dice, cons, fecl, uncl, cons_weight, grad_norm -- the trace's fp64 twin
        got = [float(out[k]) for k in ("loss", "ce", "dice", "cons", "fecl", "uncl")] + [out["cw"], float(out["gnorm"])]
        np.testing.assert_allclose(got[:7], ref[:7], rtol=1e-4, atol=1e-6, err_msg=f"loss scalars step {step}")
        np.testing.assert_allclose(got[7], ref[7], rtol=1e-4, err_msg=f"grad norm step {step}")
        np.testing.assert_array_equal(out["mask"].cpu().numpy().reshape(g[f"s{step}.mask"].shape), g[f"s{step}.mask"])
        within_budget(out["s_logits"].detach().cpu()[..., ::2, ::2, ::2].numpy(), g[f"s{step}.logits_sub"], g[f"s{step}.logits_sub.f64"],
                      f"student logits step {step}")
        within_budget(out["t_logits"].cpu()[..., ::2, ::2, ::2].numpy(), g[f"s{step}.t_logits_sub"], g[f"s{step}.t_logits_sub.f64"],
                      f"teacher logits step {step}")
        sp, tp = dict(model.named_parameters()), dict(ema_model.named_parameters())
        for k, ref_s, ref_t in zip(names, g[f"s{step}.student_stats.f64"], g[f"s{step}.teacher_stats.f64"]):
            np.testing.assert_allclose(_stats(sp[k]), ref_s, rtol=1e-4, atol=1e-4, err_msg=f"student {k} step {step}")
            np.testing.assert_allclose(_stats(tp[k]), ref_t, rtol=1e-4, atol=1e-4, err_msg=f"teacher {k} step {step}")
        if kind == "unet":
            assert sp["final.weight"].grad is None       # the discarded tanh head: grad stays None, SGD / clip skip it (as the reference)
