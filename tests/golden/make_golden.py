#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference
(rogeliorjr/DyCON_Paper_Replication at /root/reference) on CPU.

Runs only in the build container (the reference never travels to the GPU box).
Recipe: register empty stub packages for ``networks`` and
``utils`` so that the few hot-path modules import without executing the package
``__init__`` files that need monai/h5py/medpy.

What is stored is DATA only: seeded inputs, weights (small layers) or the seed of the
numpy generator that rebuilds them (full nets), and the reference's outputs / gradients /
loss values.  No reference source text is stored.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz, *.json
"""
import importlib
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/code"

for pkg in ("networks", "utils"):
    m = types.ModuleType(pkg)
    m.__path__ = [f"{REF}/{pkg}"]
    sys.modules[pkg] = m
ref_dycon = importlib.import_module("utils.dycon_losses")
ref_losses = importlib.import_module("utils.losses")
ref_ramps = importlib.import_module("utils.ramps")
ref_vnet = importlib.import_module("networks.VNet")
ref_unet = importlib.import_module("networks.UNet3D_contrastive")
ref_blocks = importlib.import_module("networks.utils")
ref_factory = importlib.import_module("networks.net_factory_3d")

from oracle import nets as onets  # noqa: E402  (parameter builders + synthetic inputs only)

torch.set_num_threads(8)
META = {"torch": torch.__version__, "numpy": np.__version__, "reference": "rogeliorjr/DyCON_Paper_Replication@2025-08-08"}


def rng_t(rng, *shape, scale=1.0):
    return torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32))


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"  {name}.npz  {sum(a.nbytes for a in out.values()) / 1024:.0f} KiB")


def blob_labels(rng, B, D, H, W):
    """1-3 random ellipsoids per volume (SURVEY.md 8d synthetic labels)."""
    zz, yy, xx = np.meshgrid(np.arange(D), np.arange(H), np.arange(W), indexing="ij")
    lab = np.zeros((B, D, H, W), np.int64)
    for b in range(B):
        for _ in range(int(rng.integers(1, 4))):
            c = rng.uniform(0.25, 0.75, 3) * (D, H, W)
            r = rng.uniform(0.12, 0.3, 3) * (D, H, W)
            lab[b] |= (((zz - c[0]) / r[0]) ** 2 + ((yy - c[1]) / r[1]) ** 2 + ((xx - c[2]) / r[2]) ** 2 <= 1)
    return torch.from_numpy(lab)


def stats(t):
    if t is None:          # parameter unused by the loss (e.g. UNet3D.final in the training step)
        return np.zeros(3)
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


# ----------------------------------------------------------------------------- schedules
def gen_schedules():
    tab = {"meta": META, "adaptive_beta": [], "threshold_rampup": [], "consistency_rampup": []}
    for e, tot in [(0, 100), (1, 3334), (50, 100), (3333, 3334), (100, 100)]:
        for mx, mn in [(5.0, 0.5), (2.0, 0.1)]:
            tab["adaptive_beta"].append([e, tot, mx, mn, ref_dycon.adaptive_beta(e, tot, mx, mn)])
    for e in [0, 1, 375, 750, 1499, 1500, 3000, -3]:
        for R in [1500, 2000, 0]:
            for lo, hi in [(0.3, 0.5), (1.3, 1.5)]:
                tab["threshold_rampup"].append([e, R, lo, hi, ref_dycon.sigmoid_rampup(e, R, lo, hi)])
    for c in [0, 1, 13, 66, 133, 199, 200, 500]:
        for R in [200.0, 40.0, 0]:
            tab["consistency_rampup"].append([c, R, ref_ramps.sigmoid_rampup(c, R)])
    with open(os.path.join(HERE, "schedules.json"), "w") as f:
        json.dump(tab, f, indent=1)
    print("  schedules.json")


# ----------------------------------------------------------------------------- UnCL
def gen_uncl():
    rng = np.random.default_rng(101)
    crit = ref_dycon.UnCLoss()
    for tag, shape, scale in [("a", (2, 2, 8, 8, 8), 1.0), ("b", (1, 2, 16, 16, 16), 3.0), ("c", (3, 2, 4, 6, 10), 8.0)]:
        s = rng_t(rng, *shape, scale=scale)
        t = rng_t(rng, *shape, scale=scale)
        out = {"s": s, "t": t, "betas": np.array([0.5, 2.5, 5.0])}
        for i, beta in enumerate([0.5, 2.5, 5.0]):
            s32 = s.clone().requires_grad_(True)
            l32 = crit(s32, t, beta)
            (g32,) = torch.autograd.grad(l32, s32)
            l64 = crit(s.double(), t.double(), beta)
            out[f"loss{i}"] = l32
            out[f"grad{i}"] = g32
            out[f"loss64_{i}"] = l64
        save(f"uncl_{tag}", **out)


# ----------------------------------------------------------------------------- FeCL
def gen_fecl():
    rng = np.random.default_rng(202)

    def unit(*shape):
        return F.normalize(rng_t(rng, *shape), dim=-1)

    cases = []
    # (tag, B, N, D, mask kind)
    for tag, B, N, D, kind in [("small", 2, 64, 16, "random"), ("mid", 1, 216, 256, "blob"),
                               ("oneclass", 1, 48, 32, "ones"), ("singleton", 2, 40, 24, "singleton"),
                               ("ragged", 3, 37, 20, "random")]:
        feat = unit(B, N, D)
        # teacher close to student so that hard negatives exist above the 0.3..0.5 threshold
        teach = F.normalize(feat + 0.35 * rng_t(rng, B, N, D) / np.sqrt(D), dim=-1)
        if kind == "random":
            mask = torch.from_numpy(rng.integers(0, 2, (B, 1, N)).astype(np.float32))
        elif kind == "blob":
            mask = torch.zeros(B, 1, N)
            mask[:, :, 50:90] = 1
        elif kind == "ones":
            mask = torch.ones(B, 1, N)
        else:
            mask = torch.zeros(B, 1, N)
            mask[:, :, 7] = 1
        # make a few features strongly correlated across classes (hard negatives / large logits)
        if kind in ("random", "blob", "singleton"):
            feat[:, 1] = F.normalize(feat[:, 0] + 0.05 * feat[:, 1], dim=-1)
            teach[:, 1] = feat[:, 0]
        gamb = torch.from_numpy(rng.uniform(0.0, 0.7, (B, N)).astype(np.float32))
        cases.append((tag, feat, teach, mask, gamb))

    for tag, feat, teach, mask, gamb in cases:
        out = {"feat": feat, "teacher": teach, "mask": mask, "gambling": gamb}
        idx = 0
        cfgs = []
        for epoch in (0, 750, 1500, 3000):
            for focal in (False, True):
                for use_t in (False, True):
                    cfgs.append((epoch, focal, use_t, False))
        cfgs.append((100, True, True, True))
        cfgs.append((100, False, False, True))
        if tag == "mid":            # big gradients: keep the fixture small
            cfgs = [(0, False, False, False), (750, True, True, False), (3000, True, True, False)]
        for epoch, focal, use_t, use_g in cfgs:
            crit = ref_dycon.FeCLoss(device="cpu", temperature=0.6, gamma=2.0, use_focal=focal, rampup_epochs=1500)
            f32 = feat.clone().requires_grad_(True)
            loss = crit(f32, mask, teach if use_t else None, gamb if use_g else None, epoch)
            (g,) = torch.autograd.grad(loss, f32)
            crit64 = ref_dycon.FeCLoss(device="cpu", temperature=0.6, gamma=2.0, use_focal=focal, rampup_epochs=1500)
            # the reference builds torch.eye in default dtype; run the fp64 twin under a dtype switch
            torch.set_default_dtype(torch.float64)
            try:
                l64 = crit64(feat.double(), mask.double(), teach.double() if use_t else None,
                             gamb.double() if use_g else None, epoch)
            finally:
                torch.set_default_dtype(torch.float32)
            out[f"cfg{idx}"] = np.array([epoch, int(focal), int(use_t), int(use_g)])
            out[f"loss{idx}"] = loss
            out[f"loss64_{idx}"] = l64
            out[f"grad{idx}"] = g
            idx += 1
        out["n_cfg"] = np.array(idx)
        save(f"fecl_{tag}", **out)


# ----------------------------------------------------------------------------- voxel losses
def gen_voxel_losses():
    rng = np.random.default_rng(303)
    a = rng_t(rng, 3, 2, 6, 8, 10, scale=2.0)
    b = rng_t(rng, 3, 2, 6, 8, 10, scale=2.0)
    lab = torch.from_numpy(rng.integers(0, 2, (3, 6, 8, 10)).astype(np.int64))
    out = {"a": a, "b": b, "label": lab}
    x = a.clone().requires_grad_(True)
    p = F.softmax(x, 1)
    d = ref_losses.dice_loss(p[:, 1], lab == 1)
    out["dice"], out["dice_grad"] = d, torch.autograd.grad(d, x)[0]
    x = a.clone().requires_grad_(True)
    ce = F.cross_entropy(x, lab)
    out["ce"], out["ce_grad"] = ce, torch.autograd.grad(ce, x)[0]
    x = a.clone().requires_grad_(True)
    dm = ref_losses.DiceLoss(2)(F.softmax(x, 1), lab.unsqueeze(1))
    out["dice_mc"], out["dice_mc_grad"] = dm, torch.autograd.grad(dm, x)[0]
    # consistency: the step passes PROBABILITIES into softmax_mse_loss, which softmaxes again
    x = a.clone().requires_grad_(True)
    mse = ref_losses.softmax_mse_loss(F.softmax(x, 1), F.softmax(b, 1)).mean()
    out["cons_mse"], out["cons_mse_grad"] = mse, torch.autograd.grad(mse, x)[0]
    x = a.clone().requires_grad_(True)
    kl = ref_losses.softmax_kl_loss(F.softmax(x, 1), F.softmax(b, 1))
    out["cons_kl"], out["cons_kl_grad"] = kl, torch.autograd.grad(kl, x)[0]
    out["mse_elem"] = ref_losses.softmax_mse_loss(a, b)
    save("voxel_losses", **out)


# ----------------------------------------------------------------------------- layers
def _layer_case(mod, x, r):
    x = x.clone().requires_grad_(True)
    y = mod(x)
    grads = torch.autograd.grad((y * r).sum(), [x] + list(mod.parameters()))
    out = {"y": y, "gx": grads[0]}
    for (k, _), g in zip(mod.named_parameters(), grads[1:]):
        out["g." + k] = g
    for k, v in mod.state_dict().items():
        out["p." + k] = v
    return out


def _randomize(mod, rng):
    with torch.no_grad():
        for p in mod.parameters():
            if p.dim() == 1:
                p.copy_(torch.from_numpy((1.0 * (p.detach().numpy() != 0) + 0.2 * rng.standard_normal(p.shape)).astype(np.float32)))


def gen_vnet_layers():
    rng = np.random.default_rng(404)
    torch.manual_seed(404)
    out = {}
    for norm in ("groupnorm", "none", "instancenorm", "batchnorm"):
        co = 32 if norm == "groupnorm" else 16
        blk = ref_vnet.ConvBlock(2, 16, co, normalization=norm)
        _randomize(blk, rng)
        x = rng_t(rng, 2, 16, 4, 6, 10)
        r = rng_t(rng, 2, co, 4, 6, 10)
        c = _layer_case(blk, x, r)
        out.update({f"convblock_{norm}.{k}": v for k, v in c.items()})
        out[f"convblock_{norm}.x"], out[f"convblock_{norm}.r"] = x, r
    dw = ref_vnet.DownsamplingConvBlock(16, 32, normalization="groupnorm")
    _randomize(dw, rng)
    x, r = rng_t(rng, 2, 16, 8, 6, 10), rng_t(rng, 2, 32, 4, 3, 5)
    out.update({f"down.{k}": v for k, v in _layer_case(dw, x, r).items()})
    out["down.x"], out["down.r"] = x, r
    up = ref_vnet.UpsamplingDeconvBlock(32, 16, normalization="groupnorm")
    _randomize(up, rng)
    x, r = rng_t(rng, 2, 32, 4, 3, 5), rng_t(rng, 2, 16, 8, 6, 10)
    out.update({f"up.{k}": v for k, v in _layer_case(up, x, r).items()})
    out["up.x"], out["up.r"] = x, r
    # first layer (1 input channel) and 1x1 head
    b1 = ref_vnet.ConvBlock(1, 1, 16, normalization="groupnorm")
    _randomize(b1, rng)
    x, r = rng_t(rng, 2, 1, 8, 8, 12), rng_t(rng, 2, 16, 8, 8, 12)
    out.update({f"first.{k}": v for k, v in _layer_case(b1, x, r).items()})
    out["first.x"], out["first.r"] = x, r
    oc = nn.Conv3d(16, 2, 1)
    x, r = rng_t(rng, 2, 16, 4, 6, 8), rng_t(rng, 2, 2, 4, 6, 8)
    out.update({f"outconv.{k}": v for k, v in _layer_case(oc, x, r).items()})
    out["outconv.x"], out["outconv.r"] = x, r
    # Dropout3d with an explicit mask: y = x * keep / (1-p)   (nn.Dropout3d(0.5), VNet.py:177)
    save("vnet_layers", **out)


def gen_unet_layers():
    rng = np.random.default_rng(505)
    torch.manual_seed(505)
    out = {}
    uc = ref_blocks.UnetConv3(8, 16, True, kernel_size=(3, 3, 3), padding_size=(1, 1, 1))
    x, r = rng_t(rng, 2, 8, 6, 8, 10), rng_t(rng, 2, 16, 6, 8, 10)
    out.update({f"unetconv.{k}": v for k, v in _layer_case(uc, x, r).items()})
    out["unetconv.x"], out["unetconv.r"] = x, r
    upc = ref_blocks.UnetUp3_CT(32, 16, True)
    skip = rng_t(rng, 1, 16, 8, 12, 10).requires_grad_(True)
    low = rng_t(rng, 1, 32, 4, 6, 5).requires_grad_(True)
    r = rng_t(rng, 1, 16, 8, 12, 10)
    y = upc(skip, low)
    grads = torch.autograd.grad((y * r).sum(), [skip, low] + list(upc.parameters()))
    out.update({"upcat.skip": skip, "upcat.low": low, "upcat.r": r, "upcat.y": y,
                "upcat.gskip": grads[0], "upcat.glow": grads[1]})
    for (k, _), g in zip(upc.named_parameters(), grads[2:]):
        out["upcat.g." + k] = g
    for k, v in upc.state_dict().items():
        out["upcat.p." + k] = v
    # max pool (ties after ReLU matter for the gradient routing)
    x = F.relu(rng_t(rng, 2, 4, 8, 6, 10)).requires_grad_(True)
    y = nn.MaxPool3d(kernel_size=(2, 2, 2))(x)
    r = rng_t(rng, *y.shape)
    out.update({"maxpool.x": x, "maxpool.r": r, "maxpool.y": y, "maxpool.gx": torch.autograd.grad((y * r).sum(), x)[0]})
    # trilinear: x2 align_corners=False (decoder) and x{2,4} align_corners=True (feature head)
    x = rng_t(rng, 1, 3, 3, 5, 4).requires_grad_(True)
    for tag, kw in [("up2", dict(scale_factor=2, align_corners=False)), ("head2", dict(scale_factor=2, align_corners=True)),
                    ("head4", dict(scale_factor=4, align_corners=True))]:
        y = F.interpolate(x, mode="trilinear", **kw)
        r = rng_t(rng, *y.shape)
        out.update({f"tri_{tag}.y": y, f"tri_{tag}.r": r, f"tri_{tag}.gx": torch.autograd.grad((y * r).sum(), x)[0]})
    out["tri.x"] = x
    # projection head in train mode (batch statistics)
    net = ref_factory.net_factory_3d("unet_3D", 1, 2, 2)
    net.load_state_dict(onets.make_unet_params(seed=31))   # weights rebuilt from the seed by the tests
    proj = net.projection
    x, r = rng_t(rng, 2, 256, 2, 3, 2), rng_t(rng, 2, 256, 2, 3, 2)
    proj.train()
    c = _layer_case(proj, x, r)   # state_dict captured AFTER the forward: running stats updated once
    for k, v in c.items():
        if k.startswith("p.") and "running" not in k:
            continue
        if k.startswith("g.") and v.numel() > 4096:
            out[f"proj.gstats.{k[2:]}"] = stats(v)
            out[f"proj.gsub.{k[2:]}"] = v.reshape(-1)[::97]
        else:
            out[f"proj.{k}"] = v
    out["proj.x"], out["proj.r"], out["proj.param_seed"] = x, r, np.array(31)
    save("unet_layers", **out)


# ----------------------------------------------------------------------------- full nets
class VNetWithHead(nn.Module):
    """Composite used as the V-Net oracle witness: reference VNet encoder/decoder +
    a projection module with the layer list of UNet3D_contrastive.py:261-267 on x5."""

    def __init__(self, scale_factor=2, has_dropout=False):
        super().__init__()
        self.vnet = ref_vnet.VNet(n_channels=1, n_classes=2, normalization="groupnorm", has_dropout=has_dropout)
        self.projection = nn.Sequential(nn.Conv3d(256, 512, 1), nn.BatchNorm3d(512), nn.ReLU(inplace=True),
                                        nn.Conv3d(512, 256, 1), nn.BatchNorm3d(256))
        self.scale_factor = scale_factor

    def load_flat(self, p):
        self.vnet.load_state_dict({k: v for k, v in p.items() if not k.startswith("projection.")})
        self.projection.load_state_dict({k[len("projection."):]: v for k, v in p.items() if k.startswith("projection.")})

    def flat_named_parameters(self):
        for k, v in self.vnet.named_parameters():
            yield k, v
        for k, v in self.projection.named_parameters():
            yield "projection." + k, v

    def forward(self, x):
        feats = self.vnet.encoder(x)
        out = self.vnet.decoder(feats)
        c = F.interpolate(feats[4], scale_factor=self.scale_factor, mode="trilinear", align_corners=True)
        return torch.tanh(out), out, self.projection(c)


def _sub(t):
    return t[..., ::2, ::2, ::2]


def _full_nets(dtype):
    """forward + objective gradients of the reference nets in `dtype` (fp32: the fixture; fp64: its twin for tolerance budgeting)"""
    rng = np.random.default_rng(606)
    c = lambda t: t.to(dtype)  # noqa: E731
    x = c(rng_t(rng, 2, 1, 32, 32, 32))
    out = {"x_seed": np.array(606)}
    # ---- V-Net: the reference class itself (no head) + composite head
    p = {k: (c(v) if v.is_floating_point() else v) for k, v in onets.make_vnet_params(seed=11).items()}
    net = VNetWithHead(2).to(dtype)
    net.load_flat(p)
    net.train()
    _, logits, feats = net(x)
    r1 = c(rng_t(rng, *logits.shape))
    r2 = c(rng_t(rng, *feats.shape))
    names = [k for k, _ in net.flat_named_parameters()]
    grads = torch.autograd.grad((logits * r1).sum() + (feats * r2).sum(), [v for _, v in net.flat_named_parameters()])
    out.update({"vnet.logits_sub": _sub(logits), "vnet.logits_stats": stats(logits), "vnet.feats": feats,
                "vnet.param_seed": np.array(11)})
    out["vnet.grad_stats"] = np.stack([stats(g) for g in grads])
    out["vnet.grad_names"] = np.array(names)
    out["vnet.r1_seed_note"] = np.array("r1,r2 drawn after x from default_rng(606): x, r1, r2 in that order")
    # plain reference VNet.forward must equal the composite's logits
    plain = net.vnet(x)
    assert torch.equal(plain, logits)
    # ---- U-Net through the reference factory
    pu = {k: (c(v) if v.is_floating_point() else v) for k, v in onets.make_unet_params(seed=12).items()}
    unet = ref_factory.net_factory_3d("unet_3D", 1, 2, 2).to(dtype)
    unet.load_state_dict(pu)
    unet.train()
    unet.dropout1.p = 0.0
    unet.dropout2.p = 0.0
    sdf, ulog, ufeat = unet(x)
    r3 = c(rng_t(rng, *ulog.shape))
    r4 = c(rng_t(rng, *ufeat.shape))
    unames = [k for k, _ in unet.named_parameters()]
    # the sdf head (tanh(final(up1))) is discarded by the training step, so it is left out of the objective:
    # final.* get no gradient (None -> zero statistics), exactly as in train_DyCON_BraTS19.py:304
    ugr = torch.autograd.grad((ulog * r3).sum() + (ufeat * r4).sum(), list(unet.parameters()), allow_unused=True)
    out.update({"unet.logits_sub": _sub(ulog), "unet.logits_stats": stats(ulog), "unet.sdf_stats": stats(sdf),
                "unet.feats": ufeat, "unet.param_seed": np.array(12)})
    out["unet.grad_stats"] = np.stack([stats(g) for g in ugr])
    out["unet.grad_names"] = np.array(unames)
    return out


F64_KEYS_FULL = ("logits_sub", "logits_stats", "feats", "grad_stats")


def gen_full_nets():
    out = _full_nets(torch.float32)
    twin = _full_nets(torch.float64)      # same inputs / weights (fp32 values), reference modules run in double
    for net in ("vnet", "unet"):
        for k in F64_KEYS_FULL:
            out[f"{net}.{k}.f64"] = twin[f"{net}.{k}"]
    save("full_nets", **out)


# ----------------------------------------------------------------------------- full step traces
def _ref_step_trace(kind, n_steps=2, dtype=torch.float32):
    """Drive the imported reference modules/losses in the exact order of train_DyCON_BraTS19.py:298-372.
    dtype=float64: the same trace with every module, input and loss in double (tolerance-budget twin)."""
    rng = np.random.default_rng(707 if kind == "unet_3D" else 708)
    B, LB, S = 2, 1, 32
    cast = lambda d: {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in d.items()}  # noqa: E731
    if kind == "unet_3D":
        model = ref_factory.net_factory_3d("unet_3D", 1, 2, 2).to(dtype)
        ema = ref_factory.net_factory_3d("unet_3D", 1, 2, 2).to(dtype)
        model.load_state_dict(cast(onets.make_unet_params(seed=21)))
        ema.load_state_dict(cast(onets.make_unet_params(seed=22)))
        for m_ in (model, ema):
            m_.dropout1.p = 0.0
            m_.dropout2.p = 0.0
        s_named = lambda: list(model.named_parameters())   # noqa: E731
        t_named = lambda: list(ema.named_parameters())     # noqa: E731
    else:
        model, ema = VNetWithHead(2).to(dtype), VNetWithHead(2).to(dtype)
        model.load_flat(cast(onets.make_vnet_params(seed=21)))
        ema.load_flat(cast(onets.make_vnet_params(seed=22)))
        s_named = lambda: list(model.flat_named_parameters())   # noqa: E731
        t_named = lambda: list(ema.flat_named_parameters())     # noqa: E731
    for _, p_ in t_named():
        p_.detach_()
    model.train()
    ema.train()
    opt = torch.optim.SGD([p_ for _, p_ in s_named()], lr=0.01, momentum=0.9, weight_decay=0.0001)
    uncl = ref_dycon.UnCLoss()
    fecl = ref_dycon.FeCLoss(device="cpu", temperature=0.6, gamma=2.0, use_focal=True, rampup_epochs=1500)
    out = {"B": np.array(B), "LB": np.array(LB), "S": np.array(S), "seeds": np.array([21, 22]),
           "rng_seed": np.array(707 if kind == "unet_3D" else 708)}
    iter_num = 0
    for step in range(n_steps):
        vol = rng_t(rng, B, 1, S, S, S).to(dtype)
        lab = blob_labels(rng, B, S, S, S)
        noise = torch.clamp(rng_t(rng, B, 1, S, S, S) * 0.1, -0.2, 0.2).to(dtype)
        epoch = step * 700          # exercise the threshold ramp
        beta = ref_dycon.adaptive_beta(epoch=epoch, total_epochs=3334, max_beta=5.0, min_beta=0.5)
        _, s_logits, s_feat = model(vol)
        with torch.no_grad():
            _, t_logits, t_feat = ema(vol + noise)
        s_prob, t_prob = F.softmax(s_logits, 1), F.softmax(t_logits, 1)
        cw = 0.1 * ref_ramps.sigmoid_rampup(iter_num // 150, 200.0)
        ce = F.cross_entropy(s_logits[:LB], lab[:LB])
        dice = ref_losses.dice_loss(s_prob[:LB, 1], lab[:LB] == 1)
        Bf, C = s_feat.shape[:2]
        s_emb = F.normalize(s_feat.view(Bf, C, -1).transpose(1, 2), dim=-1)
        t_emb = F.normalize(t_feat.view(Bf, C, -1).transpose(1, 2), dim=-1)
        mask = (F.avg_pool3d(lab.float(), kernel_size=8, stride=8) > 0.5).to(dtype).reshape(Bf, -1).unsqueeze(1)
        f_loss = fecl(feat=s_emb, mask=mask, teacher_feat=t_emb, gambling_uncertainty=None, epoch=epoch)
        u_loss = uncl(s_logits, t_logits, beta)
        cons = ref_losses.softmax_mse_loss(s_prob[LB:], t_prob[LB:]).mean()
        loss = 1.0 * (ce + dice) + cw * cons + 0.5 * (f_loss + u_loss)
        opt.zero_grad()
        loss.backward()
        gstats = np.stack([stats(p_.grad) for _, p_ in s_named()])
        gnorm = torch.nn.utils.clip_grad_norm_([p_ for _, p_ in s_named()], max_norm=1.0)
        opt.step()
        alpha = min(1 - 1 / (iter_num + 1), 0.99)
        for (_, ep), (_, sp) in zip(t_named(), s_named()):
            ep.data.mul_(alpha).add_(sp.data, alpha=1 - alpha)
        iter_num += 1
        out.update({f"s{step}.vol": vol, f"s{step}.label": lab.to(torch.uint8), f"s{step}.noise": noise,
                    f"s{step}.epoch": np.array(epoch), f"s{step}.beta": np.array(beta),
                    f"s{step}.scalars": np.array([loss.item(), ce.item(), dice.item(), cons.item(), f_loss.item(),
                                                  u_loss.item(), cw, float(gnorm)]),
                    f"s{step}.logits_sub": _sub(s_logits), f"s{step}.t_logits_sub": _sub(t_logits),
                    f"s{step}.feat_stats": stats(s_feat), f"s{step}.mask": mask,
                    f"s{step}.grad_stats": gstats,
                    f"s{step}.student_stats": np.stack([stats(p_) for _, p_ in s_named()]),
                    f"s{step}.teacher_stats": np.stack([stats(p_) for _, p_ in t_named()])})
    out["param_names"] = np.array([k for k, _ in s_named()])
    return out


F64_KEYS_STEP = ("scalars", "logits_sub", "t_logits_sub", "feat_stats", "grad_stats", "student_stats", "teacher_stats")


def gen_steps():
    for name, kind in (("step_unet", "unet_3D"), ("step_vnet", "vnet")):
        out = _ref_step_trace(kind)
        twin = _ref_step_trace(kind, dtype=torch.float64)
        for step in range(2):
            for k in F64_KEYS_STEP:
                out[f"s{step}.{k}.f64"] = twin[f"s{step}.{k}"]
        save(name, **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["schedules", "uncl", "fecl", "voxel", "vnet_layers", "unet_layers", "full", "steps"]
    fns = {"schedules": gen_schedules, "uncl": gen_uncl, "fecl": gen_fecl, "voxel": gen_voxel_losses,
           "vnet_layers": gen_vnet_layers, "unet_layers": gen_unet_layers, "full": gen_full_nets, "steps": gen_steps}
    for w in which:
        print(w)
        fns[w]()
