"""GPU parity tests, op level: every HIP kernel (called through the C ABI) against the oracle /
the golden fixtures from the reference.  fp32 storage: tolerance 1e-4 (north-star); bf16: loose.
"""
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from dycon_paper_replication_amd import ops
    from dycon_paper_replication_amd.engine import Engine
from oracle import losses as OL
from oracle import step as OS

DEV = "cuda:0"
T = torch.from_numpy


def nd(t, dtype=torch.float32):
    """NCDHW cpu -> NDHWC cuda"""
    return t.permute(0, 2, 3, 4, 1).contiguous().to(DEV, dtype)


def nc(t):
    """NDHWC cuda -> NCDHW cpu fp32"""
    return t.float().cpu().permute(0, 4, 1, 2, 3).contiguous()


def close(a, b, rtol=1e-4, atol=1e-5, msg=""):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().float().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=msg)


def relclose(a, b, tol, msg=""):
    """max |a-b| <= tol * max|b|  (for bf16 comparisons)"""
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    err = (a - b).abs().max().item()
    ref = b.abs().max().item() + 1e-12
    assert err <= tol * ref, f"{msg}: max err {err:.3e} vs scale {ref:.3e}"


def mini_engine(params, dtype=torch.float32):
    p = {k: v.to(DEV).contiguous() for k, v in params.items()}
    g = {k: torch.full_like(v, float("nan")) for k, v in p.items()}
    e = Engine("vnet", p, g, {}, dtype=dtype)
    e.recording, e.update_bn = True, True
    e.tape, e.G = [], {}
    from dycon_paper_replication_amd.engine import DropoutSpec
    e.dropout = DropoutSpec("off")
    return e


CONV_CASES = [
    # kind, cin, cout, spatial(in)
    ("k3", 16, 32, (6, 8, 10)), ("k3", 32, 16, (5, 7, 9)), ("k3", 64, 64, (4, 4, 6)), ("k3", 48, 16, (4, 6, 5)),
    ("k2s2", 16, 32, (8, 6, 10)), ("k2s2", 64, 128, (4, 4, 2)),
    ("deconv", 32, 16, (4, 3, 5)), ("deconv", 128, 64, (2, 3, 2)),
    ("1x1", 256, 512, (2, 3, 2)), ("1x1", 16, 2, (4, 6, 8)), ("k3", 1, 16, (8, 8, 12)),
    # >= 24^3 voxels: the LDS-halo kernels (all CK / NTB / wave-layout variants, partial tiles, first layer)
    ("k3", 1, 16, (24, 24, 24)), ("k3", 16, 16, (24, 28, 22)), ("k3", 16, 32, (24, 24, 24)), ("k3", 32, 16, (24, 24, 24)),
    ("k3", 32, 32, (26, 24, 24)), ("k3", 64, 64, (24, 24, 24)), ("k3", 16, 64, (24, 24, 24)), ("k3", 64, 128, (24, 24, 24)),
    ("k3", 32, 1 * 16, (20, 28, 25)),
    # small levels, >= 128 output channels (bf16): LDS-tiled split-K GEMM kernel, ragged row blocks
    ("k3", 64, 128, (4, 4, 6)), ("k3", 128, 128, (6, 6, 6)), ("k3", 128, 256, (3, 5, 4)), ("k3", 256, 256, (6, 6, 6)),
    # U-Net decoder widths on the LDS kernel: 48 input channels (chunk-major packing), 48 / 96 output channels (3 n-tiles)
    ("k3", 48, 16, (24, 24, 24)), ("k3", 96, 32, (24, 24, 24)), ("k3", 16, 48, (24, 24, 24)), ("k3", 32, 96, (24, 24, 24)),
]


@pytest.mark.parametrize("kind,cin,cout,sp", CONV_CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv(kind, cin, cout, sp, dtype):
    rng = np.random.default_rng(zlib.crc32(repr((kind, cin, cout)).encode()))   # (hash() of a str is salted per process)
    B = 2
    k = {"k3": 3, "k2s2": 2, "deconv": 2, "1x1": 1}[kind]
    wshape = (cin, cout, k, k, k) if kind == "deconv" else (cout, cin, k, k, k)
    w = T((rng.standard_normal(wshape) / np.sqrt(cin * k ** 3)).astype(np.float32))
    b = T(rng.standard_normal(cout).astype(np.float32))
    x = T(rng.standard_normal((B, cin) + sp).astype(np.float32))
    if dtype == torch.bfloat16:   # compare against the oracle on the same bf16-rounded inputs
        x = x.bfloat16().float()
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    wq = wr.bfloat16().float() if dtype == torch.bfloat16 else wr
    if kind == "k3":
        yr = F.conv3d(xr, wq, br, padding=1)
    elif kind == "k2s2":
        yr = F.conv3d(xr, wq, br, stride=2)
    elif kind == "deconv":
        yr = F.conv_transpose3d(xr, wq, br, stride=2)
    else:
        yr = F.conv3d(xr, wq, br)
    gy = T(rng.standard_normal(tuple(yr.shape)).astype(np.float32))
    out_f32 = (kind == "1x1" and cout == 2)
    if dtype == torch.bfloat16 and not out_f32:
        gy = gy.bfloat16().float()
    yr.backward(gy)

    e = mini_engine({"l.weight": w, "l.bias": b}, dtype)
    xd = nd(x, dtype)
    y = e._conv("l", xd, kind, need_gx=cin > 1, out_dtype=torch.float32 if out_f32 else None)
    e.G[id(y)] = nd(gy, y.dtype)
    for fn in reversed(e.tape):
        fn()
    torch.cuda.synchronize()
    if dtype == torch.float32:
        close(nc(y), yr, 1e-4, 1e-5, "y")
        # sums over up to 3e4 voxels: elements that nearly cancel carry the round-off of the large terms
        close(e.g["l.weight"], wr.grad, 1e-4, 2e-4 + 1e-5 * float(wr.grad.abs().max()), "gw")
        close(e.g["l.bias"], br.grad, 1e-4, 2e-4 + 1e-5 * float(br.grad.abs().max()), "gb")
        if cin > 1:
            close(nc(e.G[id(xd)]), xr.grad, 1e-4, 1e-5, "gx")
    else:
        relclose(nc(y), yr, 1e-2, "y")
        relclose(e.g["l.weight"], wr.grad, 1e-2, "gw")
        relclose(e.g["l.bias"], br.grad, 1e-2, "gb")
        if cin > 1:
            relclose(nc(e.G[id(xd)]), xr.grad, 1e-2, "gx")


def test_conv_accumulate():
    """data-gradient accumulation into an existing buffer (skip connections)"""
    rng = np.random.default_rng(5)
    w = T((rng.standard_normal((32, 16, 2, 2, 2)) * 0.1).astype(np.float32))
    b = torch.zeros(32)
    x = T(rng.standard_normal((1, 16, 4, 4, 6)).astype(np.float32))
    e = mini_engine({"l.weight": w, "l.bias": b})
    xd = nd(x)
    y = e._conv("l", xd, "k2s2")
    gy = torch.randn(y.shape, device=DEV)
    prior = torch.randn(xd.shape, device=DEV)
    e.G[id(y)] = gy
    e.G[id(xd)] = prior.clone()
    for fn in reversed(e.tape):
        fn()
    xr = x.clone().requires_grad_(True)
    F.conv3d(xr, w, b, stride=2).backward(nc(gy))
    close(nc(e.G[id(xd)]), xr.grad + nc(prior), 1e-4, 1e-5)


# spatial sizes: 120 voxels (one row per thread of the one-launch kernels), 12^3 = 1728 and 8^3 = 512 (the register-resident form with
# 8 rows per thread, partly masked), 14 x 14 x 13 = 2548 (past the one-launch limit: statistics / finalize / apply launches)
@pytest.mark.parametrize("kind,C,sp", [("gn", 32, (4, 6, 5)), ("gn", 16, (4, 6, 5)), ("in", 16, (4, 6, 5)), ("in", 48, (4, 6, 5)), ("bn", 512, (4, 6, 5)),
                                       ("gn", 128, (12, 12, 12)), ("gn", 64, (8, 8, 8)), ("in", 64, (8, 8, 8)), ("gn", 256, (7, 7, 5)),
                                       ("in", 128, (12, 12, 12)), ("gn", 64, (14, 14, 13))])
@pytest.mark.parametrize("mode", ["relu", "skip", "norelu"])
def test_norm(kind, C, sp, mode):
    rng = np.random.default_rng(C)
    B = 2
    z = T((rng.standard_normal((B, C) + sp) * 1.5 + 0.3).astype(np.float32))
    gamma = T((1 + 0.2 * rng.standard_normal(C)).astype(np.float32))
    beta = T((0.2 * rng.standard_normal(C)).astype(np.float32))
    skip = T(rng.standard_normal((B, C) + sp).astype(np.float32))
    relu = mode != "norelu"
    zr = z.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    if kind == "gn":
        yr = F.group_norm(zr, 16, gr, br, 1e-5)
    elif kind == "in":
        yr = F.instance_norm(zr, eps=1e-5)
    else:
        yr = F.batch_norm(zr, None, None, gr, br, True, 0.1, 1e-5)
    if relu:
        yr = F.relu(yr)
    if mode == "skip":
        yr = yr + skip
    gy = T(rng.standard_normal(tuple(yr.shape)).astype(np.float32))
    yr.backward(gy)
    params = {"n.weight": gamma, "n.bias": beta} if kind != "in" else {}
    e = mini_engine(params)
    zd = nd(z)
    sd = nd(skip) if mode == "skip" else None
    y = e._norm("n" if kind != "in" else None, zd, kind, relu=relu, skip=sd)
    ycopy = y.clone()
    e.G[id(y)] = nd(gy)
    for fn in reversed(e.tape):
        fn()
    close(nc(ycopy), yr, 1e-4, 2e-5, "y")
    close(nc(e.G[id(zd)]), zr.grad, 2e-4, 2e-5, "gz")
    if kind != "in":
        close(e.g["n.weight"], gr.grad, 2e-4, 2e-4, "dgamma")
        close(e.g["n.bias"], br.grad, 2e-4, 2e-4, "dbeta")
    if mode == "skip":
        close(nc(e.G[id(sd)]), gy, 0, 0, "gskip")
    if mode == "norelu" and kind != "in":
        # without ReLU the op is invertible: the backward may recover xhat from y (from_y = 1)
        nvox = sp[0] * sp[1] * sp[2]
        Nb, G, V = (1, C, B * nvox) if kind == "bn" else (B, 16, nvox)
        stats = ops.norm_stats(zd, Nb, V, C, G)
        gz2 = ops.norm_bwd(ycopy, True, nd(gy), stats, Nb, V, C, G, e.p["n.weight"], e.p["n.bias"], False)
        close(nc(gz2), zr.grad, 2e-3, 2e-4, "gz from y")


@pytest.mark.parametrize("kind,cin,cout,sp,nk", [("k3", 128, 128, (6, 6, 6), "gn"), ("k3", 128, 256, (5, 6, 4), "gn"), ("k3", 256, 256, (6, 6, 6), "in"),
                                                 ("k2s2", 64, 128, (12, 12, 12), "gn"), ("k3", 64, 64, (6, 6, 6), "gn")])
def test_split_k_finish_fused_into_norm(kind, cin, cout, sp, nk):
    """Small levels, bf16: the split-K finish of the convolution (bias + ordered slab sum + rounding) done by the one-launch norm
    that follows (dycon_conv_gemm_ex defer_finish + dycon_norm_fwd_slab) must reproduce finish launch + norm launch bit for bit:
    pre-norm tensor (kept for the backward), statistics, output, and the gradients of a backward through both."""
    rng = np.random.default_rng(cin + cout)
    B = 3
    k = 3 if kind == "k3" else 2
    params = {"l.weight": T((rng.standard_normal((cout, cin, k, k, k)) * 0.05).astype(np.float32)),
              "l.bias": T((rng.standard_normal(cout) * 0.3).astype(np.float32)),
              "n.weight": T((1 + 0.2 * rng.standard_normal(cout)).astype(np.float32)),
              "n.bias": T((0.2 * rng.standard_normal(cout)).astype(np.float32))}
    xd = nd(T(rng.standard_normal((B, cin) + sp).astype(np.float32)), torch.bfloat16)
    G = 16 if nk == "gn" else cout
    res = []
    for fuse in (False, True):
        e = mini_engine(params, torch.bfloat16)
        e.fuse_finish = fuse
        z = e._conv("l", xd, kind, norm_groups=G)
        deferred = id(z) in e._deferred
        y = e._norm("n" if nk == "gn" else None, z, nk)
        gy = torch.ones_like(y) * 0.5
        e.G[id(y)] = gy
        for fn in reversed(e.tape):
            fn()
        res.append((deferred, z.clone(), y.clone(), e.G[id(xd)].clone(), e.g["l.weight"].clone()))
    assert res[1][0] or ops.query("dycon_conv_gemm_splits", 1, 1 if kind == "k3" else 2, 0, B, *sp, cin, cout) <= 1
    assert not res[0][0]
    for a, b in zip(res[0][1:], res[1][1:]):
        assert torch.equal(a, b)


def test_golden_pool_trilinear():
    g = load_golden("unet_layers")
    x = T(g["maxpool.x"])
    y, idx = ops.maxpool2_fwd(nd(x))
    close(nc(y), g["maxpool.y"], 0, 0)
    gx = ops.maxpool2_bwd(nd(T(g["maxpool.r"])), idx, nd(x).shape)
    close(nc(gx), g["maxpool.gx"], 0, 0)
    x = T(g["tri.x"])                                     # (1, 3, 3, 5, 4): pad to 4 channels (16-byte channel groups)
    pad = lambda t: torch.cat([t, torch.zeros_like(t[:, :1])], 1)  # noqa: E731
    xd = nd(pad(x))
    for tag, scale, align in (("up2", 2, False), ("head2", 2, True), ("head4", 4, True)):
        out = tuple(s * scale for s in x.shape[2:])
        y = ops.trilinear_fwd(xd, out, align)
        close(nc(y)[:, :3], g[f"tri_{tag}.y"], 1e-5, 1e-6, tag)
        assert float(nc(y)[:, 3].abs().max()) == 0
        gx = ops.trilinear_bwd(nd(pad(T(g[f"tri_{tag}.r"]))), xd.shape, align)
        close(nc(gx)[:, :3], g[f"tri_{tag}.gx"], 1e-4, 1e-5, tag + " bwd")
    # channel-window variant (concat without a copy)
    cat = torch.zeros((1, 6, 10, 8, 12), device=DEV)
    ops.trilinear_fwd(xd, (6, 10, 8), False, out=cat, coff=4)
    close(nc(cat[..., 4:7].contiguous()), g["tri_up2.y"], 1e-5, 1e-6)
    assert float(cat[..., :4].abs().max()) == 0 and float(cat[..., 7:].abs().max()) == 0
    gwin = torch.zeros((1, 6, 10, 8, 12), device=DEV)
    gwin[..., 4:8] = nd(pad(T(g["tri_up2.r"])))
    close(nc(ops.trilinear_bwd(gwin, xd.shape, False, coff=4))[:, :3], g["tri_up2.gx"], 1e-4, 1e-5)
    # bf16 storage
    yb = ops.trilinear_fwd(torch.cat([xd, xd], -1).bfloat16(), (6, 10, 8), False)
    relclose(nc(yb)[:, :3], T(g["tri_up2.y"]), 1e-2, "tri bf16")


@pytest.mark.parametrize("shape,out,align", [((2, 7, 6, 5), (14, 12, 10), False), ((1, 6, 6, 6), (12, 12, 12), True),
                                             ((1, 5, 7, 3), (12, 9, 13), False), ((2, 4, 3, 5), (9, 11, 6), True),
                                             ((1, 3, 4, 2), (14, 18, 9), True)])
def test_trilinear_vs_torch(shape, out, align):
    """Forward and gathered adjoint against torch's upsample_trilinear3d at ragged sizes, non-integer and > 4 scale factors (the
    adjoint's candidate-output ranges and register-held per-axis weights, and the fall-back beyond 12 candidates per axis)."""
    torch.manual_seed(5)
    B, C = shape[0], 8
    x = torch.randn(B, C, *shape[1:], requires_grad=True)
    y = F.interpolate(x, size=out, mode="trilinear", align_corners=align)
    r = torch.randn_like(y)
    (gx,) = torch.autograd.grad(y, x, r)
    xd = nd(x.detach())
    close(nc(ops.trilinear_fwd(xd, out, align)), y, 1e-5, 1e-6)
    close(nc(ops.trilinear_bwd(nd(r), xd.shape, align)), gx, 1e-4, 1e-5)


def test_pointwise():
    x = torch.randn(2, 4, 5, 6, 16, device=DEV)
    scale = torch.rand(2 * 16, device=DEV)
    close(ops.scale_channels(x, scale), x * scale.view(2, 1, 1, 1, 16), 1e-6, 1e-7)
    dst = torch.zeros(2, 4, 5, 6, 40, device=DEV)
    ops.copy_channels(x, 4, dst, 10, 8)
    close(dst[..., 10:18], x[..., 4:12], 0, 0)
    assert float(dst[..., :10].abs().max()) == 0
    close(ops.add(x, x), 2 * x, 0, 0)
    close(ops.tanh(x), torch.tanh(x), 1e-5, 1e-6)
    close(ops.cast(ops.cast(x, torch.bfloat16), torch.float32), x.bfloat16().float(), 0, 0)
    mask = (torch.rand_like(x) > 0.3).float()
    close(ops.mul_mask(x, mask, 1 / 0.7), x * mask / 0.7, 1e-6, 1e-7)
    # Philox dropout: keep-rate, scaling and reproducibility
    big = torch.ones(1 << 20, device=DEV)
    y1 = ops.dropout_philox(big, 0.3, 1234, 77)
    y2 = ops.dropout_philox(big, 0.3, 1234, 77)
    assert torch.equal(y1, y2)
    keep = (y1 != 0).float().mean().item()
    assert abs(keep - 0.7) < 5e-3
    close(y1[y1 != 0], torch.full_like(y1[y1 != 0], 1 / 0.7), 1e-6, 0)
    assert not torch.equal(y1, ops.dropout_philox(big, 0.3, 1234, 78 + (1 << 20)))
    cm = ops.channel_mask_philox(4096, 0.5, 9, 0, DEV)
    assert abs((cm != 0).float().mean().item() - 0.5) < 0.05 and set(cm.unique().tolist()) <= {0.0, 2.0}
    # teacher noise: clamp(N(0,1)*0.1, +-0.2)
    z = torch.zeros(1 << 20, device=DEV)
    n = ops.add_noise(z, None, 0.1, 0.2, 42, 0)
    assert float(n.abs().max()) <= 0.2 + 1e-7
    assert abs(float(n.mean())) < 1e-3 and abs(float(n.std()) - 0.0954) < 2e-3     # std of N(0,.1) clamped at 2 sigma
    assert float((n.abs() >= 0.2 - 1e-7).float().mean()) == pytest.approx(0.0455, abs=3e-3)
    ex = torch.randn(1000, device=DEV)
    close(ops.add_noise(torch.ones(1000, device=DEV), ex), 1 + ex, 1e-6, 1e-7)


@pytest.mark.parametrize("fast", [False, True])
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_uncl_golden(tag, fast):
    """fast = the hardware exp2 / log2 / rcp sequences the bf16 step uses: same fixtures, 10x the tolerance"""
    g = load_golden(f"uncl_{tag}")
    s, t = nd(T(g["s"])), nd(T(g["t"]))
    B = s.shape[0]
    V = s.numel() // (2 * B)
    k = 10.0 if fast else 1.0
    lab = torch.zeros(s.shape[:4], dtype=torch.uint8, device=DEV)
    for i, beta in enumerate(g["betas"]):
        sums = ops.seg_losses_fwd(s, t, lab, 0, float(beta), fast=fast)
        vals = ops.seg_losses_finalize(sums, B, 0, V, float(beta))
        close(vals[5], g[f"loss{i}"], k * 1e-5, 1e-6)
        coef = torch.tensor([0, 0, 0, 0, 1.0], device=DEV)
        gs = ops.seg_losses_bwd(s, t, lab, 0, float(beta), sums, coef, fast=fast)
        close(nc(gs), g[f"grad{i}"], k * 1e-4, k * 1e-8)


@pytest.mark.parametrize("fast", [False, True])
def test_voxel_losses_golden(fast):
    g = load_golden("voxel_losses")
    a, b, lab = T(g["a"]), T(g["b"]), T(g["label"])
    B = a.shape[0]
    V = a.numel() // (2 * B)
    k = 10.0 if fast else 1.0
    ad, labd = nd(a), lab.to(DEV)
    # labelled terms: every sample labelled (LB = B)
    sums = ops.seg_losses_fwd(ad, nd(b), labd, B, 1.0, fast=fast)
    vals = ops.seg_losses_finalize(sums, B, B, V, 1.0)
    close(vals[0], g["ce"], k * 1e-5, 1e-6)
    close(vals[1], g["dice"], k * 1e-5, 1e-6)
    close(vals[2], g["dice_mc"], k * 1e-5, 1e-6)
    for j, key in ((0, "ce_grad"), (1, "dice_grad"), (2, "dice_mc_grad")):
        coef = torch.zeros(5, device=DEV)
        coef[j] = 1
        close(nc(ops.seg_losses_bwd(ad, nd(b), labd, B, 1.0, sums, coef, fast=fast)), g[key], k * 1e-4, k * 1e-8, key)
    # consistency terms: no sample labelled (LB = 0); uint8 labels exercise the other label path
    sums = ops.seg_losses_fwd(ad, nd(b), labd.to(torch.uint8), 0, 1.0, fast=fast)
    vals = ops.seg_losses_finalize(sums, B, 0, V, 1.0)
    close(vals[3], g["cons_mse"], k * 1e-5, 1e-7)
    close(vals[4], g["cons_kl"], k * 1e-4, 1e-7)
    coef = torch.tensor([0, 0, 0, 1.0, 0], device=DEV)
    close(nc(ops.seg_losses_bwd(ad, nd(b), labd.to(torch.uint8), 0, 1.0, sums, coef, 0, fast=fast)), g["cons_mse_grad"], k * 1e-4, k * 1e-9)
    close(nc(ops.seg_losses_bwd(ad, nd(b), labd.to(torch.uint8), 0, 1.0, sums, coef, 1, fast=fast)), g["cons_kl_grad"], k * 1e-4, k * 1e-9)


@pytest.mark.parametrize("tag", ["small", "mid", "oneclass", "singleton", "ragged"])
def test_fecl_golden(tag):
    """FeCL against the REFERENCE's own outputs (loss and gradient), all ramp epochs / focal / teacher / gambling."""
    g = load_golden(f"fecl_{tag}")
    feat, teach = T(g["feat"]).to(DEV), T(g["teacher"]).to(DEV)
    mask = T(g["mask"]).reshape(feat.shape[0], -1).contiguous().to(DEV)
    gamb = T(g["gambling"]).contiguous().to(DEV)
    for i in range(int(g["n_cfg"])):
        epoch, focal, use_t, use_g = [int(v) for v in g[f"cfg{i}"]]
        thr = OL.threshold_rampup(epoch, 1500, 0.3, 0.5)
        args = (feat, teach if use_t else None, mask, gamb if use_g else None, 0.6, 2.0, bool(focal), thr, 1.0)
        loss, st = ops.fecl_fwd(*args)
        close(loss[0], g[f"loss{i}"], 1e-4, 1e-6, f"loss cfg{i}")
        coef = torch.ones(1, device=DEV)
        gf = ops.fecl_bwd(*args, st, coef)
        ref = g[f"grad{i}"]
        scale = np.abs(ref).max() + 1e-12
        err = np.abs(gf.cpu().numpy() - ref).max()
        assert err <= 1e-4 * scale + 1e-8, f"grad cfg{i}: {err} vs {scale}"


def test_fecl_bf16_and_full_size():
    """N = 1728, Dm = 256 (config 2) in bf16 storage vs the fp32 oracle on the rounded inputs."""
    torch.manual_seed(3)
    B, N, Dm = 2, 1728, 256
    f = F.normalize(torch.randn(B, N, Dm), dim=-1).bfloat16()
    t = F.normalize(f.float() + 0.02 * torch.randn(B, N, Dm), dim=-1).bfloat16()
    mask = (torch.rand(B, N) > 0.8).float()
    thr = 0.35
    fr = f.float().requires_grad_(True)
    ref = OL.fecl(fr, mask.view(B, 1, N), t.float(), None, 300, 0.6, 2.0, True, 1500, 1.0)
    assert OL.threshold_rampup(300, 1500, 0.3, 0.5) == pytest.approx(0.3 + 0.2 * np.exp(-5 * 0.64))
    thr = OL.threshold_rampup(300, 1500, 0.3, 0.5)
    (gr,) = torch.autograd.grad(ref, fr)
    args = (f.to(DEV), t.to(DEV), mask.to(DEV), None, 0.6, 2.0, True, thr, 1.0)
    loss, st = ops.fecl_fwd(*args)
    close(loss[0], ref, 2e-4, 1e-6)
    gf = ops.fecl_bwd(*args, st, torch.ones(1, device=DEV))
    relclose(gf, gr, 2e-2, "fecl bf16 grad")


def test_l2norm_maskpool():
    x = torch.randn(3, 50, 256)
    x[0, 3] = 0          # below the eps clamp
    xr = x.clone().requires_grad_(True)
    yr = F.normalize(xr, dim=-1)
    gy = torch.randn_like(x)
    yr.backward(gy)
    y, nrm = ops.l2norm_fwd(x.to(DEV))
    close(y, yr, 1e-5, 1e-6)
    close(ops.l2norm_bwd(y, nrm, gy.to(DEV)), xr.grad, 1e-4, 1e-5)
    lab = (torch.rand(2, 16, 24, 8) > 0.5).long()
    close(ops.mask_pool(lab.to(DEV), 8), OL.contrast_mask(lab, 8).reshape(2, -1), 0, 0)
    close(ops.mask_pool(lab.to(DEV).to(torch.uint8), (4, 8, 2)),
          (F.avg_pool3d(lab.float().unsqueeze(1), (4, 8, 2)) > 0.5).float().reshape(2, -1), 0, 0)


def test_optimizer():
    rng = np.random.default_rng(9)
    n, n_sgd = 1000, 900
    p = T(rng.standard_normal(n).astype(np.float32))
    teacher = T(rng.standard_normal(n).astype(np.float32))
    ps, ts, mom = {"a": p[:n_sgd].clone()}, {"a": teacher.clone()}, {}
    pd, td = p.to(DEV), teacher.to(DEV)
    md = torch.zeros(n, device=DEV)
    full = p.clone()
    for step in range(3):
        g = T((rng.standard_normal(n) * (3.0 if step == 0 else 0.01)).astype(np.float32))
        gn, gc = OS.clip_grad_norm({"a": g[:n_sgd]}, 1.0)
        OS.sgd_step(ps, gc, mom, 0.01, 0.9, 1e-4)
        full = torch.cat([ps["a"], full[n_sgd:]])
        t_all = {"a": ts["a"]}
        OS.ema_update(t_all, {"a": full}, 0.99, step)
        ts = t_all
        ss = torch.zeros(1, dtype=torch.float64, device=DEV)
        gd = g.to(DEV)
        ops.sumsq(gd[:n_sgd].contiguous(), ss)
        close(ss.sqrt(), gn, 1e-5, 0)
        alpha = min(1 - 1 / (step + 1), 0.99)
        ops.sgd_ema(pd, gd, md, td, n_sgd, ss, 1.0, 1.0, 0.01, 0.9, 1e-4, alpha)
        close(pd, full, 1e-5, 1e-6, f"params step {step}")
        close(td, ts["a"], 1e-5, 1e-6, f"teacher step {step}")
    # skip flag: nothing moves
    flag = torch.ones(1, dtype=torch.int32, device=DEV)
    before = pd.clone()
    ops.sgd_ema(pd, gd, md, td, n_sgd, ss, 1.0, 1.0, 0.01, 0.9, 1e-4, 0.99, flag)
    assert torch.equal(before, pd)
    f2 = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.nonfinite_flag(torch.tensor([float("nan")], device=DEV), f2)
    assert int(f2) == 1


def test_fecl_isles_size_vs_rowblock_oracle():
    """N = 15 680 patches (ISLES, feature_scaler 4): far beyond what the reference can materialise (12 N x N fp32 tensors =
    11.8 GB per sample).  The HIP kernel (fp32 storage) against the row-block oracle, loss and gradient; bf16 storage on the loss."""
    torch.manual_seed(4)
    B, N, Dm = 1, 15680, 256
    f = F.normalize(torch.randn(B, N, Dm), dim=-1)
    t = F.normalize(f + 0.05 * torch.randn(B, N, Dm), dim=-1)
    mask = (torch.rand(B, N) > 0.9).float()
    epoch = 600
    thr = OL.threshold_rampup(epoch, 1500, 0.3, 0.5)
    ref, gref = OL.fecl_rowblocks(f, mask.view(B, 1, N), t, epoch, 0.6, 2.0, True, 1500, 1.0, block=1024)
    args = lambda ff, tt: (ff, tt, mask.to(DEV), None, 0.6, 2.0, True, thr, 1.0)  # noqa: E731
    l32, st = ops.fecl_fwd(*args(f.to(DEV), t.to(DEV)))
    close(l32[0], ref, 1e-4, 1e-6)
    g32 = ops.fecl_bwd(*args(f.to(DEV), t.to(DEV)), st, torch.ones(1, device=DEV))
    err = float((g32.cpu() - gref).abs().max())
    assert err <= 1e-3 * float(gref.abs().max()) + 1e-9, err
    l16, _ = ops.fecl_fwd(*args(f.to(DEV).bfloat16(), t.to(DEV).bfloat16()))
    ref16, _ = OL.fecl_rowblocks(f.bfloat16().float(), mask.view(B, 1, N), t.bfloat16().float(), epoch, 0.6, 2.0, True, 1500, 1.0, block=2048)
    close(l16[0], ref16, 5e-4, 1e-6)


@pytest.mark.parametrize("Dm,focal,use_t,use_g", [(256, True, True, False), (64, False, True, True), (128, True, False, False)])
def test_fecl_rows128_kernel_vs_oracle(Dm, focal, use_t, use_g):
    """N >= 1024 in bf16 storage (Dm 64 / 128 / 256, focal gamma 2 or no focal) runs on the 128-row kernels (fecl_rows128_kernel for
    passes 1-3, fecl_rows128_grad_kernel for the gradient: 32x32x16 MFMA, row fragments in registers): loss and gradient against the
    fp32 oracle on the rounded inputs, at an N that is a multiple of neither 64 nor 128 (ragged last row block, column tile and
    column split), with two column splits.  (test_fecl_bf16_and_full_size covers the headline N = 1728 on the same kernels;
    the fp32-storage cases and N < 1024 stay on fecl_kernel.)"""
    torch.manual_seed(11)
    B, N = 2, 8200
    f = F.normalize(torch.randn(B, N, Dm), dim=-1).bfloat16()
    t = F.normalize(f.float() + 0.05 * torch.randn(B, N, Dm), dim=-1).bfloat16()
    mask = (torch.rand(B, N) > 0.85).float()
    gamb = torch.rand(B, N) if use_g else None
    epoch = 300
    thr = OL.threshold_rampup(epoch, 1500, 0.3, 0.5)
    fr = f.float().requires_grad_(True)
    ref = OL.fecl(fr, mask.view(B, 1, N), t.float() if use_t else None, gamb.view(B, N) if use_g else None, epoch, 0.6, 2.0, focal, 1500, 1.0)
    (gr,) = torch.autograd.grad(ref, fr)
    args = (f.to(DEV), t.to(DEV) if use_t else None, mask.to(DEV), gamb.to(DEV) if use_g else None, 0.6, 2.0, focal, thr, 1.0)
    loss, st = ops.fecl_fwd(*args)
    close(loss[0], ref, 3e-4, 1e-6)
    gf = ops.fecl_bwd(*args, st, torch.ones(1, device=DEV))
    relclose(gf, gr, 2e-2, "fecl rows128 grad")


@pytest.mark.parametrize("norm", ["groupnorm", "none", "instancenorm", "batchnorm"])
def test_vnet_convblock_reference_fixtures(norm):
    """VNet.ConvBlock(2, 16, co, normalization=...) of the REFERENCE (VNet.py:5-31; fixture vnet_layers.npz, all four
    normalisations incl. the factory default 'none' and 'batchnorm'): conv -> [norm] -> ReLU twice through the engine's layer
    calls, forward, data gradient and every parameter gradient against the reference's outputs."""
    g = load_golden("vnet_layers")
    pre = f"convblock_{norm}."
    st = 2 if norm == "none" else 3
    nk = {"groupnorm": "gn", "instancenorm": "in", "batchnorm": "bn", "none": "none"}[norm]
    params = {k[len(pre) + 2:]: T(g[k]) for k in g.files if k.startswith(pre + "p.") and "running" not in k and "num_batches" not in k}
    e = mini_engine(params)
    e.buf = {}
    x = T(g[pre + "x"])
    xd = nd(x)
    t = xd
    for i in range(2):
        t = e._conv(f"conv.{st * i}", t, "k3")
        t = e._norm(f"conv.{st * i + 1}", t, nk)
    y = t.clone()
    e.G[id(t)] = nd(T(g[pre + "r"]))
    for fn in reversed(e.tape):
        fn()
    close(nc(y), g[pre + "y"], 1e-4, 1e-5, "y")
    close(nc(e.G[id(xd)]), g[pre + "gx"], 1e-4, 1e-5, "gx")
    for k in params:
        ref = g[pre + "g." + k]
        if k.endswith(".bias") and params[k[:-4] + "weight"].dim() == 5 and norm in ("instancenorm", "batchnorm"):
            assert float(e.g[k].abs().max()) <= 1e-4 * float(np.abs(g[pre + "g." + k[:-4] + "weight"]).max())   # analytically zero
            continue
        close(e.g[k], ref, 1e-4, 1e-4 * float(np.abs(ref).max()), k)


@pytest.mark.parametrize("kind,C,G,V,dtype", [("gn", 16, 16, 32 * 32 * 32, torch.float32), ("gn", 32, 16, 16 * 16 * 20, torch.bfloat16),
                                              ("in", 48, 48, 24 * 24 * 24, torch.float32), ("bn", 512, 512, 6912, torch.bfloat16),
                                              ("gn", 64, 16, 14 * 14 * 12, torch.float32)])
def test_norm_accumulator_form_equals_three_launch_form(kind, C, G, V, dtype):
    """dycon_norm_fwd_acc / dycon_norm_bwd_acc (statistics by double atomics into a zeroed arena slice, group statistics formed in
    the apply pass: two launches) against dycon_norm_fwd / dycon_norm_bwd (partials -> finalize -> apply): y, stats, gz, dgamma,
    dbeta and the BatchNorm running statistics agree to the last bits of the double sums."""
    gen = torch.Generator(device=DEV).manual_seed(C + V)
    Nb = 1 if kind == "bn" else 3
    x = (torch.randn(Nb, V, C, device=DEV, generator=gen) * 1.3 + 0.4).to(dtype)
    gy = torch.randn(Nb, V, C, device=DEV, generator=gen).to(dtype)
    skip = torch.randn(Nb, V, C, device=DEV, generator=gen).to(dtype)
    gamma = (1 + 0.2 * torch.randn(C, device=DEV, generator=gen)) if kind != "in" else None
    beta = (0.2 * torch.randn(C, device=DEV, generator=gen)) if kind != "in" else None
    cs = (torch.rand(Nb * C, device=DEV, generator=gen) > 0.5).float() * 2
    res = []
    for use_acc in (False, True):
        rm, rv = (torch.zeros(C, device=DEV), torch.ones(C, device=DEV)) if kind == "bn" else (None, None)
        acc = torch.zeros(ops.query("dycon_norm_acc_doubles", Nb, V, C), dtype=torch.float64, device=DEV) if use_acc else None
        y, stats = ops.norm_fwd(x, Nb, V, C, G, gamma, beta, True, skip, cs, 1e-5, rm, rv, 0.1, acc=acc)
        dg, db = (torch.empty(C, device=DEV), torch.empty(C, device=DEV)) if gamma is not None else (None, None)
        acc2 = torch.zeros(ops.query("dycon_norm_acc_doubles", Nb, V, C), dtype=torch.float64, device=DEV) if use_acc else None
        gz = ops.norm_bwd(x, False, gy, stats, Nb, V, C, G, gamma, beta, True, dg, db, chan_scale=cs, acc=acc2)
        res.append((y.float(), stats, gz.float(), dg, db, rm, rv))
    assert not ops.norm_fwd_is_fused(x, V, C, G)
    for a, b, name in zip(res[0], res[1], ("y", "stats", "gz", "dgamma", "dbeta", "running_mean", "running_var")):
        if a is None:
            continue
        tol = 1e-6 if dtype == torch.float32 else 1e-2     # (bf16: one ulp of the stored result where a statistic's last bit differs)
        assert float((a - b).abs().max()) <= tol * float(a.abs().max()) + 1e-9, name
    close(res[0][1], res[1][1], 1e-6, 1e-7, "stats")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind", ["gn", "in", "bn"])
@pytest.mark.parametrize("with_drop", [False, True])
def test_norm_head_fused_equals_norm_then_head(dtype, kind, with_drop):
    """dycon_norm_head_fwd / _bwd (block_nine's norm + ReLU + Dropout3d factor + out_conv in the norm's passes) against the
    launches they replace -- dycon_norm_fwd + dycon_conv_direct forward, dycon_conv_direct's data gradient + dycon_norm_bwd +
    dycon_conv_wgrad backward: logits bit for bit, gx / dgamma / dbeta to fp32 round-off, the head's weight / bias gradient up to
    summation order."""
    from dycon_paper_replication_amd._lib import CONV_1X1
    rng = np.random.default_rng(zlib.crc32(repr((str(dtype), kind, with_drop)).encode()))
    B, sp, C = 3, (20, 18, 16), 16
    V = sp[0] * sp[1] * sp[2]
    x = torch.from_numpy(rng.standard_normal((B,) + sp + (C,)).astype(np.float32) * 1.5 + 0.3).to(DEV, dtype)
    gl = torch.from_numpy(rng.standard_normal((B,) + sp + (2,)).astype(np.float32)).to(DEV)
    W = torch.from_numpy(rng.standard_normal((2, C, 1, 1, 1)).astype(np.float32) * 0.4).to(DEV)
    hb = torch.from_numpy(rng.standard_normal(2).astype(np.float32)).to(DEV)
    affine = kind != "in"
    gamma = torch.from_numpy(rng.standard_normal(C).astype(np.float32) * 0.5 + 1.0).to(DEV) if affine else None
    beta = torch.from_numpy(rng.standard_normal(C).astype(np.float32) * 0.3).to(DEV) if affine else None
    Nb, G, Vn = (1, C, B * V) if kind == "bn" else (B, 16, V)
    cs = None
    if with_drop:
        cs = torch.from_numpy((rng.random(Nb * C) > 0.5).astype(np.float32) * 2.0).to(DEV)
    # ---- the launches the fused pair replaces
    y, stats = ops.norm_fwd(x, Nb, Vn, C, G, gamma, beta, True, None, cs)
    w_tcn = W.reshape(2, C).t().contiguous()
    logits_ref = ops.conv_direct(y, w_tcn, hb, CONV_1X1, 2, torch.float32)
    gy = ops.conv_direct(gl, W.reshape(2, C).contiguous(), None, CONV_1X1, C, dtype)
    dg_ref, db_ref = (torch.empty(C, device=DEV), torch.empty(C, device=DEV)) if affine else (None, None)
    gx_ref = ops.norm_bwd(x, False, gy, stats, Nb, Vn, C, G, gamma, beta, True, dg_ref, db_ref, chan_scale=cs)
    gw_ref, gb_ref = torch.empty_like(W), torch.empty_like(hb)
    ops.conv_wgrad(y, gl, gw_ref, CONV_1X1, 0, 1, C, dbias=gb_ref)
    # ---- fused
    stats2 = ops.norm_stats(x, Nb, Vn, C, G)
    assert torch.equal(stats2, stats)
    logits = ops.norm_head_fwd(x, stats2, Nb, Vn, G, W, hb, gamma, beta, True, cs)
    dg, db = (torch.empty(C, device=DEV), torch.empty(C, device=DEV)) if affine else (None, None)
    gx, pend = ops.norm_head_bwd(x, gl, stats2, Nb, Vn, G, W, gamma, beta, True, dg, db, cs)
    gw, gb = torch.empty_like(W), torch.empty_like(hb)
    ops.norm_head_dparams(pend, gw, gb)
    torch.cuda.synchronize()
    assert torch.equal(logits, logits_ref)
    # same per-voxel formulas and roundings; the compiler may contract the multiply-adds of the two kernel pairs differently, which
    # moves the group sums by an fp32 ulp: gx to 1e-5 of its scale (fp32 storage) / one bf16 rounding step
    scale = float(gx_ref.float().abs().max())
    err = float((gx.float() - gx_ref.float()).abs().max())
    assert err <= (1e-5 if dtype == torch.float32 else 2.0 ** -7) * scale, (err, scale)
    if dtype == torch.bfloat16:
        assert float((gx != gx_ref).float().mean()) < 1e-3          # ... on a handful of elements
    if affine:
        close(dg, dg_ref, 1e-5, 1e-5 * float(dg_ref.abs().max()), "dgamma")
        close(db, db_ref, 1e-5, 1e-5 * float(db_ref.abs().max()), "dbeta")
    close(gw, gw_ref, 1e-4, 1e-4 * float(gw_ref.abs().max()), "head weight gradient")
    close(gb, gb_ref, 1e-4, 1e-4 * float(gb_ref.abs().max()), "head bias gradient")
    # bit-reproducibility: the same inputs again, three times, beside a second stream that keeps the memory
    # system busy with uneven bursts -- every output must repeat bit for bit (ordered two-stage reductions, no float atomics; a
    # missing barrier between a partial's write and its read-back, or an accumulator cleared late, shows up here as a changed bit)
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    for rep in range(3):
        with torch.cuda.stream(side):
            for k in range(1 + 2 * rep):
                junk[: (8 << 20) * (1 + k % 5)].add_(1)
        lo2 = ops.norm_head_fwd(x, stats2, Nb, Vn, G, W, hb, gamma, beta, True, cs)
        dg2, db2 = (torch.empty(C, device=DEV), torch.empty(C, device=DEV)) if affine else (None, None)
        gx2, pend2 = ops.norm_head_bwd(x, gl, stats2, Nb, Vn, G, W, gamma, beta, True, dg2, db2, cs)
        gw2, gb2 = torch.empty_like(W), torch.empty_like(hb)
        ops.norm_head_dparams(pend2, gw2, gb2)
        torch.cuda.synchronize()
        assert torch.equal(lo2, logits) and torch.equal(gx2, gx) and torch.equal(gw2, gw) and torch.equal(gb2, gb), f"repetition {rep} differs"
        if affine:
            assert torch.equal(dg2, dg) and torch.equal(db2, db), f"repetition {rep}: dgamma / dbeta differ"


@pytest.mark.parametrize("kind", ["gn", "in", "bn"])
@pytest.mark.parametrize("sp", [(24, 20, 28), (10, 9, 13)])
def test_first_layer_wgrad_with_norm_backward_on_load(kind, sp):
    """dycon_norm_bwd_stats + dycon_conv1_wgrad_normbwd (block_one: the normalisation's data gradient formed on load by the first
    convolution's weight gradient, never stored) against dycon_norm_bwd + dycon_conv_wgrad: dgamma / dbeta to fp32 round-off,
    dW / db to the noise of re-rounding the folded gz to bf16."""
    from dycon_paper_replication_amd._lib import CONV_K3
    rng = np.random.default_rng(zlib.crc32(repr((kind, sp)).encode()))
    B, C = 3, 16
    V = sp[0] * sp[1] * sp[2]
    x = torch.from_numpy(rng.standard_normal((B,) + sp + (1,)).astype(np.float32)).to(DEV, torch.bfloat16)
    z = torch.from_numpy(rng.standard_normal((B,) + sp + (C,)).astype(np.float32) * 1.3 + 0.2).to(DEV, torch.bfloat16)
    gy = torch.from_numpy(rng.standard_normal((B,) + sp + (C,)).astype(np.float32)).to(DEV, torch.bfloat16)
    affine = kind != "in"
    gamma = torch.from_numpy(rng.standard_normal(C).astype(np.float32) * 0.5 + 1.0).to(DEV) if affine else None
    beta = torch.from_numpy(rng.standard_normal(C).astype(np.float32) * 0.3).to(DEV) if affine else None
    Nb, G, Vn = (1, C, B * V) if kind == "bn" else (B, 16, V)
    _, stats = ops.norm_fwd(z, Nb, Vn, C, G, gamma, beta, True)
    mk = lambda: (torch.empty(C, device=DEV), torch.empty(C, device=DEV)) if affine else (None, None)   # noqa: E731
    dg_ref, db_ref = mk()
    gz = ops.norm_bwd(z, False, gy, stats, Nb, Vn, C, G, gamma, beta, True, dg_ref, db_ref)
    gw_ref, gb_ref = torch.empty(C, 1, 3, 3, 3, device=DEV), torch.empty(C, device=DEV)
    ops.conv_wgrad(x, gz, gw_ref, CONV_K3, 1, 27, 27, dbias=gb_ref)
    dg, db = mk()
    _, ab = ops.norm_bwd_stats(z, gy, stats, Nb, Vn, C, G, gamma, beta, True, dg, db)
    gw, gb = torch.empty_like(gw_ref), torch.empty_like(gb_ref)
    ops.conv1_wgrad_normbwd(x, z, gy, stats, ab, Nb, G, gw, gb, gamma, beta, True)
    torch.cuda.synchronize()
    if affine:      # same launches on the three-launch shapes; the small shape's reference is the one-launch norm backward (other summation order)
        close(dg, dg_ref, 1e-5, 1e-5 * float(dg_ref.abs().max()), "dgamma")
        close(db, db_ref, 1e-5, 1e-5 * float(db_ref.abs().max()), "dbeta")
    sw, sb = float(gw_ref.abs().max()), float(gb_ref.abs().max())
    assert float((gw - gw_ref).abs().max()) <= 2e-3 * sw, (float((gw - gw_ref).abs().max()), sw)
    # the bias gradient is sum(gz) over B*V voxels, analytically ~0 per group (the norm backward removes the mean): what is left is
    # rounding noise, so the bound is a fraction of sqrt(B*V) bf16 half-steps of rms(gz), not a relative one
    noise = 0.05 * (B * V) ** 0.5 * 2.0 ** -8 * float(gz.float().pow(2).mean().sqrt())
    assert float((gb - gb_ref).abs().max()) <= 2e-3 * sb + noise, (float((gb - gb_ref).abs().max()), sb, noise)


@pytest.mark.parametrize("kind", ["gn", "in", "bn"])
@pytest.mark.parametrize("sp", [(24, 20, 28), (10, 9, 13)])
def test_first_block_backward_in_one_pass(kind, sp):
    """dycon_first_block_bwd (block_one's whole backward as ONE pass: three tap correlations + per-channel sums, combined afterwards)
    against torch autograd in DOUBLE on the same bf16 inputs: no rounding of the data gradient enters dW, so the bound is fp32
    accumulation, not bf16."""
    rng = np.random.default_rng(zlib.crc32(repr(("fb", kind, sp)).encode()))
    B, C = 3, 16
    V = sp[0] * sp[1] * sp[2]
    xh = torch.from_numpy(rng.standard_normal((B, 1) + sp).astype(np.float32)).bfloat16()
    zh = torch.from_numpy(rng.standard_normal((B, C) + sp).astype(np.float32) * 1.3 + 0.2).bfloat16()
    gh = torch.from_numpy(rng.standard_normal((B, C) + sp).astype(np.float32)).bfloat16()
    affine = kind != "in"
    gamma = torch.from_numpy(rng.standard_normal(C).astype(np.float32) * 0.5 + 1.0) if affine else None
    beta = torch.from_numpy(rng.standard_normal(C).astype(np.float32) * 0.3) if affine else None
    # ---- reference: autograd in double
    zd = zh.double().requires_grad_(True)
    gd, bd = (gamma.double().requires_grad_(True), beta.double().requires_grad_(True)) if affine else (None, None)
    if kind == "gn":
        nd_ = F.group_norm(zd, 16, gd, bd, 1e-5)
    elif kind == "in":
        nd_ = F.instance_norm(zd, eps=1e-5)
    else:
        nd_ = F.batch_norm(zd, None, None, gd, bd, True, 0.1, 1e-5)
    F.relu(nd_).backward(gh.double())
    gz = zd.grad
    dw_ref = torch.nn.grad.conv3d_weight(xh.double(), (C, 1, 3, 3, 3), gz, padding=1)
    db_ref = gz.sum((0, 2, 3, 4))
    # ---- one pass on the GPU
    x, z, gy = nd(xh.float(), torch.bfloat16), nd(zh.float(), torch.bfloat16), nd(gh.float(), torch.bfloat16)
    Nb, G, Vn = (1, C, B * V) if kind == "bn" else (B, 16, V)
    gam, bet = (gamma.to(DEV), beta.to(DEV)) if affine else (None, None)
    _, stats = ops.norm_fwd(z, Nb, Vn, C, G, gam, bet, True)
    gw, gb = torch.empty(C, 1, 3, 3, 3, device=DEV), torch.empty(C, device=DEV)
    dg, dbt = (torch.empty(C, device=DEV), torch.empty(C, device=DEV)) if affine else (None, None)
    ops.first_block_bwd(x, z, gy, stats, Nb, G, gw, gb, gam, bet, True, dg, dbt)
    torch.cuda.synchronize()
    sw = float(dw_ref.abs().max())
    assert float((gw.cpu().double() - dw_ref).abs().max()) <= 2e-4 * sw, (float((gw.cpu().double() - dw_ref).abs().max()), sw)
    # sum(gz) is ~0 analytically: compare on the scale of the terms that cancel in it
    scale_b = float(gz.abs().sum((0, 2, 3, 4)).max())
    assert float((gb.cpu().double() - db_ref).abs().max()) <= 1e-5 * scale_b
    if affine:
        close(dg, gd.grad.float(), 2e-4, 2e-4 * float(gd.grad.abs().max()), "dgamma")
        close(dbt, bd.grad.float(), 2e-4, 2e-4 * float(bd.grad.abs().max()), "dbeta")
