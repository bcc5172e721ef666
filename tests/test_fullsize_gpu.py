"""GPU parity at the REAL sizes of BASELINE.json configs 2 and 3 (bf16 storage, B = 4 (2+2), 96^3 and 112x112x96).

Op level: the layer shapes the benchmark step actually launches -- the persistent 96^3 kernels walking ~27 tiles per workgroup
with the 2-deep halo prefetch, the LDS-halo kernel with its per-XCD tile remap, the bf16 weight-gradient kernel with its
contiguous split ranges -- against F.conv3d / F.group_norm (fp32, CPU) on the same bf16-rounded inputs.  Outputs are checked
ELEMENT-WISE (a wrong 4x8x8 tile is ~1e-4 of the voxels and invisible in a max-norm-relative bound taken over the whole tensor
only if the bound is loose; here every element must be within bf16 rounding of the fp32 value).

Step level: one full DyCON step at 96^3 / 112x112x96 / 112x112x80 (ISLES variants), B = 4 (2+2), against oracle/step.py run in
fp32 AND fp64: fp32 storage to the north-star's 1e-4 of the fp64 twin (or twice the oracle's own fp32 error) on every loss scalar
and the gradient norm, bf16 storage on the scalars and on the direction of the flat gradient measured against the ORACLE's
gradient, with PyTorch's own bf16 autocast of the oracle step as the noise budget (train_DyCON_BraTS19.py:147,
train_DyCON_Pancreas.py:99, train_DyCON_ISLES22.py:70,114,247).
"""
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from dycon_paper_replication_amd import ops
    from dycon_paper_replication_amd.engine import DropoutSpec
    from dycon_paper_replication_amd.trainer import DyconTrainer, TrainConfig
from oracle import nets as ON
from oracle import step as OS
from test_ops_gpu import mini_engine, nc, nd

DEV = "cuda:0"
T = torch.from_numpy
BF = torch.bfloat16


def assert_bf16_elementwise(got, ref, what, acc_noise=1e-3):
    """every element of a bf16-stored result within bf16 rounding (2^-8 relative, doubled) of the fp32 reference, plus fp32
    accumulation-order noise `acc_noise` x rms(ref).  Reports the worst 4x8x8 voxel tile on failure."""
    got, ref = got.float(), ref.float()
    rms = float(ref.pow(2).mean().sqrt())
    excess = (got - ref).abs() - (2.0 ** -7) * ref.abs() - acc_noise * rms
    worst = float(excess.max())
    if worst > 0:
        idx = np.unravel_index(int(excess.argmax()), excess.shape)
        bad = int((excess > 0).sum())
        tile = (idx[0], idx[2] // 4, idx[3] // 8, idx[4] // 8)
        raise AssertionError(f"{what}: {bad} elements off (worst excess {worst:.3e}, rms {rms:.3e}) first at {idx}, "
                             f"sample/tile(4x8x8) {tile}")


# kind, cin, cout, spatial -- the layers of the V-Net / U-Net step at 96^3 and 112x112x96, B = 4
FULL_CASES = [
    ("k3", 1, 16, (96, 96, 96)),      # conv_k3_c1 (persistent, first layer) and wgrad_k3_c1 (its weight gradient)
    ("k3", 1, 16, (50, 44, 46)),      # ... ragged tiles in all three directions
    ("k3", 16, 16, (96, 96, 96)),     # conv_k3_p16 (persistent, weight-stationary)
    ("k3", 16, 16, (112, 112, 96)),   # Pancreas geometry: partial tiles in H (112 = 14 x 8) ...
    ("k3", 32, 32, (48, 48, 48)),     # conv_k3_p32 (persistent, weights stationary in LDS)
    ("k3", 32, 32, (46, 50, 44)),     # ... ragged tiles in all three directions
    ("k3", 64, 64, (24, 24, 24)),     # conv_k3_lds <32,4,2>
    ("k3", 32, 32, (56, 56, 48)),     # Pancreas level 2
    ("k3", 64, 64, (28, 28, 24)),     # Pancreas level 3 (28 = 3.5 x 8: ragged tiles)
    ("k3", 48, 16, (96, 96, 96)),     # U-Net decoder conv1 (chunk-major 48-channel input)
    ("k3", 16, 32, (48, 48, 48)),     # U-Net conv2.conv1
    ("k3", 128, 128, (12, 12, 12)),   # conv_k3_tile (split-K)
    ("k3", 256, 256, (6, 6, 6)),
    ("k2s2", 16, 32, (96, 96, 96)),
    ("k2s2", 32, 64, (48, 48, 48)),
    ("deconv", 32, 16, (48, 48, 48)),
    ("deconv", 64, 32, (24, 24, 24)),
    ("1x1", 256, 512, (12, 12, 12)),  # projection head (bf16 MFMA 1x1 weight gradient, 54 row tiles over 16 splits)
    ("1x1", 512, 256, (12, 12, 12)),
    # BASELINE config 5 (ISLES22 geometry, train_DyCON_ISLES22.py:70 / the config's 112 x 112 x 80): every level of the V-Net at its
    # real grid, down to the odd 7 x 7 x 5 bottleneck and the k2s2 / transposed convolutions across the 7x7x5 <-> 14x14x10 edge
    ("k3", 16, 16, (112, 112, 80)),
    ("k3", 32, 32, (56, 56, 40)),
    ("k3", 64, 64, (28, 28, 20)),
    ("k3", 128, 128, (14, 14, 10)),
    ("k3", 256, 256, (7, 7, 5)),
    ("k2s2", 128, 256, (14, 14, 10)),
    ("deconv", 256, 128, (7, 7, 5)),
]


@pytest.mark.parametrize("kind,cin,cout,sp", FULL_CASES)
def test_conv_full_size_bf16(kind, cin, cout, sp):
    rng = np.random.default_rng(zlib.crc32(repr((kind, cin, cout, sp)).encode()))
    B = 4
    k = {"k3": 3, "k2s2": 2, "deconv": 2, "1x1": 1}[kind]
    wshape = (cin, cout, k, k, k) if kind == "deconv" else (cout, cin, k, k, k)
    w = T((rng.standard_normal(wshape) / np.sqrt(cin * k ** 3)).astype(np.float32))
    b = T(rng.standard_normal(cout).astype(np.float32))
    x = torch.randn((B, cin) + sp, generator=torch.Generator().manual_seed(int(rng.integers(1 << 30)))).bfloat16().float()
    xr = x.clone().requires_grad_(cin > 1)
    # the kernels multiply bf16-rounded weights; the rounded copy is the autograd LEAF (a gradient flowing back through a
    # .bfloat16() cast would itself be rounded to bf16)
    wr = w.bfloat16().float().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    wq = wr
    if kind == "k3":
        yr = F.conv3d(xr, wq, br, padding=1)
    elif kind == "k2s2":
        yr = F.conv3d(xr, wq, br, stride=2)
    elif kind == "1x1":
        yr = F.conv3d(xr, wq, br)
    else:
        yr = F.conv_transpose3d(xr, wq, br, stride=2)
    gy = torch.randn(tuple(yr.shape), generator=torch.Generator().manual_seed(7)).bfloat16().float()
    yr.backward(gy)

    e = mini_engine({"l.weight": w, "l.bias": b}, BF)
    xd = nd(x, BF)
    y = e._conv("l", xd, kind, need_gx=cin > 1)
    e.G[id(y)] = nd(gy, BF)
    for fn in reversed(e.tape):
        fn()
    torch.cuda.synchronize()
    assert_bf16_elementwise(nc(y), yr.detach(), f"y {kind} {cin}->{cout} @ {sp}")
    if cin > 1:
        assert_bf16_elementwise(nc(e.G[id(xd)]), xr.grad, f"gx {kind} {cin}->{cout} @ {sp}")
    # weight / bias gradients: fp32 sums over up to 3.5 M voxels of exact bf16 x bf16 products; only the summation order differs
    # from the CPU's.  One dropped 4x4x8 tile would move an element by ~1/sqrt(#tiles) ~ 1e-2 of the scale.
    for name, got, ref in (("gw", e.g["l.weight"], wr.grad), ("gb", e.g["l.bias"], br.grad)):
        err = float((got.cpu() - ref).abs().max())
        scale = float(ref.abs().max())
        assert err <= 1e-3 * scale, f"{name} {kind} {cin}->{cout} @ {sp}: max err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("C,G,sp,mode", [(16, 16, (96, 96, 96), "relu"), (16, 16, (96, 96, 96), "skip"), (32, 16, (48, 48, 48), "relu"),
                                         (16, 16, (112, 112, 96), "relu"), (64, 16, (24, 24, 24), "skip")])
def test_groupnorm_full_size_bf16(C, G, sp, mode):
    """GroupNorm(16) + ReLU (+ skip add) forward / backward at the step's real shapes, bf16 storage, B = 4."""
    gen = torch.Generator().manual_seed(C + sp[0])
    B = 4
    z = (torch.randn((B, C) + sp, generator=gen) * 1.5 + 0.3).bfloat16().float()
    gamma = 1 + 0.2 * torch.randn(C, generator=gen)
    beta = 0.2 * torch.randn(C, generator=gen)
    skip = torch.randn((B, C) + sp, generator=gen).bfloat16().float() if mode == "skip" else None
    zr = z.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    yr = F.relu(F.group_norm(zr, G, gr, br, 1e-5))
    if skip is not None:
        yr = yr + skip
    gy = torch.randn(tuple(yr.shape), generator=gen).bfloat16().float()
    yr.backward(gy)
    e = mini_engine({"n.weight": gamma, "n.bias": beta}, BF)
    zd = nd(z, BF)
    sd = nd(skip, BF) if skip is not None else None
    y = e._norm("n", zd, "gn", relu=True, skip=sd)
    ycopy = y.clone()
    e.G[id(y)] = nd(gy, BF)
    for fn in reversed(e.tape):
        fn()
    torch.cuda.synchronize()
    assert_bf16_elementwise(nc(ycopy), yr.detach(), f"gn y C={C} @ {sp}")
    assert_bf16_elementwise(nc(e.G[id(zd)]), zr.grad, f"gn gz C={C} @ {sp}", acc_noise=2e-3)
    for name, got, ref in (("dgamma", e.g["n.weight"], gr.grad), ("dbeta", e.g["n.bias"], br.grad)):
        err = float((got.cpu() - ref).abs().max())
        assert err <= 1e-3 * float(ref.abs().max()) + 1e-3, f"{name}: {err}"


def _double(p):
    return {k: (v.double() if v.is_floating_point() else v) for k, v in p.items()}


KEYS = ("loss", "ce", "dice", "cons", "fecl", "uncl")
# (name, patch, oracle StepConfig overrides, TrainConfig overrides)
STEP_CASES = [
    ("brats19", (96, 96, 96), {}, {}),                                         # BASELINE config 2 (train_DyCON_BraTS19.py:147)
    ("pancreas", (112, 112, 96), {}, {}),                                      # config 3 (train_DyCON_Pancreas.py:99)
    # config 5's per-GPU geometry with the ISLES script's variants (train_DyCON_ISLES22.py:114 teacher .eval(), :247 DiceLoss(2)) at
    # feature_scaler 2 (N = 14 x 14 x 10 = 1960 embeddings, which the CPU oracle can materialise; N = 15 680 = scaler 4 is held to
    # the row-block oracle in test_ops_gpu.py::test_fecl_isles_size_vs_rowblock_oracle)
    ("isles22", (112, 112, 80), dict(dice_variant="multiclass", teacher_bn_training=False),
     dict(dice_variant="multiclass", teacher_mode="eval")),
]


@pytest.mark.parametrize("name,patch,okw,tkw", STEP_CASES, ids=[c[0] for c in STEP_CASES])
def test_step_full_size_vs_oracle(name, patch, okw, tkw):
    """BASELINE configs 2, 3 and 5 (per-GPU geometry): one V-Net step, B = 4 (2+2), against the CPU oracle step, which is run in
    fp32, in fp64 (the footing of every fp32 bound: a value must be within max(1e-4, 2 x the oracle's OWN fp32 error) of the fp64
    twin) and under torch.autocast(bf16) (the budget of the bf16-storage step).  fp32 storage: every loss scalar, the gradient norm,
    the gradient's direction and updated parameters.  bf16 storage: scalars within 3e-2 and the flat gradient's direction against
    the ORACLE's gradient."""
    from dycon_paper_replication_amd.synthetic import make_batch
    B, LB = 4, 2
    vol, lab, noise = make_batch(77, B, patch)
    ocfg = OS.StepConfig(net_type="vnet", labeled_bs=LB, **okw)
    st = OS.StepState(student=ON.make_vnet_params(41), teacher=ON.make_vnet_params(42))
    ref = OS.train_step(ocfg, st, vol, lab, noise, 5.0, 0)
    st64 = OS.StepState(student=_double(ON.make_vnet_params(41)), teacher=_double(ON.make_vnet_params(42)))
    ref64 = OS.train_step(ocfg, st64, vol.double(), lab, noise.double(), 5.0, 0)
    exp = np.array([float(ref[k]) for k in KEYS])
    exp64 = np.array([float(ref64[k]) for k in KEYS])
    names = list(ref["grads"])
    gref = torch.cat([ref["grads"][k].reshape(-1) for k in names]).double()
    gref64 = torch.cat([ref64["grads"][k].reshape(-1) for k in names])
    with torch.autocast("cpu", dtype=torch.bfloat16):
        st_ac = OS.StepState(student=ON.make_vnet_params(41), teacher=ON.make_vnet_params(42))
        ref_ac = OS.train_step(ocfg, st_ac, vol, lab, noise, 5.0, 0)
    gac = torch.cat([ref_ac["grads"][k].float().reshape(-1) for k in names]).double()
    cos_autocast = float((gac * gref64).sum() / (gac.norm() * gref64.norm()))
    print(f"torch.autocast(bf16) oracle step vs fp64 oracle step: grad cos {cos_autocast:.6f}; oracle fp32 vs fp64: scalars "
          f"{np.abs(exp - exp64)}, grad norm {float(ref['grad_norm']):.7f} vs {float(ref64['grad_norm']):.7f}")
    off = DropoutSpec("off")
    for dt in (torch.float32, BF):
        tr = DyconTrainer(TrainConfig(model="vnet", labeled_bs=LB, batch_size=B, dtype=dt, **tkw), DEV,
                          student_init=ON.make_vnet_params(41), teacher_init=ON.make_vnet_params(42))
        out = tr.step(vol.to(DEV), lab.to(DEV), noise=noise.to(DEV), s_drop=off, t_drop=off, epoch=0, beta=5.0)
        got = np.array([float(out[k]) for k in KEYS])
        ggot = torch.cat([tr.g[k].reshape(-1) for k in names]).double().cpu()
        cos = float((ggot * gref64).sum() / (ggot.norm() * gref64.norm()))
        ratio = float(ggot.norm() / gref64.norm())
        print(f"step {name} {patch} {dt}: scalars rel err vs fp64 {np.abs(got - exp64) / (np.abs(exp64) + 1e-12)}, grad cos {cos:.7f}, norm ratio {ratio:.6f}")
        if dt == BF:      # where the direction error sits (share of |g_hip - g_ref|^2 per parameter)
            tot = float((ggot - gref64).pow(2).sum())
            rows = sorted(((float((tr.g[k].double().cpu() - ref64["grads"][k]).pow(2).sum()) / tot, k,
                            float(ref64["grads"][k].norm()), float(tr.g[k].norm())) for k in names), reverse=True)[:12]
            for share, k, nr, ng in rows:
                print(f"    {share:6.3f} of the error in {k}: |ref| {nr:.3e} |hip| {ng:.3e}")
        if dt == torch.float32:
            for k, g, e32, e64 in zip(KEYS, got, exp, exp64):
                tol = max(1e-4 * abs(e64) + 1e-6, 2 * abs(e32 - e64))
                assert abs(g - e64) <= tol, f"{k}: hip {g!r} vs fp64 oracle {e64!r} (oracle fp32 {e32!r}), tolerance {tol:.3e}"
            gn, gn32, gn64 = float(out["grad_sumsq"].sqrt()), float(ref["grad_norm"]), float(ref64["grad_norm"])
            assert abs(gn - gn64) <= max(1e-4 * gn64, 2 * abs(gn32 - gn64)), (gn, gn32, gn64)
            cos32 = float((gref * gref64).sum() / (gref.norm() * gref64.norm()))
            assert 1.0 - cos <= max(1e-5, 2 * (1.0 - cos32)), (cos, cos32)
            for k in ("block_one.conv.0.weight", "block_nine.conv.0.weight", "block_five.conv.3.weight", "out_conv.weight"):
                np.testing.assert_allclose(tr.p[k].cpu().numpy(), st64.student[k].numpy(), rtol=1e-4, atol=1e-6, err_msg=k)
        else:
            np.testing.assert_allclose(got, exp64, rtol=3e-2, atol=3e-3)
            # budget for the direction error: what PyTorch's OWN bf16 mixed precision (torch.autocast on the oracle step, same
            # inputs and weights) loses against the fp64 truth.  Measured at 96^3: autocast 0.99229, this path 0.99268 -- the
            # bf16-storage step sits at the noise floor of bf16 training itself; a tile / range bug costs far more (and is caught
            # element-wise by the op-level tests above).
            assert 1.0 - cos <= 1.5 * (1.0 - cos_autocast) + 1e-4 and cos >= 0.99 and 0.97 < ratio < 1.03, (cos, cos_autocast, ratio)
        del tr
        torch.cuda.empty_cache()


@pytest.mark.parametrize("sp", [(48, 48, 48), (46, 50, 44)])
def test_conv_p32_on_32x32x16_mfma(sp, monkeypatch):
    """conv_k3_p32x_kernel (DYCON_P32X=1: the persistent 48^3 kernel on v_mfma_f32_32x32x16_bf16 -- measured slower, off by default)
    against F.conv3d element-wise, against the default kernel within one bf16 step (the k order inside a tap differs), and its
    statistics epilogue against the default kernel's."""
    from dycon_paper_replication_amd._lib import CONV_K3
    rng = np.random.default_rng(sp[1])
    B, C = 4, 32
    x = torch.from_numpy(rng.standard_normal((B,) + sp + (C,)).astype(np.float32)).to(DEV, BF)
    w = torch.from_numpy((rng.standard_normal((C, C, 3, 3, 3)) / np.sqrt(27 * C)).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(C).astype(np.float32))
    wf = ops.pack_bfrag(w.to(DEV), BF, 27, C, C, C, 1, 27, 0, C * 27)
    chunks = ops.conv_stats_chunks(x, C, C)
    res = {}
    for v in ("0", "1"):
        monkeypatch.setenv("DYCON_P32X", v)
        y = ops.conv_gemm(x, wf, b.to(DEV), CONV_K3, C, C)
        y2, part = ops.conv_gemm_stats(x, wf, b.to(DEV), C, chunks)
        torch.cuda.synchronize()
        assert torch.equal(y, y2)
        res[v] = (y, part.reshape(B, chunks, C, 2).sum(1))
    ref = F.conv3d(x.float().cpu().permute(0, 4, 1, 2, 3), w.bfloat16().float(), b, padding=1)
    assert_bf16_elementwise(nc(res["1"][0]), ref, f"p32x y @ {sp}")
    assert_bf16_elementwise(nc(res["0"][0]), ref, f"p32 y @ {sp}")
    d = (res["1"][0].float() - res["0"][0].float()).abs()
    assert float((d > 2.0 ** -7 * res["0"][0].float().abs() + 1e-3).float().mean()) == 0.0
    np.testing.assert_allclose(res["1"][1].cpu().numpy(), res["0"][1].cpu().numpy(), rtol=2e-3, atol=0.5)


@pytest.mark.parametrize("cin,C,sp", [(32, 32, (48, 48, 48)), (32, 32, (46, 50, 44)), (16, 16, (96, 96, 96)), (16, 16, (50, 44, 46)),
                                      (1, 16, (96, 96, 96)), (1, 16, (50, 44, 46))])
@pytest.mark.parametrize("kind", ["gn", "in"])
def test_conv_takes_norm_statistics(cin, C, sp, kind):
    """dycon_conv_gemm_stats + dycon_norm_fwd_parts / dycon_norm_stats_parts (the persistent convolutions of the 48^3 and 96^3 levels add
    up {sum, sum of squares} of the values they store, the normalisation that follows skips its statistics pass) against
    dycon_conv_gemm + dycon_norm_fwd: same convolution output bit for bit, statistics to fp32 summation order, normalised output
    within one bf16 step.  Ragged shapes: partial tiles must not count their padding."""
    from dycon_paper_replication_amd._lib import CONV_K3
    rng = np.random.default_rng(zlib.crc32(repr((cin, sp, kind)).encode()))
    B = 4
    V = sp[0] * sp[1] * sp[2]
    x = torch.from_numpy(rng.standard_normal((B,) + sp + (cin,)).astype(np.float32)).to(DEV, BF)
    w = torch.from_numpy((rng.standard_normal((C, cin, 3, 3, 3)) / np.sqrt(27 * cin)).astype(np.float32)).to(DEV)
    b = torch.from_numpy(rng.standard_normal(C).astype(np.float32)).to(DEV)
    gamma = torch.from_numpy(rng.standard_normal(C).astype(np.float32) * 0.5 + 1.0).to(DEV) if kind == "gn" else None
    beta = torch.from_numpy(rng.standard_normal(C).astype(np.float32) * 0.3).to(DEV) if kind == "gn" else None
    G = 16 if kind == "gn" else C
    wf = ops.pack_bfrag(w, BF, 27, cin, C, C, 1, 27, 0, cin * 27)
    chunks = ops.conv_stats_chunks(x, cin, C)
    assert chunks > 0
    y_ref = ops.conv_gemm(x, wf, b, CONV_K3, C, C)
    n_ref, stats_ref = ops.norm_fwd(y_ref, B, V, C, G, gamma, beta, True)
    y, part = ops.conv_gemm_stats(x, wf, b, C, chunks)
    n, stats = ops.norm_fwd_parts(y, part, chunks, B, V, C, G, gamma, beta, True)
    stats2 = ops.norm_stats_parts(y, part, chunks, B, V, C, G)
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref)
    assert torch.equal(stats, stats2)
    np.testing.assert_allclose(stats.cpu().numpy(), stats_ref.cpu().numpy(), rtol=2e-5, atol=2e-6)
    err = float((n.float() - n_ref.float()).abs().max())
    assert err <= 2.0 ** -7 * float(n_ref.float().abs().max()), err
    assert float((n != n_ref).float().mean()) < 1e-3


def test_engine_with_producer_side_statistics():
    """Engine.conv_stats / conv_stats96 (the persistent convolutions of the 48^3 and 96^3 levels hand the statistics of their output to the
    normalisation that follows -- block_one through _first_block, block_nine through the fused norm + head): V-Net, bf16, B = 2 at 96^3,
    forward and parameter gradients against the default launch sequence.  Same convolution outputs; the statistics differ in
    summation order only, so the logits agree to bf16 round-off and the flat gradient keeps its direction."""
    from dycon_paper_replication_amd.engine import Engine, param_spec, projection_buffers
    p_all = ON.make_vnet_params(21)
    spec = param_spec("vnet")
    x = torch.randn(2, 96, 96, 96, 1, generator=torch.Generator().manual_seed(3)).to(DEV)
    r1 = torch.randn(2, 96, 96, 96, 2, generator=torch.Generator().manual_seed(4)).to(DEV)
    r2 = torch.randn(2, 12, 12, 12, 256, generator=torch.Generator().manual_seed(5)).to(DEV, BF)
    res = []
    for on in (False, True):
        params = {k: p_all[k].to(DEV).contiguous() for k in spec}
        grads = {k: torch.zeros_like(v) for k, v in params.items()}
        bufs = {k: p_all[k].to(DEV) for k in projection_buffers()}
        eng = Engine("vnet", params, grads, bufs, dtype=BF)
        eng.conv_stats = eng.conv_stats96 = on
        logits, feats, _ = eng.forward(x, training=True, record=True)
        assert not eng._stat_parts                      # every partial a convolution left behind was consumed by its normalisation
        lo, fe = logits.clone(), feats.float().clone()
        eng.backward(r1, r2)
        torch.cuda.synchronize()
        res.append((lo, fe, torch.cat([grads[k].reshape(-1) for k in spec if not k.startswith("final.")]).double()))
    (l0, f0, g0), (l1, f1, g1) = res
    # (a statistic whose last bit differs moves single bf16 values of the first layers by one step, 2^-8; forty layers on, that is
    #  1e-3 of the logits in the L2 norm and a few 1e-2 of their range at the worst voxel -- bf16 storage against fp32 is 8e-2 there)
    assert float((l1 - l0).norm() / l0.norm()) <= 1e-2 and float((l1 - l0).abs().max()) <= 5e-2 * float(l0.abs().max())
    assert float((f1 - f0).norm() / f0.norm()) <= 2e-2 and float((f1 - f0).abs().max()) <= 1e-1 * float(f0.abs().max())
    # two bf16-storage backward passes whose forward tensors differ in scattered last bits carry INDEPENDENT rounding noise: each sits
    # at cos ~0.9925 of the fp64 gradient (test_step_full_size_vs_oracle), so they sit at ~0.985 of one another
    cos = float((g0 * g1).sum() / (g0.norm() * g1.norm()))
    assert cos > 0.97 and 0.97 < float(g1.norm() / g0.norm()) < 1.03, cos
