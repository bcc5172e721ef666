#!/usr/bin/env python3
"""Bisect the V-Net backward: gradient w.r.t. every intermediate activation (conv outputs z, block outputs y) on the HIP engine (fp32)
against torch autograd in double on the same graph, for the student forward of shard [r, 2+r] of make_batch(9, 4, 32^3) and the
CE + Dice gradient of the labelled sample.   usage: bwd_bisect.py [r] [dtype]"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dycon_paper_replication_amd.engine import Engine, param_spec, projection_buffers  # noqa: E402
from dycon_paper_replication_amd.synthetic import make_batch  # noqa: E402
from oracle import losses as OL  # noqa: E402
from oracle import nets as ON  # noqa: E402

r = int(sys.argv[1]) if len(sys.argv) > 1 else 1
vol, lab, _ = make_batch(9, 4, (32, 32, 32))
idx = [r, 2 + r]
x, label = vol[idx].double(), lab[idx]
p = {k: (v.double() if v.is_floating_point() else v) for k, v in ON.make_vnet_params(5).items()}
names = list(ON.trainable(p))
leaves = {k: p[k].clone().requires_grad_(True) for k in names}
P = {**p, **leaves}
inter = {}


def keep(name, t):
    t.retain_grad()
    inter[name] = t
    return t


def block(t, name):
    i = 0
    while f"{name}.conv.{3 * i}.weight" in P:
        z = keep(f"{name}.conv.{3 * i}.z", F.conv3d(t, P[f"{name}.conv.{3 * i}.weight"], P[f"{name}.conv.{3 * i}.bias"], padding=1))
        t = keep(f"{name}.conv.{3 * i + 1}.y", F.relu(F.group_norm(z, 16, P[f"{name}.conv.{3 * i + 1}.weight"], P[f"{name}.conv.{3 * i + 1}.bias"], 1e-5)))
        i += 1
    return t


def down(t, name):
    z = keep(f"{name}.conv.0.z", F.conv3d(t, P[f"{name}.conv.0.weight"], P[f"{name}.conv.0.bias"], stride=2))
    return keep(f"{name}.conv.1.y", F.relu(F.group_norm(z, 16, P[f"{name}.conv.1.weight"], P[f"{name}.conv.1.bias"], 1e-5)))


def up(t, name, skip):
    z = keep(f"{name}.conv.0.z", F.conv_transpose3d(t, P[f"{name}.conv.0.weight"], P[f"{name}.conv.0.bias"], stride=2))
    return keep(f"{name}.conv.1.y", F.relu(F.group_norm(z, 16, P[f"{name}.conv.1.weight"], P[f"{name}.conv.1.bias"], 1e-5)) + skip)


x1 = block(x, "block_one")
x2 = block(down(x1, "block_one_dw"), "block_two")
x3 = block(down(x2, "block_two_dw"), "block_three")
x4 = block(down(x3, "block_three_dw"), "block_four")
x5 = block(down(x4, "block_four_dw"), "block_five")
u = up(x5, "block_five_up", x4)
u = up(block(u, "block_six"), "block_six_up", x3)
u = up(block(u, "block_seven"), "block_seven_up", x2)
u = up(block(u, "block_eight"), "block_eight_up", x1)
x9 = block(u, "block_nine")
logits = keep("out_conv.z", F.conv3d(x9, P["out_conv.weight"], P["out_conv.bias"]))
sp = F.softmax(logits, 1)
loss = F.cross_entropy(logits[:1], label[:1]) + OL.dice_loss(sp[:1, 1], label[:1] == 1)
loss.backward()

# ---- HIP engine, fp32
dev = "cuda:0"
spec = param_spec("vnet")
params = {k: p[k].float().to(dev).contiguous() for k in spec}
grads = {k: torch.full_like(v, float("nan")) for k, v in params.items()}
bufs = {k: p[k].to(dev) for k in projection_buffers()}
eng = Engine("vnet", params, grads, bufs, dtype=torch.float32)
nm = {}
oc, on_, ot = Engine._conv, Engine._norm, Engine._take


def conv_(self, name, *a, **kw):
    y = oc(self, name, *a, **kw)
    nm[id(y)] = name + ".z"
    return y


def norm_(self, prefix, *a, **kw):
    y = on_(self, prefix, *a, **kw)
    nm[id(y)] = str(prefix) + ".y"
    return y


got = {}


def take_(self, t):
    g = ot(self, t)
    if id(t) in nm:
        got[nm[id(t)]] = g.detach().clone()
    return g


Engine._conv, Engine._norm, Engine._take = conv_, norm_, take_
tens = {}
_oc2 = Engine._conv


def conv2_(self, name, *a, **kw):
    y = _oc2(self, name, *a, **kw)
    tens[name + ".z"] = y
    return y


Engine._conv = conv2_
calls = {}
_onb = ops_norm_bwd = None
from dycon_paper_replication_amd import ops as _ops  # noqa: E402
_onb = _ops.norm_bwd


def nb_(src, from_y, gy, stats, Nb, V, C, G, *a, **kw):
    out = _onb(src, from_y, gy, stats, Nb, V, C, G, *a, **kw)
    for k_, t_ in tens.items():
        if t_.data_ptr() == src.data_ptr():
            calls[k_] = (src.clone(), gy.clone(), stats.clone(), out.clone())
    return out


_ops.norm_bwd = nb_
lg, feats, _ = eng.forward(x.float().permute(0, 2, 3, 4, 1).contiguous().to(dev), training=True, record=True)
print("logits max err", float((lg.cpu().permute(0, 4, 1, 2, 3).double() - logits.detach()).abs().max()))
g_logits = inter["out_conv.z"].grad.float().permute(0, 2, 3, 4, 1).contiguous().to(dev)
eng.backward(g_logits, None)
torch.cuda.synchronize()
order = [k for k in inter][::-1]
for k in order:
    if k not in got:
        continue
    ref = inter[k].grad
    g = got[k].cpu().permute(0, 4, 1, 2, 3).double()
    e = float((g - ref).norm() / (ref.norm() + 1e-30))
    per = [float((g[b] - ref[b]).norm() / (ref[b].norm() + 1e-30)) for b in range(g.shape[0])]
    flag = "   <<<<" if e > 1e-4 else ""
    print(f"{k:28s} {tuple(ref.shape)}  rel L2 err {e:.2e}  per sample {['%.1e' % v for v in per]}{flag}")
    if k in ("block_eight_up.conv.0.z", "block_eight.conv.3.z", "block_nine.conv.0.z"):
        z = inter[k].detach()[0]                                  # pre-norm tensor of sample 0
        C = z.shape[0]
        for c in range(C):
            ec = float((g[0, c] - ref[0, c]).norm() / (ref[0].norm() / C ** 0.5 + 1e-30))
            zc = z[c]
            print(f"      ch {c:2d}: err/typical {ec:.2e}   z mean {float(zc.mean()):+.3e} std {float(zc.std()):.3e}  |mean|/std {float(zc.mean().abs() / zc.std()):.2f}"
                  f"   frac relu-active {float((inter[k.replace('.0.z', '.1.y').replace('.3.z', '.4.y')].detach()[0, c] > 0).double().mean()):.3f}")
for k in ("out_conv.weight", "block_nine.conv.0.weight", "block_eight_up.conv.0.weight", "block_eight.conv.3.weight", "block_eight.conv.4.bias",
          "block_one.conv.0.weight"):
    ref = leaves[k].grad
    print(f"param {k:30s} rel err {float((grads[k].cpu().double() - ref).norm() / ref.norm()):.2e}")

# ---- the first failing op in isolation: GroupNorm backward of block_eight_up with the oracle's own inputs
from dycon_paper_replication_amd import _lib, ops  # noqa: E402
for key, ykey, C in (("block_eight_up.conv.0.z", "block_eight_up.conv.1.y", 16), ("block_eight.conv.3.z", "block_eight.conv.4.y", 32)):
    z64, gy64, gz64 = inter[key].detach(), inter[ykey].grad, inter[key].grad
    B_, _, D_, H_, W_ = z64.shape
    V = D_ * H_ * W_
    zd = z64.float().permute(0, 2, 3, 4, 1).contiguous().to(dev)
    gyd = gy64.float().permute(0, 2, 3, 4, 1).contiguous().to(dev)
    pre = key[:-3]
    npre = ykey[:-2]
    gamma, beta = params[npre + ".weight"], params[npre + ".bias"]
    stats = ops.norm_stats(zd, B_, V, C, 16)
    m64 = z64.reshape(B_, 16, -1).mean(-1)
    v64 = z64.reshape(B_, 16, -1).var(-1, unbiased=False)
    st = stats.cpu().reshape(B_, 16, 2).double()
    print(key, "stats: mean err", float((st[..., 0] - m64).abs().max()), "rstd rel err", float(((st[..., 1] - (v64 + 1e-5).rsqrt()) / (v64 + 1e-5).rsqrt()).abs().max()))
    nws = ops.query("dycon_norm_workspace", B_, V, C)
    ws = torch.zeros(nws // 4, dtype=torch.float32, device=dev)
    gz = torch.empty_like(gyd)
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    _lib.call("dycon_norm_bwd", zd.data_ptr(), 0, gyd.data_ptr(), gz.data_ptr(), 0, B_, V, C, 16, stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
              1, None, dg.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel() * 4, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    g = gz.cpu().permute(0, 4, 1, 2, 3).double()
    for c in range(C):
        ec = float((g[0, c] - gz64[0, c]).norm() / (gz64[0].norm() / C ** 0.5))
        if ec > 1e-5:
            print(f"   standalone norm_bwd {key} ch {c}: err/typical {ec:.2e}")
    # the reduction the kernel should have produced: A_g = sum_c gamma_c sum_v g', B_g = sum_c gamma_c sum_v g' xhat
    cpg = C // 16
    xh = (z64 - m64.repeat_interleave(cpg, 1).view(B_, C, 1, 1, 1)) * (v64 + 1e-5).rsqrt().repeat_interleave(cpg, 1).view(B_, C, 1, 1, 1)
    g64 = gamma.cpu().double().view(1, C, 1, 1, 1)
    b64 = beta.cpu().double().view(1, C, 1, 1, 1)
    gm = gy64 * ((g64 * xh + b64) > 0)
    A = (g64 * gm).reshape(B_, 16, -1).sum(-1)
    Bq = (g64 * gm * xh).reshape(B_, 16, -1).sum(-1)
    import math
    chunks = -(-V // max(64, -(-V // 512)))
    ab = ws[B_ * chunks * C * 2: B_ * chunks * C * 2 + B_ * 16 * 2].cpu().double().reshape(B_, 16, 2)
    print("   A rel err per group (sample 0):", ["%.1e" % float(abs(ab[0, gq, 0] - A[0, gq]) / (A[0].abs().max())) for gq in range(16)])
    print("   B rel err per group (sample 0):", ["%.1e" % float(abs(ab[0, gq, 1] - Bq[0, gq]) / (Bq[0].abs().max())) for gq in range(16)])

# ---- what the ENGINE fed to that norm backward
for key, ykey, C in (("block_eight_up.conv.0.z", "block_eight_up.conv.1.y", 16),):
    src, gy_e, st_e, out_e = calls[key]
    z64, gy64, gz64 = inter[key].detach(), inter[ykey].grad, inter[key].grad
    ncd = lambda t: t.cpu().permute(0, 4, 1, 2, 3).double()  # noqa: E731
    print("engine z vs oracle z: max abs", float((ncd(src) - z64).abs().max()))
    for c in (11, 12, 13):
        print(f"  ch {c}: z err {float((ncd(src)[0, c] - z64[0, c]).abs().max()):.2e}  gy err {float((ncd(gy_e)[0, c] - gy64[0, c]).norm() / gy64[0, c].norm()):.2e}"
              f"  gz err {float((ncd(out_e)[0, c] - gz64[0, c]).norm() / gz64[0, c].norm()):.2e}")
    B_ = z64.shape[0]
    m64 = z64.reshape(B_, 16, -1).mean(-1)
    v64 = z64.reshape(B_, 16, -1).var(-1, unbiased=False)
    st = st_e.cpu().reshape(B_, 16, 2).double()
    print("  engine stats: mean err per group", ["%.1e" % float(abs(st[0, g_, 0] - m64[0, g_])) for g_ in range(16)])
    print("  engine stats: rstd rel err     ", ["%.1e" % float(abs(st[0, g_, 1] - (v64[0, g_] + 1e-5).rsqrt()) / (v64[0, g_] + 1e-5).rsqrt()) for g_ in range(16)])

# ---- same op, engine-captured inputs, workspace poisoned / zeroed
src, gy_e, st_e, out_e = calls["block_eight_up.conv.0.z"]
C = 16
V = src.shape[1] * src.shape[2] * src.shape[3]
npre = "block_eight_up.conv.1"
for fill in (0.0, 1e30, float("nan")):
    nws = ops.query("dycon_norm_workspace", 2, V, C)
    ws = torch.full((nws // 4,), fill, dtype=torch.float32, device=dev)
    gz = torch.empty_like(gy_e)
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    _lib.call("dycon_norm_bwd", src.data_ptr(), 0, gy_e.data_ptr(), gz.data_ptr(), 0, 2, V, C, 16, st_e.data_ptr(), params[npre + ".weight"].data_ptr(),
              params[npre + ".bias"].data_ptr(), 1, None, dg.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel() * 4, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    d = (gz - out_e).abs().max().item()
    g = gz.cpu().permute(0, 4, 1, 2, 3).double()
    gz64 = inter["block_eight_up.conv.0.z"].grad
    print(f"ws fill {fill}: max |gz - engine gz| {d:.3e}; ch12 err vs oracle {float((g[0, 12] - gz64[0, 12]).norm() / gz64[0, 12].norm()):.2e}; nan count {int(torch.isnan(gz).sum())}")

# ---- how many voxels carry that error, and where do they sit relative to the ReLU threshold?
key, ykey = "block_eight_up.conv.0.z", "block_eight_up.conv.1.y"
g = calls[key][3].cpu().permute(0, 4, 1, 2, 3).double()
gz64, z64 = inter[key].grad, inter[key].detach()
m64 = z64.reshape(2, 16, -1).mean(-1).view(2, 16, 1, 1, 1)
r64 = (z64.reshape(2, 16, -1).var(-1, unbiased=False) + 1e-5).rsqrt().view(2, 16, 1, 1, 1)
pre = params["block_eight_up.conv.1.weight"].cpu().double().view(1, 16, 1, 1, 1) * (z64 - m64) * r64 + params["block_eight_up.conv.1.bias"].cpu().double().view(1, 16, 1, 1, 1)
err = (g - gz64).abs()
big = err > 0.05 * gz64.abs().max()
print("voxels with a large gradient error:", int(big.sum()), "of", big.numel())
for ix in big.nonzero()[:8]:
    ix = tuple(int(v) for v in ix)
    zf = calls[key][0].cpu().permute(0, 4, 1, 2, 3)[ix].item()
    print(f"   at {ix}: pre-activation in double {float(pre[ix]):+.3e}  (engine fp32 z {zf:+.7f}, double z {float(z64[ix]):+.9f})  incoming gradient {float(inter[ykey].grad[ix]):+.3e}"
          f"  engine gz {float(g[ix]):+.3e} oracle gz {float(gz64[ix]):+.3e}")

e12 = err[0, 12].reshape(-1)
top = torch.topk(e12, 6)
print("ch 12: ||err|| %.3e, ||gz|| %.3e; top elementwise errors:" % (float(e12.norm()), float(gz64[0, 12].norm())))
for v_, i_ in zip(top.values, top.indices):
    ix = (0, 12) + tuple(int(q) for q in torch.unravel_index(i_, gz64.shape[2:]))
    print(f"   |err| {float(v_):.3e} at {ix}: pre-activation (double) {float(pre[ix]):+.3e}, incoming g {float(inter[ykey].grad[ix]):+.3e}, engine gz {float(g[ix]):+.3e}, oracle gz {float(gz64[ix]):+.3e}")
rest = e12.clone()
rest[top.indices] = 0
print("   ||err|| without those 6 voxels: %.3e" % float(rest.norm()))
