"""Per-op timing of the V-Net's norm / conv launches at the bench shapes (B = 4, bf16).  Back-to-back launches on one
stream, HIP events: the figure includes the launch gaps, which is what a step pays.  usage: op_micro.py [reps]"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from dycon_paper_replication_amd import ops
from dycon_paper_replication_amd._lib import CONV_K3

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = "cuda:0"
B = 4


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def both(name, fn, nbytes=0, flops=0):
    us = timeit(fn)
    print(f"{name:44s} {us:8.1f} us   {nbytes / us / 1e3:8.1f} GB/s   {flops / us / 1e6:8.1f} TFLOP/s", flush=True)


for S, C in ((96, 16), (48, 32), (24, 64), (12, 128), (6, 256)):
    V = S ** 3
    x = torch.randn(B, S, S, S, C, device=dev).bfloat16()
    gy = torch.randn(B, S, S, S, C, device=dev).bfloat16()
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    y, stats = ops.norm_fwd(x, B, V, C, 16, gamma, beta, True)
    nb = x.numel() * 2
    both(f"norm_fwd GN {C}ch @{S}^3", lambda: ops.norm_fwd(x, B, V, C, 16, gamma, beta, True), 3 * nb)
    both(f"norm_bwd GN {C}ch @{S}^3", lambda: ops.norm_bwd(x, False, gy, stats, B, V, C, 16, gamma, beta, True, dg, db), 5 * nb)
    w = torch.randn(C, C, 3, 3, 3, device=dev) * 0.02
    b = torch.zeros(C, device=dev)
    if ops.conv_uses_lds(x, C, C) or C >= 64:
        wf = ops.pack_bfrag(w, torch.bfloat16, 27, C, C, C, 1, 27, 0, C * 27)
        both(f"conv k3 {C}->{C} @{S}^3", lambda: ops.conv_gemm(x, wf, b, CONV_K3, C, C), 2 * nb, 2 * B * V * 27 * C * C)
    gw, gb = torch.empty_like(w), torch.empty_like(b)
    both(f"wgrad k3 {C}->{C} @{S}^3", lambda: ops.conv_wgrad(x, gy, gw, CONV_K3, 1, 27, C * 27, dbias=gb), 2 * nb, 2 * B * V * 27 * C * C)

# block_nine's norm + out_conv: the fused pair against the launches it replaces (96^3, 16 channels)
from dycon_paper_replication_amd._lib import CONV_1X1
S, C = 96, 16
V = S ** 3
x = torch.randn(B, S, S, S, C, device=dev).bfloat16()
gl = torch.randn(B, S, S, S, 2, device=dev)
W = torch.randn(2, C, 1, 1, 1, device=dev) * 0.3
hb = torch.zeros(2, device=dev)
gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
dg, db, gw, gb = torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty_like(W), torch.empty_like(hb)
w_tcn, w_d = W.reshape(2, C).t().contiguous(), W.reshape(2, C).contiguous()
y, stats = ops.norm_fwd(x, B, V, C, 16, gamma, beta, True)
nb = x.numel() * 2


def unfused_fwd():
    yy, st = ops.norm_fwd(x, B, V, C, 16, gamma, beta, True)
    return ops.conv_direct(yy, w_tcn, hb, CONV_1X1, 2, torch.float32)


def fused_fwd():
    st = ops.norm_stats(x, B, V, C, 16)
    return ops.norm_head_fwd(x, st, B, V, 16, W, hb, gamma, beta, True)


def unfused_bwd():
    gy = ops.conv_direct(gl, w_d, None, CONV_1X1, C, torch.bfloat16)
    ops.norm_bwd(x, False, gy, stats, B, V, C, 16, gamma, beta, True, dg, db)
    ops.conv_wgrad(y, gl, gw, CONV_1X1, 0, 1, C, dbias=gb)


def fused_bwd():
    gx, pend = ops.norm_head_bwd(x, gl, stats, B, V, 16, W, gamma, beta, True, dg, db)
    ops.norm_head_dparams(pend, gw, gb)


both("norm + head fwd, unfused (4 launches)", unfused_fwd, 4 * nb)
both("norm + head fwd, fused (3 launches)", fused_fwd, 2 * nb)
both("head dgrad + norm bwd + head wgrad, unfused", unfused_bwd, 8 * nb)
both("norm_head bwd, fused", fused_bwd, 3 * nb)

# first layer's weight gradient (the last launch of the backward: nothing overlaps it)
x1 = torch.randn(B, S, S, S, 1, device=dev).bfloat16()
g16 = torch.randn(B, S, S, S, 16, device=dev).bfloat16()
gw1, gb1 = torch.empty(16, 1, 3, 3, 3, device=dev), torch.empty(16, device=dev)
both("wgrad k3 1->16 @96^3 (first layer)", lambda: ops.conv_wgrad(x1, g16, gw1, CONV_K3, 1, 27, 27, dbias=gb1), g16.numel() * 2 + x1.numel() * 2)
