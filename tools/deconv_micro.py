"""Transposed conv k2s2 forward (scatter epilogue) and the k2s2 data-gradient at the V-Net's levels, B = 4, bf16.  usage: deconv_micro.py [reps]"""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from dycon_paper_replication_amd import ops
from dycon_paper_replication_amd._lib import CONV_1X1

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = "cuda:0"
for cin, cout, S in ((32, 16, 48), (64, 32, 24), (128, 64, 12), (256, 128, 6)):
    x = torch.randn(4, S, S, S, cin, device=dev).bfloat16()
    w = torch.randn(cin, cout, 2, 2, 2, device=dev) * 0.05
    b = torch.zeros(cout, device=dev)
    wf = ops.pack_bfrag(w, torch.bfloat16, 1, cin, 8 * cout, cout, 0, cout * 8, 1, 8)
    y = None
    for _ in range(3):
        y = ops.conv_gemm(x, wf, b, CONV_1X1, 8 * cout, cout, scatter=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        y = ops.conv_gemm(x, wf, b, CONV_1X1, 8 * cout, cout, scatter=True)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"deconv {cin}->{cout} @ {S}^3 -> {2 * S}^3 x4: {us:.1f} us  {(x.numel() + y.numel()) * 2 / us / 1e3:.0f} GB/s   checksum {float(y.float().sum()):.4f}")
