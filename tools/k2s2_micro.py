"""k=2 stride-2 convolution forward (gather) and the transposed convolution's data-gradient at the V-Net's levels, B = 4, bf16.  usage: k2s2_micro.py [reps]"""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from dycon_paper_replication_amd import ops
from dycon_paper_replication_amd._lib import CONV_K2S2

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = "cuda:0"
torch.manual_seed(0)
for cin, cout, S in ((16, 32, 96), (32, 64, 48), (64, 128, 24)):
    x = torch.randn(4, S, S, S, cin, device=dev).bfloat16()
    w = torch.randn(cout, cin, 2, 2, 2, device=dev) * 0.05
    b = torch.zeros(cout, device=dev)
    wf = ops.pack_bfrag(w, torch.bfloat16, 8, cin, cout, cout, 1, 8, 0, cin * 8)
    for _ in range(3):
        y = ops.conv_gemm(x, wf, b, CONV_K2S2, cout, cout)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        y = ops.conv_gemm(x, wf, b, CONV_K2S2, cout, cout)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"k2s2 {cin}->{cout} @ {S}^3 -> {S // 2}^3 x4: {us:.1f} us  {(x.numel() + y.numel()) * 2 / us / 1e3:.0f} GB/s   checksum {float(y.float().sum()):.4f}")
