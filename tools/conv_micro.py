"""Micro-benchmark of single conv launches (for rocprofv3 --pmc runs).  usage: conv_micro.py Cin Cout S [reps]"""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from dycon_paper_replication_amd import ops
from dycon_paper_replication_amd._lib import CONV_K3

cin, cout, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = "cuda:0"
x = torch.randn(4, S, S, S, cin, device=dev).bfloat16()
w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
b = torch.zeros(cout, device=dev)
wf = ops.pack_bfrag(w, torch.bfloat16, 27, cin, cout, cout, 1, 27, 0, cin * 27)
for _ in range(2):
    y = ops.conv_gemm(x, wf, b, CONV_K3, cout, cout)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    y = ops.conv_gemm(x, wf, b, CONV_K3, cout, cout)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
fl = 2 * 4 * S ** 3 * 27 * cin * cout
print(f"conv k3 {cin}->{cout} @ {S}^3 x4: {us:.1f} us  {fl / us / 1e6:.1f} TFLOP/s  {(x.numel() + y.numel()) * 2 / us / 1e3:.1f} GB/s")
