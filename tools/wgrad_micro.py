"""One k=3 weight-gradient launch (+ its reduce) at a V-Net level.  usage: wgrad_micro.py Cin Cout S [reps]   (B = 4, bf16)"""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from dycon_paper_replication_amd import ops
from dycon_paper_replication_amd._lib import CONV_K3
cin, cout, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
x = torch.randn(4, S, S, S, cin, device="cuda:0").bfloat16()
gy = torch.randn(4, S, S, S, cout, device="cuda:0").bfloat16()
gw, gb = torch.empty(cout, cin, 3, 3, 3, device="cuda:0"), torch.empty(cout, device="cuda:0")
ws = ops.conv_wgrad_workspace(x, gy, CONV_K3)
run = lambda: ops.conv_wgrad(x, gy, gw, CONV_K3, 1, 27, cin * 27, dbias=gb, ws=ws)   # noqa: E731
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
fl = 2 * 4 * S ** 3 * 27 * cin * cout
print(f"wgrad k3 {cin}->{cout} @ {S}^3 x4: {us:.1f} us (kernel + reduce)  {fl / us / 1e6:.1f} TFLOP/s  {(x.numel() + gy.numel()) * 2 / us / 1e3:.0f} GB/s algorithmic")
