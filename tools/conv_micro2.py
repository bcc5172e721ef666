"""conv_micro for any k3 shape incl. the split-K paths.  usage: conv_micro2.py Cin Cout S [reps]   (NOSPLIT=1: no workspace)"""
import os
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from dycon_paper_replication_amd import ops
from dycon_paper_replication_amd._lib import CONV_K3, call
cin, cout, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
x = torch.randn(4, S, S, S, cin, device="cuda:0").bfloat16()
w = torch.randn(cout, cin, 3, 3, 3, device="cuda:0") * 0.05
b = torch.zeros(cout, device="cuda:0")
wf = ops.pack_bfrag(w, torch.bfloat16, 27, cin, cout, cout, 1, 27, 0, cin * 27)
y = torch.empty(4, S, S, S, cout, device="cuda:0", dtype=torch.bfloat16)
nosplit = os.environ.get("NOSPLIT") == "1"


def run():
    if nosplit:
        call("dycon_conv_gemm", x.data_ptr(), wf.data_ptr(), b.data_ptr(), y.data_ptr(), 1, CONV_K3, 0, 0, 4, S, S, S, cin, cout, cout,
             None, 0, torch.cuda.current_stream().cuda_stream)
    else:
        ops.conv_gemm(x, wf, b, CONV_K3, cout, cout)


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
print(f"conv k3 {cin}->{cout} @ {S}^3 x4 nosplit={nosplit}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us")
