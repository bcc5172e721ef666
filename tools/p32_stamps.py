"""Diagnostic build only (tools/build_variant.sh stamp -DP32_STAMP [-DP32_DEPTH=n]; DYCON_LIB=build_variants/lib_stamp.so):
where workgroup 0 / wave 0 of conv_k3_p32_kernel spends its cycles (s_memtime segments summed over its tiles).  Shares, not times."""
import ctypes as C
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from dycon_paper_replication_amd import _lib, ops
from dycon_paper_replication_amd._lib import CONV_K3

S = int(sys.argv[1]) if len(sys.argv) > 1 else 48
x = torch.randn(4, S, S, S, 32, device="cuda:0").bfloat16()
w = torch.randn(32, 32, 3, 3, 3, device="cuda:0") * 0.05
b = torch.zeros(32, device="cuda:0")
wf = ops.pack_bfrag(w, torch.bfloat16, 27, 32, 32, 32, 1, 27, 0, 32 * 27)
for _ in range(20):
    y = ops.conv_gemm(x, wf, b, CONV_K3, 32, 32)
torch.cuda.synchronize()
out = (C.c_ulonglong * 8)()
rc = _lib.load().dycon_debug_p32_stamps(out)
assert rc == 0, rc
v = list(out)
names = ["27 taps (MFMA + fragment reads)", "write the next halo image", "geometry + request the halo after that", "output stores",
         "barrier", "-", "prologue (weights -> LDS, first halo)", "all tiles"]
tot = sum(v[:6])
for i, (n, c) in enumerate(zip(names, v)):
    print(f"{n:55s} {c:9d} cycles" + (f"  {100.0 * c / max(tot, 1):5.1f} % of the tile loop" if i < 6 else ""))
