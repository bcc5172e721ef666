"""FeCL forward + backward at a given size (default: ISLES, N = 15680, B = 2), bf16.  usage: fecl_micro.py [N] [B] [reps]"""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from dycon_paper_replication_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 15680
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
f = torch.nn.functional.normalize(torch.randn(B, N, 256, device=dev, generator=g), dim=-1).bfloat16()
t = torch.nn.functional.normalize(torch.randn(B, N, 256, device=dev, generator=g), dim=-1).bfloat16()
mask = (torch.rand(B, N, device=dev, generator=g) > 0.9).float()
coef = torch.ones(1, device=dev)
args = (f, t, mask, None, 0.6, 2.0, True, 0.4)


def run():
    loss, st = ops.fecl_fwd(*args, 1.0)
    return ops.fecl_bwd(*args, 1.0, st, coef), loss


run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    gf, loss = run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
fl = 8 * 2.0 * B * N * N * 256      # 4 self Grams + 2 cross Grams + 2 gradient GEMMs
print(f"FeCL fwd+bwd N={N} B={B}: {ms:.2f} ms, {fl / ms / 1e9:.1f} TFLOP/s executed ({fl / 1e12:.2f} TFLOP), loss {float(loss):.4f}")
