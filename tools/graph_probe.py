"""Feasibility probe: capture one whole DyCON step (all four streams, ~490 launches) into a hipGraph through torch.cuda.CUDAGraph
and time its replay against eager enqueue.  Scalars (Philox offsets, schedules) are baked in, so replays repeat the same step:
timing only.  usage: graph_probe.py [replays]"""
import sys
import time
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from dycon_paper_replication_amd.synthetic import make_batch
from dycon_paper_replication_amd.trainer import DyconTrainer, TrainConfig
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
tr = DyconTrainer(TrainConfig(model="vnet", batch_size=4, labeled_bs=2, dtype=torch.bfloat16, strict_nan_check=False), dev)
vol, lab, _ = make_batch(1337, 4, (96, 96, 96))
vol, lab = vol.to(dev), lab.to(torch.uint8).to(dev)
for _ in range(5):
    tr.step(vol, lab)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    tr.step(vol, lab)
torch.cuda.synchronize()
print(f"eager: {1e3 * (time.perf_counter() - t0) / n:.2f} ms/step", flush=True)
g = torch.cuda.CUDAGraph()
t0 = time.perf_counter()
with torch.cuda.graph(g):
    out = tr.step(vol, lab)
torch.cuda.synchronize()
print(f"capture + instantiate: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    g.replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"graph replay: host {1e3 * (t1 - t0) / n:.3f} ms/step, synchronised {1e3 * (t2 - t0) / n:.2f} ms/step, loss {float(out['loss']):.4f}", flush=True)
