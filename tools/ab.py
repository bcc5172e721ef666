#!/usr/bin/env python3
"""A/B of TrainConfig settings on the bench configuration, every case in a FRESH process (the allocator state of a process moves the
step time by up to 20 %), repeated:   python tools/ab.py fuse_finish=False fuse_finish=True [--reps 2] [--steps 100]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, torch
sys.path.insert(0, %r)
from dycon_paper_replication_amd.synthetic import make_batch
from dycon_paper_replication_amd.trainer import DyconTrainer, TrainConfig
kw = {}
for a in sys.argv[2].split(","):
    if a:
        k, v = a.split("="); kw[k] = eval(v)
steps = int(sys.argv[1])
dev = torch.device("cuda:0")
vol, lab, _ = make_batch(1337, 4, (96, 96, 96))
vol, lab = vol.to(dev), lab.to(torch.uint8).to(dev)
tr = DyconTrainer(TrainConfig(model="vnet", batch_size=4, labeled_bs=2, dtype=torch.bfloat16, seed=1337, **kw), dev)
for _ in range(10): out = tr.step(vol, lab)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): out = tr.step(vol, lab)
torch.cuda.synchronize()
print("%%-40s %%.3f ms/step   loss %%.6f" %% (sys.argv[2], (time.perf_counter() - t0) / steps * 1e3, float(out["loss"])))
''' % ROOT
args = [a for a in sys.argv[1:] if not a.startswith("--")]
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 2
steps = sys.argv[sys.argv.index("--steps") + 1] if "--steps" in sys.argv else "100"
args = [a for a in args if a not in (str(reps), steps)] or [""]
for _ in range(reps):
    for a in args:
        subprocess.run([sys.executable, "-c", CHILD, steps, a], check=True)
