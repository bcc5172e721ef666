#!/usr/bin/env python3
"""Is the step bound by the rate at which the GPU's command processor takes dispatches?  Adds N trivial one-wave launches per step on an
otherwise idle fifth stream (no dependencies on anything) and times the bench configuration; each case in a FRESH process (the
allocator state of one process changes the step time by up to 20 %).   python tools/launch_rate.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, torch
sys.path.insert(0, %r)
from dycon_paper_replication_amd.synthetic import make_batch
from dycon_paper_replication_amd.trainer import DyconTrainer, TrainConfig
n = int(sys.argv[1])
dev = torch.device("cuda:0")
vol, lab, _ = make_batch(1337, 4, (96, 96, 96))
vol, lab = vol.to(dev), lab.to(torch.uint8).to(dev)
tr = DyconTrainer(TrainConfig(model="vnet", batch_size=4, labeled_bs=2, dtype=torch.bfloat16, seed=1337), dev)
for _ in range(10): tr.step(vol, lab)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): tr.step(vol, lab)
torch.cuda.synchronize()
print("%%d extra launches per step: %%.3f ms/step" %% (n, (time.perf_counter() - t0) * 10))
''' % ROOT
for n in (0, 50, 100, 200, 0):
    env = dict(os.environ, DYCON_ABLATE="extra_launches" if n else "", DYCON_ABLATE_N=str(n))      # read once at import (engine.ABLATE)
    subprocess.run([sys.executable, "-c", CHILD, str(n)], check=True, env=env)
