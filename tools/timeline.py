#!/usr/bin/env python3
"""Coarse GPU timeline of the replayed bench step from HIP events recorded INSIDE the recorded launch list (no profiler, no extra
host work): when do the student forward, the teacher forward, the loss, the backward's dependent chain, the weight-gradient
stream and the feature stream finish, relative to the step's first launch?      python tools/timeline.py [steps] [D H W feature_scaler]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dycon_paper_replication_amd.synthetic import make_batch  # noqa: E402
from dycon_paper_replication_amd.trainer import DyconTrainer, TrainConfig  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
patch = tuple(int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (96, 96, 96)
fs = int(sys.argv[5]) if len(sys.argv) > 5 else 2
vol, lab, _ = make_batch(1337, 4, patch)
vol, lab = vol.to(dev), lab.to(torch.uint8).to(dev)
tr = DyconTrainer(TrainConfig(model="vnet", batch_size=4, labeled_bs=2, dtype=torch.bfloat16, seed=1337, feature_scaler=fs), dev)
tr.marks = {}
tr.s_eng.mark = tr._mark
for _ in range(6):
    tr.step(vol, lab)
acc = {}
for _ in range(steps):
    tr.step(vol, lab)
    torch.cuda.synchronize()
    t0 = tr.marks["step_begin"]
    for k, ev in tr.marks.items():
        acc[k] = acc.get(k, 0.0) + t0.elapsed_time(ev)
for k, v in sorted(acc.items(), key=lambda kv: kv[1]):
    print(f"{k:18s} {v / steps:7.3f} ms")
