#!/usr/bin/env python3
"""Where does the step time go?  Times the bench configuration (V-Net bf16, B = 4, 96^3) with parts of the step switched OFF
(numerically meaningless runs -- timing only): the bound each part puts on the step, contention with the dependent chain included.
    python tools/ablate.py [steps]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dycon_paper_replication_amd import engine, ops  # noqa: E402
from dycon_paper_replication_amd.synthetic import make_batch  # noqa: E402
from dycon_paper_replication_amd.trainer import DyconTrainer, TrainConfig  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
vol, lab, _ = make_batch(1337, 4, (96, 96, 96))
vol, lab = vol.to(dev), lab.to(torch.uint8).to(dev)


def run(tag, **kw):
    engine.ABLATE.clear()
    engine.ABLATE.update(kw.pop("ablate", ()))
    cfg = TrainConfig(model="vnet", batch_size=4, labeled_bs=2, dtype=torch.bfloat16, seed=1337, **kw)
    tr = DyconTrainer(cfg, dev)
    for _ in range(8):
        tr.step(vol, lab)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = 0.0
    for _ in range(steps):
        h0 = time.perf_counter()
        tr.step(vol, lab)
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"{tag:58s} {ms:7.3f} ms/step   (host in step() {host / steps * 1e3:6.3f} ms)", flush=True)
    del tr
    torch.cuda.empty_cache()


run("baseline (replay)")
run("no per-step NaN flag read", strict_nan_check=False)
run("eager (no replay)", replay=False)
run("no weight gradients (wgrad + reduce launches skipped)", ablate=("wgrad",))
run("wgrad: stream fork events only, no kernels", ablate=("wgrad_events_only",))
run("wgrad: kernels without the fork events (race; timing only)", ablate=("wgrad_no_events",))
run("baseline again")
run("no teacher forward (teacher outputs := student's)", ablate=("teacher",))
run("no wgrad, no teacher", ablate=("wgrad", "teacher"))

run("no norm backward (gz := gy)", ablate=("norm_bwd",))
run("no data-gradient convs (gx := 0-size alias)", ablate=("dgrad",))
run("single stream (all overlaps off)", overlap_teacher=False, overlap_wgrad=False, overlap_features=False)
