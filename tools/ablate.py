#!/usr/bin/env python3
"""Where does the step time go?  Times the bench configuration (V-Net bf16, B = 4, 96^3) with parts of the step switched OFF
(numerically meaningless runs -- timing only): the bound each part puts on the step, contention with the dependent chain included.
Every case runs in a FRESH process: the switches are read once at import (DYCON_ABLATE, engine.ABLATE is immutable), and the
allocator state of a process moves the step time by more than most of these effects.
    python tools/ablate.py [steps]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, torch
sys.path.insert(0, %r)
from dycon_paper_replication_amd.synthetic import make_batch
from dycon_paper_replication_amd.trainer import DyconTrainer, TrainConfig
tag, steps, kw = sys.argv[1], int(sys.argv[2]), eval(sys.argv[3])
dev = torch.device("cuda:0")
vol, lab, _ = make_batch(1337, 4, (96, 96, 96))
vol, lab = vol.to(dev), lab.to(torch.uint8).to(dev)
tr = DyconTrainer(TrainConfig(model="vnet", batch_size=4, labeled_bs=2, dtype=torch.bfloat16, seed=1337, **kw), dev)
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
print(f"{tag:62s} {ms:7.3f} ms/step   (host in step() {host / steps * 1e3:6.3f} ms)", flush=True)
''' % ROOT

steps = sys.argv[1] if len(sys.argv) > 1 else "60"


def run(tag, ablate=(), env=None, **kw):
    e = dict(os.environ, DYCON_ABLATE=",".join(ablate), **(env or {}))
    subprocess.run([sys.executable, "-c", CHILD, tag, steps, repr(kw)], check=True, env=e)


if __name__ == "__main__":
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
    for mode in ("default", "nofence", "device"):
        run(f"fork / join events created with DYCON_EVENT_FLAGS={mode}", env={"DYCON_EVENT_FLAGS": mode})
