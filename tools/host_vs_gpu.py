"""How much of a DyCON step is host enqueue time?  Times step() without the per-step flag read (host returns as soon as
the launches are enqueued) against the synchronised step time.  usage: host_vs_gpu.py [model] [steps]"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from dycon_paper_replication_amd.synthetic import make_batch
from dycon_paper_replication_amd.trainer import DyconTrainer, TrainConfig

model = sys.argv[1] if len(sys.argv) > 1 else "vnet"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
for strict in (True, False):
    cfg = TrainConfig(model=model, batch_size=4, labeled_bs=2, dtype=torch.bfloat16, strict_nan_check=strict)
    tr = DyconTrainer(cfg, dev)
    vol, lab, _ = make_batch(1337, 4, (96, 96, 96))
    vol, lab = vol.to(dev), lab.to(torch.uint8).to(dev)
    for _ in range(5):
        tr.step(vol, lab)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(vol, lab)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{model} strict_nan_check={strict}: host enqueue {1e3 * (t1 - t0) / steps:.2f} ms/step, "
          f"synchronised {1e3 * (t2 - t0) / steps:.2f} ms/step", flush=True)
