"""A/B: teacher forward on the side stream (overlap_teacher) vs in line.  usage: overlap_ab.py [steps]"""
import sys
import time
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from dycon_paper_replication_amd.synthetic import make_batch
from dycon_paper_replication_amd.trainer import DyconTrainer, TrainConfig
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
vol, lab, _ = make_batch(1337, 4, (96, 96, 96))
vol, lab = vol.to(dev), lab.to(torch.uint8).to(dev)
for ov in (True, False, True, False):
    tr = DyconTrainer(TrainConfig(model="vnet", batch_size=4, labeled_bs=2, dtype=torch.bfloat16, overlap_teacher=ov), dev)
    for _ in range(5):
        tr.step(vol, lab)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(vol, lab)
    torch.cuda.synchronize()
    print(f"overlap_teacher={ov}: {1e3 * (time.perf_counter() - t0) / steps:.2f} ms/step", flush=True)
