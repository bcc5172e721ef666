"""Soak: N steps of the bench configuration; memory must plateau, loss stay finite, no step skipped.  usage: soak.py [steps]"""
import sys
import time
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from dycon_paper_replication_amd.synthetic import make_batch
from dycon_paper_replication_amd.trainer import DyconTrainer, TrainConfig
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
tr = DyconTrainer(TrainConfig(model="vnet", batch_size=4, labeled_bs=2, dtype=torch.bfloat16), dev)
batches = [tuple(t.to(dev) for t in make_batch(s, 4, (96, 96, 96))[:2]) for s in range(4)]
t0 = time.perf_counter()
for i in range(steps):
    vol, lab = batches[i % 4]
    out = tr.step(vol, lab.to(torch.uint8))
    if i % 50 == 49 or i == 9:
        torch.cuda.synchronize()
        print(f"step {i + 1}: loss {float(out['loss']):.4f} (ce {float(out['ce']):.4f} dice {float(out['dice']):.4f} fecl {float(out['fecl']):.4f} uncl {float(out['uncl']):.4f}) "
              f"alloc {torch.cuda.memory_allocated() / 2**20:.0f} MiB peak {torch.cuda.max_memory_allocated() / 2**20:.0f} MiB reserved {torch.cuda.memory_reserved() / 2**20:.0f} MiB "
              f"{(time.perf_counter() - t0) / (i + 1) * 1e3:.2f} ms/step skipped {tr.skipped_steps}", flush=True)
