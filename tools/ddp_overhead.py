"""What the data-parallel exchange path costs a rank besides wire time: the bench configuration with a ONE-rank RCCL process group and
TrainConfig.ddp_force (4 bucketed async all-reduces of the 39 MB gradient arena issued from inside the backward, the 16 + 4-double
loss exchange, the joins) against the plain single-GPU step.  Fresh process per case."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, torch
import torch.distributed as dist
sys.path.insert(0, %r)
from dycon_paper_replication_amd.synthetic import make_batch
from dycon_paper_replication_amd.trainer import DyconTrainer, TrainConfig
force = sys.argv[1] == "1"
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="file://" + sys.argv[2], rank=0, world_size=1)
dev = torch.device("cuda:0")
vol, lab, _ = make_batch(1337, 4, (96, 96, 96))
vol, lab = vol.to(dev), lab.to(torch.uint8).to(dev)
tr = DyconTrainer(TrainConfig(model="vnet", batch_size=4, labeled_bs=2, dtype=torch.bfloat16, seed=1337, ddp_force=force), dev,
                  process_group=dist.group.WORLD if force else None)
for _ in range(10): out = tr.step(vol, lab)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): out = tr.step(vol, lab)
torch.cuda.synchronize()
print("%%-28s %%.3f ms/step   loss %%.6f" %% ("exchange through RCCL (1 rank)" if force else "plain", (time.perf_counter() - t0) / 100 * 1e3, float(out["loss"])))
dist.destroy_process_group()
''' % ROOT
for _ in range(2):
    for force in ("0", "1"):
        with tempfile.TemporaryDirectory() as d:
            subprocess.run([sys.executable, "-c", CHILD, force, os.path.join(d, "init")], check=True)
