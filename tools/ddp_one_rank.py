"""The bench step through the data-parallel exchange path with a ONE-rank RCCL process group (TrainConfig.ddp_force), for profiling:
    rocprofv3 --kernel-trace --stats -d out -o run --output-format csv -- python3 tools/ddp_one_rank.py [steps] [force 0|1]"""
import os
import sys
import tempfile
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dycon_paper_replication_amd.synthetic import make_batch  # noqa: E402
from dycon_paper_replication_amd.trainer import DyconTrainer, TrainConfig  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
force = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
torch.cuda.set_device(0)
d = tempfile.mkdtemp()
dist.init_process_group("nccl", init_method="file://" + os.path.join(d, "init"), rank=0, world_size=1)
dev = torch.device("cuda:0")
vol, lab, _ = make_batch(1337, 4, (96, 96, 96))
vol, lab = vol.to(dev), lab.to(torch.uint8).to(dev)
tr = DyconTrainer(TrainConfig(model="vnet", batch_size=4, labeled_bs=2, dtype=torch.bfloat16, seed=1337, ddp_force=force), dev,
                  process_group=dist.group.WORLD if force else None)
for _ in range(6):
    tr.step(vol, lab)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.step(vol, lab)
torch.cuda.synchronize()
print(f"force={force}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step")
dist.destroy_process_group()
