"""Loss trajectories of the bench configuration with and without the head fused into block_nine's normalisation (TrainConfig.fuse_head):
step 0 must agree exactly (same logits), later steps drift at the rate bf16 training amplifies an fp32-round-off difference in one gradient."""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from dycon_paper_replication_amd.synthetic import make_batch
from dycon_paper_replication_amd.trainer import DyconTrainer, TrainConfig
dev = torch.device("cuda:0")
vol, lab, _ = make_batch(1337, 4, (96, 96, 96))
vol, lab = vol.to(dev), lab.to(torch.uint8).to(dev)
res = {}
for fh in (False, True):
    tr = DyconTrainer(TrainConfig(model="vnet", batch_size=4, labeled_bs=2, dtype=torch.bfloat16, seed=1337, fuse_head=fh), dev)
    ls = []
    for i in range(40):
        out = tr.step(vol, lab)
        ls.append(float(out["loss"]))
    res[fh] = ls
for i in (0, 1, 2, 3, 5, 9, 19, 29, 39):
    print(i, "%.6f %.6f  diff %.2e" % (res[False][i], res[True][i], abs(res[False][i] - res[True][i])))
