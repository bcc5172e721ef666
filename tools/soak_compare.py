"""Long-run consistency: N steps eager (replay off) vs N steps with replay, same seeds and batches; also a plain long soak.
usage: soak_compare.py [steps]"""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from dycon_paper_replication_amd.synthetic import make_batch
from dycon_paper_replication_amd.trainer import DyconTrainer, TrainConfig
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
batches = [tuple(t.to(dev) for t in make_batch(s, 4, (96, 96, 96))[:2]) for s in range(3)]
res = {}
for replay in (False, True):
    tr = DyconTrainer(TrainConfig(model="vnet", batch_size=4, labeled_bs=2, dtype=torch.bfloat16, replay=replay), dev)
    losses = []
    for i in range(steps):
        vol, lab = batches[i % 3]
        out = tr.step(vol, lab)
        losses.append(float(out["loss"]))
    torch.cuda.synchronize()
    res[replay] = (losses, tr.flat_p.clone(), tr.skipped_steps)
    print(f"replay={replay}: final loss {losses[-1]:.5f}, skipped {tr.skipped_steps}", flush=True)
la, lb = res[False][0], res[True][0]
md = max(abs(a - b) for a, b in zip(la, lb))
pd = float((res[False][1] - res[True][1]).abs().max())
print(f"max |loss eager - loss replay| over {steps} steps: {md:.3e}; max |param diff|: {pd:.3e}")
