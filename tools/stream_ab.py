"""A/B of the stream assignment: 4 streams (main, teacher, features, wgrad) vs merged variants.  usage: stream_ab.py [steps]"""
import sys
import time
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from dycon_paper_replication_amd.synthetic import make_batch
from dycon_paper_replication_amd.trainer import DyconTrainer, TrainConfig
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
vol, lab, _ = make_batch(1337, 4, (96, 96, 96))
vol, lab = vol.to(dev), lab.to(torch.uint8).to(dev)
for variant in ("4 streams", "3 streams (features + wgrad merged)", "2 streams (teacher + features + wgrad merged)", "4 streams"):
    tr = DyconTrainer(TrainConfig(model="vnet", batch_size=4, labeled_bs=2, dtype=torch.bfloat16), dev)
    if variant.startswith("3"):
        tr.feat = tr.s_eng.feat_stream = tr.s_eng.wgrad_stream
    elif variant.startswith("2"):
        tr.feat = tr.s_eng.feat_stream = tr.s_eng.wgrad_stream = tr.side
    for _ in range(6):
        tr.step(vol, lab)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(vol, lab)
    torch.cuda.synchronize()
    print(f"{variant}: {1e3 * (time.perf_counter() - t0) / steps:.2f} ms/step", flush=True)
