"""Sliding-window evaluation throughput: one BraTS-sized volume (240 x 240 x 155, z-scored noise), the reference's test
settings (patch 96 x 96 x 64, stride_xy 16, stride_z 4: code/test_BraTS19.py) through the bf16 V-Net.  usage: eval_bench.py [batch]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from dycon_paper_replication_amd.networks.net_factory_3d import net_factory_3d
from dycon_paper_replication_amd.utils import test_3d_patch as T3

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
shape, patch = (240, 240, 155), (96, 96, 64)
img = np.random.default_rng(0).standard_normal(shape).astype(np.float32)
model = net_factory_3d("vnet", 1, 2, 2, dtype=torch.bfloat16).cuda()
nwin = len(T3._windows(shape, patch, 16, 4))
T3.test_single_case(model, img[:96, :96, :64], 16, 4, patch, num_classes=2, batch_size=batch)   # warm-up (weight packs, allocator)
torch.cuda.synchronize()
t0 = time.perf_counter()
lab, score = T3.test_single_case(model, img, 16, 4, patch, num_classes=2, batch_size=batch)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"sliding-window eval, vnet bf16, {shape} volume, {nwin} windows of {patch}, batch {batch}: {dt:.2f} s/volume, {nwin / dt:.0f} windows/s")
