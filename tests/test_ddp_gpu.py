"""GPU, 2 ranks sharing the one GPU of the test box (gloo rendezvous; RCCL refuses duplicate devices): the sharded DyCON step
(bucketed gradient all-reduce overlapped with the backward, all-reduced Dice sums, x world pre-scaling, 1/world folded into the
SGD kernel) reproduces the single-process step on the global batch.  u_weight = 0 switches FeCL/UnCL off so that the per-rank
BatchNorm statistics of the projection head (per-replica in the reference's DataParallel too) do not enter the comparison."""
import os
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cfg():
    from dycon_paper_replication_amd.trainer import TrainConfig
    return TrainConfig(model="vnet", labeled_bs=1, batch_size=2, dtype=torch.float32, seed=5, u_weight=0.0, base_lr=0.01)


def _worker(rank, init_file, out_file):
    import torch.distributed as dist
    from dycon_paper_replication_amd.engine import DropoutSpec
    from dycon_paper_replication_amd.synthetic import make_batch
    from dycon_paper_replication_amd.trainer import DyconTrainer
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=2)
    torch.cuda.set_device(0)
    vol, lab, noise = make_batch(9, 4, (32, 32, 32))          # global batch [lab0, lab1 | unl0, unl1]
    idx = [rank, 2 + rank]
    tr = DyconTrainer(_cfg(), DEV, process_group=dist.group.WORLD)
    assert len(tr.buckets) >= 2
    off = DropoutSpec("off")
    for _ in range(2):
        out = tr.step(vol[idx].to(DEV), lab[idx].to(DEV), noise=noise[idx].to(DEV), s_drop=off, t_drop=off)
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"p": tr.flat_p.cpu(), "t": tr.flat_t.cpu(), "loss": float(out["loss"]), "dice": float(out["dice"])}, out_file)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_equals_global_batch_step():
    import torch.multiprocessing as mp
    from dycon_paper_replication_amd.engine import DropoutSpec
    from dycon_paper_replication_amd.synthetic import make_batch
    from dycon_paper_replication_amd.trainer import DyconTrainer
    with tempfile.TemporaryDirectory() as d:
        init_file, out_file = os.path.join(d, "init"), os.path.join(d, "out.pt")
        mp.spawn(_worker, args=(init_file, out_file), nprocs=2, join=True)
        got = torch.load(out_file)
    cfg = _cfg()
    cfg.labeled_bs, cfg.batch_size = 2, 4
    cfg.base_lr = 0.02            # the 2-rank run scales LR x world (train_DyCON_BraTS19.py:108-110)
    tr = DyconTrainer(cfg, DEV)
    vol, lab, noise = make_batch(9, 4, (32, 32, 32))
    off = DropoutSpec("off")
    for _ in range(2):
        out = tr.step(vol.to(DEV), lab.to(DEV), noise=noise.to(DEV), s_drop=off, t_drop=off)
    assert got["loss"] == pytest.approx(float(out["loss"]), rel=1e-4)
    assert got["dice"] == pytest.approx(float(out["dice"]), rel=1e-4)
    np.testing.assert_allclose(got["p"].numpy(), tr.flat_p.cpu().numpy(), rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(got["t"].numpy(), tr.flat_t.cpu().numpy(), rtol=2e-4, atol=2e-6)
