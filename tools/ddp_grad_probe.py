#!/usr/bin/env python3
"""Which loss term disagrees under data parallelism?  2 ranks (gloo, sharing the GPU) run ONE step with a subset of the loss terms
switched on; the averaged gradient is compared, parameter group by parameter group, with oracle.step.ddp_train_step (double)."""
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES_ALL = {"ce+dice": dict(l_weight=1.0, u_weight=0.0, consistency=0.0),
         "cons only": dict(l_weight=0.0, u_weight=0.0, consistency=100.0),
         "uncl+fecl": dict(l_weight=0.0, u_weight=0.5, consistency=0.0),
         "uncl+fecl(no teacher loss)": dict(l_weight=0.0, u_weight=0.5, consistency=0.0, use_teacher_loss=0),
         "all": dict(l_weight=1.0, u_weight=0.5, consistency=0.1)}


CASES = {k: CASES_ALL[k] for k in ("ce+dice", "cons only")}


def worker(rank, init_file, out_file, case):
    import torch.distributed as dist
    from dycon_paper_replication_amd.engine import DropoutSpec
    from dycon_paper_replication_amd.synthetic import make_batch
    from dycon_paper_replication_amd.trainer import DyconTrainer, TrainConfig
    from oracle import nets as ON
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=2)
    torch.cuda.set_device(0)
    vol, lab, noise = make_batch(9, 4, (32, 32, 32))
    idx = [rank, 2 + rank]
    cfg = TrainConfig(model="vnet", labeled_bs=1, batch_size=2, dtype=torch.float32, seed=5, base_lr=0.01, **CASES[case])
    tr = DyconTrainer(cfg, "cuda:0", process_group=dist.group.WORLD, student_init=ON.make_vnet_params(5), teacher_init=ON.make_vnet_params(6))
    off = DropoutSpec("off")
    if os.environ.get("PROBE_LOCAL"):      # no gradient all-reduce: the arena keeps this rank's LOCAL gradient
        tr.s_eng.on_param_grads = None
        tr.buckets = []
    tr.step(vol[idx].cuda(), lab[idx].cuda(), noise=noise[idx].cuda(), s_drop=off, t_drop=off, epoch=300, beta=2.5)
    torch.cuda.synchronize()
    torch.save({k: (tr.g[k] / (1 if os.environ.get("PROBE_LOCAL") else 2)).cpu() for k in tr.names}, out_file + str(rank))
    dist.barrier()
    dist.destroy_process_group()


def main():
    import torch.multiprocessing as mp
    from dycon_paper_replication_amd.synthetic import make_batch
    from oracle import nets as ON
    from oracle import step as OS
    dbl = lambda p: {k: (v.double() if v.is_floating_point() else v) for k, v in p.items()}  # noqa: E731
    vol, lab, noise = make_batch(9, 4, (32, 32, 32))
    shards = [(vol[[r, 2 + r]].double(), lab[[r, 2 + r]], noise[[r, 2 + r]].double()) for r in range(2)]
    for case, kw in CASES.items():
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(worker, args=(os.path.join(d, "init"), os.path.join(d, "out.pt"), case), nprocs=2, join=True)
            gots = [torch.load(os.path.join(d, "out.pt" + str(r))) for r in range(2)]
        states = [OS.StepState(student=dbl(ON.make_vnet_params(5)), teacher=dbl(ON.make_vnet_params(6))) for _ in range(2)]
        okw = dict(l_weight=kw["l_weight"], u_weight=kw["u_weight"], consistency=kw["consistency"],
                   use_teacher_loss=bool(kw.get("use_teacher_loss", 1)))
        ref = OS.ddp_train_step(OS.StepConfig(net_type="vnet", labeled_bs=1, base_lr=0.02, **okw), states, shards, 2.5, 300)
        for rk in range(2):
            groups = {}
            for k, g64 in (ref["local_grads"][rk] if os.environ.get("PROBE_LOCAL") else ref["grads"]).items():
                grp = "head" if k.startswith("projection") else ("decoder" if any(s in k for s in ("six", "seven", "eight", "nine", "out_conv", "five_up")) else "encoder")
                nrm = float(g64.norm())
                if nrm < 1e-9 * float(ref["grad_norm"]) or nrm == 0:
                    continue
                e = float((gots[rk][k].double() - g64).norm()) / nrm
                groups[grp] = max(groups.get(grp, 0.0), e)
            print(f"{case:30s} rank {rk} |g| {float(ref['grad_norm']):.4e}  worst relative L2 error per group: " +
                  "  ".join(f"{g} {v:.2e}" for g, v in sorted(groups.items())), flush=True)


if __name__ == "__main__":
    main()
