#!/usr/bin/env python3
"""DyCON training-step benchmark on MI355X (contract: see the task description / DESIGN.md section 4).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One "step" = one full DyCON iteration (teacher noise, student fwd, teacher fwd, CE + Dice + consistency +
UnCL + FeCL, backward, [RCCL gradient all-reduce], clip + SGD + EMA, weight repack) on one synthetic batch
that is already resident in HBM.  Workload = BASELINE.json configs[1]: BraTS2019 geometry, V-Net (GroupNorm),
bf16 activation storage / fp32 accumulate, per-GPU batch 4 (2 labelled + 2 unlabelled), 96^3 patches.
metric = training volumes / second over all ranks (weak scaling: per-GPU batch fixed).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_TFLOPS = {"conv_gemm": 2500.0, "conv_k3_lds": 2500.0, "conv_k3_p16": 2500.0, "conv_k3_c1": 2500.0, "conv_k3_tile": 2500.0,
                    "conv_gemm_splitk": 2500.0, "wgrad_k3_bf16": 2500.0, "conv_wgrad": 157.3}   # dense bf16 MFMA / f32-input MFMA (kernel's arithmetic type)


def cpu_baseline(model, patch, seed):
    """The oracle (plain-PyTorch fp32 CPU restatement of the reference step, oracle/step.py) timed on this box's host cores,
    BASELINE.md section 3 protocol: B = 2 (1+1) and B = 4 (2+2), 1 warm-up + 3 timed steps each, median.  `value` is the B = 4
    figure (the benchmark's own batch); the B = 2 figure is quoted in `sample`.  tools/oracle_vs_reference.py (container only)
    shows the oracle's step time is within 10 % of the imported reference's."""
    import statistics
    from oracle import nets as ON
    from oracle import step as OS
    from dycon_paper_replication_amd.synthetic import make_batch
    # the box's CPU share (cgroup / affinity), not the host's logical core count: oversubscribing torch's
    # intra-op pool makes the CPU path several times slower
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:   # cgroup v2 CPU quota, e.g. "1600000 100000" -> 16 CPUs
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = min(cores, max(1, int(q) // int(per)))
    except (OSError, ValueError):
        pass
    cores = max(1, min(cores, 32))
    torch.set_num_threads(cores)
    mk = ON.make_vnet_params if model == "vnet" else ON.make_unet_params
    res = {}
    for B in (2, 4):
        cfg = OS.StepConfig(net_type=model, labeled_bs=B // 2, feature_scaler=2)
        st = OS.StepState(student=mk(1), teacher=mk(2))
        vol, lab, noise = make_batch(seed, B, patch)
        OS.train_step(cfg, st, vol, lab, noise, 5.0, 0)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            OS.train_step(cfg, st, vol, lab, noise, 5.0, 0)
            ts.append(time.perf_counter() - t0)
        res[B] = statistics.median(ts)
    return {"value": 4.0 / res[4], "unit": "volumes/s", "cores": cores, "kind": "port",
            "sample": f"oracle/step.py (torch {torch.__version__} CPU fp32), {model} at {'x'.join(map(str, patch))}: B=4 (2+2) "
                      f"{res[4]:.2f} s/step = {4.0 / res[4]:.2f} vol/s; B=2 (1+1) {res[2]:.2f} s/step = {2.0 / res[2]:.2f} vol/s; "
                      f"1 warm-up + 3 timed steps each, median"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)      # warm-up 10, time >= 50 steps (60 steps = 0.4 s on the GPU)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="vnet", choices=["vnet", "unet_3D"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--labeled", type=int, default=2)
    ap.add_argument("--patch", type=int, nargs=3, default=[96, 96, 96])
    ap.add_argument("--feature-scaler", type=int, default=2, help="2: BraTS / Pancreas (N = 1728 at 96^3); 4: ISLES (N = 15680 at 112x112x80)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--strict", action="store_true", help="exit non-zero when the per-kernel table fails its self-check")
    ap.add_argument("--cfg", default="", help="diagnostics: TrainConfig overrides, e.g. overlap_wgrad=False,overlap_teacher=False")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", local % max(ndev, 1))
    torch.cuda.set_device(dev)
    pg = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" == RCCL on ROCm (one rank per GPU over xGMI).  DYCON_DIST_BACKEND=gloo is only for rehearsing the
        # multi-rank code path on a single-GPU box (ranks then share the device; RCCL refuses duplicate GPUs).
        backend = os.environ.get("DYCON_DIST_BACKEND", "nccl")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(backend)
        pg = torch.distributed.group.WORLD

    from dycon_paper_replication_amd import ops
    from dycon_paper_replication_amd.synthetic import make_batch
    from dycon_paper_replication_amd.trainer import DyconTrainer, TrainConfig

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    extra = {k: eval(v) for k, v in (kv.split("=") for kv in args.cfg.split(",") if kv)}      # diagnostics only (tools/)
    cfg = TrainConfig(model=args.model, batch_size=args.batch, labeled_bs=args.labeled, dtype=dtype, seed=1337,
                      feature_scaler=args.feature_scaler, **extra)
    tr = DyconTrainer(cfg, dev, process_group=pg)
    patch = tuple(args.patch)
    vol, lab, _ = make_batch(1337 + rank, args.batch, patch)
    vol, lab = vol.to(dev), lab.to(torch.uint8).to(dev)        # inputs resident in HBM before the timed region

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def note(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    note("warm-up")
    for _ in range(args.warmup):
        tr.step(vol, lab)
    barrier()
    note("timed region")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = tr.step(vol, lab)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    ms_per_step = dt / args.steps * 1e3
    value = args.batch * world * args.steps / dt
    note(f"{ms_per_step:.2f} ms/step, {value:.2f} volumes/s")

    # ---- per-kernel roofline (HIP events on the launch stream, a few extra un-timed steps)
    roofline = None
    nprof = 3
    if not args.no_kernel_timing:
        # EVERY rank runs these extra steps (a step contains collectives when world > 1); only rank 0 brackets its launches
        if rank == 0:
            ops.PROFILER = ops.KernelProfiler()
        tp0 = time.perf_counter()
        for _ in range(nprof):
            tr.step(vol, lab)
        barrier()
        prof_ms_per_step = (time.perf_counter() - tp0) / nprof * 1e3
    if not args.no_kernel_timing and rank == 0:
        summ = ops.PROFILER.summary()
        ops.PROFILER = None
        # The roofline object describes ONE kernel: pick the dominant region among the entry points that are a single launch
        # (their event time is the kernel's duration, comparable with the rocprofv3 average); norm / wgrad / split-K entry points
        # enqueue a finalize or reduce launch as well and are listed, per call, in per_kernel_*.
        single = [k for k in summ if k in ("conv_k3_lds", "conv_k3_p16", "conv_k3_c1", "conv_gemm", "conv_direct")]
        dom = max(single or summ, key=lambda k: summ[k]["ms"])
        r = summ[dom]
        sec = r["ms"] * 1e-3
        gbs = r["bytes"] / sec / 1e9
        tfl = r["flops"] / sec / 1e12
        peak_t = MFMA_PEAK_TFLOPS.get(dom, 157.3) if args.dtype == "bf16" else 157.3
        f_hbm, f_mfma = gbs / HBM_PEAK_GBS, tfl / peak_t
        bound = "hbm" if (r["bytes"] / (HBM_PEAK_GBS * 1e9)) >= (r["flops"] / (peak_t * 1e12)) else "mfma"
        traffic = None     # HBM bytes per launch from the PMC passes committed under profiles/ (separate rocprofv3 --pmc runs)
        try:
            pmc_path = os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")
            if not os.path.exists(pmc_path):
                pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
            pmc = json.load(open(pmc_path))
            traffic = pmc["per_region"][dom]["hbm_bytes_per_launch"]     # same unit as `achieved`: per launch of this kernel
        except (OSError, KeyError, ValueError):
            pass
        # self-check of the table: a bracket that does not enclose its kernel shows up as a fraction above 1, and the brackets of
        # one stream cannot add up to more than the (eager, profiled) step they were taken in
        main_stream = torch.cuda.current_stream().cuda_stream
        main_ms = sum(v["ms_by_stream"].get(main_stream, 0.0) for v in summ.values()) / nprof
        fr = {k: (v["bytes"] / (v["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                  v["flops"] / (v["ms"] * 1e-3) / 1e12 / (MFMA_PEAK_TFLOPS.get(k, 157.3) if args.dtype == "bf16" else 157.3))
              for k, v in summ.items() if v["ms"] > 0}
        over = sorted(k for k, (h, m) in fr.items() if h > 1.0 or m > 1.0)
        self_check = {"ok": not over and main_ms <= prof_ms_per_step * 1.02, "fractions_above_1": over,
                      "main_stream_bracket_ms_per_step": round(main_ms, 4), "profiled_step_ms": round(prof_ms_per_step, 4)}
        if not self_check["ok"]:
            note(f"SELF-CHECK FAILED: {self_check}")
            if args.strict:
                raise SystemExit(f"bench self-check failed: {self_check}")
        roofline = {"kernel": dom, "bound": bound, "self_check": self_check,
                    "achieved": gbs if bound == "hbm" else tfl, "peak": HBM_PEAK_GBS if bound == "hbm" else peak_t,
                    "unit": "GB/s" if bound == "hbm" else "TFLOP/s", "frac": f_hbm if bound == "hbm" else f_mfma,
                    "traffic": traffic, "algorithmic_bytes_per_launch": r["bytes"] / r["launches"],
                    "algorithmic_flops_per_launch": r["flops"] / r["launches"],
                    "avg_launch_ms": r["ms"] / r["launches"], "launches_per_step": r["launches"] // nprof,
                    "hbm_frac": f_hbm, "mfma_frac": f_mfma,
                    "per_kernel_ms_per_step": {k: round(v["ms"] / nprof, 4) for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])},
                    "per_kernel_algorithmic_bytes_per_step": {k: v["bytes"] / nprof for k, v in summ.items()},
                    "per_kernel_calls_per_step": {k: v["launches"] // nprof for k, v in summ.items()},
                    # every kernel family against BOTH ceilings (algorithmic bytes / flops over the summed launch durations)
                    "per_kernel_frac": {k: {"hbm": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                            "mfma": round(v["flops"] / (v["ms"] * 1e-3) / 1e12 /
                                                          (MFMA_PEAK_TFLOPS.get(k, 157.3) if args.dtype == "bf16" else 157.3), 4)}
                                        for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]) if v["ms"] > 0}}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        note("cpu baseline (oracle on host cores)")
        cpu = cpu_baseline(args.model, patch, 1337)
        note("done")

    if rank == 0:
        line = {"metric": "train volumes/sec (96^3 patch)", "value": value, "unit": "volumes/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": args.dtype, "data": "synthetic",
                "config": {"workload": f"BraTS2019 labelnum=25 geometry, {args.model} (GroupNorm) {args.dtype}, per-GPU batch "
                                       f"{args.batch} ({args.labeled} lab + {args.batch - args.labeled} unlab), "
                                       f"{'x'.join(map(str, patch))} patches, full DyCON step",
                           "global_batch": args.batch * world, "parallelism": f"dp{world}",
                           "final_loss": float(out["loss"]), "skipped_steps": tr.skipped_steps},
                "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
