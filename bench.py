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
# dense bf16 MFMA peak 2.5 PFLOP/s; kernels on the f32-input MFMA (v_mfma_f32_16x16x4_f32) or the vector unit: 157.3 TFLOP/s
F32_MFMA_KERNELS = ("conv_wgrad_kernel", "conv_direct_kernel", "head_1x1", "grad_1x1_skinny")
# algorithmic conv-I/O bytes (bf16) and FLOPs per training volume (student fwd + bwd + teacher fwd)
ALGORITHMIC_PER_VOLUME = {("vnet", (96, 96, 96)): (1.186e9, 282.9e9), ("vnet", (112, 112, 80)): (1.345e9, 320.9e9),
                          ("vnet", (112, 112, 96)): (1.614e9, 385.0e9), ("unet_3D", (96, 96, 96)): (1.657e9, 491.8e9),
                          ("unet_3D", (112, 112, 80)): (1.879e9, 557.8e9)}


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
    ap.add_argument("--repeats", type=int, default=3, help="the timed region is run this many times; the median repeat is reported")
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
    # SURVEY 8d protocol: the timed region is run REPEATS times back to back inside this invocation (each: exactly --steps steps
    # between two barriers + synchronisations, max over ranks) and the MEDIAN repeat is reported; `ms_per_step_repeats` keeps all
    note("timed region")
    repeats = []
    for _ in range(max(1, args.repeats)):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = tr.step(vol, lab)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t)
        repeats.append(dt)
    dt = sorted(repeats)[len(repeats) // 2]
    ms_per_step = dt / args.steps * 1e3
    value = args.batch * world * args.steps / dt
    note(f"{ms_per_step:.3f} ms/step (repeats: {', '.join(f'{r / args.steps * 1e3:.3f}' for r in repeats)}), {value:.2f} volumes/s")
    alg = ALGORITHMIC_PER_VOLUME.get((args.model, patch))
    step_roofline = None
    if alg is not None and args.dtype == "bf16":      # algorithmic conv I/O bytes and FLOPs per training volume, end to end
        step_roofline = {"algorithmic_bytes_per_volume": alg[0], "algorithmic_flops_per_volume": alg[1],
                         "step_hbm_frac": alg[0] * value / world / 1e9 / HBM_PEAK_GBS,
                         "step_mfma_frac": alg[1] * value / world / 1e12 / 2500.0}

    # ---- per-kernel roofline: every launch of the library bracketed by HIP timing events on its launch stream (csrc/ktimer.cpp),
    # a few extra un-timed eager steps
    roofline = None
    nprof = 3
    if not args.no_kernel_timing:
        # EVERY rank runs these extra steps (a step contains collectives when world > 1); only rank 0 times its launches
        if rank == 0:
            ops.PROFILER = ops.KernelProfiler()
        tp0 = time.perf_counter()
        for _ in range(nprof):
            tr.step(vol, lab)
        barrier()
        prof_ms_per_step = (time.perf_counter() - tp0) / nprof * 1e3
    if not args.no_kernel_timing and rank == 0:
        ops.PROFILER.close()
        summ = ops.PROFILER.summary()
        ops.PROFILER = None
        peak_of = lambda k: 157.3 if (args.dtype != "bf16" or any(t in k for t in F32_MFMA_KERNELS)) else 2500.0   # noqa: E731

        def fracs(r, k):
            sec = r["ms"] * 1e-3
            return r["bytes"] / sec / 1e9 / HBM_PEAK_GBS, r["flops"] / sec / 1e12 / peak_of(k)
        # The roofline object describes the kernel with the LARGEST share of the step among all kernels (every record is one kernel:
        # finalize / reduce / finish launches are rows of their own, with no algorithmic bytes).  `secondary` keeps the largest
        # matrix-core kernel beside it.
        ranked = sorted(summ, key=lambda k: -summ[k]["ms"])
        dom = ranked[0]
        r = summ[dom]
        f_hbm, f_mfma = fracs(r, dom)
        bound = "hbm" if (r["bytes"] / (HBM_PEAK_GBS * 1e9)) >= (r["flops"] / (peak_of(dom) * 1e12)) else "mfma"

        def traffic_of(k):       # HBM bytes per launch from the PMC passes committed under profiles/ (separate rocprofv3 --pmc runs)
            for name in ("r03_pmc_traffic.json", "r02_pmc_traffic.json"):
                try:
                    pmc = json.load(open(os.path.join(ROOT, "profiles", name)))["per_kernel"]
                except (OSError, KeyError, ValueError):
                    continue
                base = k.replace(" ", "")
                for pk, v in pmc.items():
                    if pk.replace("void ", "").replace(" ", "") == base:
                        return v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]
            return None
        # self-check of the table: a bracket that does not enclose its kernel shows up as a fraction above 1, and the kernels of one
        # stream cannot add up to more than the (eager, profiled) step they were timed in
        main_stream = torch.cuda.current_stream().cuda_stream
        main_ms = sum(v["ms_by_stream"].get(main_stream, 0.0) for v in summ.values()) / nprof
        fr = {k: fracs(v, k) for k, v in summ.items() if v["ms"] > 0}
        over = sorted(k for k, (h, m) in fr.items() if h > 1.0 or m > 1.0)
        self_check = {"ok": not over and main_ms <= prof_ms_per_step * 1.02, "fractions_above_1": over,
                      "main_stream_kernel_ms_per_step": round(main_ms, 4), "profiled_step_ms": round(prof_ms_per_step, 4)}
        if not self_check["ok"]:
            note(f"SELF-CHECK FAILED: {self_check}")
            if args.strict:
                raise SystemExit(f"bench self-check failed: {self_check}")

        def entry(k):
            v = summ[k]
            h, m = fracs(v, k)
            b = "hbm" if (v["bytes"] / (HBM_PEAK_GBS * 1e9)) >= (v["flops"] / (peak_of(k) * 1e12)) else "mfma"
            sec = v["ms"] * 1e-3
            return {"kernel": k, "bound": b, "achieved": v["bytes"] / sec / 1e9 if b == "hbm" else v["flops"] / sec / 1e12,
                    "peak": HBM_PEAK_GBS if b == "hbm" else peak_of(k), "unit": "GB/s" if b == "hbm" else "TFLOP/s",
                    "frac": h if b == "hbm" else m, "traffic": traffic_of(k),
                    "algorithmic_bytes_per_launch": v["bytes"] / v["launches"], "algorithmic_flops_per_launch": v["flops"] / v["launches"],
                    "avg_launch_ms": v["ms"] / v["launches"], "launches_per_step": v["launches"] // nprof,
                    "ms_per_step": v["ms"] / nprof, "hbm_frac": h, "mfma_frac": m}
        mfma_kernels = [k for k in ranked if summ[k]["flops"] / (peak_of(k) * 1e12) > summ[k]["bytes"] / (HBM_PEAK_GBS * 1e9)]
        roofline = entry(dom)
        roofline.update({
            "self_check": self_check, "selection": "largest summed HIP-event duration over ALL kernels of the step",
            "secondary": entry(mfma_kernels[0]) if mfma_kernels and mfma_kernels[0] != dom else None,
            "step": step_roofline,
            "per_kernel": {k: {"ms_per_step": round(summ[k]["ms"] / nprof, 4), "launches_per_step": summ[k]["launches"] // nprof,
                               "avg_launch_us": round(summ[k]["ms"] / summ[k]["launches"] * 1e3, 2),
                               "algorithmic_bytes_per_step": summ[k]["bytes"] / nprof, "algorithmic_flops_per_step": summ[k]["flops"] / nprof,
                               "hbm_frac": round(fr[k][0], 4) if k in fr else None, "mfma_frac": round(fr[k][1], 4) if k in fr else None,
                               "entry_points": summ[k]["regions"]} for k in ranked}})

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        note("cpu baseline (oracle on host cores)")
        cpu = cpu_baseline(args.model, patch, 1337)
        note("done")

    if rank == 0:
        line = {"metric": "train volumes/sec (96^3 patch)", "value": value, "unit": "volumes/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": ms_per_step, "repeats": len(repeats),
                "ms_per_step_repeats": [round(r / args.steps * 1e3, 4) for r in repeats], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
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
