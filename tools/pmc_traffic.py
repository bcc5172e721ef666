"""Turn the two counter passes of tools/pmc_traffic.sh into profiles/<name>.json: HBM bytes per launch for every kernel and
for the kernel families bench.py reports ("regions").  Corrections as MI355X_MICROARCH.md (HBM section) prescribes for gfx950:
bytes = 2 * FETCH_SIZE * 1024 (FETCH_SIZE tallies 128-B requests at 64 B; unit KiB) + WRITE_SIZE * 1024.
usage: pmc_traffic.py <tag> <out.json> [bench_line.json [steps_in_the_pmc_run]]
With a bench line (its roofline.per_kernel_algorithmic_bytes_per_step), every family also gets measured bytes per STEP and their
ratio to the algorithmic bytes (reduce / finish / finalize launches are charged to the family they serve)."""
import collections
import csv
import json
import sys

tag, out = sys.argv[1], sys.argv[2]
bench_line = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else None
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3

# kernel-name substring -> bench.py region (ops._Region names)
REGIONS = [("head_1x1_fwd_kernel", "conv_direct"), ("head_1x1_bwd_kernel", "conv_direct"), ("grad_1x1_skinny_kernel", "conv_direct"),
           ("conv_direct_kernel", "conv_direct"), ("conv_k3_p16_kernel", "conv_k3_p16"), ("conv_k3_c1_kernel", "conv_k3_c1"), ("conv_k3_lds_kernel", "conv_k3_lds"), ("conv_k3_p32_kernel", "conv_k3_lds"),
           ("conv_k3_tile_kernel", "conv_k3_tile"), ("wgrad_k3_bf16_kernel", "wgrad_k3_bf16"), ("wgrad_k3_c1_kernel", "wgrad_k3_bf16"), ("first_block_bwd_kernel", "wgrad_k3_bf16"), ("wgrad_1x1_bf16_kernel", "conv_wgrad"), ("wgrad_k2s2_bf16_kernel", "conv_wgrad"),
           ("conv_wgrad_kernel", "conv_wgrad"), ("conv_gemm_kernel", "conv_gemm"), ("norm_apply_kernel", "norm_fwd"), ("norm_apply_head_kernel", "norm_fwd"), ("norm_head_partial_kernel", "norm_bwd"),
           ("norm_head_bwd_apply_kernel", "norm_bwd"),
           ("norm_partial_kernel<__hip_bfloat16, 0>", "norm_fwd"), ("norm_fused_fwd_kernel", "norm_fwd"),
           ("norm_partial_kernel<__hip_bfloat16, 1>", "norm_bwd"), ("norm_bwd_apply_kernel", "norm_bwd"), ("norm_fused_bwd_kernel", "norm_bwd")]
# per-step accounting: the helper launches of a family
STEP_EXTRA = [("norm_finalize_stats_kernel", "norm_fwd"), ("norm_finalize_bwd_kernel", "norm_bwd"), ("norm_sum_dparams_kernel", "norm_bwd"), ("norm_head_finalize_kernel", "norm_bwd"),
              ("splitk_finish_kernel", "conv_k3_tile+conv_gemm_splitk"), ("reduce_partials", "wgrad_k3_bf16+conv_wgrad"), ("first_block_reduce_kernel", "wgrad_k3_bf16+conv_wgrad"),
              ("first_block_finalize_kernel", "wgrad_k3_bf16+conv_wgrad"), ("colsum_kernel", "colsum")]


def load(kind):
    rows = list(csv.DictReader(open(f"gpurun_out/pmc_{tag}_{kind}/run_counter_collection.csv")))
    per = collections.defaultdict(lambda: collections.defaultdict(float))      # kernel -> dispatch -> value
    for r in rows:
        per[r["Kernel_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return per


fetch, write = load("fetch"), load("write")
kern = {}
for name in sorted(set(fetch) | set(write)):
    f = list(fetch.get(name, {}).values())
    w = list(write.get(name, {}).values())
    short = name.split("(")[0]
    kern[short] = {"launches_counted": max(len(f), len(w)),
                   "fetch_bytes_per_launch": 2 * 1024 * sum(f) / max(len(f), 1),
                   "write_bytes_per_launch": 1024 * sum(w) / max(len(w), 1)}
reg = collections.defaultdict(lambda: [0, 0.0])
for short, v in kern.items():
    for pat, region in REGIONS:
        if pat in short:
            reg[region][0] += v["launches_counted"]
            reg[region][1] += v["launches_counted"] * (v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"])
            break
per_step = collections.defaultdict(float)
for short, v in kern.items():
    tot = v["launches_counted"] * (v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) / steps
    for pat, region in REGIONS + STEP_EXTRA:
        if pat in short:
            per_step[region] += tot
            break
ratios = None
if bench_line is not None:
    alg = bench_line["roofline"]["per_kernel_algorithmic_bytes_per_step"]
    alg = dict(alg, **{"conv_k3_tile+conv_gemm_splitk": alg.get("conv_k3_tile", 0) + alg.get("conv_gemm_splitk", 0),
                       "wgrad_k3_bf16+conv_wgrad": alg.get("wgrad_k3_bf16", 0) + alg.get("conv_wgrad", 0)})
    groups = {"conv_k3_tile+conv_gemm_splitk": ("conv_k3_tile", "conv_gemm"), "wgrad_k3_bf16+conv_wgrad": ("wgrad_k3_bf16", "conv_wgrad")}
    ratios = {}
    for region in sorted(set(per_step) | set(alg)):
        if region in groups or region not in alg or alg[region] <= 0:
            continue
        meas = per_step.get(region, 0.0)
        ratios[region] = {"measured_bytes_per_step": meas, "algorithmic_bytes_per_step": alg[region], "ratio": meas / alg[region]}
    # helper launches shared by two families: charge them to the pair
    for g, members in groups.items():
        meas = per_step.get(g, 0.0) + sum(per_step.get(m, 0.0) for m in members if not (g.startswith("conv_k3_tile") and m == "conv_gemm"))
        a = sum(alg.get(m, 0) for m in (("conv_k3_tile", "conv_gemm_splitk") if g.startswith("conv_k3_tile") else members))
        if a > 0:
            ratios[g + " (incl. helper launches)"] = {"measured_bytes_per_step": meas, "algorithmic_bytes_per_step": a, "ratio": meas / a}
doc = {"steps_in_run": steps, "per_family_per_step": ratios, "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `bench.py --steps 2 --warmup 1` (tools/pmc_traffic.sh); "
                 "bytes = 2*FETCH_SIZE*1024 (gfx950 half-count correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE*1024; "
                 "the finalize / finish launches of a region are not attributed to it",
       "per_region": {k: {"launches_counted": v[0], "hbm_bytes_per_launch": v[1] / max(v[0], 1)} for k, v in sorted(reg.items())},
       "per_kernel": kern}
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps(doc["per_family_per_step"] or doc["per_region"], indent=1))
