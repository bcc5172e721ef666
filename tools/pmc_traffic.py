"""Turn the two counter passes of tools/pmc_traffic.sh into profiles/<name>.json: HBM bytes per launch of every KERNEL of the bench
step.  Corrections as MI355X_MICROARCH.md (HBM section) prescribes for gfx950: bytes = 2 * FETCH_SIZE * 1024 (FETCH_SIZE tallies
128-B requests at 64 B; unit KiB) + WRITE_SIZE * 1024.
usage: pmc_traffic.py <tag> <out.json> [bench_line.json [steps_in_the_pmc_run]]
With a bench line (roofline.per_kernel: the algorithmic bytes per step of every kernel, keyed by kernel name as ktimer.cpp reports
it), every kernel also gets its measured bytes per STEP and their ratio to the algorithmic bytes; helper launches (finalize /
reduce / finish: no algorithmic bytes of their own) are listed with the entry points they serve, and the whole step is summed."""
import collections
import csv
import json
import sys

tag, out = sys.argv[1], sys.argv[2]
bench_line = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else None
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3


def load(kind):
    rows = list(csv.DictReader(open(f"gpurun_out/pmc_{tag}_{kind}/run_counter_collection.csv")))
    per = collections.defaultdict(lambda: collections.defaultdict(float))      # kernel -> dispatch -> value
    for r in rows:
        per[r["Kernel_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return per


def norm(name):
    return name.split("(")[0].replace("void ", "").replace(" ", "")


fetch, write = load("fetch"), load("write")
kern = {}
for name in sorted(set(fetch) | set(write)):
    f = list(fetch.get(name, {}).values())
    w = list(write.get(name, {}).values())
    kern[name.split("(")[0]] = {"launches_counted": max(len(f), len(w)),
                                "fetch_bytes_per_launch": 2 * 1024 * sum(f) / max(len(f), 1),
                                "write_bytes_per_launch": 1024 * sum(w) / max(len(w), 1)}
per_step = None
if bench_line is not None:
    alg = {norm(k): v for k, v in bench_line["roofline"]["per_kernel"].items()}
    per_step = {}
    tot_meas = tot_alg = 0.0
    for short, v in kern.items():
        meas = v["launches_counted"] * (v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) / steps
        a = alg.get(norm(short))
        if a is None:
            continue                                   # not a kernel of this library (torch copies / fills)
        tot_meas += meas
        tot_alg += a["algorithmic_bytes_per_step"]
        per_step[short.replace("void ", "")] = {"measured_bytes_per_step": meas, "algorithmic_bytes_per_step": a["algorithmic_bytes_per_step"],
                                                "ratio": (meas / a["algorithmic_bytes_per_step"]) if a["algorithmic_bytes_per_step"] else None,
                                                "entry_points": a["entry_points"]}
    per_step["WHOLE STEP (all kernels of the library)"] = {"measured_bytes_per_step": tot_meas, "algorithmic_bytes_per_step": tot_alg,
                                                           "ratio": tot_meas / max(tot_alg, 1.0)}
doc = {"steps_in_run": steps, "per_kernel_per_step": per_step,
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `bench.py --steps 2 --warmup 1` (tools/pmc_traffic.sh); "
                 "bytes = 2*FETCH_SIZE*1024 (gfx950 half-count correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE*1024",
       "per_kernel": kern}
json.dump(doc, open(out, "w"), indent=1)
if per_step:
    for k, v in sorted(per_step.items(), key=lambda kv: -kv[1]["measured_bytes_per_step"]):
        r = "   -  " if v["ratio"] is None else f"{v['ratio']:6.2f}"
        print(f"{v['measured_bytes_per_step'] / 1e6:10.1f} MB measured  {v['algorithmic_bytes_per_step'] / 1e6:10.1f} MB algorithmic  x{r}  {k}")
