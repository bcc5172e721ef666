#!/bin/bash
# SQ counters of the four FeCL passes at the ISLES size (N = 15680, B = 2)
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
MICRO=fecl_micro.py REPS=1 timeout -k 10 900 bash tools/pmc_conv.sh fecl 15680 2 || exit 1
cd $R
for p in 1 2 3 4; do echo "== fecl_kernel<bf16, $p>"; python tools/pmc_summary.py fecl "fecl_kernel<__hip_bfloat16, $p>"; done > gpurun_out/s23_fecl_pmc.txt
grep -h "fecl_kernel" gpurun_out/pmc_fecl_a/run_kernel_trace.csv | awk -F, '{print $0}' | head -3 > /dev/null
python - <<'PY' >> gpurun_out/s23_fecl_pmc.txt
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/pmc_fecl_a/run_kernel_trace.csv")))
agg = collections.defaultdict(list)
for r in rows:
    if "fecl_kernel" in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:48]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    print(f"{k}  {len(v)} dispatches, us: {[round(x) for x in v]}")
PY
cat gpurun_out/s23_fecl_pmc.txt
