#!/bin/bash
# FeCL: XCD-aware workgroup -> (row block, sample, column split) mapping: parity, A/B
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py -x -q -k "fecl" 2>&1 | tail -3 || exit 1
out=gpurun_out/s26_fecl_xcd.txt
: > $out
for i in 1 2; do for e in 0 1; do
  echo "DYCON_FECL_XCD=$e" >> $out
  DYCON_FECL_XCD=$e timeout -k 10 300 python tools/fecl_micro.py 15680 2 5 2>&1 | grep -v amdgpu.ids >> $out || exit 1
  DYCON_FECL_XCD=$e timeout -k 10 300 python tools/fecl_micro.py 15680 4 3 2>&1 | grep -v amdgpu.ids >> $out || exit 1
  DYCON_FECL_XCD=$e timeout -k 10 300 python tools/fecl_micro.py 2352 4 30 2>&1 | grep -v amdgpu.ids >> $out || exit 1
  DYCON_FECL_XCD=$e timeout -k 10 300 python tools/fecl_micro.py 1728 4 50 2>&1 | grep -v amdgpu.ids >> $out || exit 1
done; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/s26_prof -o run --output-format csv -- python3 $R/tools/fecl_micro.py 15680 2 3 > $R/gpurun_out/s26_prof.log 2>&1 || exit 1
cd $R
python - <<'PY' >> $out
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/s26_prof/run_kernel_trace.csv")))
agg = collections.defaultdict(list)
for r in rows:
    if "fecl" in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:48]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    print(f"{k}  {len(v)} dispatches, us: {[round(x) for x in v]}")
PY
cat $out
