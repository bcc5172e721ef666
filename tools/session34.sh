#!/bin/bash
# FeCL (rows128 passes 1-4, branch-free epilogues): parity incl. trainer / full-size step cases, per-pass durations, config-5 step
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py tests/test_trainer_gpu.py -x -q -k "fecl or step or trainer" > gpurun_out/s34_pytest.txt 2>&1; rc=$?; tail -3 gpurun_out/s34_pytest.txt; [ $rc -eq 0 ] || exit 1
out=gpurun_out/s34_config5.txt
: > $out
B="python bench.py --no-cpu-baseline --no-kernel-timing --steps 60 --warmup 8 --repeats 1 --patch 112 112 80 --feature-scaler 4"
for i in 1 2; do for lib in build_variants/lib_head.so dycon_paper_replication_amd/libdycon_hip.so; do
  echo -n "$lib  " >> $out
  DYCON_LIB=$PWD/$lib timeout -k 10 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3),'ms', round(d['value'],1),'vol/s')" >> $out || exit 1
done; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/s34_prof -o run --output-format csv -- python3 $R/tools/fecl_micro.py 15680 2 3 > $R/gpurun_out/s34_prof.log 2>&1 || exit 1
cd $R
python - <<'PY' >> $out
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/s34_prof/run_kernel_trace.csv")))
agg = collections.defaultdict(list)
for r in rows:
    if "fecl" in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:52]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    print(f"{k}  {len(v)} dispatches, us: {[round(x) for x in v]}")
PY
cat $out
