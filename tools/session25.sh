#!/bin/bash
# hardware exp / log / rcp in the FAST (bf16-step) loss paths: parity, FeCL per pass, step A/B against the library built from HEAD
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_ops_gpu.py tests/test_trainer_gpu.py tests/test_boundary_gpu.py -x -q -k "fecl or loss or step or seg or trainer or boundary" 2>&1 | tail -3 || exit 1
out=gpurun_out/s25_fast_math.txt
: > $out
for lib in build_variants/lib_head.so dycon_paper_replication_amd/libdycon_hip.so; do
  echo "== $lib" >> $out
  DYCON_LIB=$PWD/$lib timeout -k 10 300 python tools/fecl_micro.py 15680 2 5 2>&1 | grep -v amdgpu.ids >> $out || exit 1
  DYCON_LIB=$PWD/$lib timeout -k 10 300 python tools/fecl_micro.py 1728 4 50 2>&1 | grep -v amdgpu.ids >> $out || exit 1
done
for i in 1 2 3; do for lib in build_variants/lib_head.so dycon_paper_replication_amd/libdycon_hip.so; do timeout -k 10 200 bash tools/variant_bench.sh $lib 2>&1 | grep -v amdgpu.ids >> $out || exit 1; done; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/s25_prof -o run --output-format csv -- python3 $R/tools/fecl_micro.py 15680 2 3 > $R/gpurun_out/s25_prof.log 2>&1 || exit 1
cd $R
python - <<'PY' >> $out
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/s25_prof/run_kernel_trace.csv")))
agg = collections.defaultdict(list)
for r in rows:
    if "fecl" in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:48]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    print(f"{k}  {len(v)} dispatches, us: {[round(x) for x in v]}")
PY
cat $out
