#!/bin/bash
# round 3, GPU session 9: weight-gradient staging geometry hoisted (A/B against the previous commit's library)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s9; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py "tests/test_fullsize_gpu.py::test_conv_full_size_bf16" tests/test_trainer_gpu.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for lib in build_variants/lib_base.so dycon_paper_replication_amd/libdycon_hip.so; do
  echo "=== $lib"
  for shp in "16 16 96" "32 32 48" "64 64 24" "128 128 12" "256 256 6"; do DYCON_LIB=$PWD/$lib python tools/wgrad_micro.py $shp 30; done
done 2>&1 | grep -v amdgpu.ids | tee $O/wgrad_micro.txt
for i in 1 2; do bash tools/variant_bench.sh build_variants/lib_base.so dycon_paper_replication_amd/libdycon_hip.so; done 2>&1 | tee $O/variant_bench.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/ddp1 -o run --output-format csv -- python3 $R/tools/ddp_one_rank.py 12 1 > $O/ddp1.log 2>&1
grep "ms/step" $O/ddp1.log
python3 $R/profiles/timeline.py $O/ddp1/run_kernel_trace.csv > $O/timeline_ddp1.txt 2>&1; head -8 $O/timeline_ddp1.txt
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$O/ddp1/run_kernel_trace.csv")))
byq = collections.defaultdict(collections.Counter)
for r in rows:
    byq[r["Queue_Id"]][r["Kernel_Name"].split("(")[0][-40:]] += 1
for q, c in byq.items():
    print("queue", q, sum(c.values()), "launches:", c.most_common(6))
PY
gzip -f $O/ddp1/run_kernel_trace.csv
