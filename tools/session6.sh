#!/bin/bash
# round 3, GPU session 6: 96^3 statistics epilogues (tests + A/B), DDP one-rank trace, PMC of the small-level kernels, slab caps
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s6; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_fullsize_gpu.py::test_conv_takes_norm_statistics tests/test_fullsize_gpu.py::test_conv_p32_on_32x32x16_mfma tests/test_engine_gpu.py tests/test_trainer_gpu.py tests/test_ops_gpu.py tests/test_boundary_gpu.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 600 python tools/ab.py "" conv_stats96=False --reps 3 --steps 100 2>&1 | grep "ms/step" | tee $O/ab_stats96.txt
for mb in 24 48 96; do echo -n "DYCON_WGRAD_SLAB_MB=$mb  "; DYCON_WGRAD_SLAB_MB=$mb bash tools/variant_bench.sh dycon_paper_replication_amd/libdycon_hip.so; done 2>&1 | tee $O/slab_mb.txt
hipcc -O2 --offload-arch=gfx950 tools/chain_micro.hip -o /tmp/chain_micro && timeout -k 10 300 /tmp/chain_micro > $O/chain_micro.txt 2>&1; tail -3 $O/chain_micro.txt
cd /tmp && export TMPDIR=/tmp
for f in 0 1; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/ddp$f -o run --output-format csv -- python3 $R/tools/ddp_one_rank.py 12 $f > $O/ddp$f.log 2>&1
  grep "ms/step" $O/ddp$f.log
  python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$O/ddp$f/run_kernel_stats.csv")))
print("force=$f: top kernels by total time; calls")
for r in rows[:12]:
    print("   ", r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
print("   kernels with nccl/rccl in the name:", [(r["Name"][:50], r["Calls"], r["AverageNs"]) for r in rows if "ccl" in r["Name"].lower()])
PY
  python3 $R/profiles/timeline.py $O/ddp$f/run_kernel_trace.csv > $O/timeline_ddp$f.txt 2>&1; head -8 $O/timeline_ddp$f.txt
  gzip -f $O/ddp$f/run_kernel_trace.csv
done
cd $R
MICRO=conv_micro2.py REPS=5 bash tools/pmc_conv.sh tile12 128 128 12 && python tools/pmc_summary.py tile12 conv_k3_tile > $O/pmc_tile12.txt 2>&1
MICRO=conv_micro2.py REPS=5 bash tools/pmc_conv.sh tile6 256 256 6 && python tools/pmc_summary.py tile6 conv_k3_tile > $O/pmc_tile6.txt 2>&1
MICRO=wgrad_micro.py REPS=5 bash tools/pmc_conv.sh wg24 64 64 24 && python tools/pmc_summary.py wg24 wgrad_k3_bf16 > $O/pmc_wg24.txt 2>&1
MICRO=wgrad_micro.py REPS=5 bash tools/pmc_conv.sh wg96 16 16 96 && python tools/pmc_summary.py wg96 wgrad_k3_bf16 > $O/pmc_wg96.txt 2>&1
for t in tile12 tile6 wg24 wg96; do echo "== $t"; cat $O/pmc_$t.txt; done
rm -rf gpurun_out/pmc_tile12_? gpurun_out/pmc_tile6_? gpurun_out/pmc_wg24_? gpurun_out/pmc_wg96_?
