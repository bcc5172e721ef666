#!/bin/bash
# round 3, GPU session 1: boundary / event diagnostics
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s1; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_trainer_gpu.py tests/test_ddp_gpu.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
hipcc -O2 --offload-arch=gfx950 tools/chain_micro.hip -o /tmp/chain_micro && timeout -k 10 300 /tmp/chain_micro > $O/chain_micro.txt 2>&1
cat $O/chain_micro.txt
timeout -k 10 900 python tools/ablate.py 60 > $O/ablate.txt 2>&1; cat $O/ablate.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --hip-runtime-trace -d $O/prof4 -o run --output-format csv -- python3 $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-kernel-timing > $O/prof4.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --hip-runtime-trace -d $O/prof1 -o run --output-format csv -- python3 $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-kernel-timing --cfg overlap_teacher=False,overlap_wgrad=False,overlap_features=False,split_repack=False > $O/prof1.log 2>&1
cd $R
ls $O/prof4 $O/prof1
for d in prof4 prof1; do
  python tools/gap_hist.py $O/$d/run_kernel_trace.csv $O/$d/run_hip_api_trace.csv > $O/gaps_$d.txt 2>&1
  python profiles/timeline.py $O/$d/run_kernel_trace.csv --full > $O/timeline_$d.txt 2>&1
done
head -50 $O/gaps_prof4.txt
# keep the merged-back output small: traces are tens of MB
for d in prof4 prof1; do gzip -f $O/$d/run_hip_api_trace.csv 2>/dev/null; gzip -f $O/$d/run_kernel_trace.csv 2>/dev/null; done
du -sh $O
