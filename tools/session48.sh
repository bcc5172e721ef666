#!/bin/bash
# feature-branch backward right behind its forward (coef written at the head of the step): parity, headline + config-5 A/B (DYCON_FEAT_BWD_EARLY)
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_trainer_gpu.py tests/test_ddp_gpu.py tests/test_boundary_gpu.py -x -q 2>&1 | tail -3 || exit 1
out=gpurun_out/s48_feat_bwd_early.txt
: > $out
for i in 1 2 3; do for e in 0 1; do echo -n "DYCON_FEAT_BWD_EARLY=$e  " >> $out; DYCON_FEAT_BWD_EARLY=$e timeout -k 10 200 bash tools/variant_bench.sh dycon_paper_replication_amd/libdycon_hip.so 2>&1 | grep -v amdgpu.ids >> $out || exit 1; done; done
B="python bench.py --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 8 --repeats 1 --patch 112 112 80 --feature-scaler 4"
for i in 1 2; do for e in 0 1; do echo -n "config 5 DYCON_FEAT_BWD_EARLY=$e  " >> $out; DYCON_FEAT_BWD_EARLY=$e timeout -k 10 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1),'vol/s', round(d['ms_per_step'],3),'ms', d['config']['final_loss'])" >> $out || exit 1; done; done
cat $out
