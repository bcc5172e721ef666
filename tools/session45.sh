#!/bin/bash
# secondary configurations on the round's final build
R=$GRAFT_REPO_ROOT; cd $R
B="python bench.py --no-cpu-baseline --no-kernel-timing --steps 60 --warmup 8 --repeats 1"
run() { timeout -k 10 300 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1),'vol/s', round(d['ms_per_step'],3),'ms', d['config']['workload'])"; }
{ run --model unet_3D; run --patch 112 112 96; run --patch 112 112 80 --feature-scaler 4; run --dtype f32; } | tee gpurun_out/s45_secondary.txt
