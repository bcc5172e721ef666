#!/bin/bash
# U-Net step against the norm chunk plan
R=$GRAFT_REPO_ROOT; cd $R
B="python bench.py --model unet_3D --no-cpu-baseline --no-kernel-timing --steps 60 --warmup 8 --repeats 1"
for i in 1 2; do for c in "512 64" "256 128" "256 64" "384 64"; do set -- $c; echo -n "DYCON_NORM_CHUNKS=$1 MIN_ROWS=$2  "; DYCON_NORM_CHUNKS=$1 DYCON_NORM_MIN_ROWS=$2 timeout -k 10 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1),'vol/s', round(d['ms_per_step'],3),'ms')"; done; done | tee gpurun_out/s46_unet_norm_plan.txt
