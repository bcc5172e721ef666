#!/bin/bash
# config 5 (FeCL-bound): HIP priority of the feature stream
R=$GRAFT_REPO_ROOT; cd $R
B="python bench.py --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 8 --repeats 1 --patch 112 112 80 --feature-scaler 4"
for i in 1 2; do for p in "-1,0,0" "-1,0,-1" "0,0,-1" "-1,-1,-1"; do echo -n "DYCON_SIDE_PRIORITY=$p  "; DYCON_SIDE_PRIORITY=$p timeout -k 10 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1),'vol/s', round(d['ms_per_step'],3),'ms')"; done; done | tee gpurun_out/s47_config5_priority.txt
