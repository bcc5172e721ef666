#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
B="python bench.py --no-cpu-baseline --no-kernel-timing --steps 60 --warmup 8 --repeats 1 --patch 112 112 80 --feature-scaler 4"
run() { echo -n "$1  "; env $1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3),'ms', round(d['value'],1),'vol/s')"; }
for i in 1 2 3 4; do
run "X=0"
run "DYCON_TEACHER_HEAD_STREAM=0 DYCON_SIDE_PRIORITY=0,0,0 DYCON_WGRAD_TWO_STREAMS=0"
run "DYCON_TEACHER_HEAD_STREAM=0 DYCON_SIDE_PRIORITY=0,0,0 DYCON_WGRAD_TWO_STREAMS=0 DYCON_WGRAD_W8=0 DYCON_TILE_KS64=0"
run "DYCON_TEACHER_HEAD_STREAM=0 DYCON_SIDE_PRIORITY=0,0,0 DYCON_WGRAD_TWO_STREAMS=0 DYCON_WGRAD_W8=0 DYCON_TILE_KS64=0 DYCON_EVENT_FLAGS=default"
echo -n "r02 tree  "; ( cd _r02 && python bench.py --no-cpu-baseline --no-kernel-timing --steps 60 --warmup 8 --patch 112 112 80 --feature-scaler 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3),'ms', round(d['value'],1),'vol/s')" )
done 2>&1 | tee gpurun_out/isles_switches3.txt
