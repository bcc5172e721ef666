#!/bin/bash
# same-box comparison of the round-2 tree (worktree _r02, its own library) and HEAD: headline and ISLES geometry
R=$GRAFT_REPO_ROOT; cd $R
run() { ( cd $1 && python bench.py --no-cpu-baseline --no-kernel-timing --steps $3 --warmup 8 $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3),'ms', round(d['value'],1),'vol/s')" ); }
for i in 1 2 3; do
  echo -n "r02  ISLES   "; run _r02 "--patch 112 112 80 --feature-scaler 4" 40
  echo -n "HEAD ISLES   "; run . "--patch 112 112 80 --feature-scaler 4 --repeats 1" 40
done 2>&1 | tee gpurun_out/r02_vs_head.txt
for i in 1 2; do
  echo -n "r02  headline "; run _r02 "" 100
  echo -n "HEAD headline "; run . "--repeats 1" 100
  echo -n "r02  unet     "; run _r02 "--model unet_3D" 60
  echo -n "HEAD unet     "; run . "--model unet_3D --repeats 1" 60
done 2>&1 | tee -a gpurun_out/r02_vs_head.txt
