#!/bin/bash
# config 5 (ISLES geometry): final loss + skipped steps with the old and the new FeCL kernels; stream timeline of the step
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
out=gpurun_out/s35_config5_check.txt
: > $out
B="python bench.py --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 8 --repeats 1 --patch 112 112 80 --feature-scaler 4"
for v in "DYCON_FECL_ROWS128_MIN_N=1000000" "DYCON_FECL_GRAD128=0" "DYCON_FECL_GRAD128=1"; do
  echo -n "$v  " >> $out
  env $v timeout -k 10 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3),'ms', round(d['value'],1),'vol/s', d['config']['final_loss'], d['config']['skipped_steps'])" >> $out || exit 1
done
cat $out
