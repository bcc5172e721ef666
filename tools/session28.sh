#!/bin/bash
# SQ counters of fecl_rows128_kernel at the ISLES size
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_fecl2_?
MICRO=fecl_micro.py REPS=1 timeout -k 10 900 bash tools/pmc_conv.sh fecl2 15680 2 || exit 1
cd $R
for p in 1 2 3; do echo "== fecl_rows128_kernel<$p>"; python tools/pmc_summary.py fecl2 "fecl_rows128_kernel<$p>"; done > gpurun_out/s28_fecl_pmc.txt
cat gpurun_out/s28_fecl_pmc.txt
