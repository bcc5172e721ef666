#!/bin/bash
# headline step with the 128-row FeCL kernels forced at N = 1728 (DYCON_FECL_ROWS128_MIN_N=1024) against the default (64-row kernels there)
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
out=gpurun_out/s36_fecl_small_n.txt
: > $out
for i in 1 2 3; do for v in 8192 1024; do
  echo -n "DYCON_FECL_ROWS128_MIN_N=$v  " >> $out
  DYCON_FECL_ROWS128_MIN_N=$v timeout -k 10 200 bash tools/variant_bench.sh dycon_paper_replication_amd/libdycon_hip.so 2>&1 | grep -v amdgpu.ids >> $out || exit 1
done; done
for v in 8192 1024; do DYCON_FECL_ROWS128_MIN_N=$v timeout -k 10 300 python tools/fecl_micro.py 1728 4 50 2>&1 | grep -v amdgpu.ids >> $out; done
cat $out
