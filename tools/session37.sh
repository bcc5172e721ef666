#!/bin/bash
# 128-row FeCL kernels from N = 1024 (headline N = 1728) with their own column-split plan: parity, step A/B against lib_head (HEAD~3)
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py tests/test_trainer_gpu.py tests/test_ddp_gpu.py -x -q -k "fecl or step or trainer or ddp" > gpurun_out/s37_pytest.txt 2>&1; rc=$?; tail -3 gpurun_out/s37_pytest.txt; [ $rc -eq 0 ] || exit 1
out=gpurun_out/s37_fecl_rows128_small_n.txt
: > $out
for i in 1 2 3; do for v in 8192 1024; do
  echo -n "DYCON_FECL_ROWS128_MIN_N=$v  " >> $out
  DYCON_FECL_ROWS128_MIN_N=$v timeout -k 10 200 bash tools/variant_bench.sh dycon_paper_replication_amd/libdycon_hip.so 2>&1 | grep -v amdgpu.ids >> $out || exit 1
done; done
for n in 1728 2352 4096; do for v in 1000000 1024; do echo -n "MIN_N=$v " >> $out; DYCON_FECL_ROWS128_MIN_N=$v timeout -k 10 300 python tools/fecl_micro.py $n 4 30 2>&1 | grep -v amdgpu.ids >> $out; done; done
cat $out
