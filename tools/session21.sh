#!/bin/bash
# conv_k3_tile_kernel<2, BREG>: weight fragments straight into registers (no LDS round trip): parity, micro A/B, step A/B
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -x -q -k "conv" 2>&1 | tail -5 || exit 1
out=gpurun_out/s21_tile_breg.txt
: > $out
for i in 1 2; do for e in 0 1; do
  echo "DYCON_TILE_BREG=$e" >> $out
  for shape in "128 128 12" "256 256 6" "128 128 14" "256 256 7"; do
    DYCON_TILE_BREG=$e timeout -k 10 120 python tools/conv_micro2.py $shape 200 2>&1 | grep -v amdgpu.ids >> $out || exit 1
  done
done; done
for i in 1 2 3; do for e in 0 1; do echo -n "DYCON_TILE_BREG=$e  " >> $out; DYCON_TILE_BREG=$e timeout -k 10 200 bash tools/variant_bench.sh dycon_paper_replication_amd/libdycon_hip.so 2>&1 | grep -v amdgpu.ids >> $out || exit 1; done; done
cat $out
