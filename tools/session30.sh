#!/bin/bash
# conv_k3_tile_kernel: incremental scalar address generation: parity, micro + step A/B against the library built from HEAD
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -x -q -k "conv" 2>&1 | tail -3 || exit 1
out=gpurun_out/s30_tile_incremental.txt
: > $out
for i in 1 2; do for lib in build_variants/lib_head.so dycon_paper_replication_amd/libdycon_hip.so; do
  echo "== $lib" >> $out
  for shape in "128 128 12" "256 256 6" "128 128 14" "256 256 7"; do
    DYCON_LIB=$PWD/$lib timeout -k 10 120 python tools/conv_micro2.py $shape 200 2>&1 | grep -v amdgpu.ids >> $out || exit 1
  done
done; done
for i in 1 2 3; do for lib in build_variants/lib_head.so dycon_paper_replication_amd/libdycon_hip.so; do timeout -k 10 200 bash tools/variant_bench.sh $lib 2>&1 | grep -v amdgpu.ids >> $out || exit 1; done; done
cat $out
