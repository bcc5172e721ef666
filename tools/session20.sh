#!/bin/bash
# bank-conflict-free LDS images of the k=3 / k2s2 weight-gradient kernels: parity, micro A/B against the natural pitches (-DWG_PAD=0), step A/B
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -x -q -k "wgrad or conv" 2>&1 | tail -5 || exit 1
out=gpurun_out/s20_wgrad_lds_pad.txt
: > $out
for lib in build_variants/lib_nopad.so dycon_paper_replication_amd/libdycon_hip.so; do
  echo "== $lib" >> $out
  for shape in "16 16 96" "32 32 48" "64 64 24" "128 128 12" "256 256 6" "16 32 48"; do
    DYCON_LIB=$PWD/$lib timeout -k 10 120 python tools/wgrad_micro.py $shape 50 2>&1 | grep -v amdgpu.ids >> $out || exit 1
  done
done
for i in 1 2 3; do for lib in build_variants/lib_nopad.so dycon_paper_replication_amd/libdycon_hip.so; do timeout -k 10 200 bash tools/variant_bench.sh $lib 2>&1 | grep -v amdgpu.ids >> $out || exit 1; done; done
cat $out
