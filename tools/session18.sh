#!/bin/bash
# what bounds conv_k3_tile_kernel<2>: diagnostic builds (CT_DIAG bit mask: 1 no A loads, 2 no B loads, 4 no slab stores, 8 no MFMAs)
set -o pipefail
mkdir -p gpurun_out
out=gpurun_out/s18_tile_diag.txt
: > $out
for v in "" 1 2 3 4 7 8 15; do
  lib=dycon_paper_replication_amd/libdycon_hip.so
  [ -n "$v" ] && lib=build_variants/lib_ctdiag$v.so
  echo "== CT_DIAG=${v:-0}" >> $out
  DYCON_LIB=$PWD/$lib timeout -k 10 120 python tools/conv_micro2.py 128 128 12 200 >> $out 2>&1 || exit 1
  DYCON_LIB=$PWD/$lib timeout -k 10 120 python tools/conv_micro2.py 256 256 6 200 >> $out 2>&1 || exit 1
done
cat $out
timeout -k 10 600 python -m pytest tests/test_ddp_gpu.py -x -q 2>&1 | tail -5
