#!/bin/bash
# trilinear kernels: parity, then U-Net step A/B against the library built from HEAD
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -x -q -k "trilinear or unet or pool" 2>&1 | tail -3 || exit 1
out=gpurun_out/s32_trilinear.txt
: > $out
B="python bench.py --model unet_3D --no-cpu-baseline --no-kernel-timing --steps 60 --warmup 8 --repeats 1"
for i in 1 2 3; do for lib in build_variants/lib_head.so dycon_paper_replication_amd/libdycon_hip.so; do
  echo -n "$lib  " >> $out
  DYCON_LIB=$PWD/$lib timeout -k 10 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3),'ms', round(d['value'],1),'vol/s')" >> $out || exit 1
done; done
cat $out
