#!/bin/bash
# FeCL rows128 kernel: parity (all FeCL tests + trainer/fullsize step cases), config-5 step against the library built from HEAD~ (lib_head.so)
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py tests/test_trainer_gpu.py -x -q -k "fecl or step or trainer" 2>&1 | tail -4 || exit 1
out=gpurun_out/s29_config5.txt
: > $out
B="python bench.py --no-cpu-baseline --no-kernel-timing --steps 60 --warmup 8 --repeats 1 --patch 112 112 80 --feature-scaler 4"
for i in 1 2 3; do for lib in build_variants/lib_head.so dycon_paper_replication_amd/libdycon_hip.so; do
  echo -n "$lib  " >> $out
  DYCON_LIB=$PWD/$lib timeout -k 10 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3),'ms', round(d['value'],1),'vol/s')" >> $out || exit 1
done; done
timeout -k 10 300 python tools/fecl_micro.py 15680 2 5 2>&1 | grep -v amdgpu.ids >> $out
cat $out
