#!/bin/bash
# FeCL gradient pass on 128-row blocks: parity, A/B (DYCON_FECL_GRAD128=0 is fecl_kernel<bf16, 4>)
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py -x -q -k "fecl" 2>&1 | tail -8 || exit 1
out=gpurun_out/s33_fecl_grad128.txt
: > $out
for i in 1 2; do for e in 0 1; do
  echo "DYCON_FECL_GRAD128=$e" >> $out
  DYCON_FECL_GRAD128=$e timeout -k 10 300 python tools/fecl_micro.py 15680 2 5 2>&1 | grep -v amdgpu.ids >> $out || exit 1
  DYCON_FECL_GRAD128=$e timeout -k 10 300 python tools/fecl_micro.py 15680 4 3 2>&1 | grep -v amdgpu.ids >> $out || exit 1
  DYCON_FECL_GRAD128=$e timeout -k 10 300 python tools/fecl_micro.py 8192 4 10 2>&1 | grep -v amdgpu.ids >> $out || exit 1
done; done
cat $out
