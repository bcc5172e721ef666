#!/bin/bash
# norm statistics row loop unrolled x4 (loads issued ahead): parity, step A/B against the library built from HEAD
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_boundary_gpu.py -x -q -k "norm or boundary or reference_step" 2>&1 | tail -3 || exit 1
out=gpurun_out/s43_norm_unroll.txt
: > $out
for i in 1 2 3; do for lib in build_variants/lib_head.so dycon_paper_replication_amd/libdycon_hip.so; do timeout -k 10 200 bash tools/variant_bench.sh $lib 2>&1 | grep -v amdgpu.ids >> $out || exit 1; done; done
cat $out
