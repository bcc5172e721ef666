#!/bin/bash
# grid of the norm apply passes (vectors per thread, cap on workgroups per sample), step A/B
R=$GRAFT_REPO_ROOT; cd $R
for i in 1 2; do for c in "4 2048" "8 2048" "2 2048" "4 1024" "8 1024" "16 2048" "2 4096"; do set -- $c; echo -n "APPLY_VPT=$1 CAP=$2  "; DYCON_NORM_APPLY_VPT=$1 DYCON_NORM_APPLY_CAP=$2 timeout -k 10 200 bash tools/variant_bench.sh dycon_paper_replication_amd/libdycon_hip.so; done; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s42_norm_apply_grid.txt
