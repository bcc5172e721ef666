#!/bin/bash
# low HIP priority (1) for the weight-gradient / feature streams (DYCON_SIDE_PRIORITY = teacher, weight gradients, features)
R=$GRAFT_REPO_ROOT; cd $R
for i in 1 2; do for p in "-1,0,0" "-1,1,0" "-1,1,1" "-1,0,1" "0,1,1"; do echo -n "DYCON_SIDE_PRIORITY=$p  "; DYCON_SIDE_PRIORITY=$p timeout -k 10 200 bash tools/variant_bench.sh dycon_paper_replication_amd/libdycon_hip.so; done; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s38_low_priority.txt
