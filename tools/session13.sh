#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for i in 1 2 3; do for v in 0 1; do echo -n "DYCON_WGRAD_THREE_STREAMS=$v  "; DYCON_WGRAD_THREE_STREAMS=$v bash tools/variant_bench.sh dycon_paper_replication_amd/libdycon_hip.so; done; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/wgrad_three_streams.txt
python tools/timeline.py 20 2>/dev/null | tee gpurun_out/marks_now.txt
