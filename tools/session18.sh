#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 300 env DYCON_DEFER_WGRAD=block_nine,block_eight.conv.0 python -m pytest tests/test_trainer_gpu.py -x -q -m gpu 2>&1 | tail -1
for i in 1 2 3; do for v in "" "block_nine,block_eight.conv.3" "block_nine,block_eight.conv.0" "block_nine,block_seven.conv.6" "block_nine,block_six"; do echo -n "DYCON_DEFER_WGRAD=$v  "; DYCON_DEFER_WGRAD=$v bash tools/variant_bench.sh dycon_paper_replication_amd/libdycon_hip.so; done; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/defer_wgrad.txt
