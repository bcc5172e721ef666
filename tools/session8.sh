#!/bin/bash
# round 3, GPU session 8: data-parallel run on four streams; split plan of conv_k3_tile with 64-channel k-steps
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s8; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_ddp_gpu.py tests/test_trainer_gpu.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 300 python tools/ddp_overhead.py 2>&1 | grep "ms/step" | tee $O/ddp_overhead.txt
for w in 512 384 256 768; do echo -n "DYCON_TILE_SPLIT_WGS=$w  "; DYCON_TILE_SPLIT_WGS=$w bash tools/variant_bench.sh dycon_paper_replication_amd/libdycon_hip.so; done 2>&1 | tee $O/tile_split.txt
