#!/bin/bash
# round 3, GPU session 7: 64-channel k-steps in conv_k3_tile (A/B), hardware queues for the data-parallel run
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s7; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py "tests/test_fullsize_gpu.py::test_conv_full_size_bf16" tests/test_trainer_gpu.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for v in 0 1; do for shp in "128 128 12" "256 256 6" "128 128 14"; do echo -n "DYCON_TILE_KS64=$v  "; DYCON_TILE_KS64=$v python tools/conv_micro2.py $shp 50; done; done 2>&1 | grep -v amdgpu.ids | tee $O/tile_micro.txt
for i in 1 2; do for v in 0 1; do echo -n "DYCON_TILE_KS64=$v  "; DYCON_TILE_KS64=$v bash tools/variant_bench.sh dycon_paper_replication_amd/libdycon_hip.so; done; done 2>&1 | tee $O/variant_bench.txt
for q in 4 8; do echo "GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python tools/ddp_overhead.py 2>&1 | grep "ms/step"; done | tee $O/ddp_overhead.txt
