#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s11; mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_fullsize_gpu.py::test_engine_with_producer_side_statistics -x -q -m gpu 2>&1 | tail -2
for v in 0 1; do for shp in "64 64 24" "128 128 12" "256 256 6"; do echo -n "DYCON_WGRAD_W8=$v  "; DYCON_WGRAD_W8=$v python tools/wgrad_micro.py $shp 30; done; done 2>&1 | grep -v amdgpu.ids | tee $O/wgrad_w8_micro.txt
for i in 1 2 3; do for v in 0 1; do echo -n "DYCON_WGRAD_W8=$v  "; DYCON_WGRAD_W8=$v bash tools/variant_bench.sh dycon_paper_replication_amd/libdycon_hip.so; done; done 2>&1 | tee $O/variant_bench.txt
