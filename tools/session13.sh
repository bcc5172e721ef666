#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 400 python -m pytest tests/test_trainer_gpu.py -x -q -m gpu 2>&1 | tail -1
for i in 1 2 3; do for v in 0 1; do echo -n "DYCON_FECL_EARLY=$v  "; DYCON_FECL_EARLY=$v bash tools/variant_bench.sh dycon_paper_replication_amd/libdycon_hip.so; done; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/fecl_early.txt
for i in 1 2; do for v in "" feat_bwd; do echo -n "DYCON_ABLATE=$v  "; DYCON_ABLATE=$v bash tools/variant_bench.sh dycon_paper_replication_amd/libdycon_hip.so; done; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ablate_feat_bwd.txt
