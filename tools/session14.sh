#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s14; mkdir -p $O; cd $R
timeout -k 10 500 python -m pytest tests/test_ddp_gpu.py tests/test_trainer_gpu.py tests/test_boundary_gpu.py -x -q -m gpu 2>&1 | tail -1
for i in 1 2 3; do for v in "" x1; do echo -n "DYCON_STUDENT_AFTER=$v  "; DYCON_STUDENT_AFTER=$v bash tools/variant_bench.sh dycon_paper_replication_amd/libdycon_hip.so; done; done 2>&1 | grep -v amdgpu.ids | tee $O/student_after.txt
timeout -k 10 300 python tools/ddp_overhead.py 2>&1 | grep "ms/step" | tee $O/ddp_overhead.txt
