#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 400 python -m pytest tests/test_trainer_gpu.py tests/test_ddp_gpu.py -x -q -m gpu 2>&1 | tail -1
for i in 1 2 3; do for v in "0 0" "1 0" "0 1" "1 1"; do set -- $v; echo -n "TEACHER_HEAD_STREAM=$1 TEACHER_SPLIT_PACK=$2  "; DYCON_TEACHER_HEAD_STREAM=$1 DYCON_TEACHER_SPLIT_PACK=$2 bash tools/variant_bench.sh dycon_paper_replication_amd/libdycon_hip.so; done; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/teacher_head_stream.txt
