#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for i in 1 2; do for v in "" x1 x2 x3; do echo -n "DYCON_STUDENT_AFTER=$v  "; DYCON_STUDENT_AFTER=$v timeout -k 10 200 bash tools/variant_bench.sh dycon_paper_replication_amd/libdycon_hip.so; done; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/student_after.txt
