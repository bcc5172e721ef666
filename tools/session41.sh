#!/bin/bash
# chunk plan of the two-stage norm statistics (DYCON_NORM_CHUNKS: partial workgroups per sample; DYCON_NORM_MIN_ROWS), step A/B
R=$GRAFT_REPO_ROOT; cd $R
for i in 1 2; do for c in "512 64" "256 128" "256 256" "128 128" "320 128" "256 192"; do set -- $c; echo -n "DYCON_NORM_CHUNKS=$1 MIN_ROWS=$2  "; DYCON_NORM_CHUNKS=$1 DYCON_NORM_MIN_ROWS=$2 timeout -k 10 200 bash tools/variant_bench.sh dycon_paper_replication_amd/libdycon_hip.so; done; done 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/s41_norm_chunks.txt
