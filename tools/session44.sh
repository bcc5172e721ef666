#!/bin/bash
# re-sweep of the split plans on the final kernels: DYCON_TILE_SPLIT_WGS (conv_k3_tile), DYCON_WGRAD_SLAB_MB (weight-gradient slab cap)
R=$GRAFT_REPO_ROOT; cd $R
for i in 1 2; do
  for v in 512 384 640 256; do echo -n "DYCON_TILE_SPLIT_WGS=$v  "; DYCON_TILE_SPLIT_WGS=$v timeout -k 10 200 bash tools/variant_bench.sh dycon_paper_replication_amd/libdycon_hip.so; done
  for v in 24 16 32 48; do echo -n "DYCON_WGRAD_SLAB_MB=$v  "; DYCON_WGRAD_SLAB_MB=$v timeout -k 10 200 bash tools/variant_bench.sh dycon_paper_replication_amd/libdycon_hip.so; done
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s44_split_plans.txt
