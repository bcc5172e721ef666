#!/bin/bash
# step time of the bench configuration over the split-plan tunables (DYCON_TILE_SPLIT_WGS, DYCON_WGRAD_SLAB_MB)
for t in 768 512 384 256; do
  echo -n "TILE_SPLIT_WGS=$t  "; DYCON_TILE_SPLIT_WGS=$t python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'ms')"
done
for m in 24 12 6 48; do
  echo -n "WGRAD_SLAB_MB=$m  "; DYCON_WGRAD_SLAB_MB=$m python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'ms')"
done
