#!/bin/bash
# step time of the bench configuration for each diagnostic build in build_variants/ (fresh process each; DYCON_LIB selects the build)
for lib in "$@"; do
  echo -n "$lib  "
  DYCON_LIB=$PWD/$lib python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'ms/step', d['config']['final_loss'])"
done
