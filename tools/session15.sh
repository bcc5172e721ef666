#!/bin/bash
# round 3: the measurement pass on the final code + secondary configurations
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s15; mkdir -p $O; cd $R
bash tools/measure_round.sh r03
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03/bench_line.json")); r = d["roofline"]
print({k: d[k] for k in ("value", "ms_per_step", "ms_per_step_repeats")}, d["cpu_baseline"]["value"])
print({k: r[k] for k in ("kernel", "bound", "frac", "avg_launch_ms", "launches_per_step", "ms_per_step", "traffic")}, r["self_check"], r["step"])
PY
cat gpurun_out/r03/step_marks.txt; head -8 gpurun_out/r03/timeline.txt
B="python bench.py --no-cpu-baseline --no-kernel-timing --steps 50 --warmup 10"
( $B --model unet_3D; $B --patch 112 112 96; $B --patch 112 112 80 --feature-scaler 4; $B --dtype f32 --steps 20 ) 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(round(d['value'], 1), 'vol/s', round(d['ms_per_step'], 3), 'ms', d['config']['workload'])
" | tee $O/secondary.txt
timeout -k 10 400 python tools/soak.py 1500 2>/dev/null | tail -4 | tee $O/soak.txt
