#!/bin/bash
# round 3, GPU session 2: full GPU suite on the new event / step_losses / ktimer code, bench line, DDP overhead
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s2; mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=15 > $O/pytest.log 2>&1; rc=$?
tail -25 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py > $O/bench_line.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/s2/bench_line.json"))
r = d["roofline"]
print({k: d[k] for k in ("value", "ms_per_step", "ms_per_step_repeats")}, d["cpu_baseline"]["value"])
print("roofline:", {k: r[k] for k in ("kernel", "bound", "frac", "avg_launch_ms", "launches_per_step", "ms_per_step", "traffic")}, r["self_check"], r["step"])
for k, v in list(r["per_kernel"].items())[:40]:
    print(f"  {v['ms_per_step']:7.4f} ms  n {v['launches_per_step']:3d}  avg {v['avg_launch_us']:7.2f} us  hbm {v['hbm_frac']}  mfma {v['mfma_frac']}  {k}  {v['entry_points']}")
PY
timeout -k 10 300 python tools/ddp_overhead.py > $O/ddp_overhead.txt 2>&1; cat $O/ddp_overhead.txt | grep -v amdgpu.ids
