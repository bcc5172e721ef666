#!/bin/bash
# round 3, GPU session 10: full GPU suite at HEAD, then the round's measurement pass
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s10; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log
[ $rc -ne 0 ] && { tail -40 $O/pytest.log; exit $rc; }
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/measure_round.sh r03
cat gpurun_out/r03/bench_line.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_repeats')}, d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
print({k:r[k] for k in ('kernel','bound','frac','achieved','avg_launch_ms','launches_per_step','ms_per_step','traffic')}, r['self_check'], r['step'])
print('secondary', {k:r['secondary'][k] for k in ('kernel','bound','frac','avg_launch_ms','launches_per_step')} if r['secondary'] else None)
"
head -30 gpurun_out/r03/pmc_traffic.txt
cat gpurun_out/r03/step_marks.txt
head -12 gpurun_out/r03/timeline.txt
