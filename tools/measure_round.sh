#!/bin/bash
# Everything the round's measured claims rest on, in one pass on the GPU box (tools/measure_round.sh <tag>, e.g. r03):
#   gpurun_out/<tag>/bench_line.json      the default bench run (what the driver runs)
#   gpurun_out/<tag>/prof/*               rocprofv3 --kernel-trace --stats of a short bench run
#   gpurun_out/pmc_<tag>_{fetch,write}/   the two HBM-traffic counter passes
tag=${1:-r03}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$tag
cd $R && python bench.py > gpurun_out/$tag/bench_line.json 2> gpurun_out/$tag/bench_line.err || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$tag/prof -o run --output-format csv -- python3 $R/bench.py --steps 8 --warmup 4 --repeats 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/$tag/prof.log 2>&1 || exit 1
cd $R && tools/pmc_traffic.sh $tag || exit 1
python tools/pmc_traffic.py $tag gpurun_out/$tag/pmc_traffic.json gpurun_out/$tag/bench_line.json 3 > gpurun_out/$tag/pmc_traffic.txt
python profiles/analyze_trace.py gpurun_out/$tag/prof/run_kernel_trace.csv > gpurun_out/$tag/step_by_kernel.txt
python profiles/timeline.py gpurun_out/$tag/prof/run_kernel_trace.csv > gpurun_out/$tag/timeline.txt
python tools/timeline.py 20 > gpurun_out/$tag/step_marks.txt 2>/dev/null
