#!/bin/bash
# HBM traffic of the bench step from the L2 memory-side counters (run ON the GPU box): two separate rocprofv3 --pmc passes
# (FETCH_SIZE, WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md "rocprofv3 PMC slots"), --kernel-trace only.
# usage: tools/pmc_traffic.sh <out-tag>      -> gpurun_out/pmc_<tag>_{fetch,write}/run_counter_collection.csv
tag=${1:-traffic}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  d=$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_${tag}_$d -o run --output-format csv -- \
    python3 $R/bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/pmc_${tag}_$d.log 2>&1 || exit 1
done
