#!/bin/bash
# per-kernel table of the U-Net step (rocprofv3 --kernel-trace)
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/s31_prof -o run --output-format csv -- python3 $R/bench.py --model unet_3D --steps 8 --warmup 4 --repeats 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/s31_prof.log 2>&1 || exit 1
cd $R
python profiles/analyze_trace.py gpurun_out/s31_prof/run_kernel_trace.csv > gpurun_out/s31_unet_by_kernel.txt
python profiles/timeline.py gpurun_out/s31_prof/run_kernel_trace.csv > gpurun_out/s31_unet_timeline.txt
head -50 gpurun_out/s31_unet_by_kernel.txt
rm -rf gpurun_out/s31_prof/run_kernel_trace.csv
