#!/bin/bash
# full GPU suite, smoke, then the round's measurement pass
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/s22_pytest.txt 2>&1; rc=$?; tail -5 gpurun_out/s22_pytest.txt; [ $rc -eq 0 ] || exit 1
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 || exit 1
