#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 900 python tools/ab.py "" "teacher_after='x1'" "teacher_after='x2'" "teacher_after='x3'" "teacher_after='x4'" "teacher_after='x5'" --reps 2 --steps 100 2>&1 | grep "ms/step" | tee gpurun_out/teacher_after.txt
