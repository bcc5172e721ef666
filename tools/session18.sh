#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 300 env DYCON_DEFER_WGRAD=block_nine,block_eight.conv.0 python -m pytest tests/test_trainer_gpu.py -x -q -m gpu 2>&1 | tail -40
