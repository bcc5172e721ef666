#!/bin/bash
# round 3, GPU session 4: register-resident small-level norms (A/B against the library of the previous commit), DDP issue stream,
# fuse_finish / conv_stats re-tested on top of the new norms
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s4; mkdir -p $O
cd $R
timeout -k 10 700 python -m pytest tests/test_ops_gpu.py tests/test_trainer_gpu.py tests/test_engine_gpu.py tests/test_ddp_gpu.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for lib in build_variants/lib_base.so dycon_paper_replication_amd/libdycon_hip.so; do
  echo "=== $lib" >> $O/op_micro.txt
  DYCON_LIB=$PWD/$lib timeout -k 10 200 python tools/op_micro.py 50 2>/dev/null | grep -E "norm_(fwd|bwd) GN (64|128|256)" >> $O/op_micro.txt
done
cat $O/op_micro.txt
for i in 1 2; do bash tools/variant_bench.sh build_variants/lib_base.so dycon_paper_replication_amd/libdycon_hip.so; done 2>&1 | tee $O/variant_bench.txt
timeout -k 10 600 python tools/ab.py "" fuse_finish=True conv_stats=True --reps 2 --steps 100 2>&1 | grep "ms/step" | tee $O/ab.txt
timeout -k 10 300 python tools/ddp_overhead.py 2>&1 | grep "ms/step" | tee $O/ddp_overhead.txt
