#!/bin/bash
# round 3, GPU session 5: register-resident small-level norms + 32x32x16 form of the persistent 48^3 kernel, each A/B'd
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s5; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_trainer_gpu.py tests/test_engine_gpu.py tests/test_ddp_gpu.py "tests/test_fullsize_gpu.py::test_conv_full_size_bf16" tests/test_fullsize_gpu.py::test_conv_takes_norm_statistics -x -q -m gpu > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for v in 0 1; do echo "DYCON_P32X=$v"; DYCON_P32X=$v python tools/conv_micro.py 32 32 48 100; done 2>&1 | grep -v amdgpu.ids | tee $O/p32x_micro.txt
for lib in build_variants/lib_base.so dycon_paper_replication_amd/libdycon_hip.so; do
  echo "=== $lib" >> $O/op_micro.txt
  DYCON_LIB=$PWD/$lib timeout -k 10 200 python tools/op_micro.py 50 2>/dev/null | grep -E "norm_(fwd|bwd) GN (64|128|256)" >> $O/op_micro.txt
done
cat $O/op_micro.txt
for i in 1 2; do bash tools/variant_bench.sh build_variants/lib_base.so dycon_paper_replication_amd/libdycon_hip.so; DYCON_P32X=0 bash tools/variant_bench.sh dycon_paper_replication_amd/libdycon_hip.so; done 2>&1 | tee $O/variant_bench.txt
timeout -k 10 600 python tools/ab.py "" fuse_finish=True conv_stats=True --reps 2 --steps 100 2>&1 | grep "ms/step" | tee $O/ab.txt
timeout -k 10 300 python tools/ddp_overhead.py 2>&1 | grep "ms/step" | tee $O/ddp_overhead.txt
