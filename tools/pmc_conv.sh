#!/bin/bash
# SQ counter passes for one conv_micro configuration (run ON the GPU box): tools/pmc_conv.sh <tag> <Cin> <Cout> <S>
# (rocprofv3 --pmc with --kernel-trace only; program directly after `--`)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS \
  -d $R/gpurun_out/pmc_${tag}_a -o run --output-format csv -- python3 $R/tools/${MICRO:-conv_micro.py} "$@" ${REPS:-5} > $R/gpurun_out/pmc_${tag}_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU \
  -d $R/gpurun_out/pmc_${tag}_b -o run --output-format csv -- python3 $R/tools/${MICRO:-conv_micro.py} "$@" ${REPS:-5} > $R/gpurun_out/pmc_${tag}_b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_WAVE32_LDS \
  -d $R/gpurun_out/pmc_${tag}_c -o run --output-format csv -- python3 $R/tools/${MICRO:-conv_micro.py} "$@" ${REPS:-5} > $R/gpurun_out/pmc_${tag}_c.log 2>&1
true
