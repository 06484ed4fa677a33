#!/bin/bash
# usage: tools/pmc_dcn.sh <tag> <channels> [fwd]   (three PMC passes over tools/bench_dcn.py, kernel-trace only)
tag=$1; shift
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d gpurun_out/pmc_${tag}_1 -o run --output-format csv -- python3 tools/bench_dcn.py "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU -d gpurun_out/pmc_${tag}_2 -o run --output-format csv -- python3 tools/bench_dcn.py "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL -d gpurun_out/pmc_${tag}_3 -o run --output-format csv -- python3 tools/bench_dcn.py "$@" > /dev/null 2>&1
ls gpurun_out/pmc_${tag}_1 gpurun_out/pmc_${tag}_3
