#!/bin/bash
# PMC passes over tools/microbench_conv64.py (64 -> 64, B = 8: conv64_kernel<false> forward + data gradient at [16,64,160]);
# the same counter sets and report as tools/pmc_tiled.sh.  Counters only with --kernel-trace, the program directly after `--`.
export TMPDIR=/tmp
out=gpurun_out/pmc_conv64
mkdir -p $out
python3 tools/microbench_conv64.py > $out/unprofiled.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d ${out}_f -o run --output-format csv -- python3 tools/microbench_conv64.py full > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d ${out}_w -o run --output-format csv -- python3 tools/microbench_conv64.py full > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA -d ${out}_s1 -o run --output-format csv -- python3 tools/microbench_conv64.py full > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU -d ${out}_s2 -o run --output-format csv -- python3 tools/microbench_conv64.py full > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d ${out}_g -o run --output-format csv -- python3 tools/microbench_conv64.py full > /dev/null 2>&1
PMC_KERNELS="conv64=conv64_kernel<false>" python3 tools/pmc_tiled_report.py $out
