#!/bin/bash
# PMC passes over tools/microbench_tiled.py (the B=8 full-resolution 32->32 layer: conv_tiled_kernel<2,...> and
# wgrad_tiled_kernel), counters only with --kernel-trace, the program directly after `--`:
#   pass f : FETCH_SIZE                         (HBM read bytes; doubled per MI355X_MICROARCH.md, gfx950 correction)
#   pass w : WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
#   pass s1: SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA
#   pass s2: SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU
#   pass g : GRBM_GUI_ACTIVE                    (effective clock = GRBM_GUI_ACTIVE / 8 / kernel time)
# Writes gpurun_out/pmc_traffic.json (copy it to profiles/pmc_traffic.json: bench.py quotes that file as roofline.traffic, with its `recorded` stamp) and prints the table for profiles/r02_pmc_tiled.md.
export TMPDIR=/tmp
out=gpurun_out/pmc_tiled
mkdir -p $out
python3 tools/microbench_tiled.py > $out/unprofiled.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d ${out}_f -o run --output-format csv -- python3 tools/microbench_tiled.py full > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d ${out}_w -o run --output-format csv -- python3 tools/microbench_tiled.py full > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA -d ${out}_s1 -o run --output-format csv -- python3 tools/microbench_tiled.py full > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU -d ${out}_s2 -o run --output-format csv -- python3 tools/microbench_tiled.py full > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d ${out}_g -o run --output-format csv -- python3 tools/microbench_tiled.py full > /dev/null 2>&1
python3 tools/pmc_tiled_report.py $out
