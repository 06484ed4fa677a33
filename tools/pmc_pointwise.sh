#!/bin/bash
# SQ counters of the pointwise kernels of the train step (fuse rows, up-sample adjoints, fan-in passes): two PMC passes over
# `bench.py --steps 4 --warmup 2` with every other leg switched off; per-kernel averages printed
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
A="--steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-forward --no-dcn --no-lidar --no-torch-gpu --no-other-models"
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD -d gpurun_out/pmc_pw_1 -o run --output-format csv -- python3 bench.py $A > /dev/null 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS -d gpurun_out/pmc_pw_2 -o run --output-format csv -- python3 bench.py $A > /dev/null 2>&1
python3 - <<'P'
import csv, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for n in (1, 2):
    for f in glob.glob("gpurun_out/pmc_pw_%d/**/*counter_collection.csv" % n, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not any(s in k for s in ("fuse_sum", "adjoint", "grad_combine", "tail", "fold")):
                continue
            a = acc[k[:48]][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in acc.items():
    g = lambda c: cs[c][0] / max(cs[c][1], 1)
    wc = g("SQ_WAVE_CYCLES")
    if not wc: continue
    print("%-50s waves %7.0f  wave-cycles(quad) %10.0f  active %4.1f%%  wait_any %4.1f%%  wait_inst %4.1f%%  valu %9.0f  vmem_rd %8.0f  lds %8.0f  vmem-level/wavecyc %.2f" % (
        k, g("SQ_WAVES"), wc, 100 * g("SQ_ACTIVE_INST_ANY") / wc, 100 * g("SQ_WAIT_ANY") / wc, 100 * g("SQ_WAIT_INST_ANY") / wc,
        g("SQ_INSTS_VALU"), g("SQ_INSTS_VMEM_RD"), g("SQ_INSTS_LDS"), g("SQ_INST_LEVEL_VMEM") / wc))
P
