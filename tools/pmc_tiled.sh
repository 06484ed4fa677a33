#!/bin/bash
# PMC passes over tools/microbench_tiled.py (kernel-trace only): HBM fetch / write bytes and L2 hit rate of the tiled kernels
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_tiled_f -o run --output-format csv -- python3 tools/microbench_tiled.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d gpurun_out/pmc_tiled_w -o run --output-format csv -- python3 tools/microbench_tiled.py > /dev/null 2>&1
python3 - <<'P'
import csv, collections
for d in ("f", "w"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open("gpurun_out/pmc_tiled_%s/run_counter_collection.csv" % d)):
        agg[(r["Kernel_Name"][:60], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "tiled" in k[0]:
            print(k, {c: round(sum(x) / len(x), 1) for c, x in v.items()}, len(list(v.values())[0]))
P
