#!/bin/bash
# usage: tools/pmc_dcn_bwd.sh <tag>   (PMC passes over tools/bench_dcn.py for the one-pass backward; kernel-trace only)
tag=$1; shift
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
P() { n=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/pmc_${tag}_$n -o run --output-format csv -- python3 tools/bench_dcn.py > /dev/null 2>&1; }
P 1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
P 2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU
P 3 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
P 4 FETCH_SIZE WRITE_SIZE
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
out = collections.OrderedDict()
for n in (1, 2, 3, 4):
    for f in glob.glob("gpurun_out/pmc_%s_%d/**/*counter_collection.csv" % (tag, n), recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "dcn_bwd" not in k and "dcn_gradw_fused" not in k:
                continue
            a = acc[(k[:40], r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
        for (k, c), (v, cnt) in sorted(acc.items()):
            out[(k, c)] = v / max(cnt, 1)
for (k, c), v in out.items():
    print("%-42s %-28s %16.0f" % (k, c, v))
PY
