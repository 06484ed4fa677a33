#!/bin/bash
# PMC counters of the stride-2 layers (tools/microbench_s2.py) on the round-3 LDS-tiled kernels and on the round-2 generic gather
# kernels (RTP_DISABLE_S2_FWD / _WGRAD): separate --pmc passes with --kernel-trace only, the program directly after `--`.
export TMPDIR=/tmp
o=gpurun_out/pmc_s2
mkdir -p $o
python3 tools/microbench_s2.py > $o/unprofiled_tiled.txt 2>&1
RTP_DISABLE_S2_FWD=1 RTP_DISABLE_S2_WGRAD=1 python3 tools/microbench_s2.py > $o/unprofiled_generic.txt 2>&1
for variant in tiled generic; do
  if [ $variant = generic ]; then export RTP_DISABLE_S2_FWD=1 RTP_DISABLE_S2_WGRAD=1; else unset RTP_DISABLE_S2_FWD RTP_DISABLE_S2_WGRAD; fi
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d ${o}_${variant}_f -o run --output-format csv -- python3 tools/microbench_s2.py pmc > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d ${o}_${variant}_w -o run --output-format csv -- python3 tools/microbench_s2.py pmc > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d ${o}_${variant}_s -o run --output-format csv -- python3 tools/microbench_s2.py pmc > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc TA_BUSY_avr GRBM_GUI_ACTIVE -d ${o}_${variant}_t -o run --output-format csv -- python3 tools/microbench_s2.py pmc > /dev/null 2>&1
done
python3 - <<'P'
import csv, glob, collections
for variant in ("tiled", "generic"):
    print("## %s kernels: %s" % (variant, open("gpurun_out/pmc_s2/unprofiled_%s.txt" % variant).read().strip().splitlines()[-1]))
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for f in glob.glob("gpurun_out/pmc_s2_%s_*/**/*counter_collection.csv" % variant, recursive=True) + glob.glob("gpurun_out/pmc_s2_%s_*/*counter_collection.csv" % variant):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:48]
            if not any(t in k for t in ("conv_s2_fwd", "wgrad_s2", "conv_igemm", "wgrad_kernel", "dgrad_s2")): continue
            a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, cs in sorted(agg.items()):
        row = {c: v[1] / v[0] for c, v in cs.items()}
        parts = []
        if "FETCH_SIZE" in row: parts.append("read %.1f MB (FETCH_SIZE x2)" % (row["FETCH_SIZE"] * 1024 * 2 / 1e6))
        if "WRITE_SIZE" in row: parts.append("write %.1f MB" % (row["WRITE_SIZE"] * 1024 / 1e6))
        for c in ("SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "TA_BUSY_avr", "GRBM_GUI_ACTIVE"):
            if c in row: parts.append("%s %.4g" % (c, row[c]))
        if "SQ_WAVE_CYCLES" in row and "SQ_WAIT_ANY" in row: parts.append("parked %.1f %%" % (100 * row["SQ_WAIT_ANY"] / row["SQ_WAVE_CYCLES"]))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in row and "SQ_BUSY_CYCLES" in row: parts.append("MFMA-busy/SQ-busy %.3f" % (row["SQ_VALU_MFMA_BUSY_CYCLES"] / row["SQ_BUSY_CYCLES"]))
        print("| `%s` | %s |" % (k, " | ".join(parts)))
P
