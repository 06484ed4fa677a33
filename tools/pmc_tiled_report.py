"""Summarise the PMC passes of tools/pmc_tiled.sh: per kernel mean counter values per launch, derived MFMA-busy %, LDS
figures, HBM traffic (FETCH_SIZE doubled per MI355X_MICROARCH.md for 16-B-per-lane streaming reads on gfx950, + WRITE_SIZE;
counter unit KiB), effective clock.  Writes gpurun_out/pmc_traffic.json (the only directory that travels back from the GPU box); copy it to
profiles/pmc_traffic.json, which bench.py quotes together with the file's `recorded` stamp."""
import collections
import csv
import json
import os
import sys

out = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {"conv_tiled_full": "conv_tiled_kernel<2, true, 1, false", "wgrad_tiled": "wgrad_tiled_kernel"}
if os.environ.get("PMC_KERNELS"):   # "name=substring of the kernel name;..." (tools/pmc_conv64.sh)
    KERNELS = dict(it.split("=", 1) for it in os.environ["PMC_KERNELS"].split(";") if "=" in it)
agg = {k: collections.defaultdict(list) for k in KERNELS}
dur = {k: collections.defaultdict(list) for k in KERNELS}
for tag in ("f", "w", "s1", "s2", "g"):
    path = "%s_%s/run_counter_collection.csv" % (out, tag)
    if not os.path.exists(path):
        continue
    seen = set()
    for r in csv.DictReader(open(path)):
        for k, pat in KERNELS.items():
            if pat in r["Kernel_Name"]:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                if r["Dispatch_Id"] not in seen:
                    seen.add(r["Dispatch_Id"])
                    dur[k][tag].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
res = {}
for k in KERNELS:
    m = {c: sum(v[len(v) // 2:]) / max(1, len(v[len(v) // 2:])) for c, v in agg[k].items()}   # second half: clocks settled
    d = {t: sum(v[len(v) // 2:]) / max(1, len(v[len(v) // 2:])) for t, v in dur[k].items()}
    if not m:
        continue
    row = {"counters": {c: round(v, 1) for c, v in m.items()}, "avg_us_under_profiling": {t: round(v, 2) for t, v in d.items()}}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        row["hbm_read_MB"] = round(m["FETCH_SIZE"] * 1024 * 2 / 1e6, 1)
        row["hbm_write_MB"] = round(m["WRITE_SIZE"] * 1024 / 1e6, 1)
        row["hbm_bytes_per_launch"] = int(m["FETCH_SIZE"] * 1024 * 2 + m["WRITE_SIZE"] * 1024)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CYCLES" in m:
        # SQ_BUSY_CYCLES is summed over the SEs/XCDs it is sampled on; MFMA_BUSY over SIMDs: report the guide's ratio form and
        # the instruction-count form (N_mfma x 16 cycles per 16x16x32 MFMA per SIMD vs kernel time x clock x 1024 SIMDs)
        row["mfma_busy_over_sq_busy"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["SQ_BUSY_CYCLES"], 4)
    if "GRBM_GUI_ACTIVE" in m and "g" in d:
        row["effective_clock_GHz"] = round(m["GRBM_GUI_ACTIVE"] / 8 / (d["g"] * 1e3), 3)
    if "SQ_INSTS_MFMA" in m and "s1" in d:
        clk = row.get("effective_clock_GHz", 2.4)
        row["mfma_pipe_util_from_inst_count"] = round(m["SQ_INSTS_MFMA"] * 16 / (d["s1"] * 1e3 * clk * 1024), 4)
    if "SQ_LDS_BANK_CONFLICT" in m and "SQ_LDS_IDX_ACTIVE" in m:
        row["lds_conflict_frac"] = round(m["SQ_LDS_BANK_CONFLICT"] / max(1.0, m["SQ_LDS_IDX_ACTIVE"]), 4)
    if "SQ_WAVE_CYCLES" in m:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if c in m:
                row[c.lower() + "_frac_of_wave_cycles"] = round(m[c] / m["SQ_WAVE_CYCLES"], 4)
    res[k] = row
print(json.dumps(res, indent=1))
if os.path.exists(out + "/unprofiled.txt"):
    print(open(out + "/unprofiled.txt").read())
import time
tj = {"recorded": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), "source": "tools/pmc_tiled.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/microbench_tiled.py full, B=8 [8,16,64,160,32])",
      "bytes_per_launch": {k: v["hbm_bytes_per_launch"] for k, v in res.items() if "hbm_bytes_per_launch" in v}, "detail": res}
if tj["bytes_per_launch"] and not os.environ.get("PMC_KERNELS"):
    with open(os.path.join(ROOT, "gpurun_out", "pmc_traffic.json"), "w") as f:
        json.dump(tj, f, indent=1)
    print("wrote gpurun_out/pmc_traffic.json (copy to profiles/)")
