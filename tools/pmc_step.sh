#!/bin/bash
# HBM bytes of a whole training step: FETCH_SIZE and WRITE_SIZE summed over every kernel of `bench.py --steps 6 --warmup 2`
# (only the train-step leg is active: every other bench leg, and the torch child process, is switched off)
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_step_f -o run --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-forward --no-dcn --no-lidar --no-torch-gpu --no-other-models > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_step_w -o run --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-forward --no-dcn --no-lidar --no-torch-gpu --no-other-models > /dev/null 2>&1
python3 - <<'P'
import csv, collections, json, time
totals = {}
for d, nm in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open("gpurun_out/pmc_step_%s/run_counter_collection.csv" % d)):
        if r["Counter_Name"] != nm: continue
        a = agg[r["Kernel_Name"][:50]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    tot = sum(v[1] for v in agg.values())
    totals[nm] = tot * 1024 / 8 * (2 if d == "f" else 1)   # bytes per step; FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM)
    print(nm, "total KiB over 8 steps: %.0f -> per step %.1f MB%s" % (tot, tot * 1024 / 8 / 1e6, " (x2 for reads: %.1f MB)" % (tot * 1024 / 8 / 1e6 * 2) if d == "f" else ""))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print("   %-52s calls %4d  %.1f MB/step" % (k, v[0], v[1] * 1024 / 8 / 1e6 * (2 if d == "f" else 1)))
json.dump({"recorded": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), "workload": "hr3d B=8 train step, 8 steps of bench.py (lane plan)",
           "read_bytes_per_step": totals["FETCH_SIZE"], "write_bytes_per_step": totals["WRITE_SIZE"],
           "bytes_per_step": totals["FETCH_SIZE"] + totals["WRITE_SIZE"],
           "source": "tools/pmc_step.sh: rocprofv3 --pmc FETCH_SIZE (x2) and --pmc WRITE_SIZE, separate passes, summed over every kernel"},
          open("gpurun_out/pmc_step_traffic.json", "w"), indent=1)
print("wrote gpurun_out/pmc_step_traffic.json (copy to profiles/r03_pmc_step_traffic.json)")
P
