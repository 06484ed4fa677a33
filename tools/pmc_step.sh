#!/bin/bash
# HBM bytes of a whole training step: FETCH_SIZE and WRITE_SIZE summed over every kernel of `bench.py --steps 6 --warmup 2`
# (only the train-step leg is active: every other bench leg, and the torch child process, is switched off).  bench.py times THREE
# segments of --steps, so the run holds 2 + 3 * 6 = 20 steps.  torch's own kernels (zero fills at plan construction, the input
# upload's copies) are construction-time work, not steps: they are listed and kept OUT of the per-step figure.
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_step_f -o run --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-forward --no-dcn --no-lidar --no-torch-gpu --no-other-models > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_step_w -o run --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-forward --no-dcn --no-lidar --no-torch-gpu --no-other-models > /dev/null 2>&1
python3 - <<'P'
import csv, collections, json, time
totals, setup = {}, {}
NSTEP = 20
is_setup = lambda name: name.startswith("void at::native") or "elementwise_kernel" in name or "FillFunctor" in name
for d, nm in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open("gpurun_out/pmc_step_%s/run_counter_collection.csv" % d)):
        if r["Counter_Name"] != nm: continue
        a = agg[r["Kernel_Name"][:50]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    mul = 2 if d == "f" else 1   # FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM)
    tot = sum(v[1] for k, v in agg.items() if not is_setup(k))
    setup[nm] = sum(v[1] for k, v in agg.items() if is_setup(k)) * 1024 * mul
    totals[nm] = tot * 1024 / NSTEP * mul   # bytes per step
    print(nm, "KiB of the plan's kernels over %d steps: %.0f -> per step %.1f MB%s; torch set-up kernels (not steps): %.1f MB in all"
          % (NSTEP, tot, tot * 1024 / NSTEP / 1e6 * mul, " (reads x2)" if d == "f" else "", setup[nm] / 1e6))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
        print("   %-52s calls %4d  %.1f MB/step%s" % (k, v[0], v[1] * 1024 / NSTEP / 1e6 * mul, "   [set-up, excluded]" if is_setup(k) else ""))
json.dump({"recorded": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), "workload": "hr3d B=8 train step, 20 steps of bench.py (lane plan); torch's construction-time kernels excluded",
           "setup_bytes_excluded": setup.get("FETCH_SIZE", 0) + setup.get("WRITE_SIZE", 0),
           "read_bytes_per_step": totals["FETCH_SIZE"], "write_bytes_per_step": totals["WRITE_SIZE"],
           "bytes_per_step": totals["FETCH_SIZE"] + totals["WRITE_SIZE"],
           "source": "tools/pmc_step.sh: rocprofv3 --pmc FETCH_SIZE (x2) and --pmc WRITE_SIZE, separate passes, summed over every kernel"},
          open("gpurun_out/pmc_step_traffic.json", "w"), indent=1)
print("wrote gpurun_out/pmc_step_traffic.json (copy to profiles/r04_pmc_step_traffic.json)")
P
