#!/bin/bash
# round-3 closing run on the GPU box: full GPU test suite, kernel statistics (single stream + lanes), PMC for the tiled kernels, default bench
export TMPDIR=/tmp
timeout 3400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r03_gpu_tests.txt; cat gpurun_out/r03_gpu_tests.txt
timeout 900 tools/gpu_prof.sh r03f
for k in single lanes; do
  db=$(find gpurun_out/prof_r03f_$k -name '*results.db' | head -1)
  python3 tools/prof_db.py stats $db 12 "Round 3 (final build) -- per-kernel device time of the default train step, $k" > gpurun_out/r03_step_kernel_stats_$k.md 2>&1
  python3 tools/prof_db.py tiled $db 12 >> gpurun_out/r03_step_kernel_stats_$k.md 2>&1
done
timeout 900 tools/pmc_tiled.sh > gpurun_out/r03_pmc_tiled.txt 2>&1; tail -5 gpurun_out/r03_pmc_tiled.txt
timeout 900 tools/pmc_s2.sh > gpurun_out/r03_pmc_s2.txt 2>&1; tail -12 gpurun_out/r03_pmc_s2.txt
timeout 600 python3 tools/main_lane_trace.py hr3d > gpurun_out/r03_main_lane_trace.txt 2>&1
timeout 1200 python3 bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; python3 -c "
import json; d=json.loads(open('gpurun_out/r03_bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('other_models'), d['cpu_baseline']['value'], d.get('dcn_op'))"
