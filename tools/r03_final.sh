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
# DCN operator: per-kernel device time of tools/bench_dcn.py (one-pass backward) and the model-level variant
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r03f_dcn -o run -- python3 $GRAFT_REPO_ROOT/tools/bench_dcn.py > $GRAFT_REPO_ROOT/gpurun_out/r03_dcn_bench_prof.txt 2>/dev/null )
db=$(find gpurun_out/prof_r03f_dcn -name '*results.db' | head -1)
python3 tools/prof_db.py stats $db 10 "Round 3 -- DCN operator [128,32,64,160] -> 32 (3x3, deformable_groups 4, offsets 0.5 px sigma): 14 forward + 13 forward+backward calls of tools/bench_dcn.py (miopen / Cijk rows: torch's conv2d comparison)" > gpurun_out/r03_dcn_kernel_stats.md 2>&1
head -16 gpurun_out/r03_dcn_kernel_stats.md
timeout 600 python3 bench.py --model hr3d_dcn --steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu --no-lidar --no-dcn --no-forward --no-roofline --no-other-models > gpurun_out/r03_bench_hr3d_dcn.json 2>/dev/null
python3 -c "
import json; d=json.loads(open('gpurun_out/r03_bench_hr3d_dcn.json').read().strip().splitlines()[-1]); print('hr3d_dcn', d['value'], d['ms_per_step'])"
