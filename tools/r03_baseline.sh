#!/bin/bash
# round-3 baseline diagnostics on the GPU box: per-launch serial times of three configs, the step timeline, the default bench
export TMPDIR=/tmp
mkdir -p gpurun_out
for m in hr3d hr3d_one_hm_doppler hr3d_one_hm_doppler_phase; do
  timeout 600 python3 tools/plan_times.py --model $m --convs --top 100 > gpurun_out/r03_plan_$m.txt 2>&1
done
timeout 900 tools/gpu_prof.sh r03a
python3 tools/prof_db.py step gpurun_out/prof_r03a_lanes/*/run_results.db --full > gpurun_out/r03a_step_lanes.txt 2>&1 || \
python3 tools/prof_db.py step $(find gpurun_out/prof_r03a_lanes -name '*results.db' | head -1) --full > gpurun_out/r03a_step_lanes.txt 2>&1
timeout 900 python3 bench.py > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err
tail -c 600 gpurun_out/r03a_bench.json
