#!/bin/bash
# Round-6 closing measurements on one MI355X: smoke, the driver's bench command, rocprofv3 kernel statistics (lanes / single stream),
# PMC passes of the dominant kernels and of the whole step, plan timings and lane timelines.  Everything lands in gpurun_out/r06_*.
python __graft_entry__.py --smoke 2>&1 | tail -2 > gpurun_out/r06_smoke.txt
python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
tools/gpu_prof.sh r06 > gpurun_out/r06_gpu_prof.log 2>&1
python tools/prof_db.py stats $(ls gpurun_out/prof_r06_lanes/run_results.db gpurun_out/prof_r06_lanes/*/run_results.db 2>/dev/null | head -1) 32 'Round 6 -- per-kernel device time of the default train step, lanes' > gpurun_out/r06_step_kernel_stats_lanes.md 2>&1
python tools/prof_db.py stats $(ls gpurun_out/prof_r06_single/run_results.db gpurun_out/prof_r06_single/*/run_results.db 2>/dev/null | head -1) 32 'Round 6 -- per-kernel device time of the default train step, single stream' > gpurun_out/r06_step_kernel_stats_single_stream.md 2>&1
tools/pmc_tiled.sh > gpurun_out/r06_pmc_tiled.txt 2>&1
tools/pmc_step.sh > gpurun_out/r06_pmc_step_traffic.txt 2>&1
python tools/plan_times.py > gpurun_out/r06_plan_times.txt 2>&1
python tools/lane_timeline.py > gpurun_out/r06_lane_timeline.txt 2>&1
python tools/main_lane_trace.py > gpurun_out/r06_main_lane_trace.txt 2>&1
cat gpurun_out/r06_smoke.txt
head -c 400 gpurun_out/r06_bench_default.json
