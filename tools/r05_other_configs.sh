#!/bin/bash
# Where the other shipped configs spend their step: per-launch serial cost with the plan's labels, and the lane timeline.
for m in hr3d_one_hm_doppler hr3d_one_hm_doppler_phase; do
  python tools/plan_times.py --model $m --top 50 > gpurun_out/r05_plan_times_$m.txt 2>&1
  python tools/lane_timeline.py $m > gpurun_out/r05_lane_timeline_$m.txt 2>&1
done
tail -5 gpurun_out/r05_plan_times_hr3d_one_hm_doppler_phase.txt
