#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
for m in hr3d_one_hm_doppler hr3d_one_hm_doppler_phase; do
  timeout 600 python3 tools/plan_times.py --model $m 2>/dev/null > gpurun_out/r03_plan_times_$m.txt
  head -60 gpurun_out/r03_plan_times_$m.txt
done
