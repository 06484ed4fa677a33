#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_prof.sh <tag> [bench args]  -> gpurun_out/prof_<tag>_{single,lanes}/run_results.db
tag=$1; shift
export TMPDIR=/tmp
RTP_PLAN="lanes=0,0,0,0,0,0" rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_single -o run -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-forward --no-dcn --no-lidar --no-torch-gpu --no-other-models "$@" > gpurun_out/prof_${tag}_single.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_lanes -o run -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-forward --no-dcn --no-lidar --no-torch-gpu --no-other-models "$@" > gpurun_out/prof_${tag}_lanes.log 2>&1
grep -h '"metric"' gpurun_out/prof_${tag}_lanes.log | cut -c1-160
