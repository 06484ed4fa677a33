#!/bin/bash
# round-4 baseline on a fresh box: default bench (short), main-lane trace, per-launch stand-alone times
export TMPDIR=/tmp
tag=${1:-r04_base}
timeout 600 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-torch-gpu --no-lidar --no-dcn --no-forward --no-other-models > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
python3 -c "
import json; d=json.loads(open('gpurun_out/${tag}_bench.json').read().strip().splitlines()[-1]); print('BENCH', d['value'], d['ms_per_step'], d['roofline']['frac']); [print(k, v) for k, v in d['roofline']['families'].items()]"
timeout 600 python3 tools/main_lane_trace.py hr3d > gpurun_out/${tag}_trace.txt 2>&1; head -30 gpurun_out/${tag}_trace.txt
timeout 600 python3 tools/plan_times.py --convs --top 40 > gpurun_out/${tag}_plan_times.txt 2>&1; head -70 gpurun_out/${tag}_plan_times.txt
