#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests/test_gpu_engine.py tests/test_gpu_native.py -x -q 2>&1 | tail -4
B="--steps 60 --warmup 5 --no-cpu-baseline --no-torch-gpu --no-lidar --no-dcn --no-forward --no-roofline"
timeout 900 python3 bench.py $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:v['value'] for k,v in d['other_models'].items()})"
