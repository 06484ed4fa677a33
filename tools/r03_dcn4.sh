#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests/test_gpu_dcn.py tests/test_gpu_dcn_head.py tests/test_gpu_dcn_binding.py -q 2>&1 | tail -6 > gpurun_out/r03_dcn_tests.txt
B="--model hr3d_dcn --steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu --no-lidar --no-dcn --no-forward --no-roofline --no-other-models"
timeout 600 python3 bench.py $B > gpurun_out/r03_dcn_model_fused.json 2> gpurun_out/r03_dcn_model_fused.err
RTP_DCN_NO_FUSED_BWD=1 timeout 600 python3 bench.py $B > gpurun_out/r03_dcn_model_cols.json 2> gpurun_out/r03_dcn_model_cols.err
RTP_DCN_NO_FUSED_BWD=1 RTP_DCN_NO_FUSED_GRADW=1 timeout 600 python3 bench.py $B > gpurun_out/r03_dcn_model_r2.json 2> gpurun_out/r03_dcn_model_r2.err
timeout 300 python3 tools/bench_dcn.py 2>/dev/null | head -1 > gpurun_out/r03_dcn_bench.txt
timeout 600 python3 tools/plan_times.py --model hr3d_dcn 2>/dev/null | grep -i "dcn\|adapt\|total" | head -20 > gpurun_out/r03_dcn_plan_times.txt
cat gpurun_out/r03_dcn_tests.txt gpurun_out/r03_dcn_bench.txt gpurun_out/r03_dcn_plan_times.txt
python3 - <<'PY'
import json
for n in ("fused", "cols", "r2"):
    try:
        d = json.loads(open("gpurun_out/r03_dcn_model_%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"])
    except Exception as e:
        print(n, "failed", e)
PY
