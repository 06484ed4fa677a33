#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_dcn.py tests/test_gpu_dcn_head.py -x -q -k "one_pass or head" 2>&1 | tail -5 > gpurun_out/r03_dcn_tests.txt
{
for sc in 0.5 0.2; do
  RTP_BENCH_DCN_OFF_SCALE=$sc timeout 300 python3 tools/bench_dcn.py 2>/dev/null | head -1
  RTP_DCN_FP32_MFMA=1 RTP_BENCH_DCN_OFF_SCALE=$sc timeout 300 python3 tools/bench_dcn.py 2>/dev/null | head -1
done
} > gpurun_out/r03_dcn_bench.txt 2>&1
B="--model hr3d_dcn --steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu --no-lidar --no-dcn --no-forward --no-roofline --no-other-models"
timeout 600 python3 bench.py $B > gpurun_out/r03_dcn_model_fused.json 2> gpurun_out/r03_dcn_model_fused.err
cat gpurun_out/r03_dcn_tests.txt gpurun_out/r03_dcn_bench.txt
python3 - <<'PY'
import json
for n in ("fused",):
    try:
        d = json.loads(open("gpurun_out/r03_dcn_model_%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"])
    except Exception as e:
        print(n, "failed", e)
PY
