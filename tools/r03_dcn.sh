#!/bin/bash
# round 3: the one-pass DCN backward -- parity, operator timing A/B, model-level A/B
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_dcn.py -x -q -k "one_pass" 2>&1 | tail -15 > gpurun_out/r03_dcn_tests.txt
timeout 1200 python3 -m pytest tests/test_gpu_dcn.py tests/test_gpu_dcn_head.py tests/test_gpu_dcn_binding.py -q 2>&1 | tail -15 >> gpurun_out/r03_dcn_tests.txt
{
for segs in 0 1 2 3 4; do
  echo "== fused backward, RTP_DCN_SEGS=$segs"; RTP_DCN_SEGS=$segs timeout 300 python3 tools/bench_dcn.py 2>&1 | head -1
done
echo "== column route (RTP_DCN_NO_FUSED_BWD=1)"; RTP_DCN_NO_FUSED_BWD=1 timeout 300 python3 tools/bench_dcn.py 2>&1 | head -1
echo "== column route without the fused weight gradient"; RTP_DCN_NO_FUSED_BWD=1 RTP_DCN_NO_FUSED_GRADW=1 timeout 300 python3 tools/bench_dcn.py 2>&1 | head -1
} > gpurun_out/r03_dcn_bench.txt 2>&1
B="--model hr3d_dcn --steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu --no-lidar --no-dcn --no-forward --no-roofline --no-other-models"
timeout 600 python3 bench.py $B > gpurun_out/r03_dcn_model_fused.json 2> gpurun_out/r03_dcn_model_fused.err
RTP_DCN_NO_FUSED_BWD=1 timeout 600 python3 bench.py $B > gpurun_out/r03_dcn_model_cols.json 2> gpurun_out/r03_dcn_model_cols.err
cat gpurun_out/r03_dcn_tests.txt gpurun_out/r03_dcn_bench.txt
python3 - <<'PY'
import json
for n in ("fused", "cols"):
    try:
        d = json.loads(open("gpurun_out/r03_dcn_model_%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"])
    except Exception as e:
        print(n, "failed", e)
PY
