#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_dcn.py -x -q -k "one_pass" 2>&1 | tail -3
for sc in 0.5 0.2; do
  RTP_BENCH_DCN_OFF_SCALE=$sc timeout 300 python3 tools/bench_dcn.py 2>/dev/null | head -1
done
