#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
{
for segs in 1 2 3 4; do
  echo "== fused backward, RTP_DCN_SEGS=$segs"; RTP_DCN_SEGS=$segs timeout 300 python3 tools/bench_dcn.py 2>/dev/null | head -1
done
echo "== column route (RTP_DCN_NO_FUSED_BWD=1)"; RTP_DCN_NO_FUSED_BWD=1 timeout 300 python3 tools/bench_dcn.py 2>/dev/null | head -1
echo "== column route without the fused weight gradient"; RTP_DCN_NO_FUSED_BWD=1 RTP_DCN_NO_FUSED_GRADW=1 timeout 300 python3 tools/bench_dcn.py 2>/dev/null | head -1
} > gpurun_out/r03_dcn_bench.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_dcn -o run -- python3 $GRAFT_REPO_ROOT/tools/bench_dcn.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
cat gpurun_out/r03_dcn_bench.txt
f=$(find gpurun_out/prof_dcn -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-200
