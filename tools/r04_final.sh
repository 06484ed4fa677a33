#!/bin/bash
# round-4 closing run on the GPU box: kernel statistics (single stream + lanes), PMC for the tiled kernels and the whole step,
# main-lane trace, per-launch stand-alone times, in-kernel cycle stamps, default bench.  (The GPU test suite runs in its own call.)
export TMPDIR=/tmp
tag=r04
timeout 900 tools/gpu_prof.sh ${tag}f
for k in single lanes; do   # (32 = 2 warm-up + 3 segments x 10 steps)
  db=$(find gpurun_out/prof_${tag}f_$k -name '*results.db' | head -1)
  python3 tools/prof_db.py stats $db 32 "Round 4 (final build) -- per-kernel device time of the default train step, $k" > gpurun_out/${tag}_step_kernel_stats_$k.md 2>&1
  python3 tools/prof_db.py tiled $db 32 >> gpurun_out/${tag}_step_kernel_stats_$k.md 2>&1
done
timeout 900 tools/pmc_tiled.sh > gpurun_out/${tag}_pmc_tiled.txt 2>&1; tail -5 gpurun_out/${tag}_pmc_tiled.txt
timeout 900 tools/pmc_step.sh > gpurun_out/${tag}_pmc_step_traffic.txt 2>&1; tail -3 gpurun_out/${tag}_pmc_step_traffic.txt
timeout 600 python3 tools/main_lane_trace.py hr3d > gpurun_out/${tag}_main_lane_trace.txt 2>&1
timeout 600 python3 tools/lane_timeline.py hr3d > gpurun_out/${tag}_lane_timeline.txt 2>&1
timeout 600 python3 tools/plan_times.py --convs --order --top 30 > gpurun_out/${tag}_plan_times.txt 2>&1
RTP_LIB=rt_pose_amd/lib/librtp_hip_prof.so timeout 300 python3 tools/tiled_prof.py > gpurun_out/${tag}_tiled_prof.txt 2>&1
RTP_LIB=rt_pose_amd/lib/librtp_hip_wgtprof.so timeout 300 python3 tools/wgt_prof.py > gpurun_out/${tag}_wgt_prof.txt 2>&1
timeout 1500 python3 bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err; python3 -c "
import json; d=json.loads(open('gpurun_out/${tag}_bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('segments_ms_per_step'), d['roofline']['frac'], d.get('other_models'), d['cpu_baseline']['value'], d.get('dcn_op'))"
timeout 600 python3 bench.py --model hr3d_dcn --steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu --no-lidar --no-dcn --no-forward --no-roofline --no-other-models > gpurun_out/${tag}_bench_hr3d_dcn.json 2>/dev/null
python3 -c "
import json; d=json.loads(open('gpurun_out/${tag}_bench_hr3d_dcn.json').read().strip().splitlines()[-1]); print('hr3d_dcn', d['value'], d['ms_per_step'])"
