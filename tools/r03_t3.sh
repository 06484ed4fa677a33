#!/bin/bash
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-roofline --no-forward --no-dcn --no-lidar --no-torch-gpu --steps 20 --warmup 5"
timeout 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "stride2_forward or test_wgrad" 2>&1 | tail -15
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -x -q 2>&1 | tail -3
for m in hr3d hr3d_one_hm_doppler hr3d_one_hm_doppler_phase; do
  timeout 300 python3 bench.py $Q --model $m 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', d['value'], d['ms_per_step'], d.get('final_loss'))"
done
RTP_DISABLE_S2_WGRAD=1 timeout 300 python3 bench.py $Q --model hr3d 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hr3d no-s2-wgrad', d['value'], d['ms_per_step'], d.get('final_loss'))"
timeout 600 python3 tools/plan_times.py --model hr3d --convs --top 30 > gpurun_out/r03_plan_hr3d_c.txt 2>&1
grep -E " s2 " gpurun_out/r03_plan_hr3d_c.txt | cut -c1-170; head -20 gpurun_out/r03_plan_hr3d_c.txt
