#!/bin/bash
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-roofline --no-forward --no-dcn --no-lidar --no-torch-gpu --steps 20 --warmup 5"
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "wide_convs or conv_forward or wgrad or transposed or split_conv" 2>&1 | tail -15
for m in hr3d_one_hm_doppler_phase hr3d; do
  python3 bench.py $Q --model $m 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', d['value'], d['ms_per_step'], d.get('final_loss'))"
  RTP_DISABLE_SLICED=1 python3 bench.py $Q --model $m 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m nosliced', d['value'], d['ms_per_step'], d.get('final_loss'))"
done
timeout 600 python3 tools/plan_times.py --model hr3d_one_hm_doppler_phase --convs --top 40 > gpurun_out/r03_plan_phase_sliced.txt 2>&1
head -24 gpurun_out/r03_plan_phase_sliced.txt
