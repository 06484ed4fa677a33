#!/bin/bash
# the DCN-head config (BASELINE config 4): forward of the deformable conv on the plan's layout (rtp_dcn_cl_forward) on / off, same box
for rep in 1 2; do for v in 1 0; do echo -n "hr3d_dcn RTP_DCN_CL=$v "; RTP_DCN_CL=$v python bench.py --model hr3d_dcn --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-forward --no-dcn --no-lidar --no-torch-gpu --no-other-models 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done; done
python tools/plan_times.py --model hr3d_dcn --top 8 2>&1 | tail -9
