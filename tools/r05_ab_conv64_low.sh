#!/bin/bash
# the 64 -> 64 convs of levels 2 / 3 (ragged W) on conv64_tiled.hip (default) or on the generic kernel (RTP_CONV64=0), same box
for rep in 1 2 3; do for v in "" "RTP_CONV64=0"; do for m in hr3d hr3d_one_hm_doppler; do echo -n "$m ${v:-default} "; env $v python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-forward --no-dcn --no-lidar --no-torch-gpu --no-other-models 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done; done; done
