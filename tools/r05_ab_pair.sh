#!/bin/bash
# paired head towers (Graph.conv_pair) on / off, same box: doppler and phase configs
for rep in 1 2; do for m in hr3d_one_hm_doppler hr3d_one_hm_doppler_phase; do for v in 1 0; do echo -n "$m pair=$v "; RTP_PAIR_HEADS=$v python bench.py --model $m --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-forward --no-dcn --no-lidar --no-torch-gpu --no-other-models 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done; done; done
