#!/bin/bash
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-roofline --no-forward --no-dcn --no-lidar --no-torch-gpu --steps 40 --warmup 8"
o=gpurun_out/r03_exp2.txt
: > $o
run() { echo "== $1" >> $o; shift; ( "$@" python3 bench.py $Q 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" ) >> $o 2>&1; }
run default env
for q in 1 2 3 5 6; do run hwq$q env GPU_MAX_HW_QUEUES=$q; done
for l in 0,0,2,3,4,5 0,1,2,0,4,5 0,1,1,3,4,1 0,1,2,3,4,2 0,1,2,3,1,5 0,1,2,3,2,5 0,1,2,1,4,5 0,1,2,3,3,5 0,1,1,3,4,5 0,1,2,2,2,5; do run lanes$l env RTP_LANES=$l; done
run prio0 env RTP_MAIN_PRIORITY=0
run default2 env
cat $o
