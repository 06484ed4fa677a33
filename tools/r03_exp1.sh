#!/bin/bash
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-roofline --no-forward --no-dcn --no-lidar --no-torch-gpu --steps 40 --warmup 8"
o=gpurun_out/r03_exp1.txt
: > $o
echo "== host issue" >> $o; timeout 300 python3 tools/host_issue_time.py >> $o 2>&1
run() { echo "== $1" >> $o; shift; ( "$@" python3 bench.py $Q 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" ) >> $o 2>&1; }
run default env
run default_again env
run hwq8 env GPU_MAX_HW_QUEUES=8
run hwq16 env GPU_MAX_HW_QUEUES=16
run kernarg env HIP_FORCE_DEV_KERNARG=1
run hwq8_kernarg env GPU_MAX_HW_QUEUES=8 HIP_FORCE_DEV_KERNARG=1
run single env RTP_LANES=0,0,0,0,0,0
echo "== graph" >> $o; ( python3 bench.py $Q --graph 2>/dev/null | cut -c1-200 ) >> $o 2>&1
cat $o
