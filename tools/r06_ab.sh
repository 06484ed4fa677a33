#!/bin/bash
# A/B of plan options on one box: tools/r06_ab.sh "<RTP_PLAN A>" "<RTP_PLAN B>" ... -- every variant `reps` times, interleaved
# (A B C A B C ...), the train-step leg of bench.py only; prints ms/step per run.  Output: gpurun_out/r06_ab.txt
reps=${REPS:-3}
out=gpurun_out/r06_ab.txt
: > $out
flags="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-forward --no-dcn --no-lidar --no-torch-gpu --no-other-models ${BENCH_ARGS:-}"
for r in $(seq 1 $reps); do
  for plan in "$@"; do
    ms=$(RTP_PLAN="$plan" python bench.py $flags 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['segments_ms_per_step'])")
    echo "rep $r  [$plan]  $ms" | tee -a $out
  done
done
