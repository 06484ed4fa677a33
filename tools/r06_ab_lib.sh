#!/bin/bash
# A/B of two builds on one box: tools/r06_ab_lib.sh <libA> <libB> ...  (paths for RTP_LIB; "" = the product library): the tiled
# microbenchmark and the train-step leg of bench.py, `REPS` times interleaved.  Output: gpurun_out/r06_ab_lib.txt
reps=${REPS:-3}
out=gpurun_out/r06_ab_lib.txt
: > $out
flags="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-forward --no-dcn --no-lidar --no-torch-gpu --no-other-models"
for r in $(seq 1 $reps); do
  for lib in "$@"; do
    mb=$(RTP_LIB="$lib" python tools/microbench_tiled.py 2>/dev/null | head -3 | tr '\n' ' ')
    ms=$(RTP_LIB="$lib" python bench.py $flags 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['segments_ms_per_step'])")
    echo "rep $r  [${lib:-product}]  step $ms | $mb" | tee -a $out
  done
done
