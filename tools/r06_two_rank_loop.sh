#!/bin/bash
# the two-process tests in a loop on a fresh box, the FIRST python work of the call (what the driver's run was),
# then the A/B settings.  Output: gpurun_out/r06_loop.txt
out=gpurun_out/r06_loop.txt
: > $out
N=${1:-12}
for i in $(seq 1 $N); do
  t0=$(date +%s.%N)
  timeout 400 python -m pytest tests/test_gpu_bench_two_ranks.py -x -q -p no:cacheprovider > gpurun_out/r06_loop_$i.log 2>&1
  rc=$?
  t1=$(date +%s.%N)
  echo "bench_two_ranks iter $i rc $rc $(echo "$t1 - $t0" | bc) s" >> $out
  if [ $rc -ne 0 ]; then tail -80 gpurun_out/r06_loop_$i.log >> $out; fi
done
for i in $(seq 1 4); do
  t0=$(date +%s.%N)
  timeout 600 python -m pytest tests/test_gpu_dp_one_device.py -x -q -p no:cacheprovider > gpurun_out/r06_loop_dp_$i.log 2>&1
  rc=$?
  t1=$(date +%s.%N)
  echo "dp_one_device iter $i rc $rc $(echo "$t1 - $t0" | bc) s" >> $out
  if [ $rc -ne 0 ]; then tail -80 gpurun_out/r06_loop_dp_$i.log >> $out; fi
done
cat $out | grep -v "^ " | head -60
