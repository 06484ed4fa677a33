#!/bin/bash
# the two-process tests in a loop on a fresh box, the FIRST python work of the call (what the driver's run was):
# N iterations of the default (small-size) bench two-rank test, N/2 of the native-shape B = 8 variant (RTP_SLOW=1), 4 of the
# data-parallel one-device test.  Output: gpurun_out/r06_loop.txt
out=gpurun_out/r06_loop.txt
: > $out
N=${1:-20}
run() {  # label, iterations, env, test id
  for i in $(seq 1 $2); do
    t0=$(date +%s%N)
    env $3 timeout 400 python -m pytest "$4" -x -q -p no:cacheprovider > gpurun_out/r06_loop_last.log 2>&1
    rc=$?
    echo "$1 iter $i rc $rc $(( ($(date +%s%N) - t0) / 1000000 )) ms" >> $out
    if [ $rc -ne 0 ]; then tail -60 gpurun_out/r06_loop_last.log >> $out; fi
  done
}
run bench_two_ranks_small $N RTP_X=1 "tests/test_gpu_bench_two_ranks.py"
run bench_two_ranks_native_b8 $((N / 2)) RTP_SLOW=1 "tests/test_gpu_bench_two_ranks.py"
run dp_one_device 4 RTP_X=1 "tests/test_gpu_dp_one_device.py"
grep -c " rc 0 " $out; grep -v " rc 0 " $out | head -40
