#!/bin/bash
# The driver's GPU tier, as the driver runs it (fresh box, `pytest tests/ -x -q -m gpu` as the first python work), plus --durations.
# Output: gpurun_out/r06_gpu_suite<tag>.txt ; then the default bench line: gpurun_out/r06_bench<tag>.json
tag=${1:-}
t0=$(date +%s)
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=25 -p no:cacheprovider > gpurun_out/r06_gpu_suite$tag.txt 2>&1
echo "rc $? wall $(( $(date +%s) - t0 )) s" >> gpurun_out/r06_gpu_suite$tag.txt
tail -5 gpurun_out/r06_gpu_suite$tag.txt
if [ "${2:-bench}" = "bench" ]; then
  timeout 900 python bench.py > gpurun_out/r06_bench$tag.json 2> gpurun_out/r06_bench$tag.err
  echo "bench rc $?"; head -c 600 gpurun_out/r06_bench$tag.json
fi
