#!/bin/bash
# round 3: GPU tests of the new native-shape cases, the key-point artefact, the whole-step PMC traffic, the default bench
export TMPDIR=/tmp
timeout 2400 python3 -m pytest tests/test_gpu_native.py tests/test_gpu_engine.py -x -q -s 2>&1 | grep -v "^$" | tail -120 > gpurun_out/r03_native_tests.txt
tail -5 gpurun_out/r03_native_tests.txt
timeout 900 python3 -m tests.keypoint_agreement --steps 1500 --eval-batches 16 --seeds 2 --out gpurun_out/keypoint_agreement.json > gpurun_out/r03_keypoint.log 2>&1
tail -c 1500 gpurun_out/r03_keypoint.log
timeout 900 tools/pmc_step.sh > gpurun_out/r03_pmc_step.txt 2>&1; tail -3 gpurun_out/r03_pmc_step.txt
timeout 900 python3 bench.py > gpurun_out/r03b_bench.json 2> gpurun_out/r03b_bench.err; python3 -c "
import json; d=json.loads(open('gpurun_out/r03b_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('other_models'), d['cpu_baseline'])"
