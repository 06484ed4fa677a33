#!/bin/bash
# tools/r04_ab_model.sh <model> <tag> "ENV=.." ... : same-box A/B of bench.py --model <model> under environment settings
export TMPDIR=/tmp
model=$1; tag=$2; shift 2
i=0
for envs in "$@"; do
  out=gpurun_out/${tag}_$i.json
  env $envs timeout 200 python3 bench.py --model $model --steps 10 --warmup 4 --no-cpu-baseline --no-torch-gpu --no-lidar --no-dcn --no-forward --no-other-models --no-roofline > $out 2> gpurun_out/${tag}_$i.err
  python3 -c "
import json,sys
try:
    d=json.loads(open('$out').read().strip().splitlines()[-1]); print('AB [$envs] $model ms/step %.3f %s f/s %.1f' % (d['ms_per_step'], d.get('segments_ms_per_step'), d['value']))
except Exception as e:
    print('AB [$envs] FAILED', e); print(open('gpurun_out/${tag}_$i.err').read()[-800:])
"
  i=$((i+1))
done
