#!/bin/bash
# A/B of bench.py under environment settings: tools/r04_ab.sh <tag> "ENV1=a ENV2=b" "ENV3=c" ...   (one short bench per setting)
export TMPDIR=/tmp
tag=$1; shift
i=0
for envs in "$@"; do
  out=gpurun_out/${tag}_$i.json
  env $envs timeout 150 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-torch-gpu --no-lidar --no-dcn --no-forward --no-other-models > $out 2> gpurun_out/${tag}_$i.err
  python3 - "$envs" $out <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    f = d["roofline"]["families"]
    k = [v for kk, v in f.items() if kk.startswith("conv_tiled_kernel")][0]
    print("AB [%s] ms/step %.3f %s  f/s %.1f  tiled_full %.1f us  frac %.3f  wgrad_tiled %.1f us" % (sys.argv[1], d["ms_per_step"], d.get("segments_ms_per_step"), d["value"], k["avg_us_per_launch"], d["roofline"]["frac"], f["wgrad_tiled (32ch 3x3x3)"]["avg_us_per_launch"]))
except Exception as e:
    print("AB [%s] FAILED %r" % (sys.argv[1], e)); print(open(sys.argv[2].replace(".json", ".err")).read()[-1500:])
PY
  i=$((i+1))
done
