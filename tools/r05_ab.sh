#!/bin/bash
# Round-5 A/B runs of bench.py on one MI355X (each in its own process; a crash or hang of one does not stop the others).
# usage: tools/r05_ab.sh <out file> 'label@@ENV1=..|ENV2=..@@bench args' ...
out=$1; shift
: > "$out"
FAST="--no-other-models --no-dcn --no-lidar --no-torch-gpu --no-cpu-baseline --no-forward"
for spec in "$@"; do
  label=${spec%%@@*}; rest=${spec#*@@}; envs=${rest%%@@*}; args=${rest#*@@}
  echo "== $label  env[$envs]  args[$args]" >> "$out"
  ( IFS='|'; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done; IFS=' '
    timeout 300 python bench.py $FAST $args > /tmp/ab_out.txt 2> /tmp/ab_err.txt; rc=$?
    python - "$rc" >> "$out" <<'PY'
import sys, json
rc = int(sys.argv[1])
ok = False
for ln in open('/tmp/ab_out.txt'):
    ln = ln.strip()
    if ln.startswith('{'):
        d = json.loads(ln); r = d.get('roofline') or {}; ok = True
        print('   value %.1f f/s  ms/step %.3f  segments %s  hip_graph %s  frac %s' % (d['value'], d['ms_per_step'], d.get('segments_ms_per_step'), d['config'].get('hip_graph'), r.get('frac')))
if not ok:
    print('   NO RESULT rc %d; stderr tail:' % rc)
    print(''.join('      ' + l for l in open('/tmp/ab_err.txt').readlines()[-12:]))
PY
  )
done
cat "$out"
