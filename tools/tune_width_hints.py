#!/usr/bin/env python
"""Coordinate descent over the plan's width hints (engine.DEFAULT_WIDTH_HINTS) on one box: for every rule in turn, the train-step leg of
bench.py with that rule's workgroup count replaced by each candidate (256 = no hint for that tag), the others fixed; a change is kept
when it beats the incumbent by more than `MARGIN` in a re-run pair.  Prints every measurement; the last line is the best string.
    python tools/tune_width_hints.py [extra tag prefixes to consider ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rt_pose_amd.engine import DEFAULT_WIDTH_HINTS, parse_width_hints  # noqa: E402

FLAGS = "--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-forward --no-dcn --no-lidar --no-torch-gpu --no-other-models".split()
CANDS = [160, 176, 192, 208, 224, 256]
MARGIN = 0.004


def run(rules):
    spec = ",".join("%s=%d" % (k, v) for k, v in rules if v < 256)
    env = dict(os.environ, RTP_PLAN="width_hints=" + spec)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *FLAGS], env=env, capture_output=True, text=True, timeout=600).stdout
    d = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    return min(d["segments_ms_per_step"]), d["ms_per_step"]


def main():
    rules = [list(r) for r in parse_width_hints(DEFAULT_WIDTH_HINTS)] + [[t, 256] for t in sys.argv[1:]]
    best = min(run(rules)[0] for _ in range(2))
    print("start %.3f ms  %s" % (best, rules), flush=True)
    for i in range(len(rules)):
        keep = rules[i][1]
        for c in CANDS:
            if c == keep:
                continue
            trial = [list(r) for r in rules]
            trial[i][1] = c
            t = run(trial)[0]
            print("  %-18s %3d -> %3d : %.3f ms (incumbent %.3f)" % (rules[i][0], keep, c, t, best), flush=True)
            if t < best * (1 - MARGIN):
                t2, b2 = run(trial)[0], run(rules)[0]     # confirm against a fresh run of the incumbent
                print("     confirm: %.3f vs incumbent %.3f" % (t2, b2), flush=True)
                if t2 < b2 * (1 - MARGIN / 2):
                    rules, best, keep = trial, min(t, t2), c
    print("best %.3f ms" % best)
    print(";".join("%s=%d" % (k, v) for k, v in rules if v < 256))


if __name__ == "__main__":
    main()
