#!/usr/bin/env python
"""Timeline of EVERY launch of a training step's forward and backward plans (timing events on the launch's own lane stream):
start and end relative to the plan's first launch, per lane.  The step with the median duration of R traced steps is printed.
tools/lane_timeline.py [config] > timeline.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rt_pose_amd import configs, synth
from rt_pose_amd.trainer import DataParallelTrainer
name = sys.argv[1] if len(sys.argv) > 1 else "hr3d"
spec = configs.spec(name)
tr = DataParallelTrainer(name, 8, configs.NATIVE_DIMS, total_steps=1000, use_graph=False)
tr.load(synth.make_batch(8, spec["cin"], configs.NATIVE_DIMS, seed=1234, one_hm=spec["heads"]["hm"] == 1))
for _ in range(10):
    tr.step()
torch.cuda.synchronize()
plans = (("fwd", tr.engine.fwd_plan), ("bwd", tr.engine.bwd_plan))
R = 7
runs = []
for r in range(R):
    for _, pl in plans:
        pl.trace = {i: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for i in range(len(pl.launches))}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with tr._on_stream():
        e0.record(tr.stream)
    tr.step()
    with tr._on_stream():
        e1.record(tr.stream)
    torch.cuda.synchronize()
    rows = {}
    for pname, pl in plans:
        first = pl.trace[0][0]
        rows[pname] = [(first.elapsed_time(pl.trace[i][0]) * 1e3, first.elapsed_time(pl.trace[i][1]) * 1e3) for i in range(len(pl.launches))]
    runs.append((e0.elapsed_time(e1) * 1e3, rows))
runs.sort(key=lambda t: t[0])
step, rows = runs[R // 2]
print("step %.0f us (all %s)" % (step, [int(t[0]) for t in runs]))
for pname, pl in plans:
    print("-- %s" % pname)
    for i, L in enumerate(pl.launches):
        a, b = rows[pname][i]
        lane = pl.lane_of[i]
        deps = [pl.launches[j].tag for j in pl.waits[i]]
        print("%8.0f %8.0f  %6.0f  %s lane %d  %-30s %s" % (a, b, b - a, "    " * lane, lane, L.tag, ("<- " + ", ".join(deps)) if deps else ""))
