#!/usr/bin/env python
"""Busy time and gaps of the MAIN lane (stream 0) during an unprofiled training step: timing events around every lane-0 launch of
the forward and backward plans (rocprofv3's kernel trace slows the host's launch calls enough to distort exactly these gaps).
Prints, per plan, the lane's busy time, the sum of the gaps between consecutive lane-0 launches, and the largest gaps with the
launch that had to wait (its cross-lane dependencies are what the main lane was waiting for)."""
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
for _, pl in plans:
    pl.trace = {i: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for i in range(len(pl.launches)) if pl.lane_of[i] == 0}
acc = {}
R = 8
for r in range(R):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with tr._on_stream():
        e0.record(tr.stream)
    tr.step()
    with tr._on_stream():
        e1.record(tr.stream)
    torch.cuda.synchronize()
    acc.setdefault("step", []).append(e0.elapsed_time(e1) * 1e3)
    for pname, pl in plans:
        idx = sorted(pl.trace)
        for a, b in zip(idx[:-1], idx[1:]):
            acc.setdefault((pname, "gap", b), []).append(pl.trace[a][1].elapsed_time(pl.trace[b][0]) * 1e3)
        for a in idx:
            acc.setdefault((pname, "dur", a), []).append(pl.trace[a][0].elapsed_time(pl.trace[a][1]) * 1e3)
        acc.setdefault((pname, "span"), []).append(pl.trace[idx[0]][0].elapsed_time(pl.trace[idx[-1]][1]) * 1e3)
med = lambda v: sorted(v)[len(v) // 2]
print("step (events on the step stream, traced): %.0f us" % med(acc["step"]))
for pname, pl in plans:
    busy = sum(med(v) for k, v in acc.items() if isinstance(k, tuple) and k[0] == pname and k[1] == "dur")
    gaps = {k[2]: med(v) for k, v in acc.items() if isinstance(k, tuple) and k[0] == pname and k[1] == "gap"}
    print("%s: main lane span %.0f us, busy %.0f us, gaps %.0f us over %d launches" % (pname, med(acc[(pname, "span")]), busy, sum(gaps.values()), len(pl.trace)))
    for i, gp in sorted(gaps.items(), key=lambda kv: -kv[1])[:12]:
        deps = [pl.launches[j].tag for j in pl.waits[i]]
        print("   gap %6.0f us before %-28s (%.0f us)  waits for: %s" % (gp, pl.launches[i].tag, med(acc[(pname, "dur", i)]), ", ".join(deps) or "-"))
