#!/usr/bin/env python
"""Which lane -> stream foldings survive HIP-graph capture of the training plan?  Each candidate runs in a child process
(hipStreamEndCapture has crashed the process on some fork/join shapes, ROCm 7.2)."""
import os
import subprocess
import sys

CHILD = r"""
import sys, torch
sys.path.insert(0, %r)
from rt_pose_amd import synth
from rt_pose_amd.lanes import LanePlan
from rt_pose_amd.trainer import DataParallelTrainer
gmap = %r
dims = (8, 16, 32)
tr = DataParallelTrainer("hr3d", 2, dims, total_steps=20, use_graph=True, seed=3)
tr.engine.fwd_plan = LanePlan(tr.be, tr.engine.fwd, gmap)
tr.engine.bwd_plan = LanePlan(tr.be, tr.engine.bwd, gmap)
tr.load(synth.make_batch(2, 1, dims, seed=77))
for _ in range(3):
    tr.step()
torch.cuda.synchronize()
print("ok %%.5f" %% float(tr.losses()["loss"]))
"""

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
maps = [[0] * 6, [0, 1, 1, 1, 1, 1], [0, 0, 0, 1, 1, 0], [0, 1, 1, 2, 2, 1], [0, 1, 2, 3, 3, 2], [0, 1, 2, 3, 4, 5]]
if len(sys.argv) > 1:   # repeat a few foldings to see whether a crash is deterministic
    maps = [[0, 1, 1, 2, 2, 1]] * int(sys.argv[1]) + [[0, 1, 1, 1, 1, 1]] * int(sys.argv[1])
for m in maps:
    r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, m)], capture_output=True, text=True, timeout=600)
    print(m, "rc", r.returncode, r.stdout.strip()[-40:], flush=True)
