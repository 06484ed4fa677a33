#!/usr/bin/env python
"""Host-side cost of issuing one training step (Python + ctypes + HIP launches and event calls, no synchronisation inside
the loop) against the step's wall time: 2.2 ms of 6.8 ms on the GPU box, i.e. the step is not launch-bound on the host."""
import sys, time, os
sys.path.insert(0, '/root/repo')
import torch
from rt_pose_amd import configs, synth
from rt_pose_amd.trainer import DataParallelTrainer
spec = configs.spec("hr3d")
tr = DataParallelTrainer("hr3d", 8, configs.NATIVE_DIMS, total_steps=1000, use_graph=False)
ex = synth.make_batch(8, spec["cin"], configs.NATIVE_DIMS, seed=1234, one_hm=False)
tr.load(ex)
for _ in range(5): tr.step()
torch.cuda.synchronize()
# host-side issue time: no sync inside the loop; the queue depth lets the CPU run ahead
t0 = time.perf_counter(); hs = []
for _ in range(40):
    a = time.perf_counter(); tr.step(); hs.append(time.perf_counter() - a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host issue per step: median %.2f ms, min %.2f, max %.2f; loop %.2f ms/step; with final sync %.2f ms/step" % (
    sorted(hs)[20] * 1e3, min(hs) * 1e3, max(hs) * 1e3, (t1 - t0) / 40 * 1e3, (t2 - t0) / 40 * 1e3))
