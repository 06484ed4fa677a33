#!/usr/bin/env python
"""Run-to-run reproducibility of the training step: per-step losses of several fresh trainers on the same seeded batch
(lane mode twice, one stream once).  Differences should start at rounding level (the only unordered sums are the LDS
float atomics of class_sums) and grow slowly; an early jump would point at a missing cross-lane dependency."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def run(steps, lanes=True, batch=8):
    from rt_pose_amd import configs, synth
    from rt_pose_amd.trainer import DataParallelTrainer
    spec = configs.spec("hr3d")
    tr = DataParallelTrainer("hr3d", batch, configs.NATIVE_DIMS, total_steps=100, use_graph=False)
    tr.engine.use_lanes = lanes
    ex = synth.make_batch(batch, spec["cin"], configs.NATIVE_DIMS, seed=1234, one_hm=False)
    tr.load(ex)
    out = []
    for _ in range(steps):
        tr.step()
        torch.cuda.synchronize()
        out.append(float(tr.losses()["loss"]))
    gn = float(tr.flat.g.double().norm())
    return out, gn


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    a, ga = run(n)
    b, gb = run(n)
    c, gc = run(n, lanes=False)
    for i in range(n):
        print("step %2d  lanes %.6f  lanes' %.6f  one-stream %.6f   |d| %.2e %.2e" % (i, a[i], b[i], c[i], abs(a[i] - b[i]), abs(a[i] - c[i])))
    print("final grad norms", ga, gb, gc)
