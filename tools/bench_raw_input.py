#!/usr/bin/env python
"""PCIe-inclusive training rate: every step's batch arrives as raw fp16 cubes + key-points in host memory and goes through
rt_pose_amd.input_pipeline (pinned ring, one async H2D, crop/normalise and label kernels) -- the number DESIGN.md quotes
beside bench.py's inputs-resident `value` (never instead of it)."""
import argparse
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    from rt_pose_amd import configs
    from rt_pose_amd.input_pipeline import DeviceInputPipeline
    from rt_pose_amd.trainer import DataParallelTrainer
    tr = DataParallelTrainer("hr3d", a.batch, configs.NATIVE_DIMS, total_steps=1000, use_graph=False)
    pipe = DeviceInputPipeline(tr.engine, configs.ROI1, configs.VOXEL_SIZE, (20000, 45000), "zyx_real")
    tr.attach_input_pipeline(pipe)
    rng = np.random.default_rng(0)
    nb = 4   # distinct host batches, cycled
    cubes = [rng.uniform(0, 60000, size=(a.batch, 32, 128, 256)).astype(np.float16) for _ in range(nb)]
    poses = [[[(np.array([rng.uniform(1.5, 7), rng.uniform(-4, 4), rng.uniform(-0.5, 4)]) + rng.normal(0, 0.3, (15, 3))).tolist()]
              for _ in range(a.batch)] for _ in range(nb)]
    tr.feed_raw(cubes[0], poses[0])
    for i in range(a.warmup):
        tr.step_fed()
        tr.feed_raw(cubes[(i + 1) % nb], poses[(i + 1) % nb])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        tr.step_fed()
        tr.feed_raw(cubes[(i + 1) % nb], poses[(i + 1) % nb])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(json.dumps({"metric": "radar frames/sec (train), raw fp16 cubes + key-points from host memory each step",
                      "value": round(a.batch * a.steps / el, 2), "ms_per_step": round(1e3 * el / a.steps, 3),
                      "h2d_MB_per_step": round(a.batch * 32 * 128 * 256 * 2 / 1e6, 1), "final_loss": round(float(tr.losses()["loss"]), 4)}))


if __name__ == "__main__":
    main()
