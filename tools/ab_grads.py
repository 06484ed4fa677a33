#!/usr/bin/env python
"""A/B helper: run `steps` training steps of a config at the native shape and save the flat gradient of the last
step + the loss trajectory to an .npz (compare two builds / env settings with --cmp a.npz b.npz)."""
import argparse
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out")
    ap.add_argument("--model", default="hr3d")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--cmp", nargs=2)
    a = ap.parse_args()
    if a.cmp:
        x, y = np.load(a.cmp[0]), np.load(a.cmp[1])
        g0, g1 = x["g"].astype(np.float64), y["g"].astype(np.float64)
        print("loss", x["loss"], y["loss"])
        print("grad norm %.6f %.6f  rel diff %.3e  max abs diff %.3e  identical %s" % (
            np.linalg.norm(g0), np.linalg.norm(g1), np.linalg.norm(g0 - g1) / np.linalg.norm(g0), np.abs(g0 - g1).max(),
            np.array_equal(x["g"], y["g"])))
        return
    import torch
    from rt_pose_amd import configs, synth
    from rt_pose_amd.trainer import DataParallelTrainer
    spec = configs.spec(a.model)
    tr = DataParallelTrainer(a.model, a.batch, configs.NATIVE_DIMS, total_steps=100, use_graph=False)
    ex = synth.make_batch(a.batch, spec["cin"], configs.NATIVE_DIMS, seed=1234, one_hm=spec["heads"]["hm"] == 1)
    tr.load(ex)
    losses = []
    for _ in range(a.steps):
        tr.step()
        torch.cuda.synchronize()
        losses.append(float(tr.losses()["loss"]))
    np.savez(a.out, g=tr.flat.g.cpu().numpy(), loss=np.array(losses))
    print(a.out, losses)


if __name__ == "__main__":
    main()
