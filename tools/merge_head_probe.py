#!/usr/bin/env python
"""Which head launches RTP_MERGE_HEAD=1 manages to merge, and the step time with / without (run under both settings)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rt_pose_amd import configs, synth
from rt_pose_amd.trainer import DataParallelTrainer
tr = DataParallelTrainer("hr3d", 8, configs.NATIVE_DIMS, total_steps=1000, use_graph=False)
spec = configs.spec("hr3d")
tr.load(synth.make_batch(8, spec["cin"], configs.NATIVE_DIMS, seed=1234, one_hm=False))
print("merged:", getattr(tr.engine, "merged_head", None))
for _ in range(10):
    tr.step()
torch.cuda.synchronize()
print("loss", float(tr.losses()["loss"]))
