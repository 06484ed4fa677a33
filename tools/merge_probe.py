import sys, os
sys.path.insert(0, "/root/repo")
import torch
from rt_pose_amd import configs, synth
from rt_pose_amd.trainer import DataParallelTrainer
tr = DataParallelTrainer("hr3d", 8, configs.NATIVE_DIMS, total_steps=100, use_graph=False)
print("merged:", tr.engine.merged)
print("fwd launches", len(tr.engine.fwd), "bwd", len(tr.engine.bwd))
