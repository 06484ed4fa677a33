import os, sys
sys.path.insert(0, '/root/repo')
import torch
from rt_pose_amd import configs, synth
from rt_pose_amd.trainer import DataParallelTrainer
spec = configs.spec("hr3d")
ex = synth.make_batch(8, spec["cin"], configs.NATIVE_DIMS, seed=1234, one_hm=False)
def grads(lanes, steps=1):
    tr = DataParallelTrainer("hr3d", 8, configs.NATIVE_DIMS, total_steps=100, use_graph=False)
    tr.engine.use_lanes = lanes
    tr.load(ex)
    out = []
    for _ in range(steps):
        with tr._on_stream():
            tr._fwd_bwd()
        torch.cuda.synchronize()
        out.append(tr.flat.g.detach().clone())
    return out
ref = grads(False, 1)[0]
rn = float(ref.double().norm())
print("ref norm", rn)
for trial in range(3):
    gs = grads(True, 6)   # same weights (no optimiser step): every replay must give the same gradient
    for i, g in enumerate(gs):
        d = (g - ref).double()
        print("trial %d replay %d: rel l2 %.3e  max abs %.3e (max |ref| %.3e)" % (trial, i, float(d.norm()) / rn, float(d.abs().max()), float(ref.abs().max())))
g1 = grads(False, 2)
d = (g1[1] - ref).double()
print("one-stream replay: rel l2 %.3e max abs %.3e" % (float(d.norm()) / rn, float(d.abs().max())))
