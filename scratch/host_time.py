import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rt_pose_amd.trainer import DataParallelTrainer
from rt_pose_amd import synth, configs
use_graph = "--graph" in sys.argv
tr = DataParallelTrainer("hr3d", batch_per_gpu=8, use_graph=use_graph)
spec = configs.spec("hr3d")
ex = synth.make_batch(8, spec["cin"], configs.NATIVE_DIMS, seed=1234, one_hm=spec["heads"]["hm"] == 1, rank=0)
tr.load(ex)
for _ in range(5):
    tr.step()
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K):
    tr.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("lanes", os.environ.get("RTP_LANES"), "graph", use_graph, "host enqueue ms/step %.3f, total ms/step %.3f" % ((t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3))
print("fwd launches", len(tr.engine.fwd), "waits", sum(len(w) for w in tr.engine.fwd_plan.waits), "bwd launches", len(tr.engine.bwd), "waits", sum(len(w) for w in tr.engine.bwd_plan.waits))
