import sys; sys.path.insert(0, '.')
import torch
from rt_pose_amd import configs, synth
from rt_pose_amd.trainer import DataParallelTrainer
B = 8
for graph in (True, False):
    tr = DataParallelTrainer('hr3d', B, configs.NATIVE_DIMS, total_steps=100, use_graph=graph)
    ex = synth.make_batch(B, 1, configs.NATIVE_DIMS, seed=1234)
    tr.load(ex)
    for i in range(3):
        tr.step(); torch.cuda.synchronize()
        g = tr.flat.g
        print('graph', graph, 'step', i, 'hyper', [round(float(v), 6) for v in tr.opt.hyper.cpu()], 'partial sum', float(tr.opt.partial.sum()),
              'g finite', bool(torch.isfinite(g).all()), 'g norm', float(g.norm()), 'g absmax', float(g.abs().max()), 'norm', float(tr.opt.norm[0]),
              'm absmax', float(tr.flat.m.abs().max()), 'v absmax', float(tr.flat.v.abs().max()))
