import sys; sys.path.insert(0, '.')
import torch
from rt_pose_amd import configs, synth
from rt_pose_amd.trainer import DataParallelTrainer
for graph in (False, True):
    tr = DataParallelTrainer('hr3d', 8, configs.NATIVE_DIMS, total_steps=100, use_graph=graph)
    ex = synth.make_batch(8, 1, configs.NATIVE_DIMS, seed=1234)
    tr.load(ex)
    for i in range(8):
        tr.step()
        torch.cuda.synchronize()
        l = tr.losses()
        print('graph', graph, i, float(l['loss']), float(l['hm_loss']), 'gnorm', float(tr.opt.norm[0]), 'pnorm', float(tr.flat.p.norm()))
    del tr
