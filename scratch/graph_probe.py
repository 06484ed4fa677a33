import sys; sys.path.insert(0, '.')
import torch
from rt_pose_amd import configs, synth
from rt_pose_amd.trainer import DataParallelTrainer
B = 8
tr = DataParallelTrainer('hr3d', B, configs.NATIVE_DIMS, total_steps=100, use_graph=True)
ex = synth.make_batch(B, 1, configs.NATIVE_DIMS, seed=1234)
tr.load(ex)
tr.step(); torch.cuda.synchronize()          # step 0 (captures)
eng = tr.engine
def snap():
    d = {}
    for a in eng.graph.acts:
        d['act:' + a.name] = a.buf.float().clone()
        if a.grad is not None: d['grad:' + a.name] = a.grad.buf.float().clone()
    for k, v in tr.flat.grads.items(): d['pg:' + k] = v.clone()
    return d
tr._graph.replay(); torch.cuda.synchronize()
r = snap()
tr._fwd_bwd(); torch.cuda.synchronize()
e = snap()
bad = 0
for k in r:
    a, b = r[k], e[k]
    nf = (~torch.isfinite(a)).sum().item()
    diff = float((a - b).abs().max()) if nf == 0 else float('nan')
    if nf or diff > 0:
        print('%-70s nonfinite %d maxdiff %.3e (absmax eager %.3e)' % (k, nf, diff, float(b.abs().max())))
        bad += 1
        if bad > 25: break
print('differing tensors:', bad)
