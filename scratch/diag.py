import sys; sys.path.insert(0, '.')
import torch
from oracle import hrradarpose_ref as O
from tests.test_gpu_engine import make, DIMS
from tests.emu_backend import EmuBackend
from rt_pose_amd.backend import HipBackend
from tests.util import rel_err
hip = HipBackend('cuda:0')
name='hr3d'
arch, fin, fout, fuse, heads, weight, cw = O.MODEL_CONFIGS[name]
ex = O.synth_example(2, 1, DIMS, seed=1234)
res = {}
for tag, be in (('hip', hip), ('emu', EmuBackend())):
    eng, flat, sd = make(be, name, 2, DIMS)
    eng.load_input(ex['rdr']['rdr_tensor']); eng.load_targets(ex['rdr']); eng.run_forward(); eng.run_loss_backward()
    if tag == 'hip': torch.cuda.synchronize()
    res[tag] = (eng, flat)
eh, fh = res['hip']; ee, fe = res['emu']
# activation + grad comparison in order
for ah, ae in zip(eh.graph.acts, ee.graph.acts):
    e = rel_err(ah.buf.float().cpu(), ae.buf.float())
    ge = -1
    if ah.grad is not None and ae.grad is not None:
        ge = rel_err(ah.grad.buf.float().cpu(), ae.grad.buf.float())
    print('%-14s fwd %.2e  grad %.2e' % (ah.name, e, ge))
for k in fh.grads:
    if k in eh.live_params:
        print('%-60s %.3e' % (k, rel_err(fh.grads[k].cpu(), fe.grads[k])))
