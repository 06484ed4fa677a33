"""Row T on the GPU against the REFERENCE's optimiser: rtp_sqnorm + rtp_adam_step (rt_pose_amd.engine.FlatAdam) replay the seven
steps of tests/golden/optim_golden.npz, captured from det3d/solver/fastai_optim.py:121-175 (OptimWrapper, true_wd, bn_wd) around
torch.optim.Adam, det3d/solver/learning_schedules_fastai.py:53-95 (OneCycle) and clip_grad_norm_(35) by
tests/golden/gen_golden_optim.py.  fp32 state: 1e-5 relative per tensor (the kernel fuses the decay, the clip coefficient and the
bias corrections into one pass; the products associate differently from torch's)."""
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_adam_true_wd_one_cycle_against_the_reference_optimiser():
    from rt_pose_amd.backend import HipBackend
    from rt_pose_amd.engine import FlatAdam, FlatParams, one_cycle
    z = np.load(os.path.join(HERE, "golden", "optim_golden.npz"))
    names, total = [str(n) for n in z["names"]], int(z["total_steps"])
    be = HipBackend("cuda:0")
    shapes = OrderedDict((k, tuple(z["init/" + k].shape)) for k in names)
    flat = FlatParams(shapes, be.alloc)
    flat.load_state_dict({k: torch.tensor(z["init/" + k]) for k in names})
    live = {k for k in names if "grad/0/" + k in z.files}
    assert live and len(live) < len(names), "the fixture has parameters with and without gradients"
    opt = FlatAdam(be, flat, live)
    for step in range(len(z["lr"])):
        lr, b1 = one_cycle(step, total, 1e-3)
        assert abs(lr - float(z["lr"][step])) < 1e-15 and abs(b1 - float(z["mom"][step])) < 1e-15
        flat.g.zero_()
        for k in live:
            flat.grads[k].copy_(torch.tensor(z["grad/%d/%s" % (step, k)]))
        opt.set_hyper(lr, b1)
        opt.run()
        torch.cuda.synchronize()
        assert abs(float(opt.norm[0]) - float(z["grad_norm"][step])) < 1e-5 * float(z["grad_norm"][step])
        for k in names:
            want = torch.tensor(z["after/%d/%s" % (step, k)]).double()
            got = flat.values[k].cpu().double()
            assert float((got - want).norm() / want.norm()) < 1e-5, (step, k)
            assert float((got - want).abs().max()) < 2e-6 + 1e-5 * float(want.abs().max()), (step, k)
