"""Golden vectors for row T (the train-step rule) captured from the REFERENCE's own optimiser files.

Runs only in the authoring container (needs /root/reference).  det3d/solver/fastai_optim.py cannot be imported on Python >= 3.10
because of its first line (`from collections import Iterable`); the alias `collections.Iterable = collections.abc.Iterable` is set
here before a file-level import -- the same kind of stand-in as the `numba.jit` identity stub of gen_golden.py.  Nothing else is
replaced: OptimWrapper (fastai_optim.py:121-175), OneCycle / LRSchedulerStep (learning_schedules_fastai.py:7-95),
torch.optim.Adam and torch.nn.utils.clip_grad_norm_ are the real ones, driven in the order the reference's trainer drives them:

    lr_scheduler.step(global_step)                      det3d/torchie/trainer/trainer.py:408-412
    optimizer.zero_grad(); backward; clip_grad_norm_(max_norm=35, norm_type=2); optimizer.step()
                                                        det3d/torchie/trainer/hooks/optimizer.py:14-24
    optimizer = OptimWrapper.create(partial(Adam, betas=(0.9, 0.99), amsgrad=0.0), 3e-3, get_layer_groups(model),
                                    wd=0.01, true_wd=True, bn_wd=True)        det3d/torchie/apis/train.py:157-174
    OneCycle(optimizer, total_steps, lr_max=0.001, moms=[0.95, 0.85], div_factor=10.0, pct_start=0.4)
                                                        configs/cruw_pose/hr3d.py:176-181, apis/train.py:270-275

The "model" is a small module tree with the reference's parameter kinds (Conv3d weight + bias, GroupNorm affine, a parameter that
never receives a gradient); the gradients are seeded noise written into .grad (the optimiser rule does not care where they come
from), one step large enough for the clip to bite.  The fixture holds the inputs (initial parameters, per-step gradients) and the
reference's outputs (lr, momentum and every parameter after every step).

    python tests/golden/gen_golden_optim.py        # rewrites tests/golden/optim_golden.npz
"""
import collections
import collections.abc
import importlib.util
import os
import sys
from functools import partial

import numpy as np
import torch
from torch import nn

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
TOTAL_STEPS, STEPS = 10, 7          # pct_start 0.4 -> the peak is at step 4: both cosine phases are crossed
BIG_STEP = 2                        # the step whose gradient norm exceeds max_norm = 35


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


class Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.norm = nn.GroupNorm(8, 16)
        self.conv = nn.Conv3d(16, 8, 3, padding=1, bias=False)
        self.point = nn.Conv3d(8, 5, 1, bias=True)
        self.dead = nn.Conv3d(4, 3, 1, bias=True)     # stage-4 fuse rows under 'top': parameters without a gradient


def main():
    collections.Iterable = collections.abc.Iterable      # fastai_optim.py:1 (Python < 3.10 spelling)
    fo = _load("ref_fastai_optim", "det3d/solver/fastai_optim.py")
    ls = _load("ref_learning_schedules_fastai", "det3d/solver/learning_schedules_fastai.py")

    torch.manual_seed(5)
    model = Tiny()
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(torch.randn_like(p) * 0.3)
    names = [k for k, _ in model.named_parameters()]
    out = {"names": np.array(names), "total_steps": np.int64(TOTAL_STEPS)}
    for k, p in model.named_parameters():
        out["init/" + k] = p.detach().numpy().copy()

    # apis/train.py:150-174: one layer group of the flattened leaf modules
    def flatten_model(m):
        return sum(map(flatten_model, m.children()), []) if len(list(m.children())) else [m]
    layer_groups = [nn.Sequential(*flatten_model(model))]
    opt = fo.OptimWrapper.create(partial(torch.optim.Adam, betas=(0.9, 0.99), amsgrad=0.0), 3e-3, layer_groups,
                                 wd=0.01, true_wd=True, bn_wd=True)
    sched = ls.OneCycle(opt, TOTAL_STEPS, 0.001, [0.95, 0.85], 10.0, 0.4)

    g = torch.Generator().manual_seed(77)
    lrs, moms, norms = [], [], []
    for step in range(STEPS):
        sched.step(step)
        opt.zero_grad()
        for k, p in model.named_parameters():
            if k.startswith("dead."):
                continue
            p.grad = torch.randn(p.shape, generator=g) * (4.0 if step == BIG_STEP else 0.05)
            out["grad/%d/%s" % (step, k)] = p.grad.numpy().copy()
        total = torch.nn.utils.clip_grad_norm_(filter(lambda q: q.requires_grad, model.parameters()), max_norm=35, norm_type=2)
        opt.step()
        lrs.append(opt.lr)
        moms.append(opt.mom)
        norms.append(float(total))
        for k, p in model.named_parameters():
            out["after/%d/%s" % (step, k)] = p.detach().numpy().copy()
    assert norms[BIG_STEP] > 35 and max(n for i, n in enumerate(norms) if i != BIG_STEP) < 35
    out["lr"], out["mom"], out["grad_norm"] = np.asarray(lrs, np.float64), np.asarray(moms, np.float64), np.asarray(norms, np.float64)
    np.savez_compressed(os.path.join(HERE, "optim_golden.npz"), **out)
    print("steps", STEPS, "lr", lrs, "mom", moms, "norms", [round(n, 3) for n in norms])


if __name__ == "__main__":
    main()
