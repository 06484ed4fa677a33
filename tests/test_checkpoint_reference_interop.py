"""Checkpoint interop against the REFERENCE's own functions (SURVEY 8b "checkpoint compatibility", 8f row N4), on CPU in the authoring
container: det3d/torchie/trainer/checkpoint.py is loaded at file level (torchvision / terminaltables / det3d.torchie are stand-in
modules for imports the two functions never touch; `torchie.mkdir_or_exist` is os.makedirs) and its save_checkpoint (:235-260) /
load_checkpoint (:166-217) run as written, on the reference's own module tree (RefNet of tests/golden/gen_golden.py: the reference's
HighResolution3DNet + CenterHead) and its own OptimWrapper(Adam) (det3d/solver/fastai_optim.py).

  ours -> reference:  rt_pose_amd.checkpoint.save_epoch  ->  reference load_checkpoint(strict=True) into the reference modules, and the
                      `optimizer` entry into the reference's OptimWrapper.load_state_dict
  reference -> ours:  reference save_checkpoint(model, optimizer=OptimWrapper)  ->  rt_pose_amd.checkpoint.load_checkpoint / resume

Skipped where /root/reference is absent (the GPU box): this is a `not gpu` test."""
import collections
import collections.abc
import importlib.util
import os
import sys
import types
from functools import partial

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="needs /root/reference (authoring container)")
DIMS, B = (4, 8, 16), 1


@pytest.fixture(autouse=True)
def clean_modules():
    """The file-level import of the reference plants synthetic det3d.* packages and stand-ins in sys.modules: removed again afterwards
    (other tests install rt_pose_amd's own det3d shim in the same process)."""
    before = dict(sys.modules)
    had_iterable = hasattr(collections, "Iterable")
    yield
    for k in list(sys.modules):
        if k not in before:
            del sys.modules[k]
    sys.modules.update({k: v for k, v in before.items() if sys.modules.get(k) is not v})
    if not had_iterable and hasattr(collections, "Iterable"):
        del collections.Iterable


def _ref():
    from tests.golden import gen_golden as G
    R = G.import_reference()
    for name in ("torchvision", "terminaltables"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["terminaltables"].AsciiTable = None
    if "spconv" not in sys.modules:     # checkpoint.py:44-47, :58: only an isinstance test against SparseConvolution (no module here is one)
        sp = types.ModuleType("spconv")
        sp.__path__ = []
        sp.conv = types.SimpleNamespace(SparseConvolution=type("SparseConvolution", (), {}))
        sys.modules["spconv"], sys.modules["spconv.pytorch"] = sp, sp
        sp.pytorch = sp
    tor = sys.modules["det3d.torchie"]
    tor.mkdir_or_exist = lambda d, mode=0o777: os.makedirs(d, mode=mode, exist_ok=True) if d else None
    G._pkg("det3d.torchie.trainer")
    u = types.ModuleType("det3d.torchie.trainer.utils")
    u.get_dist_info = lambda: (0, 1)
    sys.modules["det3d.torchie.trainer.utils"] = u
    ck = G._load("det3d.torchie.trainer.checkpoint", "det3d/torchie/trainer/checkpoint.py")
    collections.Iterable = collections.abc.Iterable      # det3d/solver/fastai_optim.py:1 (Python < 3.10 spelling)
    spec = importlib.util.spec_from_file_location("ref_fastai_optim", os.path.join(REF, "det3d/solver/fastai_optim.py"))
    fo = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fo)
    return G, R, ck, fo


def _ref_optimizer(fo, model):
    def flatten_model(m):   # det3d/torchie/apis/train.py:150-155
        return sum(map(flatten_model, m.children()), []) if len(list(m.children())) else [m]
    return fo.OptimWrapper.create(partial(torch.optim.Adam, betas=(0.9, 0.99), amsgrad=0.0), 3e-3, [torch.nn.Sequential(*flatten_model(model))],
                                  wd=0.01, true_wd=True, bn_wd=True)


def _trainer(seed=0):
    from rt_pose_amd.trainer import DataParallelTrainer
    from tests.emu_backend import EmuBackend
    return DataParallelTrainer("hr3d", B, DIMS, total_steps=10, backend=EmuBackend(exact=True), seed=seed)


@pytest.mark.timeout(900)
def test_our_checkpoint_loads_with_the_reference_loader(tmp_path):
    from rt_pose_amd import checkpoint as ours, synth
    G, R, ck, fo = _ref()
    tr = _trainer()
    for s in range(2):
        tr.step(synth.make_batch(B, 1, DIMS, seed=50 + s))
    path = ours.save_epoch(tr, str(tmp_path))
    model = G.RefNet(R, "hr3d")
    raw = ck.load_checkpoint(model, path, map_location="cpu", strict=True)     # strict: any missing / unexpected key raises
    assert set(raw) == {"meta", "state_dict", "optimizer"} and raw["meta"]["iter"] == 2
    for k, p in model.state_dict().items():
        assert torch.equal(p, tr.flat.values[k].detach().cpu()), k
    # the optimizer entry is what the reference's own wrapper takes (trainer.py:494-509 resume)
    opt = _ref_optimizer(fo, model)
    opt.load_state_dict(raw["optimizer"])
    names = list(model.state_dict().keys())
    params = [p for g in opt.opt.param_groups for p in g["params"]]
    assert len(params) == len(names)
    by_param = {id(p): n for n, p in zip(names, (p for _, p in model.named_parameters()))}
    seen = 0
    for p in params:
        st = opt.opt.state.get(p)
        if st:
            k = by_param[id(p)]
            assert torch.equal(st["exp_avg"], tr.flat._view(tr.flat.m, k).detach().cpu()), k
            assert float(st["step"]) == 2.0
            seen += 1
    assert seen == len(tr.engine.live_params)


@pytest.mark.timeout(900)
def test_a_checkpoint_written_by_the_reference_resumes_here(tmp_path):
    from rt_pose_amd import checkpoint as ours
    G, R, ck, fo = _ref()
    torch.manual_seed(3)
    model = G.RefNet(R, "hr3d")
    opt = _ref_optimizer(fo, model)
    g = torch.Generator().manual_seed(9)
    tr0 = _trainer()
    live = tr0.engine.live_params
    for step in range(2):        # two reference optimizer steps on seeded gradients (dead parameters get none, like under 'top')
        opt.lr, opt.mom = 1e-3, 0.9
        for k, p in model.named_parameters():
            p.grad = torch.randn(p.shape, generator=g) * 0.01 if k in live else None
        opt.step()
    path = os.path.join(tmp_path, "epoch_7.pth")
    ck.save_checkpoint(model, path, optimizer=opt, meta={"epoch": 7, "iter": 1234})
    tr = _trainer(seed=5)
    meta = ours.resume(tr, path)["meta"]
    assert meta["epoch"] == 7 and meta["iter"] == 1234 and tr.step_idx == 1234
    sd = model.state_dict()
    for k in sd:
        assert torch.equal(tr.flat.values[k].detach().cpu(), sd[k]), k
    params = dict(model.named_parameters())
    for k in live:
        st = opt.opt.state[params[k]]
        assert torch.equal(tr.flat._view(tr.flat.m, k).detach().cpu(), st["exp_avg"]), k
        assert torch.equal(tr.flat._view(tr.flat.v, k).detach().cpu(), st["exp_avg_sq"]), k
    assert tr.opt.t == 2
