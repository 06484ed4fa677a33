"""Reference-format checkpoint interop (rt_pose_amd/checkpoint.py; det3d/torchie/trainer/checkpoint.py:67-260,
trainer.py:354-368, 494-509).  CPU: the trainer runs on the emulated kernels."""
import os
from collections import OrderedDict

import pytest
import torch

from rt_pose_amd import checkpoint as ck
from rt_pose_amd import configs, synth
from rt_pose_amd.trainer import DataParallelTrainer
from tests.emu_backend import EmuBackend

DIMS, B = (4, 8, 16), 1


def make(seed=0):
    return DataParallelTrainer("hr3d", B, DIMS, total_steps=10, backend=EmuBackend(exact=True), seed=seed)


@pytest.mark.timeout(600)
def test_save_resume_round_trip_and_torch_adam_accepts_the_state(tmp_path):
    tr = make()
    for s in range(2):
        tr.step(synth.make_batch(B, 1, DIMS, seed=50 + s))
    path = ck.save_epoch(tr, str(tmp_path))
    assert os.path.basename(path) == "epoch_1.pth" and os.readlink(os.path.join(tmp_path, "latest.pth")) == "epoch_1.pth"
    raw = torch.load(path, weights_only=False)
    assert set(raw) == {"meta", "state_dict", "optimizer"} and raw["meta"]["iter"] == 2 and raw["meta"]["epoch"] == 1
    assert list(raw["state_dict"]) == list(configs.param_shapes("hr3d"))            # reference names, reference order
    assert raw["state_dict"]["backbone.backbone.layer1.conv2.conv.weight"].shape == (32, 32, 3, 3, 3)
    # torch's own Adam (what the reference's OptimWrapper wraps) takes the optimizer entry as is
    ps = [torch.nn.Parameter(v.clone()) for v in raw["state_dict"].values()]
    adam = torch.optim.Adam([{"params": ps, "lr": 0}, {"params": [], "lr": 0}], betas=(0.9, 0.99))
    adam.load_state_dict(raw["optimizer"])
    live = tr.engine.live_params
    names = list(raw["state_dict"])
    assert set(raw["optimizer"]["state"]) == {i for i, k in enumerate(names) if k in live}   # unused params: no state, like torch
    assert len(live) < len(names)                                                              # 'top' fuse: dead stage-4 rows
    i0 = names.index("backbone.backbone.layer1.conv2.conv.weight")
    assert torch.equal(adam.state[ps[i0]]["exp_avg"], tr.flat._view(tr.flat.m, names[i0]).cpu())
    # resume into a differently initialised trainer: identical weights, moments, position -> identical next step
    tr2 = make(seed=7)
    ck.resume(tr2, os.path.join(tmp_path, "latest.pth"))
    assert tr2.step_idx == 2 and tr2.opt.t == tr.opt.t == 2
    assert torch.equal(tr2.flat.p, tr.flat.p) and torch.equal(tr2.flat.m, tr.flat.m) and torch.equal(tr2.flat.v, tr.flat.v)
    nxt = synth.make_batch(B, 1, DIMS, seed=99)
    tr.step(nxt)
    tr2.step(nxt)
    assert torch.equal(tr2.flat.p, tr.flat.p)


def test_loads_ddp_prefixed_checkpoint_non_strict(tmp_path, capsys):
    tr = make()
    sd = OrderedDict(("module." + k, torch.full(tuple(s), 0.25)) for k, s in configs.param_shapes("hr3d").items())
    sd["module.extra.weight"] = torch.zeros(3)                                   # unexpected key: reported, skipped
    k_bad = "module.pose_head.tasks.0.hm.2.bias"
    sd[k_bad] = torch.zeros(7)                                                   # shape mismatch: reported, skipped
    before = tr.flat.values[k_bad[7:]].clone()
    p = os.path.join(tmp_path, "ref.pth")
    torch.save({"meta": {"epoch": 3, "iter": 120}, "state_dict": sd}, p)
    ck.load_checkpoint(tr, p)
    out = capsys.readouterr().out
    assert "unexpected key" in out and "extra.weight" in out and "mismatched shape" in out
    assert float(tr.flat.values["backbone.backbone.layer1.conv2.conv.weight"].mean()) == 0.25
    assert torch.equal(tr.flat.values[k_bad[7:]], before)
    with pytest.raises(RuntimeError):
        ck.load_checkpoint(tr, p, strict=True)
    with pytest.raises(IOError):
        ck.load_checkpoint(tr, os.path.join(tmp_path, "nope.pth"))
    torch.save({"weights": 1}, p)
    with pytest.raises(RuntimeError):
        ck.load_checkpoint(tr, p)
    with pytest.raises(TypeError):
        ck.save_checkpoint(tr, p, meta=[1])


def test_dcn_head_checkpoint_round_trip_keeps_current_format_offset_keys(tmp_path):
    """'<pack>.conv_offset.weight' also ends in '_offset.weight'; the pre-v2 migration must not rename it."""
    from rt_pose_amd.engine import FlatParams
    shapes = configs.param_shapes("hr3d_dcn")
    offs = [k for k in shapes if ".conv_offset." in k]
    assert len(offs) == 4, offs
    alloc = lambda shape, dt: torch.zeros(shape, dtype=torch.float32)
    src, dst = FlatParams(shapes, alloc), FlatParams(shapes, alloc)
    g = torch.Generator().manual_seed(3)
    src.p.copy_(torch.randn(src.numel, generator=g))
    p = os.path.join(tmp_path, "dcn.pth")
    torch.save({"meta": {"epoch": 1, "iter": 5}, "state_dict": ck.model_state_dict(src)}, p)   # what save_checkpoint writes
    unexpected, missing, mismatched = ck.load_state_dict(dst, torch.load(p, weights_only=False)["state_dict"])
    assert not unexpected and not missing and not mismatched
    dst.p.zero_()
    ck.load_checkpoint(dst, p, strict=True)
    assert torch.equal(dst.p, src.p)
    for k in offs:
        assert torch.equal(dst.values[k], src.values[k]) and float(dst.values[k].abs().sum()) > 0


def test_pre_v2_offset_keys_are_migrated_only_onto_existing_targets(tmp_path):
    from rt_pose_amd.dcn import migrate_pre_v2_keys
    from rt_pose_amd.engine import FlatParams
    shapes = configs.param_shapes("hr3d_dcn")
    alloc = lambda shape, dt: torch.zeros(shape, dtype=torch.float32)
    src, dst = FlatParams(shapes, alloc), FlatParams(shapes, alloc)
    src.p.copy_(torch.randn(src.numel, generator=torch.Generator().manual_seed(4)))
    sd = OrderedDict()
    for k, v in src.state_dict().items():      # write the DCN packs' offset convs under their pre-version-2 names
        sd[k.replace(".conv_offset.", "_offset.") if ".conv_offset." in k else k] = v
    assert not any(".conv_offset." in k for k in sd)
    sd["something_offset.weight"] = torch.zeros(2)          # no such pack in the target: must stay as it is (unexpected key)
    p = os.path.join(tmp_path, "old.pth")
    torch.save({"meta": {"epoch": 0, "iter": 0}, "state_dict": sd}, p)
    ck.load_checkpoint(dst, p)
    assert torch.equal(dst.p, src.p)
    out = migrate_pre_v2_keys(dict(sd), target_keys=set(dst.values))
    assert "something_offset.weight" in out and all(k in out for k in shapes)
