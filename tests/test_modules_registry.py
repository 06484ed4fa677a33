"""Drop-in boundary on CPU: registry semantics, config loading through the det3d shim, state_dict
names/shapes, and the RadarPoseNet call convention -- with the emulated kernels injected so no GPU is needed."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import hrradarpose_ref as O
from rt_pose_amd import configs, modules, registry
from rt_pose_amd.registry import Config, Registry, build_detector, build_from_cfg
from tests.emu_backend import EmuBackend

REF_CFG_DIR = "/root/reference/configs/cruw_pose"


@pytest.fixture(autouse=True)
def emu():
    modules.set_backend_factory(lambda device: EmuBackend())
    yield
    modules.set_backend_factory(None)


def test_registry_semantics():
    reg = Registry("thing")

    @reg.register_module
    class A:
        def __init__(self, x=1, y=2):
            self.x, self.y = x, y

    assert reg.get("A") is A and reg.get("B") is None
    with pytest.raises(KeyError):
        reg.register_module(A)                      # duplicate
    with pytest.raises(TypeError):
        reg._register_module(lambda: 0)             # not a class
    obj = build_from_cfg(dict(type="A", x=5), reg, dict(y=7, x=9))
    assert (obj.x, obj.y) == (5, 7)                 # default_args only fill gaps
    assert build_from_cfg(dict(type=A), reg).x == 1
    with pytest.raises(KeyError):
        build_from_cfg(dict(type="Nope"), reg)
    with pytest.raises(AssertionError):
        build_from_cfg(["type"], reg)
    with pytest.raises(TypeError):
        build_from_cfg(dict(type=3), reg)
    for name in ("RadarFeatureNet",):
        assert registry.READERS.get(name)
    assert registry.BACKBONES.get("HRNet3D") and registry.HEADS.get("CenterHead") and registry.DETECTORS.get("RadarPoseNet")


@pytest.mark.parametrize("name", configs.NAMES)
def test_state_dict_matches_reference_schema(name, schema):
    model = build_detector(configs.model_dict(name), train_cfg=None, test_cfg=configs.test_cfg())
    sd = model.state_dict()
    assert list(sd.keys()) == list(schema[name].keys())
    for k, v in sd.items():
        assert list(v.shape) == schema[name][k], k
    assert {k: list(v) for k, v in configs.param_shapes(name).items()} == schema[name]
    assert float(sd["pose_head.tasks.0.hm.2.bias"][0]) == pytest.approx(-2.19)


@pytest.mark.skipif(not os.path.isdir(REF_CFG_DIR), reason="reference configs only exist in the authoring container")
@pytest.mark.parametrize("path", sorted(glob.glob(REF_CFG_DIR + "/*.py")))
def test_reference_configs_load_unchanged(path):
    cfg = Config.fromfile(path)
    name = os.path.basename(path)[:-3]
    assert cfg.model.type == "RadarPoseNet"
    assert cfg.enable_amp in (False, True)          # missing in hr3d.py / hr3d_one_hm.py -> False
    with pytest.raises(AttributeError):
        cfg.no_such_key
    assert dict(cfg.model.backbone) == configs.model_dict(name)["backbone"]
    model = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    assert sum(p.numel() for p in model.parameters()) == sum(int(np.prod(s)) for s in configs.param_shapes(name).values())
    assert cfg.test_cfg.voxel_size == configs.VOXEL_SIZE and list(cfg.test_cfg.pc_range) == configs.test_cfg()["pc_range"]


@pytest.mark.parametrize("name", ["hr3d", "hr3d_one_hm_doppler"])
def test_radar_pose_net_call_convention(name):
    spec = configs.spec(name)
    arch, fin, fout, fuse, heads, weight, cw = O.MODEL_CONFIGS[name]
    model = build_detector(configs.model_dict(name), train_cfg=None, test_cfg=configs.test_cfg())
    sd = O.seeded_state_dict(O.param_shapes(arch, fin, fout, fout, heads), seed=1)
    model.load_state_dict(sd)                       # reference-named checkpoint loads strictly
    ex = O.synth_example(2, spec["cin"], (8, 16, 16), seed=1234, one_hm=heads["hm"] == 1)
    # ---- training call: loss dict with the reference's keys, loss.backward() fills p.grad
    out = model(ex, return_loss=True)
    assert set(out.keys()) == {"loss", "hm_loss", "loc_loss", "loc_loss_elem", "num_positive"}
    loss = sum(out["loss"])
    loss.backward()
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.radar_pose_net(sdr, ex, fuse, weight, cw)
    ref["loss"][0].backward()
    assert abs(float(loss) - float(ref["loss"][0].detach())) < 2e-2 * abs(float(ref["loss"][0].detach()))
    named = dict(model.named_parameters())
    live = [k for k in sd if sdr[k].grad is not None]
    assert all(named[k].grad is not None for k in live)
    assert all(named[k].grad is None for k in sd if sdr[k].grad is None)   # stage-4 fuse rows 1..3 under 'top'
    gm = torch.cat([named[k].grad.reshape(-1) for k in live])
    gr = torch.cat([sdr[k].grad.reshape(-1) for k in live])
    assert float(torch.dot(gm, gr) / (gm.norm() * gr.norm())) > 0.97
    # a plain torch optimiser can step the flattened parameters
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    before = named["pose_head.tasks.0.hm.2.weight"].detach().clone()
    opt.step()
    assert not torch.equal(before, named["pose_head.tasks.0.hm.2.weight"].detach())
    # ---- inference call: list of {'keypoints': [(id,x,y,z,score)...], 'metadata': meta}
    model.load_state_dict(sd)
    with torch.no_grad():
        preds = model(ex, return_loss=False)
    assert len(preds) == 2 and set(preds[0]) == {"keypoints", "metadata"}
    assert preds[1]["metadata"] == ex["meta"][1]
    assert len(preds[0]["keypoints"]) == 15 and len(preds[0]["keypoints"][0]) == 5


def test_standalone_backbone_and_head_forward():
    name = "hr3d"
    arch, fin, fout, fuse, heads, weight, cw = O.MODEL_CONFIGS[name]
    md = configs.model_dict(name)
    bb = registry.build_backbone(md["backbone"])
    hd = registry.build_head(md["pose_head"])
    sd = O.seeded_state_dict(O.param_shapes(arch, fin, fout, fout, heads), seed=1)
    bb.load_state_dict({k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")})
    hd.load_state_dict({k[len("pose_head."):]: v for k, v in sd.items() if k.startswith("pose_head.")})
    x = O.synth_example(1, 1, (8, 16, 16), seed=3)["rdr"]["rdr_tensor"]
    feats = bb(x)
    preds, same = hd(feats)
    with torch.no_grad():
        rf = O.hrnet3d(sd, x, fuse)
        rp, _ = O.center_head(sd, rf)
    assert tuple(feats.shape) == tuple(rf.shape)
    assert float((feats.float() - rf).norm() / rf.norm()) < 3e-2
    for k in ("reg", "hm"):
        assert tuple(preds[0][k].shape) == tuple(rp[0][k].shape)
        assert float((preds[0][k].float() - rp[0][k]).norm() / rp[0][k].norm()) < 4e-2


def test_mpjpe_matches_reference_vectors(golden):
    from rt_pose_amd.evaluate import abs_pjpe, evaluate, pjpe
    np.testing.assert_allclose(abs_pjpe(golden["pjpe.pred"], golden["pjpe.gt"]), golden["pjpe.abs"], rtol=1e-12)
    np.testing.assert_allclose(pjpe(golden["pjpe.pred"], golden["pjpe.gt"]), golden["pjpe.rel"], rtol=1e-12)
    pred, gt = golden["pjpe.pred"], golden["pjpe.gt"]
    det = {"s1/0/0": {"keypoints": [(i, *pred[i], 0.9) for i in range(15)]},
           "s1/1/1": {"keypoints": [(i, *gt[i], 0.9) for i in range(15)]},
           "s2/0/0": {"keypoints": [(i, *pred[i], 0.9) for i in range(15)]}}
    g = {"s1": {"0": [{"pose": gt.tolist()}], "1": [{"pose": gt.tolist()}]}, "s2": {"0": [{"pose": gt.tolist()}]}}
    res = evaluate(det, g)
    m = float(np.mean(golden["pjpe.rel"])) * 1000
    assert res["seq_results"]["s1"]["MPJPE"] == pytest.approx(m / 2) and res["seq_results"]["s2"]["MPJPE"] == pytest.approx(m)
    assert res["results"]["MPJPE"] == pytest.approx(0.75 * m)


def test_registry_behaves_like_the_reference_registry_class():
    """Differential test against the reference's own det3d/utils/registry.py (loaded at file level; it imports nothing but inspect
    and det3d.torchie.is_str): the same operations on both Registry classes give the same results and the same exception TYPES --
    registration (decorator without parens), duplicate, non-class, lookup, build_from_cfg with default_args, unknown type, non-dict
    cfg, non-str / non-class type."""
    import importlib.util
    import sys
    import types
    ref_path = "/root/reference/det3d/utils/registry.py"
    if not os.path.exists(ref_path):
        pytest.skip("needs /root/reference (authoring container)")
    before = dict(sys.modules)
    try:
        tor = types.ModuleType("det3d.torchie")
        tor.is_str = lambda x: isinstance(x, str)
        pkg = types.ModuleType("det3d")
        pkg.__path__ = []
        pkg.torchie = tor
        sys.modules.update({"det3d": pkg, "det3d.torchie": tor})
        spec = importlib.util.spec_from_file_location("ref_registry", ref_path)
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)
    finally:
        for k in list(sys.modules):
            if k not in before:
                del sys.modules[k]
        sys.modules.update(before)

    def script(Reg, build):
        out = []

        def attempt(f):
            try:
                out.append(("ok", f()))
            except Exception as e:   # noqa: BLE001 -- the exception TYPE is the observable
                out.append(("raise", type(e).__name__))
        reg = Reg("thing")

        class A:
            def __init__(self, x=1, y=2):
                self.x, self.y = x, y

        attempt(lambda: reg.register_module(A) is A)
        attempt(lambda: reg.register_module(A))                                   # duplicate
        attempt(lambda: reg._register_module(lambda: 0))                          # not a class
        attempt(lambda: (reg.get("A") is A, reg.get("B")))
        attempt(lambda: reg.name)
        attempt(lambda: sorted(reg.module_dict))
        attempt(lambda: (lambda o: (o.x, o.y))(build(dict(type="A", x=5), reg, dict(y=7, x=9))))
        attempt(lambda: build(dict(type=A), reg).x)
        attempt(lambda: build(dict(type="Nope"), reg))
        attempt(lambda: build(["type"], reg))
        attempt(lambda: build(dict(x=1), reg))
        attempt(lambda: build(dict(type=3), reg))
        attempt(lambda: build(dict(type="A"), reg, default_args=[1]))
        return out

    assert script(Registry, build_from_cfg) == script(ref.Registry, ref.build_from_cfg)
