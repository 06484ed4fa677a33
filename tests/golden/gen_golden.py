"""Generate the golden vectors by importing the REFERENCE's own hot-path files.

Runs only in the authoring container (needs /root/reference).  The reference's
package ``det3d`` cannot be imported as a whole (addict, spconv, numba, yacs ...
are absent), so its hot-path *files* are loaded one by one with importlib into
synthetic ``det3d.*`` packages -- numba.jit becomes the identity, and the yacs
arch tables are read from the reference file with a 10-line ``CfgNode`` stand-in
(attribute dict).  Nothing from the reference is written to this repo except the
numeric inputs/outputs below (small .npz fixtures) and the parameter name/shape
lists.

    python tests/golden/gen_golden.py           # rewrites tests/golden/*.npz, *.json

Weights come from ``oracle.hrradarpose_ref.seeded_state_dict`` (a by-name seeded
recipe), so fixtures hold only inputs' seeds and the reference's outputs.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import hrradarpose_ref as O  # noqa: E402


def _pkg(name):
    if name not in sys.modules:
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
        if "." in name:
            parent, child = name.rsplit(".", 1)
            setattr(_pkg(parent), child, m)
    return sys.modules[name]


def _load(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    if "." in modname:
        parent, child = modname.rsplit(".", 1)
        setattr(_pkg(parent), child, m)
    spec.loader.exec_module(m)
    return m


def import_reference():
    # stand-ins for absent third-party packages
    numba = types.ModuleType("numba")
    numba.jit = lambda *a, **k: (lambda f: f)
    sys.modules["numba"] = numba

    class CfgNode(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

    yacs = types.ModuleType("yacs")
    yacs_config = types.ModuleType("yacs.config")
    yacs_config.CfgNode = CfgNode
    yacs.config = yacs_config
    sys.modules["yacs"] = yacs
    sys.modules["yacs.config"] = yacs_config

    for p in ["det3d", "det3d.core", "det3d.core.utils", "det3d.torchie", "det3d.torchie.cnn", "det3d.models",
              "det3d.models.losses", "det3d.models.utils", "det3d.models.backbones", "det3d.models.backbones.hr_util",
              "det3d.models.pose_heads", "det3d.utils", "det3d.ops", "det3d.torchie.trainer"]:
        _pkg(p)
    sys.modules["det3d.torchie"].is_str = lambda x: isinstance(x, str)
    sys.modules["det3d.core"].box_torch_ops = None
    reg = _load("det3d.utils.registry", "det3d/utils/registry.py")
    sys.modules["det3d.utils"].Registry = reg.Registry
    sys.modules["det3d.utils"].build_from_cfg = reg.build_from_cfg
    _load("det3d.models.registry", "det3d/models/registry.py")
    _load("det3d.core.utils.circle_nms_jit", "det3d/core/utils/circle_nms_jit.py")
    cu = _load("det3d.core.utils.center_utils", "det3d/core/utils/center_utils.py")
    wi = _load("det3d.torchie.cnn.weight_init", "det3d/torchie/cnn/weight_init.py")
    sys.modules["det3d.torchie.cnn"].kaiming_init = wi.kaiming_init
    misc = _load("det3d.models.utils.misc", "det3d/models/utils/misc.py")
    sys.modules["det3d.models.utils"].Sequential = misc.Sequential
    loss = _load("det3d.models.losses.centernet_loss", "det3d/models/losses/centernet_loss.py")
    _load("det3d.models.backbones.hr_util.common", "det3d/models/backbones/hr_util/common.py")
    cfgs = _load("det3d.models.backbones.hrnet3D_config", "det3d/models/backbones/hrnet3D_config.py")
    hr3d = _load("det3d.models.backbones.hr_util.hr3d", "det3d/models/backbones/hr_util/hr3d.py")
    head = _load("det3d.models.pose_heads.center_head", "det3d/models/pose_heads/center_head.py")
    ev = _load("ref_eval_util", "eval_util.py")
    return dict(center_utils=cu, loss=loss, hr3d=hr3d, cfgs=cfgs, head=head, eval=ev)


class RefHRNet3D(torch.nn.Module):
    """Replays det3d/models/backbones/hrnet3d.py:11-43 around the imported HighResolution3DNet
    (hrnet3d.py itself needs the un-importable det3d.models.builder)."""

    def __init__(self, R, arch, final_in, final_out, final_fuse):
        super().__init__()
        self.backbone = R["hr3d"].HighResolution3DNet(R["cfgs"].MODEL_CONFIGS[arch], full_res_stem=True)
        self.final_conv = torch.nn.Identity() if final_in == final_out else torch.nn.Conv3d(final_in, final_out, 1)
        self.final_fuse = final_fuse

    def forward(self, x_):
        x = self.backbone(x_)
        size = x[0].shape[2:]
        if self.final_fuse == "top":
            return self.final_conv(x[0])
        ups = [torch.nn.functional.interpolate(t, size=size, mode="trilinear", align_corners=True) for t in x[1:]]
        feats = torch.cat([x[0], *ups], 1)
        if self.final_fuse == "conat_conv":
            feats = self.final_conv(feats)
        return feats


class RefNet(torch.nn.Module):
    def __init__(self, R, name):
        super().__init__()
        arch, fin, fout, fuse, heads, weight, cw = O.MODEL_CONFIGS[name]
        self.backbone = RefHRNet3D(R, arch, fin, fout, fuse)
        ncls = heads["hm"]
        tasks = [dict(num_class=ncls, class_names=[f"j{i}" for i in range(ncls)])]
        self.pose_head = R["head"].CenterHead(
            tasks=tasks, in_channels=fout, share_conv_channel=fout, dataset="cruw_pose", weight=weight,
            code_weights=cw, common_heads={"reg": (heads["reg"], 2)}, dcn_head=False)


class _TestCfg(dict):
    __getattr__ = dict.__getitem__


TEST_CFG = dict(  # configs/cruw_pose/hr3d.py:113-131
    post_center_limit_range=[0.7703125, -5.0250000000000234, -1.0875000000000021, 8.0203125, 5.024999999999931, 4.7125],
    score_threshold=0.0, pc_range=[0.7703125, -5.0250000000000234, -1.0875000000000021],
    out_size_factor=[1, 1, 1], voxel_size=[0.0453125, 0.15703125, 0.3625])


def put(out, key, t, limit=40000, nsample=8192):
    """Full tensor when small, otherwise seeded flat samples + moments (tests/test_oracle_golden.py mirrors this)."""
    a = np.asarray(t.detach().numpy() if torch.is_tensor(t) else t)
    if a.size <= limit:
        out[key] = a
        return
    idx = sample_index(a.size, nsample)
    out[key + "#samples"] = a.reshape(-1)[idx]
    out[key + "#moments"] = np.asarray([a.mean(), a.std(), np.abs(a).max()], np.float64)


def sample_index(numel, nsample=8192):
    return np.random.RandomState(numel % 2**31).randint(0, numel, size=nsample)


def clone_example(ex):
    return {"rdr": {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in ex["rdr"].items()},
            "meta": ex["meta"]}


def main():
    R = import_reference()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    out = {}
    schema = {}
    small = (8, 16, 16)

    for name, (arch, fin, fout, fuse, heads, weight, cw) in O.MODEL_CONFIGS.items():
        net = RefNet(R, name)
        ref_sd = net.state_dict()
        shapes = O.param_shapes(arch, fin, fout, fout, heads)
        assert list(ref_sd.keys()) == list(shapes.keys()), (name, set(ref_sd) ^ set(shapes))
        for k in shapes:
            assert tuple(ref_sd[k].shape) == tuple(shapes[k]), (k, ref_sd[k].shape, shapes[k])
        schema[name] = {k: list(v) for k, v in shapes.items()}
        sd = O.seeded_state_dict(shapes, seed=1)
        net.load_state_dict(sd)
        cin = O.ARCHS[arch]["inplanes"]
        one_hm = heads["hm"] == 1
        ex = O.synth_example(2, cin, small, seed=1234, one_hm=one_hm)
        x = ex["rdr"]["rdr_tensor"]

        # G3/G4: backbone levels + fused feature; G5: head
        ys = net.backbone.backbone(x)
        feats = net.backbone(x)
        preds, _ = net.pose_head(feats)
        for i, y in enumerate(ys):
            put(out, f"{name}.bb{i}", y)
        put(out, f"{name}.feats", feats)
        put(out, f"{name}.reg", preds[0]["reg"])
        put(out, f"{name}.hm", preds[0]["hm"])

        # G6: loss dict + grads (loss() applies sigmoid_ in place -> recompute preds)
        net.zero_grad()
        preds, _ = net.pose_head(net.backbone(x))
        exr = clone_example(ex)["rdr"]
        losses = net.pose_head.loss(exr, preds, None)
        losses["loss"][0].backward()
        for k in ("loss", "hm_loss", "loc_loss", "loc_loss_elem", "num_positive"):
            out[f"{name}.loss.{k}"] = np.asarray(losses[k][0].detach().numpy(), np.float64)
        gnames = ["backbone.backbone.layer1.conv2.conv.weight", "backbone.backbone.layer1.conv2.groupnorm.weight",
                  "backbone.backbone.layer1.conv2.groupnorm.bias", "backbone.backbone.stage3.0.fuse_layers.2.0.1.1.weight",
                  "backbone.backbone.stage4.0.fuse_layers.0.3.1.weight", "backbone.backbone.transition3.3.0.0.weight",
                  "pose_head.tasks.0.hm.2.weight", "pose_head.tasks.0.reg.0.bias"]
        params = dict(net.named_parameters())
        for g in gnames:
            put(out, f"{name}.grad.{g}", params[g].grad)
        out[f"{name}.gradnorm"] = np.asarray(
            [float(p.grad.norm()) if p.grad is not None else -1.0 for p in params.values()], np.float64)

        # G7: predict
        with torch.no_grad():
            preds, _ = net.pose_head(net.backbone(x))
            ret = net.pose_head.predict(clone_example(ex), preds, _TestCfg(TEST_CFG))
        out[f"{name}.predict"] = np.asarray([[list(kp) for kp in r["keypoints"]] for r in ret], np.float64)

    # native-shape pin for the primary arch: sampled outputs + moments at [1,1,16,64,160]
    name = "hr3d"
    arch, fin, fout, fuse, heads, weight, cw = O.MODEL_CONFIGS[name]
    net = RefNet(R, name)
    net.load_state_dict(O.seeded_state_dict(O.param_shapes(arch, fin, fout, fout, heads), seed=1))
    ex = O.synth_example(1, 1, (16, 64, 160), seed=1234)
    with torch.no_grad():
        preds, _ = net.pose_head(net.backbone(ex["rdr"]["rdr_tensor"]))
    idx = torch.randint(0, 16 * 64 * 160, (512,), generator=torch.Generator().manual_seed(7))
    for k in ("reg", "hm"):
        t = preds[0][k][0].reshape(preds[0][k].shape[1], -1)
        out[f"native.{k}.samples"] = t[:, idx].numpy()
        out[f"native.{k}.moments"] = np.asarray([float(t.mean()), float(t.abs().max()), float(t.std())])
    out["native.idx"] = idx.numpy()

    # G1: ResNetBlock stem (BASELINE config 1: the 2-layer 3-D conv stem) fwd + grads
    for tag, cin in (("stem1", 1), ("stem32", 32)):
        blk = sys.modules["det3d.models.backbones.hr_util.common"].ResNetBlock(cin, 32, order="gcr")
        shapes = {k: tuple(v.shape) for k, v in blk.state_dict().items()}
        sd = O.seeded_state_dict(shapes, seed=3)
        blk.load_state_dict(sd)
        x = torch.relu(torch.randn(2, cin, 8, 16, 32, generator=torch.Generator().manual_seed(5)) * 0.5 + 0.1)
        x.requires_grad_(True)
        y = blk(x)
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(6))
        y.backward(gy)
        schema[tag] = {k: list(v) for k, v in shapes.items()}
        put(out, f"{tag}.y", y)
        put(out, f"{tag}.gx", x.grad)
        put(out, f"{tag}.gw2", blk.conv2.conv.weight.grad)
        put(out, f"{tag}.ggn3", blk.conv3.groupnorm.weight.grad)

    # G9: label gaussians; G10: PJPE
    cu = R["center_utils"]
    for r in (1, 2):
        d = 2 * r + 1
        out[f"gauss3d.r{r}"] = cu.gaussian3D((d, d, d), sigma=d / 6)
    hm = np.zeros((8, 16, 16), np.float32)
    for c in ((0, 0, 0), (15, 15, 7), (5, 9, 3), (6, 9, 3)):
        cu.draw_gaussian3D(hm, c, 2)
    out["gauss3d.drawn"] = hm
    rng = np.random.RandomState(0)
    pred, gt = rng.randn(15, 3), rng.randn(15, 3)
    out["pjpe.pred"], out["pjpe.gt"] = pred.copy(), gt.copy()
    out["pjpe.abs"] = R["eval"].ABS_PJPE(pred.copy(), gt.copy())
    out["pjpe.rel"] = R["eval"].PJPE(pred.copy(), gt.copy())

    np.savez_compressed(os.path.join(HERE, "hrradarpose_golden.npz"), **out)
    with open(os.path.join(HERE, "param_schema.json"), "w") as f:
        json.dump(schema, f, indent=0)
    print("wrote", len(out), "arrays;", os.path.getsize(os.path.join(HERE, "hrradarpose_golden.npz")) / 1e6, "MB")


if __name__ == "__main__":
    main()
