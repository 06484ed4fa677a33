"""BASELINE config 4 at model level on a real MI355X: HRRadarPose + the DCN head (dcn_head=True: two FeatureAdaption modules,
center_head.py:24-62, per (frame, z) slice in front of the SepHead towers -- SURVEY.md 8d C4) on the HIP kernels: the plan
hands the bf16 channels-last feature to the fp32 NCHW deformable-convolution operator (include/rtp.h section D) and back.
Checked against the composition oracle.hrradarpose_ref + oracle.dcn_ref.  PARITY UNPINNED for the deformable convolution
itself (oracle/dcn_ref.py header): the reference's DCNSepHead cannot run on a 5-D feature, and its CUDA op cannot be built
here."""
import numpy as np
import pytest
import torch

from oracle import hrradarpose_ref as O
from rt_pose_amd import configs
from rt_pose_amd.engine import FlatParams, PoseEngine
from tests.golden.gen_golden import TEST_CFG
from tests.util import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from rt_pose_amd.backend import HipBackend
    return HipBackend("cuda:0")


@pytest.mark.parametrize("dims,batch", [((4, 8, 16), 2), ((8, 16, 32), 3)])
def test_dcn_head_train_step_vs_oracle_composition(hip, dims, batch):
    arch, fin, fout, fuse, heads, weight, cw = O.MODEL_CONFIGS["hr3d"]
    shapes = O.param_shapes(arch, fin, fout, fout, heads, dcn_head=True)
    assert {k: tuple(v) for k, v in shapes.items()} == {k: tuple(v) for k, v in configs.param_shapes("hr3d_dcn").items()}
    sd = O.seeded_state_dict(shapes, seed=1)
    flat = FlatParams(shapes, hip.alloc)
    flat.load_state_dict(sd)
    eng = PoseEngine(hip, flat.values, arch, fuse, heads, weight, cw, batch, dims, pgrads=flat.grads, test_cfg=TEST_CFG)
    ex = O.synth_example(batch, 1, dims, seed=1234)
    eng.load_input(ex["rdr"]["rdr_tensor"])
    eng.load_targets(ex["rdr"])
    eng.run_forward()
    eng.run_loss_backward()
    torch.cuda.synchronize()
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.radar_pose_net(sdr, ex, fuse, weight, cw)
    ref["loss"][0].backward()
    with torch.no_grad():
        preds, _ = O.center_head(sd, O.hrnet3d(sd, ex["rdr"]["rdr_tensor"], fuse))
    for k in ("reg", "hm"):
        assert rel_err(eng.output(k).float().cpu(), preds[0][k]) < 3e-2, k
    got, want = float(eng.losses()["loss"]), float(ref["loss"][0].detach())
    assert abs(got - want) < 2e-2 * abs(want), (got, want)
    live = [k for k in sd if sdr[k].grad is not None]
    assert set(live) == eng.live_params
    errs = {k: rel_err(flat.grads[k], sdr[k].grad) for k in live if "feature_adapt" in k}
    print("\nadaption-module gradient errors vs the oracle composition:", {k.split("tasks.0.")[1]: round(v, 4) for k, v in errs.items()})
    for k, e in errs.items():
        # each tensor on its own.  The deformable conv's weight sees the bf16 feature through a linear map (a few %); the
        # OFFSET conv's gradient goes through d(bilinear sample)/d(coordinate) = differences of neighbouring bf16 feature
        # values, which amplifies their rounding (2^-9 relative on values whose differences are ~10x smaller)
        assert e < (8e-2 if "conv_adaption" in k else 0.3), (k, e)
    gh = torch.cat([flat.grads[k].detach().float().cpu().reshape(-1) for k in live])
    gr = torch.cat([sdr[k].grad.reshape(-1) for k in live])
    assert float(torch.dot(gh, gr) / (gh.norm() * gr.norm())) > 0.97


def test_dcn_head_through_the_registry_door(hip):
    """CenterHead(dcn_head=True) built by build_detector from the model dict: loss.backward() fills p.grad of the adaption
    modules; inference returns key-points."""
    from rt_pose_amd import registry
    registry.install_det3d_shim()
    from det3d.models import build_detector
    md = configs.model_dict("hr3d_dcn")
    assert md["pose_head"]["dcn_head"] is True
    model = build_detector(md, train_cfg=None, test_cfg=configs.test_cfg())
    names = [n for n, _ in model.named_parameters()]
    assert "pose_head.tasks.0.feature_adapt_cls.conv_adaption.weight" in names
    dims, b = (4, 8, 16), 2
    ex = O.synth_example(b, 1, dims, seed=5)
    exd = {"rdr": {k: (v.to("cuda:0") if torch.is_tensor(v) else [t.to("cuda:0") for t in v]) for k, v in ex["rdr"].items()},
           "meta": ex["meta"]}
    out = model(exd, return_loss=True)
    sum(out["loss"]).backward()
    torch.cuda.synchronize()
    named = dict(model.named_parameters())
    g = named["pose_head.tasks.0.feature_adapt_reg.conv_adaption.weight"].grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().max()) > 0
    with torch.no_grad():
        preds = model(exd, return_loss=False)
    assert len(preds) == b and len(preds[0]["keypoints"]) == 15
