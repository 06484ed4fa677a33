"""Plan logic (rt_pose_amd/graph.py, net.py, engine.py) checked on CPU with the emulated kernels
(tests/emu_backend.py) against the oracle's autograd.  No GPU, no HIP code involved: this pins the fold algebra,
the backward emission and the optimiser wiring; the HIP kernels themselves are checked in the `-m gpu` tests.

exact=True keeps 'bf16' buffers in fp32, so the plan must match the oracle to fp32 round-off
(except a handful of ReLU ties); exact=False rounds like the kernels do and is compared loosely.
"""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import hrradarpose_ref as O
from rt_pose_amd.engine import FlatAdam, FlatParams, PoseEngine, one_cycle
from tests.emu_backend import EmuBackend
from tests.golden.gen_golden import TEST_CFG
from tests.util import rel_err

DIMS = (8, 16, 16)


def make(name, exact, batch=2, train=True):
    arch, fin, fout, fuse, heads, weight, cw = O.MODEL_CONFIGS[name]
    shapes = O.param_shapes(arch, fin, fout, fout, heads)
    sd = O.seeded_state_dict(shapes, seed=1)
    be = EmuBackend(exact=exact)
    flat = FlatParams(shapes, be.alloc)
    flat.load_state_dict(sd)
    eng = PoseEngine(be, flat.values, arch, fuse, heads, weight, cw, batch, DIMS, train=train, pgrads=flat.grads,
                     test_cfg=TEST_CFG)
    ex = O.synth_example(batch, O.ARCHS[arch]["inplanes"], DIMS, seed=1234, one_hm=heads["hm"] == 1)
    return eng, flat, sd, ex, (fuse, weight, cw)


@pytest.mark.parametrize("name", ["hr3d", "hr3d_one_hm_doppler"])
def test_exact_forward_loss_backward(name):
    eng, flat, sd, ex, (fuse, weight, cw) = make(name, exact=True)
    eng.load_input(ex["rdr"]["rdr_tensor"])
    eng.load_targets(ex["rdr"])
    eng.run_forward()
    eng.run_loss_backward()
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.radar_pose_net(sdr, ex, fuse, weight, cw, return_loss=True)
    ref["loss"][0].backward()
    got = eng.losses()
    for k in ("loss", "hm_loss", "loc_loss", "loc_loss_elem", "num_positive"):
        np.testing.assert_allclose(got[k].numpy(), ref[k][0].detach().numpy(), rtol=2e-4, atol=1e-6, err_msg=k)
    live = eng.live_params
    for k in sd:
        if sdr[k].grad is None:
            assert k not in live, k
            assert float(flat.grads[k].abs().max()) == 0.0
        else:
            assert k in live, k
            # a ReLU tie can flip one mask element on these tiny tensors -> 1e-2, not 1e-5
            assert rel_err(flat.grads[k], sdr[k].grad) < 1e-2, (k, rel_err(flat.grads[k], sdr[k].grad))
    errs = [rel_err(flat.grads[k], sdr[k].grad) for k in sd if sdr[k].grad is not None]
    assert np.median(errs) < 2e-4


@pytest.mark.parametrize("name", list(O.MODEL_CONFIGS))
def test_bf16_forward_and_predict(name):
    eng, flat, sd, ex, (fuse, weight, cw) = make(name, exact=False, train=False)
    eng.load_input(ex["rdr"]["rdr_tensor"])
    eng.run_forward()
    eng.run_decode()
    with torch.no_grad():
        feats = O.hrnet3d(sd, ex["rdr"]["rdr_tensor"], fuse)
        preds, _ = O.center_head(sd, feats)
    assert rel_err(eng.features().float(), feats) < 3e-2
    for k in ("reg", "hm"):
        assert rel_err(eng.output(k), preds[0][k]) < 3e-2, k
    # decode parity on the engine's own logits (argmax must agree exactly; coordinates to fp32 round-off)
    mine = eng.keypoints()
    own = [{"reg": eng.output("reg").float(), "hm": eng.output("hm").float()}]
    ref = O.center_head_predict(own, TEST_CFG)
    for a, b in zip(mine, ref):
        np.testing.assert_allclose(np.asarray(a["keypoints"]), np.asarray(b["keypoints"]), rtol=1e-5, atol=1e-5)


def test_bf16_gradients_are_close():
    eng, flat, sd, ex, (fuse, weight, cw) = make("hr3d", exact=False)
    eng.load_input(ex["rdr"]["rdr_tensor"])
    eng.load_targets(ex["rdr"])
    eng.run_forward()
    eng.run_loss_backward()
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.radar_pose_net(sdr, ex, fuse, weight, cw, return_loss=True)
    ref["loss"][0].backward()
    assert abs(float(eng.losses()["loss"]) - float(ref["loss"][0].detach())) < 2e-2 * abs(float(ref["loss"][0].detach()))
    gm = torch.cat([flat.grads[k].reshape(-1) for k in sd if sdr[k].grad is not None])
    gr = torch.cat([sdr[k].grad.reshape(-1) for k in sd if sdr[k].grad is not None])
    cos = float(torch.dot(gm, gr) / (gm.norm() * gr.norm()))
    assert cos > 0.97, cos  # bf16 forward (ReLU-mask flips dominate); fp32-exact mode is checked above
    assert abs(float(gm.norm() / gr.norm()) - 1) < 0.05


def test_optimizer_step_matches_rule():
    eng, flat, sd, ex, (fuse, weight, cw) = make("hr3d", exact=True)
    opt = FlatAdam(eng.be, flat, eng.live_params)
    sdr = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd.items())
    ref_opt = O.AdamTrueWD(list(sdr.values()))
    total = 10
    for step in range(2):
        eng.load_input(ex["rdr"]["rdr_tensor"])
        eng.load_targets(ex["rdr"])
        eng.run_forward()
        eng.run_loss_backward()
        lr, b1 = one_cycle(step, total, 1e-3)
        assert (lr, b1) == pytest.approx(O.one_cycle(step, total, 1e-3))
        opt.set_hyper(lr, b1)
        opt.run()
        for p in sdr.values():
            p.grad = None
        O.radar_pose_net(sdr, ex, fuse, weight, cw)["loss"][0].backward()
        norm = ref_opt.step(lr, b1, max_norm=35.0)
        assert abs(float(opt.norm[0]) - norm) < 1e-2 * norm
    for k in sd:
        d_ref = (sdr[k].detach() - sd[k])
        d_got = (flat.values[k] - sd[k])
        # Adam's first steps are +-lr per element, so compare the parameter DELTAS
        assert rel_err(d_got, d_ref) < 0.08, (k, rel_err(d_got, d_ref))


def test_flat_params_runs():
    shapes = OrderedDict(a=(4,), b=(2, 3), c=(5,), d=(1,))
    fp = FlatParams(shapes, EmuBackend().alloc)
    assert fp.numel == 16
    assert fp.runs({"a", "b", "d"}) == [(0, 10, True), (10, 15, False), (15, 16, True)]


def test_no_tail_switch_keeps_every_deferred_item(monkeypatch):
    """PlanOptions.no_tail (the per-item A/B switch, RTP_PLAN="no_tail") must emit one launch per deferred item, not drop them:
    parameter gradients are identical with and without it."""
    grads = {}
    for flag in (None, "1"):
        if flag:
            monkeypatch.setenv("RTP_PLAN", "no_tail")
        else:
            monkeypatch.delenv("RTP_PLAN", raising=False)
        eng, flat, sd, ex, _ = make("hr3d", exact=True)
        eng.load_input(ex["rdr"]["rdr_tensor"])
        eng.load_targets(ex["rdr"])
        eng.run_forward()
        eng.run_loss_backward()
        grads[flag] = flat.g.clone()
        ntail = sum(1 for L in eng.bwd if L.tag == "tail")
        assert ntail > 50 if flag else ntail <= 3, ntail
    assert float(grads[None].abs().max()) > 0
    assert torch.equal(grads[None], grads["1"])


def test_dcn_head_plan_matches_oracle_composition():
    """BASELINE config 4 at model level: hr3d + dcn_head (two FeatureAdaption modules per (frame, z) slice in front of the
    towers) against oracle.hrradarpose_ref composed with oracle.dcn_ref -- plan wiring, parameter gradients of the adaption
    modules, the extra gradient paths into the feature.  (The DCN arithmetic itself is parity-unpinned: oracle/dcn_ref.py.)"""
    arch, fin, fout, fuse, heads, weight, cw = O.MODEL_CONFIGS["hr3d"]
    shapes = O.param_shapes(arch, fin, fout, fout, heads, dcn_head=True)
    sd = O.seeded_state_dict(shapes, seed=1)
    be = EmuBackend(exact=True)
    flat = FlatParams(shapes, be.alloc)
    flat.load_state_dict(sd)
    dims = (4, 8, 16)
    eng = PoseEngine(be, flat.values, arch, fuse, heads, weight, cw, 2, dims, pgrads=flat.grads, test_cfg=TEST_CFG)
    ex = O.synth_example(2, 1, dims, seed=1234)
    eng.load_input(ex["rdr"]["rdr_tensor"])
    eng.load_targets(ex["rdr"])
    eng.run_forward()
    eng.run_loss_backward()
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.radar_pose_net(sdr, ex, fuse, weight, cw)
    ref["loss"][0].backward()
    assert abs(float(eng.losses()["loss"]) - float(ref["loss"][0])) < 2e-4 * abs(float(ref["loss"][0]))
    adapt = [k for k in sd if "feature_adapt" in k]
    assert len(adapt) == 6 and all(k in eng.live_params for k in adapt)
    for k in sd:
        if sdr[k].grad is not None:
            assert rel_err(flat.grads[k], sdr[k].grad) < 1e-2, (k, rel_err(flat.grads[k], sdr[k].grad))
    assert float(flat.grads["pose_head.tasks.0.feature_adapt_cls.conv_offset.weight"].abs().max()) > 0


def test_lidar_fusion_head_matches_oracle_concat():
    """BASELINE config 5 at model level (this repo's composition, no reference counterpart): the towers' first conv
    reads concat(radar feature, dense LiDAR grid).  The plan runs it as two input-channel slices of the LDS-tiled conv (the
    second slice has 4 real channels) and never builds the concatenation; the oracle concatenates.  Loss, every parameter
    gradient (including the 36-input-channel tower weights) and the data gradient into the backbone must agree."""
    from rt_pose_amd import synth
    arch, fin, fout, fuse, heads, weight, cw = O.MODEL_CONFIGS["hr3d"]
    C_L = 4
    shapes = O.param_shapes(arch, fin, fout, fout + C_L, heads)
    assert shapes["pose_head.tasks.0.hm.0.weight"] == (32, 36, 3, 3, 3)
    sd = O.seeded_state_dict(shapes, seed=1)
    be = EmuBackend(exact=True)
    flat = FlatParams(shapes, be.alloc)
    flat.load_state_dict(sd)
    dims = (4, 8, 16)
    eng = PoseEngine(be, flat.values, arch, fuse, heads, weight, cw, 2, dims, pgrads=flat.grads, test_cfg=TEST_CFG, lidar_channels=C_L)
    tags = [L.tag for L in eng.fwd]
    assert "pack:lidar" in tags and "conv:head.hm.0.0" in tags and "conv:head.hm.0.1" in tags
    ex = O.synth_example(2, 1, dims, seed=1234)
    ex["rdr"]["lidar_grid"] = synth.lidar_grid(2, C_L, dims, seed=5, occupancy=0.3)
    eng.load_input(ex["rdr"]["rdr_tensor"])
    eng.load_lidar(ex["rdr"]["lidar_grid"])
    eng.load_targets(ex["rdr"])
    eng.run_forward()
    eng.run_loss_backward()
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.radar_pose_net(sdr, ex, fuse, weight, cw)
    ref["loss"][0].backward()
    assert abs(float(eng.losses()["loss"]) - float(ref["loss"][0])) < 2e-4 * abs(float(ref["loss"][0]))
    for k in sd:
        if sdr[k].grad is not None:
            assert rel_err(flat.grads[k], sdr[k].grad) < 1e-2, (k, rel_err(flat.grads[k], sdr[k].grad))
    gw = flat.grads["pose_head.tasks.0.reg.0.weight"]
    assert float(gw[:, 32:].abs().max()) > 0, "the LiDAR input channels of the tower weights receive gradient"
    # without the LiDAR stream the result differs (the grid really is read)
    ex2 = {"rdr": dict(ex["rdr"], lidar_grid=torch.zeros_like(ex["rdr"]["lidar_grid"])), "meta": ex["meta"]}
    ref2 = O.radar_pose_net(sd, ex2, fuse, weight, cw)
    assert abs(float(ref2["loss"][0]) - float(ref["loss"][0])) > 1e-6


def test_fused_stride2_data_gradient_route(monkeypatch):
    monkeypatch.setenv("RTP_PLAN", "fused_s2=1")   # built and tested, off by default (graph.ConvOp._fusable)
    """Volumes wide enough for the stride-2 parity-class kernel (Wo % 16 == 0): the first conv of a fuse chain from branch 0
    is the last gradient contribution to the branch-0 output and writes its finished gradient (weight gradient -> slab
    contraction -> data gradient); gradients still equal the oracle's."""
    arch, fin, fout, fuse, heads, weight, cw = O.MODEL_CONFIGS["hr3d"]
    shapes = O.param_shapes(arch, fin, fout, fout, heads)
    sd = O.seeded_state_dict(shapes, seed=1)
    dims = (4, 8, 32)
    be = EmuBackend(exact=True)
    flat = FlatParams(shapes, be.alloc)
    flat.load_state_dict(sd)
    eng = PoseEngine(be, flat.values, arch, fuse, heads, weight, cw, 2, dims, pgrads=flat.grads)
    tags = [L.tag for L in eng.bwd]
    assert "qslab:s2.f10.0" in tags and "qslab:s3.f20.0" in tags, "stride-2 data gradients of the fuse chains take the fused route"
    assert "combine:s2.b0.c3" not in tags and "combine:s3.b0.c3" not in tags, "no fan-in pass left for the branch-0 outputs"
    ex = O.synth_example(2, 1, dims, seed=1234)
    eng.load_input(ex["rdr"]["rdr_tensor"])
    eng.load_targets(ex["rdr"])
    eng.run_forward()
    eng.run_loss_backward()
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.radar_pose_net(sdr, ex, fuse, weight, cw)
    ref["loss"][0].backward()
    for k in sd:
        if sdr[k].grad is not None:
            assert rel_err(flat.grads[k], sdr[k].grad) < 1e-2, (k, rel_err(flat.grads[k], sdr[k].grad))


def test_launch_list_reorderings_keep_dependencies_and_results(monkeypatch):
    """lanes.main_row_first / lanes.hoist_tagged permute a launch list only inside the read/write relations of the original
    order (checked here with an independent walk), refuse a permutation that would break one, and the plan computes the same
    losses and gradients with the forward list in creation order (PlanOptions.fwd_row0_first = 0) and with the backward hoist switched off."""
    from rt_pose_amd.lanes import Launch, _order_preds, hoist_tagged, main_row_first

    def respects(orig, new):
        pos = {id(L): i for i, L in enumerate(new)}
        preds = _order_preds(orig)
        return len(new) == len(orig) and all(pos[id(orig[q])] < pos[id(orig[k])] for k in range(len(orig)) for q in preds[k])

    monkeypatch.setenv("RTP_PLAN", "fwd_row0_first=0")      # creation order (the engine applies main_row_first itself by default)
    eng, flat, sd, ex, _ = make("hr3d", exact=True)
    monkeypatch.delenv("RTP_PLAN")
    fwd2 = main_row_first(eng.fwd)
    assert [L.tag for L in fwd2] != [L.tag for L in eng.fwd] and respects(eng.fwd, fwd2)
    assert [L.tag for L in make("hr3d", exact=True)[0].fwd] == [L.tag for L in fwd2], "the default forward list is the re-ordered one"
    tags = [L.tag for L in fwd2]
    assert tags.index("conv:s3.f01") < tags.index("fuse:s3.row0") < tags.index("conv:s3.f20.0")
    # a hoist across a true dependency is refused: b reads what a writes
    bx, by, bz = torch.zeros(1), torch.zeros(1), torch.zeros(1)
    a, b, c = Launch(None, 0, [], [bx], "w:a"), Launch(None, 1, [bx], [by], "r:b"), Launch(None, 2, [], [bz], "o:c")
    assert hoist_tagged([a, c, b], r"r:b", r"w:a") == [a, c, b]
    assert [L.tag for L in hoist_tagged([a, c, b], r"r:b", r"o:c")] == ["w:a", "r:b", "o:c"]

    def run(plan):
        if plan:
            monkeypatch.setenv("RTP_PLAN", plan)
        e, f, _, x, _ = make("hr3d", exact=True)
        e.load_input(x["rdr"]["rdr_tensor"])
        e.load_targets(x["rdr"])
        e.run_forward()
        e.run_loss_backward()
        monkeypatch.delenv("RTP_PLAN", raising=False)
        return e.losses()["loss"].clone(), {k: v.clone() for k, v in f.grads.items()}

    l0, g0 = run("")
    for env in ("fwd_row0_first=0", "bwd_f10_first=0"):
        l1, g1 = run(env)
        assert torch.equal(l0, l1), env
        assert all(torch.equal(g0[k], g1[k]) for k in g0), env


def test_head_towers_over_a_wide_feature_share_their_launches(monkeypatch):
    """Graph.conv_pair: SepHead's two first convs over the 128-channel feature of the one-heat-map configs (center_head.py:86-93)
    run 64 wide -- one launch per 64-channel slice forward, one per 64 input channels for the data gradient, which writes the SUM
    over both towers -- and give what the per-tower slice route (PlanOptions.pair_heads = 0) gives."""
    def grads(paired):
        monkeypatch.setenv("RTP_PLAN", "pair_heads=%d" % int(paired))
        eng, flat, sd, ex, _ = make("hr3d_one_hm_doppler", exact=True)
        eng.load_input(ex["rdr"]["rdr_tensor"])
        eng.load_targets(ex["rdr"])
        eng.run_forward()
        eng.run_loss_backward()
        return [L.tag for L in eng.fwd + eng.bwd], {k: v.clone() for k, v in flat.grads.items()}, eng.losses()
    tags, g1, l1 = grads(True)
    assert [t for t in tags if t.startswith("conv:head.reg.0")] == ["conv:head.reg.0+head.hm.0.0", "conv:head.reg.0+head.hm.0.1"]
    assert [t for t in tags if t.startswith("dgrad:head.reg.0")] == ["dgrad:head.reg.0+head.hm.0.0", "dgrad:head.reg.0+head.hm.0.1"]
    assert not any(t.startswith(("conv:head.hm.0", "dgrad:head.hm.0")) for t in tags)
    assert sum(t.startswith("wgrad:head.reg.0.") for t in tags) == 4 and sum(t.startswith("wgrad:head.hm.0.") for t in tags) == 4
    assert "combine:final.sum" not in tags, "one gradient tensor for the feature: no fan-in pass"
    tags0, g0, l0 = grads(False)
    assert sum(t.startswith("conv:head.reg.0.") for t in tags0) == 4 and "combine:final.sum" in tags0
    assert abs(float(l1["loss"]) - float(l0["loss"])) < 1e-5 * abs(float(l0["loss"]))
    for k in g0:
        assert rel_err(g1[k], g0[k]) < 1e-4, k
