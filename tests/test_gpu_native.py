"""Test reach on a real MI355X that the engine-level tests do not give (round-1 review):

  * the REGISTRY DOOR -- RadarPoseNet / HRNet3D / CenterHead built by build_detector from the reference's model dict
    (through install_det3d_shim) running on HipBackend: model(example, return_loss=True) -> loss.backward() -> p.grad,
    model(example, return_loss=False) -> key-point list (det3d/models/detectors/radar_pose_net.py:36-46,
    det3d/models/pose_heads/center_head.py:232-330);
  * the BENCH WORKLOAD itself -- a train step at the dataset-native shape with B = 8 (one sample per XCD, 32 workgroups
    per sample on the tiled kernels: a different placement path from B = 2) against the oracle's fp32 autograd and
    against the emulated plan, with PER-PARAMETER-TENSOR gates instead of one global cosine.

Per-tensor tolerances (stated, from bf16 rounding): a tensor's gradient is a sum over >= 10^6 voxel products of bf16
activations (relative rounding 2^-9 = 2e-3 each, independent) times gradients that crossed up to ~40 bf16-stored layers;
rounding errors add in quadrature along a chain, ReLU-mask flips add a sparse term.  Large, well-averaged tensors (conv
weights) measure 1-4 % relative error against fp32 autograd; the worst are GroupNorm beta/gamma of low-resolution branches
(32-128 numbers, each a difference of large cancelling sums).  Gates: relative error <= GATE_REL of the tensor's norm
OR absolute error <= GATE_ABS of the largest tensor norm of the model (tiny-norm tensors), cosine >= GATE_COS.
"""
import contextlib
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import hrradarpose_ref as O
from rt_pose_amd import configs, synth
from rt_pose_amd.engine import FlatParams, PoseEngine
from tests.emu_backend import EmuBackend
from tests.util import rel_err

pytestmark = pytest.mark.gpu
NATIVE = configs.NATIVE_DIMS

# per-tensor gates (relative error, cosine) per configuration: vs the oracle's fp32 autograd, and vs the emulated bf16 plan
# (tests/emu_backend.py: same rounding points as the kernels, fp32 CPU arithmetic in between -- differs from the HIP path
# only by summation order and the ReLU / L1-sign decisions that a last-bit difference flips).
#   hr3d: measured worst tensor 0.094 / 0.067 (transition1 / layer1.conv2, round 2).
#   hr3d_one_hm_doppler (and the other one-heat-map configurations): ONE supervised voxel per frame (1 heat-map class + 45 L1
#   offsets there), so the whole gradient field grows out of 8 voxels per batch: no averaging over voxels, a ReLU decided
#   differently near one of them moves a layer-1 tensor by percents.  Measured at B = 8 (round 3): worst tensor 0.28 vs the
#   oracle, 0.195 vs the emulated plan (layer1 / stage2 GroupNorm weights), median 0.022 / 0.016, global cosine 0.9999.
#   Round 2 blamed L1 sign flips at ties; the "+far" variant refutes that: with every regression target moved >= 1.0 away from
#   the oracle's own prediction (far_targets: no sign can flip) the same tensors are off by the same amounts (0.283 / 0.193).
#   What "+far" does gate tightly is the regression tower and the head itself (HEAD_REL): the tensors next to the loss.
#   Round 4: the spread is MEASURED (test_one_heat_map_gate_is_summation_order_spread): two runs of the emulated plan that differ only
#   by fp32 noise of another summation order (5e-7 relative in front of every bf16 rounding) are 0.1765 apart on the worst tensor
#   (median 0.0169) -- the same tensors, the same size as HIP-vs-emulation (0.1774 / 0.0181) -- and eight supervised voxels per frame
#   do not shrink it.  The fixed bounds below (0.25 vs the emulated plan = 1.4 x that spread) stay as coarse guards; the derived
#   gates (1.5 x the spread measured in the same run) are the sharp ones.
GATES = {"hr3d": {"oracle": (0.13, 0.991), "emu": (0.10, 0.994)},
         "hr3d_one_hm": {"oracle": (0.18, 0.985), "emu": (0.13, 0.99)},          # measured 0.135 / 0.097 (Cin = 1: dense radar input)
         "hr3d_one_hm_doppler": {"oracle": (0.35, 0.94), "emu": (0.25, 0.97)},   # measured 0.283 / 0.195
         "hr3d_one_hm_doppler+far": {"oracle": (0.35, 0.94), "emu": (0.25, 0.97)},
         "hr3d_one_hm_doppler_phase": {"oracle": (0.35, 0.94), "emu": (0.25, 0.97)},   # measured 0.260 at B = 2
         "hr3d_one_hm_doppler_phase+far": {"oracle": (0.35, 0.94), "emu": (0.25, 0.97)}}
HEAD_REL = 0.10   # "+far": every pose_head.* tensor within 10 % of the oracle's fp32 autograd
GATE_ABS = 2e-3   # of the model's largest per-tensor gradient norm (tensors whose own norm is tiny)
MEDIAN_REL = {"hr3d": (0.04, 0.04), "hr3d_one_hm": (0.08, 0.06), "hr3d_one_hm_doppler": (0.08, 0.06), "hr3d_one_hm_doppler+far": (0.08, 0.06),
              "hr3d_one_hm_doppler_phase": (0.08, 0.06), "hr3d_one_hm_doppler_phase+far": (0.08, 0.06)}
COSINES = {"hr3d": (0.997, 0.995)}   # global cosine (emu, oracle); the one-heat-map configurations: (0.985, 0.96)


@pytest.fixture(scope="module")
def hip():
    from rt_pose_amd.backend import HipBackend
    return HipBackend("cuda:0")


def to_dev(v, dev):
    if torch.is_tensor(v):
        return v.to(dev)
    if isinstance(v, dict):
        return {k: to_dev(x, dev) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return type(v)(to_dev(x, dev) for x in v)
    return v


def tensor_report(got: dict, want: dict, names):
    """-> rows (name, numel, |want|, rel err, cosine) sorted by rel err descending."""
    rows = []
    for k in names:
        a, b = got[k].detach().double().cpu().reshape(-1), want[k].detach().double().cpu().reshape(-1)
        nb = float(b.norm())
        rel = float((a - b).norm()) / (nb + 1e-30)
        cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))
        rows.append((k, a.numel(), nb, rel, cos, float((a - b).norm())))
    rows.sort(key=lambda r: -r[3])
    return rows


def gate(rows, kind, label, name):
    rel_max, cos_min = GATES[name][kind]
    top = max(r[2] for r in rows)
    fmt = "   %-62s n=%-7d |g|=%.3e rel=%.4f cos=%.5f"
    print("\n[%s] worst 5 parameter tensors vs %s:\n%s" % (label, kind, "\n".join(fmt % r[:5] for r in rows[:5])))
    bad = [r for r in rows if not ((r[3] <= rel_max and r[4] >= cos_min) or r[5] <= GATE_ABS * top)]
    med = float(np.median([r[3] for r in rows]))
    return med, ["%s vs %s: %s" % (label, kind, fmt % r[:5]) for r in bad]


# ------------------------------------------------------------------------------------------------ registry door
@pytest.mark.parametrize("name", ["hr3d", "hr3d_one_hm_doppler"])
def test_registry_door_on_hip(hip, name):
    """build_detector(model dict) -> RadarPoseNet on HipBackend: the reference's call convention end to end, compared with
    the PoseEngine driven directly (same kernels: equal up to the class-sum atomics) and with the oracle."""
    from rt_pose_amd import registry
    registry.install_det3d_shim()
    from det3d.models import build_detector   # the name a reference tools/train.py imports

    dims, b = (8, 16, 32), 2
    arch, fin, fout, fuse, heads, weight, cw = O.MODEL_CONFIGS[name]
    spec = configs.spec(name)
    model = build_detector(configs.model_dict(name), train_cfg=None, test_cfg=configs.test_cfg())
    sd = O.seeded_state_dict(O.param_shapes(arch, fin, fout, fout, heads), seed=1)
    model.load_state_dict(sd)
    ex = O.synth_example(b, spec["cin"], dims, seed=1234, one_hm=heads["hm"] == 1)
    exd = to_dev(ex, "cuda:0")
    # ---- training call
    out = model(exd, return_loss=True)
    assert set(out.keys()) == {"loss", "hm_loss", "loc_loss", "loc_loss_elem", "num_positive"}
    loss = sum(out["loss"])
    assert loss.requires_grad and loss.is_cuda
    loss.backward()
    torch.cuda.synchronize()
    named = dict(model.named_parameters())
    assert all(p.is_cuda for p in named.values()), "parameters were re-pointed at the flat device buffer"
    # the same step on the engine directly
    flat = FlatParams(O.param_shapes(arch, fin, fout, fout, heads), hip.alloc)
    flat.load_state_dict(sd)
    eng = PoseEngine(hip, flat.values, arch, fuse, heads, weight, cw, b, dims, pgrads=flat.grads, test_cfg=configs.test_cfg())
    eng.load_input(ex["rdr"]["rdr_tensor"])
    eng.load_targets(ex["rdr"])
    eng.run_forward()
    eng.run_loss_backward()
    torch.cuda.synchronize()
    el = eng.losses()
    for k in ("loss", "hm_loss", "loc_loss"):
        got = float(sum(out[k]).detach().float().sum())
        assert abs(got - float(el[k])) <= 1e-5 * abs(float(el[k])) + 1e-7, (k, got, float(el[k]))
    assert torch.allclose(out["loc_loss_elem"][0].float().cpu(), el["loc_loss_elem"].float().cpu(), rtol=1e-5, atol=1e-7)
    assert float(out["num_positive"][0]) == float(el["num_positive"])
    # oracle: loss dict 2 %, live-parameter set, per-tensor gradients
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.radar_pose_net(sdr, ex, fuse, weight, cw)
    ref["loss"][0].backward()
    for k in ("loss", "hm_loss", "loc_loss"):
        want = float(ref[k][0].detach())
        assert abs(float(sum(out[k]).detach().float().sum()) - want) < 2e-2 * abs(want) + 1e-4, k
    live = [k for k in sd if sdr[k].grad is not None]
    assert all(named[k].grad is not None for k in live)
    assert all(named[k].grad is None for k in sd if sdr[k].grad is None)   # stage-4 fuse rows 1..3 under 'top'
    for k in live:   # module door == engine door (identical launches; only the class-sum LDS atomics reorder)
        assert rel_err(named[k].grad, flat.grads[k]) < 1e-5, k
    gm = torch.cat([named[k].grad.detach().float().cpu().reshape(-1) for k in live])
    gr = torch.cat([sdr[k].grad.reshape(-1) for k in live])
    assert float(torch.dot(gm, gr) / (gm.norm() * gr.norm())) > 0.97
    # a plain torch optimiser steps the flattened parameters
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    before = named["pose_head.tasks.0.hm.2.weight"].detach().clone()
    opt.step()
    assert not torch.equal(before, named["pose_head.tasks.0.hm.2.weight"].detach())
    # ---- inference call on the same (reloaded) weights == engine decode == oracle decode of the engine's logits
    model.load_state_dict(sd)
    with torch.no_grad():
        preds = model(exd, return_loss=False)
    torch.cuda.synchronize()
    assert len(preds) == b and set(preds[0]) == {"keypoints", "metadata"}
    assert preds[1]["metadata"] == ex["meta"][1]
    inf = PoseEngine(hip, flat.values, arch, fuse, heads, weight, cw, b, dims, train=False, test_cfg=configs.test_cfg())
    inf.load_input(ex["rdr"]["rdr_tensor"])
    inf.run_forward()
    inf.run_decode()
    torch.cuda.synchronize()
    for a, c in zip(preds, inf.keypoints()):
        np.testing.assert_allclose(np.asarray(a["keypoints"]), np.asarray(c["keypoints"]), rtol=1e-6, atol=1e-6)
    own = [{"reg": inf.output("reg").float().cpu(), "hm": inf.output("hm").float().cpu()}]
    for a, c in zip(preds, O.center_head_predict(own, configs.test_cfg())):
        np.testing.assert_allclose(np.asarray(a["keypoints"]), np.asarray(c["keypoints"]), rtol=1e-5, atol=1e-4)


def test_standalone_backbone_and_head_on_hip(hip):
    """HRNet3D.forward and CenterHead.forward / predict as registered modules of their own (hrnet3d.py:29-56,
    center_head.py:232-238, 272-330)."""
    from rt_pose_amd import registry
    name = "hr3d"
    arch, fin, fout, fuse, heads, weight, cw = O.MODEL_CONFIGS[name]
    md = configs.model_dict(name)
    bb = registry.build_backbone(md["backbone"])
    hd = registry.build_head(md["pose_head"])
    sd = O.seeded_state_dict(O.param_shapes(arch, fin, fout, fout, heads), seed=1)
    bb.load_state_dict({k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")})
    hd.load_state_dict({k[len("pose_head."):]: v for k, v in sd.items() if k.startswith("pose_head.")})
    bb, hd = bb.to("cuda:0"), hd.to("cuda:0")
    x = O.synth_example(1, 1, (8, 16, 32), seed=3)["rdr"]["rdr_tensor"]
    feats = bb(x.to("cuda:0"))
    preds, same = hd(feats)
    kps = hd.predict({"meta": [{"frame": 0}]}, preds, configs.test_cfg())
    torch.cuda.synchronize()
    with torch.no_grad():
        rf = O.hrnet3d(sd, x, fuse)
        rp, _ = O.center_head(sd, rf)
    assert feats.is_cuda and tuple(feats.shape) == tuple(rf.shape)
    assert rel_err(feats.float().cpu(), rf) < 3e-2
    for k in ("reg", "hm"):
        assert tuple(preds[0][k].shape) == tuple(rp[0][k].shape)
        assert rel_err(preds[0][k].float().cpu(), rp[0][k]) < 4e-2
    own = [{k: preds[0][k].float().cpu() for k in ("reg", "hm")}]
    np.testing.assert_allclose(np.asarray(kps[0]["keypoints"]), np.asarray(O.center_head_predict(own, configs.test_cfg())[0]["keypoints"]),
                               rtol=1e-5, atol=1e-4)


# ------------------------------------------------------------------------------------------------ the bench workload
DEV = "cuda:0"   # where the checkers of the native-shape tests run (see oracle_on_device)


def oracle_on_device(sd, ex, fuse, weight, cw, dev=DEV):
    """The oracle's fp32 train step (oracle/hrradarpose_ref.py: plain torch functional code, device-agnostic) with its tensors on
    `dev`: at the dataset-native shape and B = 8 its autograd takes ~40 s on the host's cores and ~3 s through ATen / MIOpen on the
    GPU (still fp32, still the oracle's code -- only the summation order inside torch's conv kernels differs, which the 2 % / per-
    tensor gates below do not see).  The small-shape tests (test_gpu_engine.py, the registry door above) keep the CPU oracle.
    -> (ref loss dict with CPU scalars, {name: CPU grad or None})."""
    sdr = {k: v.detach().to(dev).requires_grad_(True) for k, v in sd.items()}
    ref = O.radar_pose_net(sdr, to_dev(ex, dev), fuse, weight, cw)
    ref["loss"][0].backward()
    torch.cuda.synchronize()
    out = {k: [v.detach().float().cpu() for v in vs] for k, vs in ref.items()}
    grads = {k: (None if v.grad is None else v.grad.detach().float().cpu()) for k, v in sdr.items()}
    del sdr, ref
    torch.cuda.empty_cache()
    return out, grads


def far_targets(sd, ex, fuse, nreg, seed=77):
    """Regression targets at least 1.0 away from the oracle's fp32 prediction at the supervised voxel (random side, 1 .. 4 away):
    the L1 loss's sign(pred - target) is then the same in fp32 and in bf16, for every one of the 45 offsets."""
    with torch.no_grad():
        sdd = {k: v.to(DEV) for k, v in sd.items()}
        preds, _ = O.center_head(sdd, O.hrnet3d(sdd, ex["rdr"]["rdr_tensor"].to(DEV), fuse))
    reg = preds[0]["reg"].float().cpu()                          # [B, nreg, Z, Y, X]
    del sdd, preds
    b = reg.shape[0]
    flatr = reg.reshape(b, nreg, -1)
    ind = ex["rdr"]["ind"][0]                                    # [B, m]
    g = torch.Generator().manual_seed(seed)
    anno = ex["rdr"]["anno_pose"][0].clone()
    m = ind.shape[1]
    at = torch.gather(flatr, 2, ind.view(b, 1, m).expand(b, nreg, m)).permute(0, 2, 1)   # [B, m, nreg]
    side = torch.where(torch.rand(b, m, nreg, generator=g) < 0.5, -1.0, 1.0)
    anno.copy_((at + side * (1.0 + 3.0 * torch.rand(b, m, nreg, generator=g))).reshape(anno.shape))
    ex["rdr"]["anno_pose"] = [anno]
    return ex


@pytest.mark.parametrize("name,b,far", [("hr3d", 8, False), ("hr3d_one_hm", 8, False),
                                        pytest.param("hr3d_one_hm_doppler", 8, False, marks=pytest.mark.slow),
                                        pytest.param("hr3d_one_hm_doppler", 8, True, marks=pytest.mark.slow),
                                        ("hr3d_one_hm_doppler_phase", 2, False),
                                        pytest.param("hr3d_one_hm_doppler_phase", 2, True, marks=pytest.mark.slow)])
def test_native_b8_train_step_per_tensor(hip, name, b, far):
    """The bench's own workload (B = 8 frames of [Cin,16,64,160]; the 64-channel phase configuration at B = 2): loss dict vs the
    oracle (2 %), every live parameter tensor's gradient vs the oracle's fp32 autograd AND vs the emulated bf16 plan, worst five
    reported.  far: regression targets away from the L1 ties (far_targets) -- the tight gates of hr3d then apply.
    hr3d_one_hm_doppler at B = 8 runs by default in test_one_heat_map_gate_is_summation_order_spread, whose gates are DERIVED per
    tensor from the spread measured in the same run (the phase configuration there with RTP_SLOW=1); here the Doppler configurations
    keep the coarse fixed guards of GATES (phase at B = 2 by default, the others with RTP_SLOW=1)."""
    arch, fin, fout, fuse, heads, weight, cw = O.MODEL_CONFIGS[name]
    spec = configs.spec(name)
    shapes = O.param_shapes(arch, fin, fout, fout, heads)
    sd = O.seeded_state_dict(shapes, seed=1)
    ex = synth.make_batch(b, spec["cin"], NATIVE, seed=1234, one_hm=heads["hm"] == 1)
    if far:
        ex = far_targets(sd, ex, fuse, heads["reg"])
        name = name + "+far"
    res = {}
    for tag, be in (("hip", hip), ("emu", EmuBackend(fast=True, device=DEV))):   # the emulated plan: torch ops, on the device
        with (be.on_device() if tag == "emu" else contextlib.nullcontext()):
            flat = FlatParams(shapes, be.alloc)
            flat.load_state_dict(sd)
            eng = PoseEngine(be, flat.values, arch, fuse, heads, weight, cw, b, NATIVE, pgrads=flat.grads)
            eng.load_input(ex["rdr"]["rdr_tensor"])
            eng.load_targets(ex["rdr"])
            eng.run_forward()
            eng.run_loss_backward()
        torch.cuda.synchronize()
        res[tag] = (eng, flat, {k: v.detach().float().cpu() for k, v in eng.losses().items()})
        hm = eng.output("hm").float().cpu()
        res[tag + "_hm"] = hm
        if tag == "emu":   # keep the gradients, drop the emulated plan's device buffers
            res["emu_grads"] = OrderedDict((k, v.detach().float().cpu()) for k, v in flat.grads.items())
            res[tag] = (None, None, res[tag][2])
            del eng, flat
            torch.cuda.empty_cache()
    ref, rgrad = oracle_on_device(sd, ex, fuse, weight, cw)
    eng, flat, losses = res["hip"]
    live = [k for k in sd if rgrad[k] is not None]
    assert set(live) == eng.live_params
    for k in ("loss", "hm_loss", "loc_loss"):
        want = float(ref[k][0].detach())
        assert abs(float(losses[k].sum()) - want) < 2e-2 * abs(want) + 1e-4, (k, float(losses[k].sum()), want)
        assert abs(float(losses[k].sum()) - float(res["emu"][2][k].sum())) < 1e-2 * abs(want) + 1e-4, k
    np.testing.assert_allclose(losses["loc_loss_elem"].numpy(), ref["loc_loss_elem"][0].detach().numpy(), rtol=3e-2, atol=1e-4)
    assert float(losses["num_positive"]) == float(ref["num_positive"][0])
    assert rel_err(res["hip_hm"], res["emu_hm"]) < 1e-2      # same rounding points, different summation order
    got = OrderedDict((k, flat.grads[k]) for k in live)
    med_o, bad_o = gate(tensor_report(got, {k: rgrad[k] for k in live}, live), "oracle", "%s B=%d native" % (name, b), name)
    med_e, bad_e = gate(tensor_report(got, res["emu_grads"], live), "emu", "%s B=%d native" % (name, b), name)
    gh = torch.cat([got[k].detach().float().cpu().reshape(-1) for k in live])
    gr = torch.cat([rgrad[k].reshape(-1) for k in live])
    ge = torch.cat([res["emu_grads"][k].detach().float().reshape(-1) for k in live])
    cos = lambda a, b: float(torch.dot(a, b) / (a.norm() * b.norm()))
    print("median rel: oracle %.4f emu %.4f; global cosine: oracle %.5f emu %.5f; norm ratio %.4f"
          % (med_o, med_e, cos(gh, gr), cos(gh, ge), float(gh.norm() / gr.norm())))
    assert not (bad_o + bad_e), "tensors outside the per-tensor gates:\n" + "\n".join(bad_o + bad_e)
    if far:   # the regression tower and the rest of the head: next to the loss, no sign flips possible -> a tight gate
        rows = tensor_report(got, {k: rgrad[k] for k in live}, [k for k in live if k.startswith("pose_head.")])
        print("head tensors vs oracle (worst 3):\n" + "\n".join("   %-50s rel=%.4f cos=%.5f" % (r[0], r[3], r[4]) for r in rows[:3]))
        assert rows and rows[0][3] <= HEAD_REL, rows[0]
    assert med_o < MEDIAN_REL[name][0] and med_e < MEDIAN_REL[name][1], (med_o, med_e)
    ce, co = COSINES.get(name, (0.985, 0.96))
    assert cos(gh, ge) > ce and cos(gh, gr) > co
    assert abs(float(gh.norm() / gr.norm()) - 1) < (0.02 if name in COSINES else 0.06)
    dead = [k for k in sd if rgrad[k] is None]
    assert all(float(flat.grads[k].abs().max()) == 0 for k in dead)


# ------------------------------------------------------------------------------------------------ MPJPE proxy
def test_keypoint_agreement_bf16_vs_fp32_on_trained_weights():
    """Train the product path until the heat-maps peak, then decode held-out frames with the HIP bf16 plan and with the
    oracle's fp32 forward on the same weights (tests/keypoint_agreement.py): the bf16 path must not move key-points by more
    than a voxel, and the MPJPE difference must stay far inside the north star's 0.5 cm budget."""
    from tests.keypoint_agreement import run
    r = run(steps=400, eval_batches=4)
    print("\nkey-point agreement:", r)
    assert r["loss_history"][-1][1] < 0.5 * r["loss_history"][0][1], "training did not make progress"
    assert r["mean_peak_score"]["oracle_fp32"] > 0.2, "heat-maps did not peak: the comparison would be meaningless"
    assert r["argmax_within_1_voxel"] >= 0.97
    assert r["argmax_agreement"] >= 0.95
    assert abs(r["abs_mpjpe_cm"]["delta"]) < 0.5
    # MPJPE is ROOT-relative (eval_util.py:5-10): one frame whose pelvis decodes to the neighbouring voxel (a near-tie of two
    # heat-map values; 36 cm in z) moves that frame's other 14 joints by a voxel -- the 0.5 cm budget applies to everything
    # else (profiles/r03_keypoint_agreement.json: 0.06 cm over 256 frames)
    slack = r["root_argmax_disagreements"] * max(r["voxel_size_cm"]) * 14.0 / 15.0 / r["frames"]
    assert abs(r["mpjpe_cm"]["delta"]) < 0.5 + slack, (r["mpjpe_cm"], r["root_argmax_disagreements"])


# ------------------------------------------------------------------------------------------------ the one-heat-map gate, measured
def multi_pose_batch(batch, cin, dims, seed, poses):
    """synth.make_batch for the one-heat-map heads with `poses` people per frame (the reference's max_poses > 1 label layout,
    datasets/pipelines/pose.py:407-451: ind / mask / cat / anno_pose of shape [B, max_poses(, 45)], every centre splatted into the one
    heat-map): `poses` supervised voxels per frame instead of one."""
    Z, Y, X = dims
    g = torch.Generator().manual_seed(seed)
    rdr = torch.relu(torch.randn(batch, cin, Z, Y, X, generator=g) * 0.5 + 0.1)
    prof = synth.splat_profile(2)
    hm = torch.zeros(batch, 1, Z, Y, X)
    ind = torch.zeros(batch, poses, dtype=torch.int64)
    for b in range(batch):
        seen = set()
        for m in range(poses):
            while True:
                cz, cy, cx = (int(torch.randint(0, n, (1,), generator=g)) for n in (Z, Y, X))
                if (cz, cy, cx) not in seen:
                    break
            seen.add((cz, cy, cx))
            synth.draw_splat(hm[b, 0], cz, cy, cx, 2, prof)
            ind[b, m] = (cz * Y + cy) * X + cx
    ex = dict(rdr_tensor=rdr, hm=[hm], ind=[ind], mask=[torch.ones(batch, poses, dtype=torch.uint8)],
              cat=[torch.zeros(batch, poses, dtype=torch.int64)], anno_pose=[torch.rand(batch, poses, 45, generator=g) * 16 - 8])
    return {"rdr": ex, "meta": [{"seq": "synth", "frame": b, "rdr_frame": b} for b in range(batch)]}


def _plan_grads(be, name, sd, ex, b, max_objs=None):
    arch, fin, fout, fuse, heads, weight, cw = O.MODEL_CONFIGS[name]
    shapes = O.param_shapes(arch, fin, fout, fout, heads)
    flat = FlatParams(shapes, be.alloc)
    flat.load_state_dict(sd)
    eng = PoseEngine(be, flat.values, arch, fuse, heads, weight, cw, b, NATIVE, pgrads=flat.grads, max_objs=max_objs)
    eng.load_input(ex["rdr"]["rdr_tensor"])
    eng.load_targets(ex["rdr"])
    eng.run_forward()
    eng.run_loss_backward()
    torch.cuda.synchronize()
    return eng, flat


SUM_ORDER_NOISE = 5e-7   # relative fp32 noise that stands for another summation order of a ~10^3-term fp32 dot product (~sqrt(864) * 2^-24 / 3)


@pytest.mark.parametrize("name,b,poses", [("hr3d_one_hm_doppler", 8, 1), pytest.param("hr3d_one_hm_doppler_phase", 2, 1, marks=pytest.mark.slow),
                                          pytest.param("hr3d_one_hm_doppler", 8, 8, marks=pytest.mark.slow)])
def test_one_heat_map_gate_is_summation_order_spread(hip, name, b, poses):
    """hr3d_one_hm_doppler, B = 8: is the 15-20 % worst-tensor distance between the HIP kernels and the emulated plan "summation
    order only"?  Measured instead of asserted by hand: the emulated plan is run TWICE with fp32 noise of
    another summation order in front of every bf16 rounding (EmuBackend(noise=...), two seeds).  Two correct implementations of one
    plan may differ by exactly that spread, so the gates are DERIVED from it: HIP-vs-emulation per tensor <= 1.5 x the worst
    emulation-vs-emulation distance (+ 1 %), medians likewise, and HIP-vs-oracle <= 1.5 x emulation-vs-oracle.
    poses = 1: the shipped label layout, one supervised voxel per frame.  poses = 8: eight people per frame (64 supervised voxels per
    batch, the reference's max_poses > 1 layout) -- round 3 blamed the ONE supervised voxel for the wide spread; the measurement
    says otherwise: with eight the emulation differs from itself by almost as much (the sensitivity belongs to this configuration's
    dense 32-channel input path, layer1 / stage2 tensors), and the kernels stay inside that spread either way.
    Measured (round 4): poses 1: emu-emu worst 0.1765 / median 0.0169, HIP-emu 0.1774 / 0.0181, the same five tensors on top.
    Round 6: (a) the gate is PER TENSOR -- tensor k of the HIP plan may be as far from its nearest emulation as
    2 x the largest distance between any two of THREE emulations on that same tensor (+ 1 %); a mis-scaled small tensor, which a fixed
    35 % bound lets through, is far outside its own spread.  Three samples estimate a tensor's spread coarsely, so up to 2 % of the
    tensors may exceed their own bound as long as none exceeds 1.5 x the worst spread of all.  (b) "it is rounding noise" as a test:
    the same plan with every bf16 buffer kept in fp32 (EmuBackend(exact=True): same fold / un-fold algebra, same launch list) lands
    on the oracle -- the worst tensor within 2 % of the norm and at least 8 x closer than the bf16 emulations are -- so what
    separates the bf16 plan from the oracle is the rounding of stored activations, not the plan."""
    arch, fin, fout, fuse, heads, weight, cw = O.MODEL_CONFIGS[name]
    spec = configs.spec(name)
    sd = O.seeded_state_dict(O.param_shapes(arch, fin, fout, fout, heads), seed=1)
    ex = synth.make_batch(b, spec["cin"], NATIVE, seed=1234, one_hm=True) if poses == 1 else multi_pose_batch(b, spec["cin"], NATIVE, 4321, poses)
    mo = None if poses == 1 else poses
    eng, flat = _plan_grads(hip, name, sd, ex, b, max_objs=mo)
    ref, rgrad = oracle_on_device(sd, ex, fuse, weight, cw)
    live = [k for k in sd if rgrad[k] is not None]
    assert set(live) == eng.live_params
    losses = {k: v.detach().float().cpu() for k, v in eng.losses().items()}
    for k in ("loss", "hm_loss", "loc_loss"):
        want = float(ref[k][0].detach())
        assert abs(float(losses[k].sum()) - want) < 2e-2 * abs(want) + 1e-4, (k, float(losses[k].sum()), want)
    assert float(losses["num_positive"]) == float(ref["num_positive"][0]) == b * poses
    got = OrderedDict((k, flat.grads[k].detach().float().cpu()) for k in live)
    dead = [k for k in sd if rgrad[k] is None]
    assert all(float(flat.grads[k].abs().max()) == 0 for k in dead)
    del eng, flat
    torch.cuda.empty_cache()
    orc = OrderedDict((k, rgrad[k]) for k in live)

    def emulate(**kw):   # (the emulated plan = torch ops on the device: seconds instead of minutes on the host's cores)
        be = EmuBackend(fast=True, device=DEV, **kw)
        with be.on_device():
            e2, fl = _plan_grads(be, name, sd, ex, b, max_objs=mo)
        g = OrderedDict((k, fl.grads[k].detach().float().cpu()) for k in live)
        del e2, fl
        torch.cuda.empty_cache()
        return g
    emu = [emulate(noise=SUM_ORDER_NOISE, seed=seed) for seed in (11, 22, 33)]
    pairs = [(0, 1), (0, 2), (1, 2)]
    r_ee = [tensor_report(emu[i], emu[j], live) for i, j in pairs]
    r_he = [tensor_report(got, e, live) for e in emu]
    r_ho, r_eo = tensor_report(got, orc, live), [tensor_report(e, orc, live) for e in emu]
    top = max(r[2] for r in r_ee[0])
    sig = lambda rows: [r for r in rows if r[5] > GATE_ABS * top]          # tensors that are not tiny-norm
    worst = lambda rows: max(r[3] for r in sig(rows))
    med = lambda rows: float(np.median([r[3] for r in rows]))
    worst_ee, med_ee = max(worst(rows) for rows in r_ee), max(med(rows) for rows in r_ee)
    worst_he, med_he = max(worst(rows) for rows in r_he), max(med(rows) for rows in r_he)
    worst_ho, worst_eo = worst(r_ho), max(worst(rows) for rows in r_eo)
    fmt = "   %-62s n=%-7d |g|=%.3e rel=%.4f cos=%.5f"
    print("\n[%s B=%d, %d pose(s) per frame] emulation (seed 11) vs emulation (seed 22), noise %.0e -- worst 5:\n%s"
          % (name, b, poses, SUM_ORDER_NOISE, "\n".join(fmt % r[:5] for r in r_ee[0][:5])))
    print("HIP vs emulation (seed 11) -- worst 5:\n%s" % "\n".join(fmt % r[:5] for r in r_he[0][:5]))
    print("HIP vs oracle -- worst 3:\n%s" % "\n".join(fmt % r[:5] for r in r_ho[:3]))
    print("worst tensor: emu-emu %.4f  hip-emu %.4f  emu-oracle %.4f  hip-oracle %.4f ; median: emu-emu %.4f  hip-emu %.4f"
          % (worst_ee, worst_he, worst_eo, worst_ho, med_ee, med_he))
    assert worst_he <= 1.5 * worst_ee + 0.01, ("HIP differs from the emulated plan by more than two emulations differ from each other",
                                              worst_he, worst_ee)
    assert med_he <= 1.5 * med_ee + 0.005, (med_he, med_ee)
    assert worst_ho <= 1.5 * worst_eo + 0.01, (worst_ho, worst_eo)
    # ---- (a) per tensor: HIP-to-nearest-emulation against that tensor's own emulation-to-emulation spread
    spread = {k: max(next(r[3] for r in rows if r[0] == k) for rows in r_ee) for k in live}
    tiny = {r[0] for r in r_he[0] if min(next(q[5] for q in rows if q[0] == r[0]) for rows in r_he) <= GATE_ABS * top}
    dist = {k: min(next(r[3] for r in rows if r[0] == k) for rows in r_he) for k in live}
    over = sorted(((dist[k] / (2.0 * spread[k] + 0.01), k, dist[k], spread[k]) for k in live if k not in tiny and dist[k] > 2.0 * spread[k] + 0.01),
                  reverse=True)
    print("per-tensor gate (2 x own spread + 1 %%): %d of %d tensors over it%s"
          % (len(over), len(live), "".join("\n   %-62s hip-emu %.4f  own spread %.4f" % (k, d, sp) for _, k, d, sp in over[:5])))
    assert len(over) <= max(1, len(live) // 50), over[:5]
    assert all(d <= 1.5 * worst_ee + 0.01 for _, _, d, _ in over), over[:5]
    # ---- (b) the same plan with fp32 storage: the spread collapses, i.e. it IS the rounding of the stored activations
    exact = emulate(exact=True)
    r_xo = tensor_report(exact, orc, live)
    worst_xo, med_xo = max(r[3] for r in r_xo), med(r_xo)   # (every tensor: nothing is "tiny" by absolute error here)
    print("fp32-storage plan vs oracle: worst %.5f median %.6f (bf16 emulations vs oracle: worst %.4f)\n%s"
          % (worst_xo, med_xo, worst_eo, "\n".join(fmt % r[:5] for r in r_xo[:3])))
    assert worst_xo <= 0.02 and worst_xo * 8 <= worst_eo, (worst_xo, worst_eo)
    assert med_xo <= 2e-3, med_xo
