"""End-to-end parity on a real MI355X: the whole HRRadarPose plan (forward, losses, backward, optimiser) on the
HIP kernels against (a) the golden vectors captured from the reference, (b) the oracle run on the same seeded
inputs, (c) the emulated plan; plus size-independent properties at the dataset-native shape [B,1,16,64,160].

Stated tolerances (bf16 storage, fp32 accumulation, vs the reference's fp32): logits norm-wise 3e-2 and
max-abs 0.25; losses 2e-2 relative; parameter-gradient cosine > 0.97 vs fp32 autograd, > 0.98 vs the emulated
plan; decoded key-points: identical argmax voxel on the engine's own logits.
"""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import hrradarpose_ref as O
from rt_pose_amd.engine import FlatAdam, FlatParams, PoseEngine, one_cycle
from tests.emu_backend import EmuBackend
from tests.golden.gen_golden import TEST_CFG
from tests.util import check_golden_like_emulation, check_golden, rel_err

pytestmark = pytest.mark.gpu
DIMS = (8, 16, 16)


@pytest.fixture(scope="module")
def hip():
    from rt_pose_amd.backend import HipBackend
    return HipBackend("cuda:0")


def make(be, name, batch, dims, train=True, seed=1):
    arch, fin, fout, fuse, heads, weight, cw = O.MODEL_CONFIGS[name]
    shapes = O.param_shapes(arch, fin, fout, fout, heads)
    sd = O.seeded_state_dict(shapes, seed=seed)
    flat = FlatParams(shapes, be.alloc)
    flat.load_state_dict(sd)
    eng = PoseEngine(be, flat.values, arch, fuse, heads, weight, cw, batch, dims, train=train, pgrads=flat.grads,
                     test_cfg=TEST_CFG)
    return eng, flat, sd


def cat_grads(flat, names):
    return torch.cat([flat.grads[k].detach().float().cpu().reshape(-1) for k in names])


@pytest.mark.parametrize("name", list(O.MODEL_CONFIGS))
def test_forward_vs_golden_and_oracle(hip, name, golden):
    eng, flat, sd = make(hip, name, 2, DIMS, train=False)
    arch, fin, fout, fuse, heads, weight, cw = O.MODEL_CONFIGS[name]
    ex = O.synth_example(2, O.ARCHS[arch]["inplanes"], DIMS, seed=1234, one_hm=heads["hm"] == 1)
    eng.load_input(ex["rdr"]["rdr_tensor"])
    eng.run_forward()
    eng.run_decode()
    torch.cuda.synchronize()
    # bounds derived from the emulated bf16 plan (same rounding points, CPU arithmetic): the kernels must be as close to the
    # reference's values as the precision choice itself is -- norm-wise and per element (round-2 review: no more atol = 0.25)
    emu, _, _ = make(EmuBackend(), name, 2, DIMS, train=False)
    emu.load_input(ex["rdr"]["rdr_tensor"])
    emu.run_forward()
    for k in ("reg", "hm"):
        r_h, r_e = check_golden_like_emulation(golden, "%s.%s" % (name, k), eng.output(k).float().cpu().contiguous(), emu.output(k).float().contiguous())
        print("%s.%s: rms error vs the reference's values: kernels %.4g, emulated plan %.4g" % (name, k, r_h, r_e))
    check_golden_like_emulation(golden, "%s.feats" % name, eng.features().float().cpu().contiguous(), emu.features().float().contiguous())
    with torch.no_grad():
        preds, _ = O.center_head(sd, O.hrnet3d(sd, ex["rdr"]["rdr_tensor"], fuse))
    for k in ("reg", "hm"):
        assert rel_err(eng.output(k).float().cpu(), preds[0][k]) < 3e-2, k
    own = [{"reg": eng.output("reg").float().cpu(), "hm": eng.output("hm").float().cpu()}]
    for a, b in zip(eng.keypoints(), O.center_head_predict(own, TEST_CFG)):
        np.testing.assert_allclose(np.asarray(a["keypoints"]), np.asarray(b["keypoints"]), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("name", ["hr3d", "hr3d_one_hm_doppler"])
def test_train_step_vs_oracle_and_emulated_plan(hip, name):
    arch, fin, fout, fuse, heads, weight, cw = O.MODEL_CONFIGS[name]
    ex = O.synth_example(2, O.ARCHS[arch]["inplanes"], DIMS, seed=1234, one_hm=heads["hm"] == 1)
    results = {}
    for tag, be in (("hip", hip), ("emu", EmuBackend())):
        eng, flat, sd = make(be, name, 2, DIMS)
        eng.load_input(ex["rdr"]["rdr_tensor"])
        eng.load_targets(ex["rdr"])
        eng.run_forward()
        eng.run_loss_backward()
        if tag == "hip":
            torch.cuda.synchronize()
        results[tag] = (eng, flat, {k: float(v.float().sum()) for k, v in eng.losses().items()})
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.radar_pose_net(sdr, ex, fuse, weight, cw)
    ref["loss"][0].backward()
    live = [k for k in sd if sdr[k].grad is not None]
    eng, flat, losses = results["hip"]
    assert set(live) == eng.live_params
    for k in ("loss", "hm_loss", "loc_loss"):
        assert abs(losses[k] - float(ref[k][0].detach())) < 2e-2 * abs(float(ref[k][0].detach())) + 1e-4, k
        assert abs(losses[k] - results["emu"][2][k]) < 1.5e-2 * abs(losses[k]) + 1e-4, k
    gh, ge = cat_grads(flat, live), cat_grads(results["emu"][1], live)
    gr = torch.cat([sdr[k].grad.reshape(-1) for k in live])
    cos = lambda a, b: float(torch.dot(a, b) / (a.norm() * b.norm()))
    assert cos(gh, ge) > 0.98, cos(gh, ge)  # two bf16 evaluations differing only in summation order (ReLU-mask flips)
    assert cos(gh, gr) > 0.97, cos(gh, gr)
    assert abs(float(gh.norm() / gr.norm()) - 1) < 0.05
    dead = [k for k in sd if sdr[k].grad is None]
    assert all(float(flat.grads[k].abs().max()) == 0 for k in dead)


@pytest.mark.parametrize("name,dims,batch", [("hr3d", (8, 24, 40), 3), ("hr3d", (24, 8, 16), 1), ("hr3d_one_hm", (16, 16, 48), 1),
                                             ("hr3d_one_hm_doppler", (8, 8, 8), 3)])
def test_train_step_other_shapes(hip, name, dims, batch):
    """Volumes with other aspect ratios (non-tiled widths, the smallest volume the three stride-2 levels allow) and odd
    batch sizes against the oracle's fp32 autograd: loss and parameter-gradient direction (24 such combinations were
    clean when this was added)."""
    arch, fin, fout, fuse, heads, weight, cw = O.MODEL_CONFIGS[name]
    ex = O.synth_example(batch, O.ARCHS[arch]["inplanes"], dims, seed=4321, one_hm=heads["hm"] == 1)
    eng, flat, sd = make(hip, name, batch, dims)
    eng.load_input(ex["rdr"]["rdr_tensor"])
    eng.load_targets(ex["rdr"])
    eng.run_forward()
    eng.run_loss_backward()
    torch.cuda.synchronize()
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.radar_pose_net(sdr, ex, fuse, weight, cw)
    ref["loss"][0].backward()
    live = [k for k in sd if sdr[k].grad is not None]
    gh, gr = cat_grads(flat, live), torch.cat([sdr[k].grad.reshape(-1) for k in live])
    loss, lref = float(eng.losses()["loss"].float().sum()), float(ref["loss"][0].detach())
    assert abs(loss - lref) < 3e-2 * abs(lref) + 1e-4
    assert float(torch.dot(gh, gr) / (gh.norm() * gr.norm())) > 0.96


def test_native_shape_properties(hip):
    """[2,1,16,64,160]: determinism, batch independence (GroupNorm is per-sample), loss decreases under the
    reference's optimiser rule, oracle agreement of the forward on one frame."""
    dims = (16, 64, 160)
    eng, flat, sd = make(hip, "hr3d", 2, dims)
    ex = O.synth_example(2, 1, dims, seed=1234)
    x = ex["rdr"]["rdr_tensor"]
    eng.load_input(x)
    eng.load_targets(ex["rdr"])
    eng.run_forward()
    torch.cuda.synchronize()
    hm1 = eng.output("hm").float().cpu().clone()
    eng.run_forward()
    torch.cuda.synchronize()
    assert torch.equal(hm1, eng.output("hm").float().cpu()), "forward is deterministic"
    # frame 0 alone == frame 0 in the batch (swap the frames, outputs swap)
    eng.load_input(x.flip(0))
    eng.run_forward()
    torch.cuda.synchronize()
    assert torch.equal(hm1.flip(0), eng.output("hm").float().cpu()), "frames are independent"
    with torch.no_grad():
        preds, _ = O.center_head(sd, O.hrnet3d(sd, x[:1], "top"))
    assert rel_err(hm1[:1], preds[0]["hm"]) < 3e-2
    # a few optimiser steps reduce the loss
    eng.load_input(x)
    opt = FlatAdam(hip, flat, eng.live_params)
    hist = []
    for step in range(6):
        eng.run_forward()
        eng.run_loss_backward()
        lr, b1 = one_cycle(step, 20, 1e-3)
        opt.set_hyper(lr, b1)
        opt.run()
        hist.append(float(eng.losses()["loss"]))
    assert all(np.isfinite(hist)) and hist[-1] < hist[0], hist


_GRAPH_CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from rt_pose_amd import synth
from rt_pose_amd.trainer import DataParallelTrainer
dims = (8, 16, 32)
ex = synth.make_batch(2, 1, dims, seed=77)
tr = DataParallelTrainer("hr3d", 2, dims, total_steps=20, use_graph=%s, seed=3)
tr.load(ex)
losses = []
for _ in range(3):
    tr.step()
    torch.cuda.synchronize()
    losses.append(float(tr.losses()["loss"]))
np.savez(sys.argv[1], losses=np.array(losses), p=tr.flat.p.float().cpu().numpy())
print("ok")
"""


def test_graph_replay_equals_eager_lane_mode(tmp_path):
    """DataParallelTrainer: a captured-and-replayed HIP graph (lanes folded onto fewer streams) and eager replay on one
    stream per lane walk the same plan -- same losses and parameters after three steps up to the run-to-run float noise of
    the class-sum atomics.  Each mode runs in a child process: hipStreamEndCapture has crashed the process on some
    fork/join shapes (ROCm 7.2), and a crash there must fail THIS test, not take the session down."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = []
    for use_graph in (False, True):
        f = str(tmp_path / ("g%d.npz" % use_graph))
        r = subprocess.run([sys.executable, "-c", _GRAPH_CHILD % (root, use_graph), f], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, "use_graph=%s child failed (rc %d):\n%s" % (use_graph, r.returncode, r.stderr[-2000:])
        d = np.load(f)
        out.append((d["losses"], torch.from_numpy(d["p"])))
    (l0, p0), (l1, p1) = out
    assert np.allclose(l0, l1, rtol=1e-4), (l0, l1)
    assert rel_err(p1, p0) < 1e-4


@pytest.mark.parametrize("env", [{"defer_wg": ""}, {"lazy_coef": 0}, {"fuse_stats": 0, "fused_fold": 0}])
def test_schedule_and_fusion_switches_do_not_change_the_result(hip, env):
    """The plan-level choices of round 2 only move work between launches or change where a launch is issued (rt_pose_amd.options.
    PlanOptions): the weight-gradient lane on the main lane (defer_wg), GroupNorm-backward coefficients in the combine's prologue
    (lazy_coef), statistics from the fuse rows and the fold inside the tiled conv (fuse_stats, fused_fold).  One train step
    with each switch flipped must give the default plan's loss and parameter gradients up to summation-order noise."""
    from rt_pose_amd.options import PlanOptions
    name, dims, batch = "hr3d", (8, 16, 32), 2
    arch, fin, fout, fuse, heads, weight, cw = O.MODEL_CONFIGS[name]
    shapes = O.param_shapes(arch, fin, fout, fout, heads)
    sd = O.seeded_state_dict(shapes, seed=1)
    ex = O.synth_example(batch, 1, dims, seed=1234)

    def one(options=None):
        flat = FlatParams(shapes, hip.alloc)
        flat.load_state_dict(sd)
        eng = PoseEngine(hip, flat.values, arch, fuse, heads, weight, cw, batch, dims, pgrads=flat.grads, options=options)
        eng.load_input(ex["rdr"]["rdr_tensor"])
        eng.load_targets(ex["rdr"])
        eng.run_forward()
        eng.run_loss_backward()
        torch.cuda.synchronize()
        return float(eng.losses()["loss"]), flat.g.detach().float().cpu().clone(), [L.tag for L in eng.bwd], [L.tag for L in eng.fwd]

    l0, g0, bwd0, fwd0 = one()
    l1, g1, bwd1, fwd1 = one(PlanOptions(**env))
    assert (bwd0, fwd0) != (bwd1, fwd1), "the switch changes the launch lists"
    # forward switches re-round a few folded weights / class-bias sums: two bf16 evaluations of one plan.  Measured against the
    # oracle's fp32 loss 36.4926 on this input: round 3 36.5615 / 36.5524 (default / un-fused fold), round 4 36.5047 / 36.6357 --
    # each within 0.4 % of fp32, up to 0.36 % from each other; backward-only switches agree to 1e-5 below
    assert abs(l0 - l1) <= (8e-3 if "fused_fold" in env else 2e-3) * abs(l0), (l0, l1)
    if "defer_wg" in env or "lazy_coef" in env:      # backward-only: same operands, same arithmetic
        assert rel_err(g1, g0) < 1e-5, rel_err(g1, g0)
    else:   # forward statistics summed in another order: a few folded weights move by a bf16 ulp, ReLU masks of single voxels
            # flip -- the same gate as two bf16 evaluations of one plan elsewhere in this file
        cos = float(torch.dot(g0, g1) / (g0.norm() * g1.norm()))
        print("\nforward switches: gradient cosine %.5f, norm ratio %.4f, loss %.6f vs %.6f" % (cos, float(g1.norm() / g0.norm()), l1, l0))
        assert cos > 0.98 and abs(float(g1.norm() / g0.norm()) - 1) < 0.05, (cos, rel_err(g1, g0))
