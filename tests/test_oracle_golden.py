"""The oracle (oracle/hrradarpose_ref.py) against the golden vectors captured from the reference
(tests/golden/gen_golden.py).  CPU only.  fp32 vs fp32: tolerance 2e-5 abs / 1e-4 rel."""
import numpy as np
import pytest
import torch

from oracle import hrradarpose_ref as O
from tests.golden.gen_golden import TEST_CFG
from tests.util import check_golden

RT, AT = 1e-4, 2e-5
CONFIGS = list(O.MODEL_CONFIGS)


def _setup(name, schema):
    arch, fin, fout, fuse, heads, weight, cw = O.MODEL_CONFIGS[name]
    shapes = O.param_shapes(arch, fin, fout, fout, heads)
    assert {k: list(v) for k, v in shapes.items()} == schema[name]
    sd = O.seeded_state_dict(shapes, seed=1)
    ex = O.synth_example(2, O.ARCHS[arch]["inplanes"], (8, 16, 16), seed=1234, one_hm=heads["hm"] == 1)
    return sd, ex, fuse, weight, cw


@pytest.mark.parametrize("name", CONFIGS)
def test_forward(name, golden, schema):
    sd, ex, fuse, _, _ = _setup(name, schema)
    x = ex["rdr"]["rdr_tensor"]
    with torch.no_grad():
        ys = O.hr3d_backbone(sd, x)
        feats = O.hrnet3d(sd, x, fuse)
        preds, _ = O.center_head(sd, feats)
    for i, y in enumerate(ys):
        check_golden(golden, f"{name}.bb{i}", y, RT, AT)
    check_golden(golden, f"{name}.feats", feats, RT, AT)
    check_golden(golden, f"{name}.reg", preds[0]["reg"], RT, AT)
    check_golden(golden, f"{name}.hm", preds[0]["hm"], RT, AT)


@pytest.mark.parametrize("name", CONFIGS)
def test_loss_and_grads(name, golden, schema):
    sd, ex, fuse, weight, cw = _setup(name, schema)
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    losses = O.radar_pose_net(sd, ex, fuse, weight, cw, return_loss=True)
    losses["loss"][0].backward()
    for k in ("loss", "hm_loss", "loc_loss", "loc_loss_elem", "num_positive"):
        np.testing.assert_allclose(losses[k][0].detach().numpy(), golden[f"{name}.loss.{k}"], rtol=1e-4, atol=1e-6)
    gkeys = [k for k in golden.files if k.startswith(f"{name}.grad.")]
    assert gkeys
    for gk in {k.split("#")[0] for k in gkeys}:
        pname = gk[len(name) + 6:]
        check_golden(golden, gk, sd[pname].grad, 2e-3, 1e-6)
    gn = np.asarray([float(p.grad.norm()) if p.grad is not None else -1.0 for p in sd.values()])
    ref = golden[f"{name}.gradnorm"]
    assert ((gn < 0) == (ref < 0)).all(), "same set of unused parameters"
    np.testing.assert_allclose(gn, ref, rtol=2e-3, atol=1e-7)


@pytest.mark.parametrize("name", CONFIGS)
def test_predict(name, golden, schema):
    sd, ex, fuse, weight, cw = _setup(name, schema)
    with torch.no_grad():
        ret = O.radar_pose_net(sd, ex, fuse, weight, cw, return_loss=False, test_cfg=TEST_CFG)
    got = np.asarray([[list(kp) for kp in r["keypoints"]] for r in ret])
    np.testing.assert_allclose(got, golden[f"{name}.predict"], rtol=1e-4, atol=1e-4)


def test_native_shape_pin(golden):
    """hr3d at the dataset-native [1,1,16,64,160]: sampled logits + moments."""
    arch, fin, fout, fuse, heads, _, _ = O.MODEL_CONFIGS["hr3d"]
    sd = O.seeded_state_dict(O.param_shapes(arch, fin, fout, fout, heads), seed=1)
    ex = O.synth_example(1, 1, (16, 64, 160), seed=1234)
    with torch.no_grad():
        preds, _ = O.center_head(sd, O.hrnet3d(sd, ex["rdr"]["rdr_tensor"], fuse))
    idx = torch.from_numpy(golden["native.idx"])
    for k in ("reg", "hm"):
        t = preds[0][k][0].reshape(preds[0][k].shape[1], -1)
        np.testing.assert_allclose(t[:, idx].numpy(), golden[f"native.{k}.samples"], rtol=1e-4, atol=3e-5)
        np.testing.assert_allclose([float(t.mean()), float(t.abs().max()), float(t.std())],
                                   golden[f"native.{k}.moments"], rtol=1e-4)


@pytest.mark.parametrize("tag,cin", [("stem1", 1), ("stem32", 32)])
def test_stem_block(tag, cin, golden, schema):
    """BASELINE config 1: one radar tensor through the 2-layer 3-D conv stem (ResNetBlock) on CPU."""
    shapes = {k: tuple(v) for k, v in schema[tag].items()}
    sd = {"blk." + k: v.requires_grad_(True) for k, v in O.seeded_state_dict(shapes, seed=3).items()}
    x = torch.relu(torch.randn(2, cin, 8, 16, 32, generator=torch.Generator().manual_seed(5)) * 0.5 + 0.1)
    x.requires_grad_(True)
    y = O.resnet_block(sd, "blk", x)
    y.backward(torch.randn(y.shape, generator=torch.Generator().manual_seed(6)))
    check_golden(golden, f"{tag}.y", y, RT, AT)
    check_golden(golden, f"{tag}.gx", x.grad, 1e-3, 1e-5)
    check_golden(golden, f"{tag}.gw2", sd["blk.conv2.conv.weight"].grad, 1e-3, 1e-4)
    check_golden(golden, f"{tag}.ggn3", sd["blk.conv3.groupnorm.weight"].grad, 1e-3, 1e-4)


def test_gaussian_and_pjpe(golden):
    for r in (1, 2):
        d = 2 * r + 1
        np.testing.assert_allclose(O.gaussian3d((d, d, d), sigma=d / 6), golden[f"gauss3d.r{r}"], rtol=1e-12)
    hm = np.zeros((8, 16, 16), np.float32)
    for c in ((0, 0, 0), (15, 15, 7), (5, 9, 3), (6, 9, 3)):
        O.draw_gaussian3d(hm, c, 2)
    np.testing.assert_array_equal(hm, golden["gauss3d.drawn"])
    np.testing.assert_allclose(O.abs_pjpe(golden["pjpe.pred"], golden["pjpe.gt"]), golden["pjpe.abs"], rtol=1e-12)
    np.testing.assert_allclose(O.pjpe(golden["pjpe.pred"], golden["pjpe.gt"]), golden["pjpe.rel"], rtol=1e-12)


def test_one_cycle_and_adam_rule():
    """Row T (formula-pinned): endpoints of the OneCycle schedule and one Adam/true-wd step vs torch.optim.Adam."""
    lr0, m0 = O.one_cycle(0, 1000, 1e-3)
    assert abs(lr0 - 1e-4) < 1e-12 and abs(m0 - 0.95) < 1e-12
    lrp, mp = O.one_cycle(400, 1000, 1e-3)
    assert abs(lrp - 1e-3) < 1e-12 and abs(mp - 0.85) < 1e-12
    lre, me = O.one_cycle(999, 1000, 1e-3)
    assert lre < 2e-8 + 1e-8 and abs(me - 0.95) < 1e-4
    torch.manual_seed(0)
    p = torch.randn(50, requires_grad=True)
    q = p.detach().clone().requires_grad_(True)
    g = torch.randn(50)
    p.grad, q.grad = g.clone(), g.clone()
    opt = torch.optim.Adam([q], lr=3e-4, betas=(0.9, 0.99), eps=1e-8)
    mine = O.AdamTrueWD([p])
    for _ in range(3):
        with torch.no_grad():
            q.mul_(1 - 0.01 * 3e-4)
        opt.step()
        mine.step(3e-4, 0.9, max_norm=1e9)
    np.testing.assert_allclose(p.detach().numpy(), q.detach().numpy(), rtol=1e-6, atol=1e-7)


def test_train_step_rule_against_the_reference_optimiser():
    """Row T pinned by import, not by formula: tests/golden/optim_golden.npz was captured from the reference's OWN OptimWrapper
    (fastai_optim.py:121-175, true_wd) around torch.optim.Adam + OneCycle (learning_schedules_fastai.py:53-95) + clip_grad_norm_(35),
    driven in the trainer's order (gen_golden_optim.py).  O.one_cycle and O.AdamTrueWD must reproduce lr, momentum and every
    parameter after each of the 7 steps (both cosine phases, one clipped step, parameters that never get a gradient)."""
    import os
    import numpy as np
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "optim_golden.npz"))
    names, total = [str(n) for n in z["names"]], int(z["total_steps"])
    params = [torch.tensor(z["init/" + k]).requires_grad_(True) for k in names]
    opt = O.AdamTrueWD(params)
    for step in range(len(z["lr"])):
        lr, b1 = O.one_cycle(step, total, 1e-3)
        assert abs(lr - float(z["lr"][step])) < 1e-15 and abs(b1 - float(z["mom"][step])) < 1e-15, (step, lr, b1)
        for k, p in zip(names, params):
            key = "grad/%d/%s" % (step, k)
            p.grad = torch.tensor(z[key]) if key in z.files else None
        norm = opt.step(lr, b1)
        assert abs(norm - float(z["grad_norm"][step])) < 1e-4 * float(z["grad_norm"][step])
        for k, p in zip(names, params):
            np.testing.assert_allclose(p.detach().numpy(), z["after/%d/%s" % (step, k)], rtol=2e-6, atol=2e-7, err_msg="%d %s" % (step, k))
