"""Per-kernel parity against the ORACLE (oracle/hrradarpose_ref.py: the restatement pinned by the reference-captured golden vectors,
tests/test_oracle_golden.py) -- not against tests/emu_backend.py, the builder's own emulation of the kernels that
tests/test_gpu_kernels.py compares with (that layer is self-referential).  One test per module-level operation of the
path, each through the C ABI entry points the plan uses for it, on seeded inputs the oracle finishes in milliseconds:

  GroupNorm -> Conv3d(3x3x3) [-> ReLU]            O.single_conv        hr_util/common.py:73-96     rtp_conv_gn_fused | rtp_fold_fwd + rtp_conv_igemm
  GroupNorm -> Conv3d(3x3x3, stride 2) [-> ReLU]  O._gn_conv_seq       hr_util/hr3d.py:168-197     rtp_fold_fwd + rtp_conv_igemm (tiled stride-2 / generic)
  ResNetBlock                                      O.resnet_block       hr_util/common.py:98-148    the two launches above + residual + ReLU epilogue
  fuse row: sum of up-sampled terms, ReLU          F.interpolate(...)   hr_util/hr3d.py:213-228     rtp_fuse_sum_stats
  ... and its adjoint                              autograd of the same                              rtp_upsample_bwd
  sigmoid + clamp + FastFocalLoss, RegLoss         O.fast_focal_loss / O.reg_loss  centernet_loss.py:17-54   rtp_focal_loss_ex / rtp_reg_loss (values AND logit gradients)
  decode                                           O.center_head_predict           center_head.py:272-360    rtp_decode
  optimiser rule                                   O.AdamTrueWD + one_cycle        fastai_optim.py:154-172   rtp_sqnorm + rtp_adam_step

Stated tolerances: activations are stored in bf16 and GroupNorm is folded into bf16 weights, so a conv output carries ~2^-9 relative
rounding per stored operand: norm-wise 5e-3 per conv against the oracle's fp32 on the same (bf16-representable) inputs -- the emulation,
which rounds at the same points, sits at 2.5-2.7e-3 against the same values, a two-conv block at 3.3e-3; bf16-stored point-wise results
and gradients 4e-3 (measured 1.5-2.9e-3); fp32 quantities (losses, decoded coordinates, optimiser state) 1e-5."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import hrradarpose_ref as O
from rt_pose_amd.graph import Geom, View, pad_to
from tests.util import rel_err

pytestmark = pytest.mark.gpu
BF_CONV = 5e-3


@pytest.fixture(scope="module")
def hip():
    from rt_pose_amd.backend import HipBackend
    return HipBackend("cuda:0")


def rnd(shape, seed, scale=1.0, relu=False):
    g = torch.Generator().manual_seed(seed)
    t = torch.randn(*shape, generator=g) * scale
    return torch.relu(t + 0.1) if relu else t


def to_view(hip, t_ncdhw, c_pad=None):
    """fp32 NCDHW CPU tensor -> (bf16-rounded fp32 NCDHW copy the oracle gets, channels-last bf16 device View the kernels get)."""
    n, c, d, h, w = t_ncdhw.shape
    cp = c_pad or c
    cl = torch.zeros(n, d, h, w, cp, dtype=torch.bfloat16)
    cl[..., :c] = t_ncdhw.permute(0, 2, 3, 4, 1).to(torch.bfloat16)
    buf = cl.to(hip.device)
    return cl[..., :c].float().permute(0, 4, 1, 2, 3).contiguous(), View(buf, n, d, h, w, cp, 0, cp)


def from_view(v, c):
    return v.buf[..., :c].float().cpu().permute(0, 4, 1, 2, 3).contiguous()


def input_stats(hip, xv, nsplit=3):
    st = hip.alloc((xv.n, nsplit, xv.c, 2), "f32")
    hip.chan_stats(xv, None, nsplit, st)(hip.stream())
    return st, nsplit


def gn_conv(hip, xv, sd, pgn, pconv, stride, relu, res=None, groups=8):
    """GroupNorm(pgn) -> Conv3d(pconv, 3x3x3, pad 1, stride) [+ res] [ReLU] on the device, as the plan launches it."""
    w = sd[pconv + ".weight"].to(hip.device)
    gamma, beta = sd[pgn + ".weight"].to(hip.device), sd[pgn + ".bias"].to(hip.device)
    co_real, ci = w.shape[0], w.shape[1]
    do, ho, wo = [(s + 2 - 3) // stride + 1 for s in (xv.d, xv.h, xv.w)]
    co = pad_to(co_real, 16)
    geom = Geom(xv.n, xv.d, xv.h, xv.w, do, ho, wo, ci, co, 3, stride, 1)
    st, ns = input_stats(hip, xv)
    yc = co if co_real % 16 == 0 else pad_to(co_real, 32)
    y = View(hip.alloc((xv.n, do, ho, wo, yc), "bf16"), xv.n, do, ho, wo, yc, 0, yc)
    mr = hip.alloc((xv.n, groups, 2), "f32")
    s = hip.stream()
    fused = stride == 1 and ci == 32 and co in (16, 32) and hip.conv_tiled_ok(xv, geom, False)
    if fused:
        wt = hip.alloc((27, co, ci), "f32")
        hip.tail([("pack_wt", w, co_real, co, ci, 27, wt)])(s)
        hip.conv_gn_fused(xv, wt, None, gamma, beta, st, ns, groups, 1e-5, co_real, mr, res, y, geom, relu)(s)
    else:
        wf, bt = hip.alloc((xv.n, 27, co, ci), "bf16"), hip.alloc((xv.n, 64, co), "f32")
        hip.fold_fwd(w, None, gamma, beta, st, ns, groups, 1e-5, geom, ci, co_real, wf, bt, mr, None)(s)
        # (wide layers: the slice route / the native 64 -> 64 kernel, as the plan launches them -- with a workspace)
        ws = hip.alloc((xv.n * do * ho * wo * 32,), "f32") if hip.conv_sliced_ok(xv, geom, False) else None
        hip.conv(xv, wf, True, bt, res, y, geom, relu, False, False, ws=ws)(s)
    torch.cuda.synchronize()
    return y, fused


@pytest.mark.parametrize("n,dims,ci,co,relu,tiled", [(2, (4, 8, 32), 32, 32, True, True), (1, (2, 12, 48), 32, 32, False, True), (2, (3, 6, 40), 32, 32, False, False),
                                                     (2, (4, 8, 16), 64, 64, True, False), (3, (2, 8, 32), 32, 16, True, True),
                                                     (1, (8, 16, 64), 32, 32, True, True),
                                                     # 64 -> 64 on the 64-wide kernel behind a fold launch, ragged W (levels 2 / 3 of the native shape)
                                                     (8, (4, 16, 40), 64, 64, True, False), (16, (2, 8, 20), 64, 64, False, False),
                                                     (2, (3, 6, 10), 64, 64, True, False)])
def test_single_conv_against_the_oracle(hip, n, dims, ci, co, relu, tiled):
    """O.single_conv = GroupNorm(8) -> Conv3d(3x3x3, bias=False) [-> ReLU] (hr_util/common.py:73-96): the LDS-tiled kernel with the fold
    in its prologue where the geometry is its own, fold launch + generic kernel otherwise."""
    d, h, w = dims
    sd = {"p.groupnorm.weight": rnd((ci,), 1) * 0.2 + 1.0, "p.groupnorm.bias": rnd((ci,), 2) * 0.2,
          "p.conv.weight": rnd((co, ci, 3, 3, 3), 3, 0.05)}
    xo, xv = to_view(hip, rnd((n, ci, d, h, w), 4, relu=True))
    y, fused = gn_conv(hip, xv, sd, "p.groupnorm", "p.conv", 1, relu)
    assert fused == tiled, "which kernel this geometry takes"
    want = O.single_conv(sd, "p", xo, relu)
    assert rel_err(from_view(y, co), want) < BF_CONV


@pytest.mark.parametrize("n,dims,ci,co", [(2, (4, 8, 32), 32, 32), (1, (8, 16, 64), 32, 32), (2, (4, 8, 16), 64, 64), (2, (2, 4, 8), 32, 64)])
def test_stride2_gn_conv_against_the_oracle(hip, n, dims, ci, co):
    """O._gn_conv_seq(stride 2, padding 1) [-> ReLU]: the transitions and the fuse chains' stride-2 convs (hr_util/hr3d.py:168-197, 297-305)."""
    d, h, w = dims
    sd = {"p.0.weight": rnd((ci,), 11) * 0.2 + 1.0, "p.0.bias": rnd((ci,), 12) * 0.2, "p.1.weight": rnd((co, ci, 3, 3, 3), 13, 0.05)}
    xo, xv = to_view(hip, rnd((n, ci, d, h, w), 14, relu=True))
    y, _ = gn_conv(hip, xv, sd, "p.0", "p.1", 2, True)
    want = O._gn_conv_seq(sd, "p", xo, 2, 1, relu=True)
    assert tuple(want.shape[2:]) == (y.d, y.h, y.w)
    assert rel_err(from_view(y, co), want) < BF_CONV


@pytest.mark.parametrize("n,c,dims", [(2, 32, (4, 8, 32)), (8, 64, (4, 16, 40)), (16, 64, (2, 8, 20))])
def test_resnet_block_against_the_oracle(hip, n, c, dims):
    """O.resnet_block with Cin == Cout (hr_util/common.py:98-148: no conv1): gcr -> gc -> + x -> ReLU = two tiled launches, the second
    with the residual and the ReLU in its epilogue; the intermediate is stored in bf16 as in the plan.  32 channels: conv_tiled.hip;
    64 channels at the level-2 / level-3 shapes: fold launch + conv64_tiled.hip (ragged brick columns)."""
    d, h, w = dims
    sd = {}
    for k, nm in enumerate(("conv2", "conv3")):
        sd["b.%s.groupnorm.weight" % nm] = rnd((c,), 20 + k) * 0.2 + 1.0
        sd["b.%s.groupnorm.bias" % nm] = rnd((c,), 22 + k) * 0.2
        sd["b.%s.conv.weight" % nm] = rnd((c, c, 3, 3, 3), 24 + k, 0.05)
    xo, xv = to_view(hip, rnd((n, c, d, h, w), 26, relu=True))
    mid, _ = gn_conv(hip, xv, sd, "b.conv2.groupnorm", "b.conv2.conv", 1, True)
    out, _ = gn_conv(hip, mid, sd, "b.conv3.groupnorm", "b.conv3.conv", 1, True, res=xv)
    want = O.resnet_block(sd, "b", xo)
    assert rel_err(from_view(out, c), want) < 1.5 * BF_CONV     # two bf16-stored layers


def gn_conv_backward(hip, xv, gyv, w, gamma, beta, stride, co_real, fused, groups=8):
    """The backward launches of one GroupNorm -> Conv3d layer as the plan issues them; -> (dx View, dW, dgamma, dbeta).
    fused: rtp_wgrad_q -> rtp_gn_bwd_coeffs_cls -> rtp_conv_dgrad_fused (the LDS-tiled geometries); otherwise data gradient ->
    rtp_chan_stats -> rtp_gn_bwd_coeffs -> rtp_grad_combine, and rtp_wgrad.  Both end in rtp_wgrad_fold."""
    n, ci = xv.n, xv.c
    co = pad_to(co_real, 16)
    co32 = pad_to(co, 32)
    geom = Geom(n, xv.d, xv.h, xv.w, gyv.d, gyv.h, gyv.w, ci, co, 3, stride, 1)
    s = hip.stream()
    wdev, gam, bet = w.to(hip.device), gamma.to(hip.device), beta.to(hip.device)
    # what the forward pass leaves behind: (mean, rstd) per sample and group
    st, ns = input_stats(hip, xv)
    wf, bt, mr = hip.alloc((n, 27, co, ci), "bf16"), hip.alloc((n, 64, co), "f32"), hip.alloc((n, groups, 2), "f32")
    hip.fold_fwd(wdev, None, gam, bet, st, ns, groups, 1e-5, geom, ci, co_real, wf, bt, mr, None)(s)
    wd = hip.alloc((27, ci, co32), "bf16")
    hip.pack_dgrad_w(wdev.reshape(co_real, ci, 27), geom, ci, co_real, wd)(s)
    csp, csum = hip.alloc((n, 3, 64, co32), "f32"), hip.alloc((n, 64, co32), "f32")
    hip.class_sums(gyv, 3, csp, None if fused else csum)(s)     # fused: the partials only, reduced in the coefficient launch
    dx = View(hip.alloc((n, xv.d, xv.h, xv.w, ci), "bf16"), n, xv.d, xv.h, xv.w, ci, 0, ci)
    coeff = hip.alloc((n * ci * 5,), "f32")
    if fused:
        S = hip.wgrad_nsplit(geom)
        assert S > 0 and hip.conv_dgrad_fused_ok(gyv, geom)
        gp, qp = hip.alloc((n, S, 27, co32, ci), "f32"), hip.alloc((n, S, ci), "f32")
        hip.wgrad_q(gyv, xv, geom, S, gp, wd, qp)(s)
        hip.gn_bwd_coeffs_cls(qp, S, csp, 3, csum, wd, mr, gam, geom, ci, co_real, groups, coeff)(s)
        hip.conv_dgrad_fused(gyv, wd, xv, coeff, [], False, dx, geom)(s)
    else:
        S = 2
        gp = hip.alloc((n, S, 27, co32, ci), "f32")
        hip.wgrad(gyv, xv, geom, S, gp)(s)
        dxh = View(hip.alloc((n, xv.d, xv.h, xv.w, ci), "bf16"), n, xv.d, xv.h, xv.w, ci, 0, ci)
        ws = hip.alloc((n * xv.vox * 32,), "f32") if hip.conv_sliced_ok(gyv, geom, True) else None
        hip.conv(gyv, wd, False, None, None, dxh, geom, False, True, False, ws=ws)(s)
        pq = hip.alloc((n, 4, ci, 2), "f32")
        hip.chan_stats(dxh, xv, 4, pq)(s)
        hip.gn_bwd_coeffs(pq, 4, mr, gam, n, ci, groups, xv.vox, coeff, None, None, 0)(s)
        hip.grad_combine([(dxh, coeff)], xv, None, dx)(s)
    dw = hip.alloc((co_real, ci, 27), "f32")
    hip.wgrad_fold(gp, S, csum, mr, gam, bet, groups, geom, ci, co_real, dw, None, 0)(s)
    torch.cuda.synchronize()
    part = coeff[n * ci * 3:].view(n, ci, 2).cpu()
    return dx, dw.cpu().view(co_real, ci, 3, 3, 3), part[..., 0].sum(0), part[..., 1].sum(0)


@pytest.mark.parametrize("n,dims,ci,co,stride,fused", [(2, (4, 8, 32), 32, 32, 1, True), (1, (8, 16, 64), 32, 32, 1, True), (3, (2, 8, 32), 32, 16, 1, True),
                                                       (2, (4, 8, 32), 32, 32, 1, False), (2, (3, 6, 40), 32, 32, 1, False),
                                                       (2, (4, 8, 32), 32, 64, 2, False), (1, (8, 16, 64), 32, 32, 2, False),
                                                       (2, (4, 8, 16), 64, 64, 1, False)])
def test_gn_conv_backward_against_the_oracle(hip, n, dims, ci, co, stride, fused):
    """The gradients of y = Conv3d(GroupNorm(x)) (O.single_conv without the ReLU / O._gn_conv_seq) with respect to x, the conv weight and
    the GroupNorm affine pair, as autograd gives them for the oracle, against the launches the backward plan issues for such a layer --
    the fused chain (GroupNorm-backward coefficients from slab contractions and class sums: no pass over a stored dxhat) and the
    unfused one.  The weight gradient is produced in fp32 from bf16 operands the oracle sees exactly: 1e-3 (emulation: 5e-7).  dgamma / dbeta
    contract data-gradient weights that are rounded to bf16 (and, unfused, a dxhat stored in bf16): 8e-3 (emulation: 1.3-3.0e-3); the
    data gradient is stored in bf16 on top of that: 6e-3 (emulation: 2.3-2.9e-3)."""
    d, h, w = dims
    do, ho, wo = [(v + 2 - 3) // stride + 1 for v in dims]
    W, gamma, beta = rnd((co, ci, 3, 3, 3), 40, 0.05), rnd((ci,), 41) * 0.2 + 1.0, rnd((ci,), 42) * 0.2
    xo, xv = to_view(hip, rnd((n, ci, d, h, w), 43, relu=True))
    gyo, gyv = to_view(hip, rnd((n, co, do, ho, wo), 44), pad_to(pad_to(co, 16), 32))
    dx, dw, dgamma, dbeta = gn_conv_backward(hip, xv, gyv, W, gamma, beta, stride, co, fused)
    xr, Wr, gr, br = [t.clone().requires_grad_(True) for t in (xo, W, gamma, beta)]
    sd = {"p.0.weight": gr, "p.0.bias": br, "p.1.weight": Wr}
    O._gn_conv_seq(sd, "p", xr, stride, 1, relu=False).backward(gyo)
    assert rel_err(dw, Wr.grad) < 1e-3, "weight gradient"
    assert rel_err(dgamma, gr.grad) < 8e-3 and rel_err(dbeta, br.grad) < 8e-3, "GroupNorm affine gradients"
    assert rel_err(from_view(dx, ci), xr.grad) < 6e-3, "data gradient"


def test_fuse_row_and_its_adjoint_against_the_oracle(hip):
    """A HighResolutionModule fuse row, relu(x0 + up(t1) + up(t2)) with trilinear align_corners=True up-sampling (hr3d.py:213-228;
    O.hr_module), and the adjoint of that up-sampling as autograd computes it."""
    n, c = 2, 32
    hi, lo1, lo2 = (8, 16, 32), (4, 8, 16), (2, 4, 8)
    x0o, x0v = to_view(hip, rnd((n, c, *hi), 30))
    t1o, t1v = to_view(hip, rnd((n, c, *lo1), 31))
    t2o, t2v = to_view(hip, rnd((n, c, *lo2), 32))
    out = View(hip.alloc((n, *hi, c), "bf16"), n, *hi, c, 0, c)
    hip.fuse_sum([x0v, t1v, t2v], None, out, True)(hip.stream())
    torch.cuda.synchronize()
    up = lambda t: F.interpolate(t, size=hi, mode="trilinear", align_corners=True)
    want = F.relu(x0o + up(t1o) + up(t2o))
    assert rel_err(from_view(out, c), want) < 4e-3                # one bf16 rounding of an fp32 sum
    # adjoint: d/dt1 of <g, up(t1)>
    go, gv = to_view(hip, rnd((n, c, *hi), 33))
    glow = View(hip.alloc((n, *lo1, c), "bf16"), n, *lo1, c, 0, c)
    hip.upsample_bwd(gv, glow)(hip.stream())
    torch.cuda.synchronize()
    z = torch.zeros(n, c, *lo1, requires_grad=True)
    up(z).backward(go)
    assert rel_err(from_view(glow, c), z.grad) < 4e-3


@pytest.mark.parametrize("ncls,nreg", [(15, 3), (1, 45)])
def test_losses_and_decode_against_the_oracle(hip, ncls, nreg):
    """CenterHead._sigmoid + FastFocalLoss + RegLoss (center_head.py:240-258, centernet_loss.py:17-54): loss values and the gradients with
    respect to the raw logits / regression outputs as autograd gives them for the oracle; CenterHead.predict (center_head.py:272-360)."""
    n, (d, h, w) = 2, (4, 8, 16)
    ex = O.synth_example(n, 1, (d, h, w), seed=5, one_hm=ncls == 1)["rdr"]
    m = ex["ind"][0].shape[1]
    hm_c, rg_c = pad_to(ncls, 16), pad_to(nreg, 16)
    logits = (rnd((n, ncls, d, h, w), 50, 2.0) - 2.0).requires_grad_(True)
    reg = rnd((n, nreg, d, h, w), 51).requires_grad_(True)
    dev = hip.device

    def cl_f32(t, cp):
        b = torch.zeros(n, d, h, w, cp)
        b[..., :t.shape[1]] = t.detach().permute(0, 2, 3, 4, 1)
        return View(b.to(dev), n, d, h, w, cp, 0, cp)
    hv, rv = cl_f32(logits, hm_c), cl_f32(reg, rg_c)
    # oracle
    p = torch.clamp(torch.sigmoid(logits), 1e-4, 1 - 1e-4)
    hm_loss = O.fast_focal_loss(p, ex["hm"][0], ex["ind"][0], ex["mask"][0], ex["cat"][0])
    rl = O.reg_loss(reg, ex["mask"][0], ex["ind"][0], ex["anno_pose"][0].reshape(n, m, nreg))
    cw = torch.linspace(1, 2, nreg)
    loc = (rl * cw).sum()
    (hm_loss + 0.25 * loc).backward()
    # kernels (gscale = the weight each loss enters the total with)
    gh_c, gr_c = pad_to(hm_c, 32), pad_to(rg_c, 32)
    ghv = View(torch.zeros(n, d, h, w, gh_c, dtype=torch.bfloat16, device=dev), n, d, h, w, gh_c, 0, gh_c)
    grv = View(torch.zeros(n, d, h, w, gr_c, dtype=torch.bfloat16, device=dev), n, d, h, w, gr_c, 0, gr_c)
    lh, lr = torch.zeros(1, device=dev), torch.zeros(nreg + 1, device=dev)
    s = hip.stream()
    hip.focal_loss(hv, ex["hm"][0].to(dev), ex["ind"][0].to(dev), ex["mask"][0].to(dev), ex["cat"][0].to(dev), ncls, 1.0,
                   hip.focal_scratch(n), lh, ghv)(s)
    hip.reg_loss(rv, ex["anno_pose"][0].reshape(n, m, nreg).contiguous().to(dev), ex["ind"][0].to(dev), ex["mask"][0].to(dev),
                 cw.to(dev), nreg, 0.25, lr, grv)(s)
    torch.cuda.synchronize()
    assert abs(float(lh[0]) - float(hm_loss)) < 1e-4 * abs(float(hm_loss)) + 1e-6
    np.testing.assert_allclose(lr[:nreg].cpu().numpy(), rl.detach().numpy(), rtol=1e-5, atol=1e-7)
    assert abs(float(lr[nreg]) - float(loc)) < 1e-5 * abs(float(loc)) + 1e-7
    assert rel_err(from_view(ghv, ncls), logits.grad) < 4e-3      # gradients are handed to the backward plan in bf16
    assert rel_err(from_view(grv, nreg), reg.grad) < 4e-3
    # decode
    tcfg = dict(out_size_factor=(1, 1, 1), voxel_size=(0.05, 0.15, 0.36), pc_range=(0.77, -5.0, -1.1, 0, 0, 0), score_threshold=0.0)
    out = torch.zeros(n, ncls, 2 + nreg, device=dev)
    hip.decode(hv, rv, ncls, nreg, (0.05, 0.15, 0.36), (0.77, -5.0, -1.1), hip.decode_scratch(n, ncls), out)(s)
    torch.cuda.synchronize()
    want = O.center_head_predict([{"hm": logits.detach(), "reg": reg.detach()}], tcfg)
    got = out.cpu()
    for b in range(n):
        kps = want[b]["keypoints"]
        if nreg == 3:
            for cidx in range(ncls):
                np.testing.assert_allclose(got[b, cidx, 2:5].numpy(), np.asarray(kps[cidx][1:4]), rtol=1e-5, atol=1e-5)
                assert abs(float(got[b, cidx, 1]) - kps[cidx][4]) < 1e-6
        else:
            np.testing.assert_allclose(got[b, 0, 2:2 + nreg].numpy().reshape(-1, 3), np.asarray([k[1:4] for k in kps]), rtol=1e-5, atol=1e-5)


def test_optimiser_rule_against_the_oracle(hip):
    """clip_grad_norm_(35) -> p *= 1 - wd * lr (every parameter) -> Adam with scheduled beta1 (fastai_optim.py:154-172, restated as
    O.AdamTrueWD) for three steps of the one-cycle schedule, one tensor without a gradient among them."""
    from rt_pose_amd.engine import FlatAdam, FlatParams, one_cycle
    from collections import OrderedDict
    shapes = OrderedDict(a=(32, 32, 3, 3, 3), dead=(16,), b=(64,), c=(7, 5))
    flat = FlatParams(shapes, hip.alloc)
    g = torch.Generator().manual_seed(9)
    sd = OrderedDict((k, torch.randn(s, generator=g)) for k, s in shapes.items())
    flat.load_state_dict(sd)
    ref = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd.items())
    opt_o = O.AdamTrueWD(list(ref.values()))
    live = {"a", "b", "c"}
    opt = FlatAdam(hip, flat, live)
    for step in range(3):
        lr, b1 = one_cycle(step, 10, 1e-3)
        lr_o, b1_o = O.one_cycle(step, 10, 1e-3)
        assert abs(lr - lr_o) < 1e-12 and abs(b1 - b1_o) < 1e-12
        for k in shapes:
            gk = torch.randn(shapes[k], generator=g) * (30.0 if step == 1 else 0.1)   # step 1: the norm exceeds 35, the clip bites
            ref[k].grad = gk.clone() if k in live else None
            flat.grads[k].copy_(gk if k in live else torch.zeros(shapes[k]))
        opt_o.step(lr_o, b1_o)
        opt.set_hyper(lr, b1)
        opt.run()
        torch.cuda.synchronize()
        for k in shapes:
            assert rel_err(flat.values[k].cpu(), ref[k].detach()) < 1e-5, (step, k)
