"""Per-kernel parity on a real MI355X: every C-ABI entry point (through rt_pose_amd.backend.HipBackend) against the
torch-CPU emulation of the same kernel on identical seeded buffers, including the edge cases the shapes of the
path produce (stride 2, 1x1x1, depth-1 volumes, channel-padded heads, channel-slice views, ragged voxel counts).

Tolerances: bf16 outputs -- norm-wise 4e-3 (one bf16 ulp is 2^-8 relative; fp32 accumulation order differs);
fp32 outputs -- norm-wise 2e-4.
"""
import os

import numpy as np
import pytest
import torch

from rt_pose_amd.graph import Geom, View, pad_to
from tests.emu_backend import EmuBackend
from tests.util import rel_err


def same_partials(a, b, dim=1):
    """Per-workgroup partials of two launches of one problem: bit-equal (bricks are dealt statically)."""
    return torch.equal(a, b)

pytestmark = pytest.mark.gpu

BF, F32 = 4e-3, 2e-4


@pytest.fixture(scope="module")
def hip():
    from rt_pose_amd.backend import HipBackend
    return HipBackend("cuda:0")


EMU = EmuBackend()


class Pair:
    """The same tensor on both backends."""

    def __init__(self, hip, t):
        self.c = t.clone()
        self.g = t.to(hip.device)

    def sync_back(self):
        return self.g.detach().cpu()


def rnd(shape, seed, dtype=torch.bfloat16, scale=1.0, relu=False):
    g = torch.Generator().manual_seed(seed)
    t = torch.randn(*shape, generator=g) * scale
    if relu:
        t = torch.relu(t + 0.1)
    return t.to(dtype)


def views(hip, t, n, d, h, w, co=0, c=None):
    p = Pair(hip, t)
    cs = t.shape[-1]
    c = c or cs - co
    return p, View(p.c, n, d, h, w, cs, co, c), View(p.g, n, d, h, w, cs, co, c)


def run(hip, fc, fg):
    fc(None)
    fg(hip.stream())
    torch.cuda.synchronize()


def check(pair, tol, what=""):
    got, ref = pair.sync_back().float(), pair.c.float()
    assert torch.isfinite(got).all(), what
    e = rel_err(got, ref)
    assert e < tol, (what, e)


CONV_CASES = [
    # n, dims, ci, co_real, ks, stride, per_sample, res, relu, fp32
    (2, (4, 8, 16), 32, 32, 3, 1, True, True, True, False),
    (2, (4, 8, 16), 32, 64, 3, 2, True, False, True, False),
    (1, (2, 4, 20), 64, 32, 1, 1, True, False, False, False),
    (2, (4, 8, 8), 32, 15, 3, 1, False, False, False, True),
    (1, (1, 2, 4), 64, 64, 3, 2, True, False, True, False),   # depth-1 volume: first AND last flags together
    (1, (3, 5, 7), 128, 128, 3, 1, True, True, False, False),  # ragged voxel count (105)
    (2, (4, 8, 8), 32, 45, 3, 1, False, False, False, True),
    # geometries that take the LDS-tiled kernel (csrc/conv_tiled.hip): D%2, H%8, W%32, Cin 32, Cout 16|32
    (2, (4, 8, 32), 32, 32, 3, 1, True, True, True, False),
    (3, (2, 16, 64), 32, 32, 3, 1, True, False, False, False),
    (1, (2, 16, 64), 32, 15, 3, 1, False, False, False, True),
    (2, (6, 24, 96), 32, 3, 3, 1, False, False, True, True),
    (2, (8, 32, 80), 32, 32, 3, 1, True, True, True, False),   # W = 80: the last brick column is half padding
    (1, (2, 4, 16), 32, 32, 3, 1, True, False, True, False),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_forward(hip, case):
    n, dims, ci, co_real, ks, stride, per_sample, has_res, relu, fp32 = case
    d, h, w = dims
    pad = ks // 2
    do, ho, wo = [(s + 2 * pad - ks) // stride + 1 for s in dims]
    co = pad_to(co_real, 16)
    geom = Geom(n, d, h, w, do, ho, wo, ci, co, ks, stride, pad)
    xp, xc, xg = views(hip, rnd((n, d, h, w, ci), 1, relu=True), n, d, h, w)
    nw = n if per_sample else 1
    wf = Pair(hip, rnd((nw, ks ** 3, co, ci), 2, scale=0.05))
    bt = Pair(hip, rnd((nw, 64, co), 3, torch.float32))
    yc_ch = co if fp32 else (co if co_real % 16 == 0 else pad_to(co_real, 32))
    yp, yc, yg = views(hip, torch.zeros(n, do, ho, wo, yc_ch, dtype=torch.float32 if fp32 else torch.bfloat16), n, do, ho, wo)
    rc = rg = None
    if has_res:
        rp, rc, rg = views(hip, rnd((n, do, ho, wo, co), 4), n, do, ho, wo)
    run(hip, EMU.conv(xc, wf.c, per_sample, bt.c, rc, yc, geom, relu, False, fp32),
        hip.conv(xg, wf.g, per_sample, bt.g, rg, yg, geom, relu, False, fp32))
    check(yp, F32 if fp32 else BF, "conv fwd %r" % (case,))
    # fused statistics epilogue (sum y, sum y^2) where the LDS-tiled kernel offers it: compared as the sum over partials
    # with a chan_stats pass over the tensor the kernel itself stored (fp32 accuracy), and with the emulation (bf16-level)
    S = 0 if fp32 or co != yc_ch else hip.conv_stats_nsplit(xg, geom, False)
    if S:
        st = hip.alloc((n, S, co, 2), "f32")
        hip.conv(xg, wf.g, per_sample, bt.g, rg, yg, geom, relu, False, fp32, (None, st))(hip.stream())
        ref = hip.alloc((n, 3, co, 2), "f32")
        hip.chan_stats(yg, None, 3, ref)(hip.stream())
        torch.cuda.synchronize()
        check(yp, BF, "conv fwd (stats variant) %r" % (case,))
        assert rel_err(st.sum(1).cpu(), ref.sum(1).cpu()) < F32, "fused fwd stats %r" % (case,)
        yv = yc.buf.float().reshape(n, -1, yc_ch)
        emu = torch.stack([yv.sum(1), (yv * yv).sum(1)], -1)
        assert rel_err(st.sum(1).cpu(), emu) < BF * 2
    else:
        assert fp32 or co != yc_ch   # both kernels (LDS-tiled and generic) emit statistics for bf16 outputs


@pytest.mark.parametrize("case", [
    # n, dims, co_real, groups, nsplit of the input's statistics, bias, residual, relu
    (2, (4, 8, 32), 32, 8, 3, False, True, True), (3, (2, 16, 64), 32, 8, 1, True, False, False),
    (1, (2, 16, 64), 15, 8, 5, True, False, True), (2, (8, 32, 80), 32, 4, 37, False, True, True),
    (2, (6, 24, 96), 3, 32, 2, True, False, False), (8, (4, 8, 32), 16, 8, 16, False, False, True),
])
def test_conv_with_groupnorm_fold_in_the_prologue(hip, case):
    """rtp_conv_gn_fused (GroupNorm fold inside the LDS-tiled conv kernel) against rtp_fold_fwd + rtp_conv_igemm on the SAME
    device inputs (same arithmetic: results agree to bf16 rounding of a handful of weights at most), against the emulation,
    and the (mean, rstd) it saves for the backward pass."""
    n, dims, co_real, groups, nsplit, has_bias, has_res, relu = case
    d, h, w = dims
    ci, co = 32, pad_to(co_real, 16)
    geom = Geom(n, d, h, w, d, h, w, ci, co, 3, 1, 1)
    xp, xc, xg = views(hip, rnd((n, d, h, w, ci), 11, relu=True), n, d, h, w)
    W = Pair(hip, rnd((co_real, ci, 3, 3, 3), 12, torch.float32, scale=0.05))
    bias = Pair(hip, rnd((co_real,), 13, torch.float32)) if has_bias else None
    gamma, beta = Pair(hip, rnd((ci,), 14, torch.float32) * 0.2 + 1.0), Pair(hip, rnd((ci,), 15, torch.float32) * 0.2)
    # the input's statistics as a producer's epilogue leaves them: nsplit partials per sample (any split sums to the totals)
    xv = xp.c.float().reshape(n, -1, ci)
    tot = torch.stack([xv.sum(1), (xv * xv).sum(1)], -1)                      # [n, ci, 2]
    frac = torch.rand(nsplit, generator=torch.Generator().manual_seed(16)) + 0.1
    st = Pair(hip, (tot[:, None] * (frac / frac.sum())[None, :, None, None]).contiguous())
    yc_ch = co if co_real % 16 == 0 else pad_to(co_real, 32)
    yp, yc, yg = views(hip, torch.zeros(n, d, h, w, yc_ch, dtype=torch.bfloat16), n, d, h, w)
    rc = rg = None
    if has_res:
        rp, rc, rg = views(hip, rnd((n, d, h, w, co), 17), n, d, h, w)
    mr = Pair(hip, torch.zeros(n, groups, 2))
    b = lambda P: (P.c, P.g) if P is not None else (None, None)
    wt = Pair(hip, torch.zeros(27, co, ci))
    EMU.tail([("pack_wt", W.c, co_real, co, ci, 27, wt.c)])(None)
    hip.tail([("pack_wt", W.g, co_real, co, ci, 27, wt.g)])(hip.stream())
    torch.cuda.synchronize()
    assert torch.equal(wt.g.cpu(), wt.c)
    run(hip, EMU.conv_gn_fused(xc, wt.c, b(bias)[0], gamma.c, beta.c, st.c, nsplit, groups, 1e-5, co_real, mr.c, rc, yc, geom, relu),
        hip.conv_gn_fused(xg, wt.g, b(bias)[1], gamma.g, beta.g, st.g, nsplit, groups, 1e-5, co_real, mr.g, rg, yg, geom, relu))
    check(yp, BF, "conv + GroupNorm fold %r" % (case,))
    assert rel_err(mr.sync_back(), mr.c) < 1e-5
    # the two-launch route on the device
    wf = hip.alloc((n, 27, co, ci), "bf16"); bt = hip.alloc((n, 64, co), "f32"); mr2 = hip.alloc((n, groups, 2), "f32")
    y2 = torch.zeros_like(yp.g)
    y2v = View(y2, n, d, h, w, yc_ch, 0, yc_ch)
    hip.fold_fwd(W.g, b(bias)[1], gamma.g, beta.g, st.g, nsplit, groups, 1e-5, geom, ci, co_real, wf, bt, mr2, None)(hip.stream())
    hip.conv(xg, wf, True, bt, rg, y2v, geom, relu, False, False)(hip.stream())
    torch.cuda.synchronize()
    assert rel_err(mr.g.cpu(), mr2.cpu()) < 1e-6
    assert rel_err(yp.g.float().cpu(), y2.float().cpu()) < 1e-3, "fused fold vs fold + conv"
    # with the statistics epilogue
    S = hip.conv_stats_nsplit(xg, geom, False) if co == yc_ch else 0
    if S:
        so = hip.alloc((n, S, co, 2), "f32")
        hip.conv_gn_fused(xg, wt.g, b(bias)[1], gamma.g, beta.g, st.g, nsplit, groups, 1e-5, co_real, mr.g, rg, yg, geom, relu, so)(hip.stream())
        ref = hip.alloc((n, 3, co, 2), "f32")
        hip.chan_stats(yg, None, 3, ref)(hip.stream())
        torch.cuda.synchronize()
        assert rel_err(so.sum(1).cpu(), ref.sum(1).cpu()) < F32


@pytest.mark.parametrize("case", [
    (2, (4, 8, 16), 32, 32, 3, 1), (2, (4, 8, 16), 32, 64, 3, 2), (1, (2, 4, 20), 64, 32, 1, 1),
    (2, (4, 8, 8), 32, 15, 3, 1), (1, (1, 2, 4), 64, 64, 3, 2), (1, (5, 6, 7), 32, 32, 3, 2),
    (2, (4, 8, 32), 32, 32, 3, 1), (1, (2, 16, 64), 32, 15, 3, 1), (2, (8, 32, 80), 32, 32, 3, 1),  # LDS-tiled kernel, flipped taps
    # stride-2 data gradients on the parity-class kernel (csrc/dgrad_s2_tiled.hip): output dims = 2 x gy dims, Ho % 2, Wo % 16
    (2, (4, 8, 32), 32, 32, 3, 2), (3, (8, 16, 64), 32, 32, 3, 2), (1, (2, 4, 32), 32, 32, 3, 2), (8, (16, 64, 160), 32, 32, 3, 2)])
def test_conv_transposed_is_data_gradient(hip, case):
    n, dims, ci, co_real, ks, stride = case
    d, h, w = dims
    pad = ks // 2
    do, ho, wo = [(s + 2 * pad - ks) // stride + 1 for s in dims]
    co = pad_to(co_real, 16)
    cok = pad_to(co, 32)
    geom = Geom(n, d, h, w, do, ho, wo, ci, co, ks, stride, pad)
    gt = rnd((n, do, ho, wo, cok), 5)
    gt[..., co_real:] = 0
    gp_, gc, gg = views(hip, gt, n, do, ho, wo)
    w32 = Pair(hip, rnd((co_real, ci, ks, ks, ks), 6, torch.float32, 0.05))
    wd = Pair(hip, torch.zeros(ks ** 3, ci, cok, dtype=torch.bfloat16))
    run(hip, EMU.pack_dgrad_w(w32.c, geom, ci, co_real, wd.c), hip.pack_dgrad_w(w32.g, geom, ci, co_real, wd.g))
    assert torch.equal(wd.sync_back(), wd.c)
    yp, yc, yg = views(hip, torch.zeros(n, d, h, w, ci, dtype=torch.bfloat16), n, d, h, w)
    run(hip, EMU.conv(gc, wd.c, False, None, None, yc, geom, False, True, False),
        hip.conv(gg, wd.g, False, None, None, yg, geom, False, True, False))
    check(yp, BF, "dgrad %r" % (case,))
    S = hip.conv_stats_nsplit(gg, geom, True)
    if S:  # fused P = sum dxhat, Q = sum dxhat * x
        _, sxc, sxg = views(hip, rnd((n, d, h, w, 2 * ci), 9, relu=True), n, d, h, w, co=ci, c=ci)   # channel-slice view
        st = hip.alloc((n, S, ci, 2), "f32")
        hip.conv(gg, wd.g, False, None, None, yg, geom, False, True, False, (sxg, st))(hip.stream())
        ref = hip.alloc((n, 2, ci, 2), "f32")
        hip.chan_stats(yg, sxg, 2, ref)(hip.stream())
        torch.cuda.synchronize()
        check(yp, BF, "dgrad (stats variant) %r" % (case,))
        assert rel_err(st.sum(1).cpu(), ref.sum(1).cpu()) < F32, "fused P/Q %r" % (case,)
    # and it IS the autograd data gradient of F.conv3d
    x = torch.zeros(n, ci, d, h, w, requires_grad=True)
    wq = w32.c.to(torch.bfloat16).float()
    y = torch.nn.functional.conv3d(x, wq, None, stride, pad)
    y.backward(gt[..., :co_real].float().permute(0, 4, 1, 2, 3))
    assert rel_err(yp.sync_back().float().permute(0, 4, 1, 2, 3), x.grad) < BF


@pytest.mark.parametrize("case", [
    (2, (4, 8, 16), 32, 32, 3, 1, 3), (2, (8, 8, 16), 32, 64, 3, 2, 2), (1, (2, 4, 20), 64, 32, 1, 1, 1),
    (2, (4, 8, 8), 32, 15, 3, 1, 3), (1, (3, 5, 7), 128, 128, 3, 1, 1), (1, (1, 2, 4), 64, 64, 3, 2, 1),
    # 1x1x1: the all-channels-per-block kernel (wgrad_1x1_kernel)
    (2, (4, 8, 16), 64, 256, 1, 1, 3), (1, (3, 5, 7), 128, 64, 1, 1, 2), (2, (4, 8, 16), 32, 32, 1, 1, 4), (3, (2, 4, 20), 32, 45, 1, 1, 1),
    (1, (4, 16, 40), 64, 32, 1, 1, 10),
    # nsplit=None: geometries of the LDS-tiled kernel (csrc/wgrad_tiled.hip), which picks its own slab count
    (2, (4, 8, 32), 32, 32, 3, 1, None), (1, (2, 16, 64), 32, 15, 3, 1, None), (3, (6, 12, 96), 32, 32, 3, 1, None), (2, (8, 32, 80), 32, 32, 3, 1, None), (1, (2, 4, 16), 32, 32, 3, 1, None),
    # stride 2 on the LDS-tiled kernel (csrc/wgrad_s2_tiled.hip): 32 -> 32, ragged Wo, wide convs as slices (>= 65536 output voxels)
    (2, (4, 8, 32), 32, 32, 3, 2, None), (1, (8, 16, 64), 32, 32, 3, 2, None), (3, (4, 4, 80), 32, 15, 3, 2, None), (2, (4, 8, 40), 32, 32, 3, 2, None),
    (2, (2, 4, 8), 32, 32, 3, 2, None), (8, (16, 32, 128), 64, 64, 3, 2, None), (2, (8, 32, 80), 32, 64, 3, 2, None)])
def test_wgrad(hip, case):
    n, dims, ci, co_real, ks, stride, nsplit = case
    d, h, w = dims
    pad = ks // 2
    do, ho, wo = [(s + 2 * pad - ks) // stride + 1 for s in dims]
    co = pad_to(co_real, 16)
    co32 = pad_to(co, 32)
    geom = Geom(n, d, h, w, do, ho, wo, ci, co, ks, stride, pad)
    gt = rnd((n, do, ho, wo, co32), 7)
    gt[..., co_real:] = 0
    _, gc, gg = views(hip, gt, n, do, ho, wo)
    _, xc, xg = views(hip, rnd((n, d, h, w, ci), 8, relu=True), n, d, h, w)
    tiled = nsplit is None
    if tiled:
        nsplit = hip.wgrad_nsplit(geom)
        assert nsplit > 0, "expected the tiled kernel for %r" % (case,)
    gp = Pair(hip, torch.full((n, nsplit, ks ** 3, co32, ci), 7.0))
    run(hip, EMU.wgrad(gc, xc, geom, nsplit, gp.c), hip.wgrad(gg, xg, geom, nsplit, gp.g))
    if tiled:  # the tiled kernel partitions voxels by brick, the emulation by flat ranges: compare per-sample sums
        assert rel_err(gp.sync_back().sum(1), gp.c.sum(1)) < F32 * 5
    else:
        check(gp, F32 * 5, "wgrad %r" % (case,))
    # slabs sum to the autograd weight gradient
    x = xc.buf.float().permute(0, 4, 1, 2, 3)
    wz = torch.zeros(co_real, ci, ks, ks, ks, requires_grad=True)
    torch.nn.functional.conv3d(x, wz, None, stride, pad).backward(gt[..., :co_real].float().permute(0, 4, 1, 2, 3))
    got = gp.sync_back().sum((0, 1))[:, :co_real].permute(1, 2, 0).reshape(wz.shape)
    assert rel_err(got, wz.grad) < 1e-3


@pytest.mark.parametrize("seed", range(16))
def test_conv_kernels_random_geometries(hip, seed):
    """Random volumes (tiled-friendly and ragged, down to single voxels), channel counts, kernel sizes, strides and
    epilogue options through the three convolution entry points (forward, data gradient, weight gradient); a run of
    360 such draws was clean when this was added."""
    rs = np.random.RandomState(4200 + seed)
    if rs.randint(0, 3) == 0:
        d3 = (int(rs.choice([2, 4, 6])), int(rs.choice([8, 16, 24])), int(rs.choice([32, 64, 80, 96])))
    else:
        d3 = (int(rs.randint(1, 7)), int(rs.randint(1, 20)), int(rs.randint(1, 50)))
    n = int(rs.randint(1, 4))
    ci = int(rs.choice([32, 32, 64, 128]))
    co_real = int(rs.choice([3, 15, 16, 32, 32, 45, 64, 128]))
    ks, stride = int(rs.choice([1, 3, 3])), int(rs.choice([1, 1, 2]))
    per_sample, has_res, relu = bool(rs.rand() < 0.6), bool(rs.rand() < 0.4), bool(rs.rand() < 0.5)
    fp32 = bool(co_real % 16 != 0 or rs.rand() < 0.2)
    test_conv_forward(hip, (n, d3, ci, co_real, ks, stride, per_sample, has_res, relu, fp32))
    test_conv_transposed_is_data_gradient(hip, (n, d3, ci, co_real, ks, stride))
    test_wgrad(hip, (n, d3, ci, co_real, ks, stride, int(rs.randint(1, 5))))


@pytest.mark.parametrize("c,vox,nsplit,with_b", [(32, 1000, 3, False), (64, 333, 1, True), (128, 2048, 7, True), (32, 64, 4, False)])
def test_chan_stats(hip, c, vox, nsplit, with_b):
    n = 2
    _, ac, ag = views(hip, rnd((n, 1, 1, vox, c), 9, relu=True), n, 1, 1, vox)
    bc = bg = None
    if with_b:
        _, bc, bg = views(hip, rnd((n, 1, 1, vox, c), 10), n, 1, 1, vox)
    out = Pair(hip, torch.zeros(n, nsplit, c, 2))
    run(hip, EMU.chan_stats(ac, bc, nsplit, out.c), hip.chan_stats(ag, bg, nsplit, out.g))
    check(out, F32, "chan_stats")


@pytest.mark.parametrize("ci,co_real,ks,stride,norm,bias,dims,slc", [
    (32, 32, 3, 1, True, False, (4, 8, 8), None), (32, 64, 3, 2, True, False, (4, 8, 8), None),
    (64, 32, 1, 1, True, False, (2, 4, 4), None), (32, 15, 3, 1, False, True, (4, 4, 4), None),
    (64, 128, 1, 1, False, True, (2, 2, 2), (192, 64)), (32, 32, 3, 2, True, False, (1, 2, 4), None)])
def test_fold_and_wgrad_fold(hip, ci, co_real, ks, stride, norm, bias, dims, slc):
    n, groups = 2, 8
    d, h, w = dims
    pad = ks // 2
    do, ho, wo = [(s + 2 * pad - ks) // stride + 1 for s in dims]
    co = pad_to(co_real, 16)
    cit, cio = slc if slc else (0, 0)
    geom = Geom(n, d, h, w, do, ho, wo, ci, co, ks, stride, pad, cit, cio)
    ntap = ks ** 3
    w32 = Pair(hip, rnd((co_real, cit or ci, ks, ks, ks), 11, torch.float32, 0.1))
    b32 = Pair(hip, rnd((co_real,), 12, torch.float32)) if bias else None
    gam = Pair(hip, 1 + 0.2 * rnd((ci,), 13, torch.float32)) if norm else None
    bet = Pair(hip, 0.2 * rnd((ci,), 14, torch.float32)) if norm else None
    nsplit = 3
    stats = None
    if norm:
        x = torch.relu(rnd((n, d * h * w, ci), 15, torch.float32) + 0.3)
        st = torch.zeros(n, nsplit, ci, 2)
        st[:, 0, :, 0] = x.sum(1)
        st[:, 0, :, 1] = (x * x).sum(1)
        stats = Pair(hip, st)
    nw = n if norm else 1
    wf = Pair(hip, torch.zeros(nw, ntap, co, ci, dtype=torch.bfloat16))
    bt = Pair(hip, torch.zeros(nw, 64, co))
    mr = Pair(hip, torch.zeros(n, groups, 2)) if norm else None

    def g_(p, side):
        return None if p is None else getattr(p, side)
    run(hip,
        EMU.fold_fwd(w32.c, g_(b32, "c"), g_(gam, "c"), g_(bet, "c"), g_(stats, "c"), nsplit, groups, 1e-5, geom, ci, co_real, wf.c, bt.c, g_(mr, "c")),
        hip.fold_fwd(w32.g, g_(b32, "g"), g_(gam, "g"), g_(bet, "g"), g_(stats, "g"), nsplit, groups, 1e-5, geom, ci, co_real, wf.g, bt.g, g_(mr, "g")))
    check(wf, BF, "fold wf")
    # the optional data-gradient packing out of the same launch equals rtp_pack_dgrad_w
    cok = pad_to(co, 32)
    wd1, wd2 = Pair(hip, torch.zeros(ntap, ci, cok, dtype=torch.bfloat16)), Pair(hip, torch.ones(ntap, ci, cok, dtype=torch.bfloat16))
    hip.pack_dgrad_w(w32.g, geom, ci, co_real, wd1.g)(hip.stream())
    hip.fold_fwd(w32.g, g_(b32, "g"), g_(gam, "g"), g_(bet, "g"), g_(stats, "g"), nsplit, groups, 1e-5, geom, ci, co_real, wf.g, bt.g, g_(mr, "g"), wd2.g)(hip.stream())
    torch.cuda.synchronize()
    assert torch.equal(wd1.sync_back(), wd2.sync_back())
    check(bt, F32 * 5, "fold btab")
    if norm:
        check(mr, F32, "fold mr")
    # backward fold
    co32 = pad_to(co, 32)
    gp = Pair(hip, rnd((n, 2, ntap, co32, ci), 16, torch.float32))
    cs = Pair(hip, rnd((n, 64, co32), 17, torch.float32))
    dw = Pair(hip, rnd((co_real, cit or ci, ks, ks, ks), 18, torch.float32))
    db = Pair(hip, rnd((co_real,), 19, torch.float32)) if bias else None
    for acc in (0, 1):
        run(hip,
            EMU.wgrad_fold(gp.c, 2, cs.c, g_(mr, "c"), g_(gam, "c"), g_(bet, "c"), groups, geom, ci, co_real, dw.c, g_(db, "c"), acc),
            hip.wgrad_fold(gp.g, 2, cs.g, g_(mr, "g"), g_(gam, "g"), g_(bet, "g"), groups, geom, ci, co_real, dw.g, g_(db, "g"), acc))
        check(dw, F32 * 5, "wgrad_fold dw acc=%d" % acc)
        if bias:
            check(db, F32 * 5, "wgrad_fold db")


def test_gn_bwd_coeffs_and_class_sums(hip):
    n, c, groups, vox, nsplit = 3, 64, 8, 5000, 4
    pq = Pair(hip, rnd((n, nsplit, c, 2), 20, torch.float32))
    mr = Pair(hip, torch.rand(n, groups, 2, generator=torch.Generator().manual_seed(1)) + 0.5)
    gam = Pair(hip, 1 + 0.2 * rnd((c,), 21, torch.float32))
    co = Pair(hip, torch.zeros(n * c * 5))
    dg, db = Pair(hip, rnd((c,), 22, torch.float32)), Pair(hip, rnd((c,), 23, torch.float32))
    for acc in (0, 1):
        run(hip, EMU.gn_bwd_coeffs(pq.c, nsplit, mr.c, gam.c, n, c, groups, vox, co.c, dg.c, db.c, acc),
            hip.gn_bwd_coeffs(pq.g, nsplit, mr.g, gam.g, n, c, groups, vox, co.g, dg.g, db.g, acc))
        assert rel_err(co.sync_back()[:n * c * 3], co.c[:n * c * 3]) < F32, "coeff"
        check(dg, F32, "dgamma")
        check(db, F32, "dbeta")
    for dims, ch in (((4, 6, 10), 32), ((1, 2, 4), 64), ((2, 1, 3), 128)):
        d, h, w = dims
        _, gc, gg = views(hip, rnd((2, d, h, w, ch), 24), 2, d, h, w)
        out = Pair(hip, torch.ones(2, 64, ch))
        run(hip, EMU.class_sums(gc, 3, None, out.c), hip.class_sums(gg, 3, hip.alloc((2, 3, 64, ch), "f32"), out.g))
        check(out, F32 * 5, "class_sums %r" % (dims,))


def test_split_conv_slices_on_the_tiled_kernels(hip):
    """A 128-channel 3x3x3 conv as four 32-channel input slices (graph.SplitConvOp): forward with the fp32 partial-sum
    chain (rtp_conv_igemm_acc), data gradient into channel slices of one buffer, weight gradient over slices of x --
    each against the emulation, and the forward against one plain conv over all 128 channels."""
    n, d, h, w, ci, co = 2, 4, 8, 32, 128, 32
    xp, xc, xg = views(hip, rnd((n, d, h, w, ci), 301, relu=True), n, d, h, w)
    w32 = rnd((co, ci, 3, 3, 3), 302, torch.float32, 0.05)
    bias = rnd((co,), 303, torch.float32)
    yp, yc, yg = views(hip, torch.zeros(n, d, h, w, co, dtype=torch.bfloat16), n, d, h, w)
    acc = Pair(hip, torch.zeros(n, d * h * w, co))
    accv_c = View(acc.c.view(n, d, h, w, co), n, d, h, w, co, 0, co)
    accv_g = View(acc.g.view(n, d, h, w, co), n, d, h, w, co, 0, co)
    for k in range(4):
        geom = Geom(n, d, h, w, d, h, w, 32, co, 3, 1, 1, ci, 32 * k)
        sl_c, sl_g = View(xc.buf, n, d, h, w, ci, 32 * k, 32), View(xg.buf, n, d, h, w, ci, 32 * k, 32)
        assert hip.conv_tiled_ok(sl_g, geom, False) and EMU.conv_tiled_ok(sl_c, geom, False)
        wf = Pair(hip, w32[:, 32 * k:32 * k + 32].permute(2, 3, 4, 0, 1).reshape(1, 27, co, 32).contiguous().to(torch.bfloat16))
        last = k == 3
        bt = Pair(hip, bias.view(1, 1, co).expand(1, 64, co).contiguous()) if last else None
        a_c, a_g = ((acc.c, co), (acc.g, co)) if k else (None, None)
        run(hip, EMU.conv(sl_c, wf.c, False, bt.c if last else None, None, yc if last else accv_c, geom, last, False, not last, None, a_c),
            hip.conv(sl_g, wf.g, False, bt.g if last else None, None, yg if last else accv_g, geom, last, False, not last, None, a_g))
        if not last:
            check(acc, F32 * 5, "partial sum after slice %d" % k)
    check(yp, BF, "split conv forward")
    ref = torch.relu(torch.nn.functional.conv3d(xc.buf.float().permute(0, 4, 1, 2, 3), w32.to(torch.bfloat16).float(), bias, 1, 1))
    assert rel_err(yp.sync_back().float().permute(0, 4, 1, 2, 3), ref) < BF
    # backward: data gradient slices + weight gradient over x slices
    gp_, gc, gg = views(hip, rnd((n, d, h, w, 32), 304), n, d, h, w)
    dxp, _, _ = views(hip, torch.zeros(n, d, h, w, ci, dtype=torch.bfloat16), n, d, h, w)
    for k in range(4):
        geom = Geom(n, d, h, w, d, h, w, 32, co, 3, 1, 1, ci, 32 * k)
        wd = Pair(hip, w32[:, 32 * k:32 * k + 32].permute(2, 3, 4, 1, 0).reshape(27, 32, co).contiguous().to(torch.bfloat16))
        dk_c, dk_g = View(dxp.c, n, d, h, w, ci, 32 * k, 32), View(dxp.g, n, d, h, w, ci, 32 * k, 32)
        run(hip, EMU.conv(gc, wd.c, False, None, None, dk_c, geom, False, True, False), hip.conv(gg, wd.g, False, None, None, dk_g, geom, False, True, False))
        S = hip.wgrad_nsplit(geom)
        assert S > 0
        gpk = Pair(hip, torch.zeros(n, S, 27, 32, 32))
        sl_c, sl_g = View(xc.buf, n, d, h, w, ci, 32 * k, 32), View(xg.buf, n, d, h, w, ci, 32 * k, 32)
        run(hip, EMU.wgrad(gc, sl_c, geom, S, gpk.c), hip.wgrad(gg, sl_g, geom, S, gpk.g))
        assert rel_err(gpk.sync_back().sum(1), gpk.c.sum(1)) < F32 * 5, "wgrad slice %d" % k
    check(dxp, BF, "split conv data gradient")


def test_deferred_tail_batches(hip):
    """rtp_tail_*: several independent items per launch equal the single-item entry points' emulation, including slab
    counts that take the 8-deep unrolled loops (nsplit 37) and every supported channel width."""
    n, groups = 3, 8
    stage_a, stage_a_emu, stage_b, stage_b_emu, outs = [], [], [], [], []
    for k, (ci, co_real, ks, stride, norm, bias, nsplit) in enumerate([
            (32, 32, 3, 1, True, False, 37), (64, 64, 3, 2, True, False, 5), (128, 64, 1, 1, True, False, 2),
            (32, 15, 3, 1, False, True, 33), (256, 256, 3, 1, True, False, 1), (32, 32, 3, 1, False, False, 9)]):
        d, h, w = 4, 8, 8
        pad = ks // 2
        do, ho, wo = [(s + 2 * pad - ks) // stride + 1 for s in (d, h, w)]
        co = pad_to(co_real, 16)
        co32 = pad_to(co, 32)
        geom = Geom(n, d, h, w, do, ho, wo, ci, co, ks, stride, pad, 0, 0)
        ntap = ks ** 3
        gp = Pair(hip, rnd((n, nsplit, ntap, co32, ci), 100 + k, torch.float32))
        cs_part = Pair(hip, rnd((n, 7, 64, co32), 120 + k, torch.float32)) if (norm or bias) else None
        cs = Pair(hip, torch.zeros(n, 64, co32)) if (norm or bias) else None
        mr = Pair(hip, torch.rand(n, groups, 2, generator=torch.Generator().manual_seed(k)) + 0.5) if norm else None
        gam = Pair(hip, 1 + 0.2 * rnd((ci,), 140 + k, torch.float32)) if norm else None
        bet = Pair(hip, 0.2 * rnd((ci,), 160 + k, torch.float32)) if norm else None
        dw = Pair(hip, torch.zeros(co_real, ci, ks, ks, ks))
        db = Pair(hip, torch.zeros(co_real)) if bias else None
        outs += [(dw, "dw%d" % k)] + ([(db, "db%d" % k)] if bias else [])

        def side(p, sd):
            return None if p is None else getattr(p, sd)
        for sd, sa, sb in (("c", stage_a_emu, stage_b_emu), ("g", stage_a, stage_b)):
            if cs is not None:
                sa.append(("class_reduce", side(cs_part, sd), 7, n, co32, side(cs, sd)))
            sb.append(("wgrad_fold", side(gp, sd), nsplit, side(cs, sd), side(mr, sd), side(gam, sd), side(bet, sd), groups,
                       geom, ci, co_real, side(dw, sd), side(db, sd), 0))
        if norm:
            c = ci
            pq = Pair(hip, rnd((n, 3, c, 2), 180 + k, torch.float32))
            coeff = Pair(hip, torch.zeros(n * c * 5))
            dg, dbt = Pair(hip, torch.zeros(c)), Pair(hip, torch.zeros(c))
            run(hip, EMU.gn_bwd_coeffs(pq.c, 3, mr.c, gam.c, n, c, groups, 1000, coeff.c, None, None, 0),
                hip.gn_bwd_coeffs(pq.g, 3, mr.g, gam.g, n, c, groups, 1000, coeff.g, None, None, 0))
            stage_a_emu.append(("gn_param", coeff.c, n, c, dg.c, dbt.c, 0))
            stage_a.append(("gn_param", coeff.g, n, c, dg.g, dbt.g, 0))
            outs += [(dg, "dgamma%d" % k), (dbt, "dbeta%d" % k)]
    run(hip, EMU.tail(stage_a_emu), hip.tail(stage_a))
    run(hip, EMU.tail(stage_b_emu), hip.tail(stage_b))
    for pair, what in outs:
        check(pair, F32 * 5, what)


def test_grad_combine_fuse_upsample(hip):
    n, d, h, w, c = 2, 4, 8, 16, 32
    _, xc, xg = views(hip, rnd((n, d, h, w, c), 30, relu=True), n, d, h, w)
    _, t1c, t1g = views(hip, rnd((n, d, h, w, c), 31), n, d, h, w)
    _, t2c, t2g = views(hip, rnd((n, d, h, w, 2 * c), 32), n, d, h, w, co=c, c=c)  # channel-slice view
    cf = Pair(hip, rnd((n * c * 5,), 33, torch.float32))
    op, oc, og = views(hip, torch.zeros(n, d, h, w, c, dtype=torch.bfloat16), n, d, h, w)
    run(hip, EMU.grad_combine([(t1c, None), (t2c, cf.c)], xc, xc, oc), hip.grad_combine([(t1g, None), (t2g, cf.g)], xg, xg, og))
    check(op, BF, "grad_combine")
    # the same combine with the per-boundary-class sums fused in (incl. depth-1 / width-1 volumes, 64 channels, 2 GN terms)
    for (dd, hh, ww), ch, ns in (((4, 8, 16), 32, 5), ((1, 6, 1), 64, 2), ((16, 8, 24), 32, 64), ((3, 5, 80), 64, 4)):
        _, x2c, x2g = views(hip, rnd((n, dd, hh, ww, ch), 130, relu=True), n, dd, hh, ww)
        _, a1c, a1g = views(hip, rnd((n, dd, hh, ww, ch), 131), n, dd, hh, ww)
        _, a2c, a2g = views(hip, rnd((n, dd, hh, ww, 2 * ch), 132), n, dd, hh, ww, co=ch, c=ch)
        _, a3c, a3g = views(hip, rnd((n, dd, hh, ww, ch), 133), n, dd, hh, ww)
        cf2, cf3 = Pair(hip, rnd((n * ch * 5,), 134, torch.float32)), Pair(hip, rnd((n * ch * 5,), 135, torch.float32))
        o2p, o2c, o2g = views(hip, torch.zeros(n, dd, hh, ww, ch, dtype=torch.bfloat16), n, dd, hh, ww)
        sc_c, sc_g = torch.zeros(n, ns, 64, ch), hip.alloc((n, ns, 64, ch), "f32")
        cs = Pair(hip, torch.ones(n, 64, ch))
        run(hip, EMU.grad_combine([(a1c, None), (a2c, cf2.c), (a3c, cf3.c)], x2c, x2c, o2c, (ns, sc_c)),
            hip.grad_combine([(a1g, None), (a2g, cf2.g), (a3g, cf3.g)], x2g, x2g, o2g, (ns, sc_g)))
        check(o2p, BF, "grad_combine_cls out %r" % ((dd, hh, ww),))
        run(hip, EMU.class_sums_reduce(sc_c, ns, n, ch, cs.c), hip.class_sums_reduce(sc_g, ns, n, ch, cs.g))
        # the sums are of the bf16-rounded outputs, which may differ by an ulp between the two -> bf16-level tolerance
        check(cs, BF * 2, "grad_combine_cls sums %r" % ((dd, hh, ww),))
        # and they must equal the standalone scan of the tensor the kernel itself wrote, to fp32 accuracy
        ref = hip.alloc((n, 64, ch), "f32")
        hip.class_sums(o2g, 3, hip.alloc((n, 3, 64, ch), "f32"), ref)(hip.stream())
        torch.cuda.synchronize()
        assert rel_err(cs.g.cpu(), ref.cpu()) < F32, "fused vs standalone class sums"
    # fuse rows: same-res + three lower resolutions (the stage-4 row-0 pattern)
    lows = [((2, 4, 8), 35), ((1, 2, 4), 36), ((1, 1, 2), 37)]
    tc, tg = [t1c], [t1g]
    for (ld, lh, lw), seed in lows:
        _, lc, lg = views(hip, rnd((n, ld, lh, lw, c), seed), n, ld, lh, lw)
        tc.append(lc)
        tg.append(lg)
    for relu in (True, False):
        run(hip, EMU.fuse_sum(tc, None, oc, relu), hip.fuse_sum(tg, None, og, relu))
        check(op, BF, "fuse_sum relu=%s" % relu)
        # with the statistics epilogue: same row, and the partials sum to a chan_stats pass over the tensor the kernel stored
        S = hip.fuse_stats_nsplit(og)
        assert S > 0
        st = hip.alloc((n, S, c, 2), "f32")
        before = op.g.clone()
        hip.fuse_sum(tg, None, og, relu, (S, st))(hip.stream())
        ref = hip.alloc((n, 3, c, 2), "f32")
        hip.chan_stats(og, None, 3, ref)(hip.stream())
        torch.cuda.synchronize()
        assert torch.equal(op.g, before), "statistics variant stores the same row"
        assert rel_err(st.sum(1).cpu(), ref.sum(1).cpu()) < F32, "fused fuse-row statistics"
    # native row width (160 <- 80, 40, 20: the row-run kernel) and a width that takes the point-per-thread kernel
    for (dd, hh, ww), lowdims in (((2, 4, 160), [(1, 2, 80), (1, 1, 40), (2, 1, 20)]), ((2, 3, 12), [(1, 2, 6), (1, 1, 5)])):
        _, b0c, b0g = views(hip, rnd((n, dd, hh, ww, c), 230), n, dd, hh, ww)
        bp, boc, bog = views(hip, torch.zeros(n, dd, hh, ww, c, dtype=torch.bfloat16), n, dd, hh, ww)
        tc2, tg2 = [b0c], [b0g]
        for k, (ld, lh, lw) in enumerate(lowdims):
            _, lc, lg = views(hip, rnd((n, ld, lh, lw, c), 231 + k), n, ld, lh, lw)
            tc2.append(lc)
            tg2.append(lg)
        run(hip, EMU.fuse_sum(tc2, None, boc, True), hip.fuse_sum(tg2, None, bog, True))
        check(bp, BF, "fuse_sum %r" % ((dd, hh, ww),))
        S = hip.fuse_stats_nsplit(bog)
        st = hip.alloc((n, S, c, 2), "f32")
        hip.fuse_sum(tg2, None, bog, True, (S, st))(hip.stream())
        ref = hip.alloc((n, 2, c, 2), "f32")
        hip.chan_stats(bog, None, 2, ref)(hip.stream())
        torch.cuda.synchronize()
        assert rel_err(st.sum(1).cpu(), ref.sum(1).cpu()) < F32, "fused fuse-row statistics %r" % ((dd, hh, ww),)
    # upsample adjoint for x2, x4, x8 and a non-integer ratio
    for (ld, lh, lw), ch in (((2, 4, 8), 32), ((1, 2, 4), 64), ((1, 1, 2), 128), ((3, 5, 7), 32)):
        _, gc, gg = views(hip, rnd((n, d, h, w, ch), 38), n, d, h, w)
        lp, lc, lg = views(hip, torch.zeros(n, ld, lh, lw, ch, dtype=torch.bfloat16), n, ld, lh, lw)
        run(hip, EMU.upsample_bwd(gc, lc), hip.upsample_bwd(gg, lg))
        check(lp, BF, "upsample_bwd %r" % ((ld, lh, lw),))


@pytest.mark.parametrize("dims,ch,groups,S", [((4, 8, 16), 32, 8, 5), ((2, 6, 10), 64, 8, 80), ((3, 5, 80), 32, 4, 1), ((2, 4, 8), 64, 8, 37)])
def test_grad_combine_with_lazy_groupnorm_coefficients(hip, dims, ch, groups, S):
    """rtp_grad_combine_cls_lazy: the fan-in pass computes the GroupNorm-backward coefficients of a GN term in its own prologue
    from the statistics partials (P, Q) of the data gradient.  Must equal rtp_gn_bwd_coeffs + rtp_grad_combine_cls on the same
    inputs (same arithmetic), write the same coefficient / dgamma-dbeta-partial table, and agree with the emulation."""
    from rt_pose_amd.graph import LazyCoeff
    n = 2
    d, h, w = dims
    _, xc, xg = views(hip, rnd((n, d, h, w, ch), 330, relu=True), n, d, h, w)
    _, a1c, a1g = views(hip, rnd((n, d, h, w, ch), 331), n, d, h, w)
    _, a2c, a2g = views(hip, rnd((n, d, h, w, ch), 332), n, d, h, w)
    pq = Pair(hip, rnd((n, S, ch, 2), 333, torch.float32))
    mr = Pair(hip, torch.stack([rnd((n, groups), 334, torch.float32) * 0.1, torch.rand(n, groups) + 0.5], -1).contiguous())
    gam = Pair(hip, rnd((ch,), 335, torch.float32) * 0.2 + 1.0)
    vox = d * h * w
    ns = 3
    res = {}
    for mode in ("lazy", "launch"):
        cf = Pair(hip, torch.zeros(n * ch * 5))
        op, oc, og = views(hip, torch.zeros(n, d, h, w, ch, dtype=torch.bfloat16), n, d, h, w)
        sc_c, sc_g = torch.zeros(n, ns, 64, ch), hip.alloc((n, ns, 64, ch), "f32")
        if mode == "lazy":
            lzc = LazyCoeff(None, "t", 0, pq.c, S, mr.c, gam.c, n, ch, groups, vox, cf.c)
            lzg = LazyCoeff(None, "t", 0, pq.g, S, mr.g, gam.g, n, ch, groups, vox, cf.g)
            run(hip, EMU.grad_combine([(a1c, None), (a2c, lzc)], xc, xc, oc, (ns, sc_c)),
                hip.grad_combine([(a1g, None), (a2g, lzg)], xg, xg, og, (ns, sc_g)))
            check(op, BF, "lazy combine vs emulation")
            assert rel_err(cf.g.cpu(), cf.c) < 1e-4, "coefficient table vs emulation"
        else:
            hip.gn_bwd_coeffs(pq.g, S, mr.g, gam.g, n, ch, groups, vox, cf.g, None, None, 0)(hip.stream())
            hip.grad_combine([(a1g, None), (a2g, cf.g)], xg, xg, og, (ns, sc_g))(hip.stream())
            torch.cuda.synchronize()
        res[mode] = (op.g.float().cpu(), cf.g.cpu().clone(), sc_g.cpu().clone())
    assert rel_err(res["lazy"][1], res["launch"][1]) < 1e-6, "same coefficient / partial table as the stand-alone launch"
    assert rel_err(res["lazy"][0], res["launch"][0]) < 1e-3
    assert rel_err(res["lazy"][2].sum(1), res["launch"][2].sum(1)) < 1e-3


@pytest.mark.parametrize("n,c,cs,co,vox", [(2, 72, 80, 0, 64 * 160), (3, 5, 16, 4, 777), (1, 32, 32, 0, 63), (2, 7, 12, 4, 65)])
def test_unpack_f32_channels_last_to_ncdhw(hip, n, c, cs, co, vox):
    """rtp_unpack_ncdhw_f32 (the DCN head's offsets: fp32 channels-last conv output -> the operator's NCHW): exact copy, channel
    slices, ragged voxel counts, channel counts that are not a multiple of 4."""
    import ctypes as C
    x = torch.randn(n, vox, cs, device=hip.device)
    y = torch.full((n, c, vox), -7.0, device=hip.device)
    rc = hip.lib.rtp_unpack_ncdhw_f32(C.c_void_p(x.data_ptr()), cs, co, C.c_void_p(y.data_ptr()), n, c, vox, hip.stream())
    torch.cuda.synchronize()
    assert rc == 0
    assert torch.equal(y.cpu(), x[:, :, co:co + c].permute(0, 2, 1).contiguous().cpu())


@pytest.mark.parametrize("n,c,cpad,cs,co,vox,relu,two", [(2, 72, 96, 96, 0, 64 * 160, False, False), (3, 32, 32, 64, 32, 777, True, True),
                                                         (1, 5, 8, 16, 8, 63, False, True), (2, 32, 32, 32, 0, 65, True, False)])
def test_pack_ex_and_unpack_through_lds_tiles(hip, n, c, cpad, cs, co, vox, relu, two):
    """rtp_pack_ncdhw_ex (fp32 NC(D)HW [+ second addend] [ReLU] -> bf16 channels-last slice, padding channels zero) and
    rtp_unpack_ncdhw (bf16 channels-last slice -> fp32 NC(D)HW): exact against torch on ragged voxel counts and channel slices."""
    import ctypes as C
    from rt_pose_amd.backend import _act
    x = torch.randn(n, c, vox, device=hip.device)
    x2 = torch.randn(n, c, vox, device=hip.device) if two else None
    y = torch.full((n, vox, cs), 3.0, dtype=torch.bfloat16, device=hip.device)
    yv = View(y, n, 1, 1, vox, cs, co, cpad)
    rc = hip.lib.rtp_pack_ncdhw_ex(C.c_void_p(x.data_ptr()), C.c_void_p(x2.data_ptr()) if two else None, _act(yv), n, c, vox, int(relu), hip.stream())
    torch.cuda.synchronize()
    assert rc == 0
    ref = x + x2 if two else x.clone()
    if relu:
        ref = ref.clamp_min(0)
    want = torch.full((n, vox, cs), 3.0, dtype=torch.bfloat16)
    want[:, :, co:co + cpad] = 0
    want[:, :, co:co + c] = ref.permute(0, 2, 1).to(torch.bfloat16).cpu()
    assert torch.equal(y.cpu(), want)
    back = torch.full((n, c, vox), -1.0, device=hip.device)
    rc = hip.lib.rtp_unpack_ncdhw(_act(View(y, n, 1, 1, vox, cs, co, cpad)), C.c_void_p(back.data_ptr()), n, c, vox, hip.stream())
    torch.cuda.synchronize()
    assert rc == 0
    assert torch.equal(back.cpu(), want[:, :, co:co + c].float().permute(0, 2, 1).contiguous())


def test_stem_and_pack(hip):
    n, d, h, w = 2, 4, 8, 16
    x = Pair(hip, torch.relu(rnd((n, 1, d, h, w), 40, torch.float32)))
    wt, b = Pair(hip, rnd((32, 1, 1, 1, 1), 41, torch.float32)), Pair(hip, rnd((32,), 42, torch.float32))
    yp, yc, yg = views(hip, torch.zeros(n, d, h, w, 32, dtype=torch.bfloat16), n, d, h, w)
    run(hip, EMU.stem_fwd(x.c, wt.c, b.c, yc), hip.stem_fwd(x.g, wt.g, b.g, yg))
    check(yp, BF, "stem_fwd")
    # the same stem with the statistics of its stored output as an epilogue (rtp_stem_fwd_stats): output bit for bit, statistics = a
    # read pass over that output (rtp_chan_stats) up to the summation order
    for nn, dims2 in ((n, (d, h, w)), (3, (6, 10, 20))):
        d2, h2, w2 = dims2
        x2 = Pair(hip, torch.relu(rnd((nn, 1, d2, h2, w2), 45, torch.float32)))
        ya = views(hip, torch.zeros(nn, d2, h2, w2, 32, dtype=torch.bfloat16), nn, d2, h2, w2)
        yb = views(hip, torch.zeros(nn, d2, h2, w2, 32, dtype=torch.bfloat16), nn, d2, h2, w2)
        S = hip.stem_stats_nsplit(nn, 32, d2 * h2 * w2)
        assert S > 0
        st = hip.alloc((nn, S, 32, 2), "f32")
        st.fill_(float("nan"))     # every partial must be written
        ref = hip.alloc((nn, 3, 32, 2), "f32")
        hip.stem_fwd(x2.g, wt.g, b.g, ya[2])(hip.stream())
        hip.stem_fwd_stats(x2.g, wt.g, b.g, yb[2], st, S)(hip.stream())
        hip.chan_stats(yb[2], None, 3, ref)(hip.stream())
        torch.cuda.synchronize()
        assert torch.equal(ya[0].g, yb[0].g)
        assert rel_err(st.sum(1).cpu(), ref.sum(1).cpu()) < F32
    _, gc, gg = views(hip, rnd((n, d, h, w, 32), 43), n, d, h, w)
    dw, db = Pair(hip, torch.zeros(32, 1, 1, 1, 1)), Pair(hip, torch.zeros(32))
    sc = hip.alloc((hip.stem_bwd_blocks(), 32, 2), "f32")
    run(hip, EMU.stem_bwd(x.c, gc, None, dw.c, db.c, 0), hip.stem_bwd(x.g, gg, sc, dw.g, db.g, 0))
    check(dw, F32 * 5, "stem dw")
    check(db, F32 * 5, "stem db")
    x32 = Pair(hip, rnd((n, 32, d, h, w), 44, torch.float32))
    pp, pc, pg = views(hip, torch.zeros(n, d, h, w, 32, dtype=torch.bfloat16), n, d, h, w)
    run(hip, EMU.pack_ncdhw(x32.c, pc, 32), hip.pack_ncdhw(x32.g, pg, 32))
    assert torch.equal(pp.sync_back(), pp.c)
    back = Pair(hip, torch.zeros(n, 32, d, h, w))
    run(hip, EMU.unpack_ncdhw(pc, back.c, 32), hip.unpack_ncdhw(pg, back.g, 32))
    assert torch.equal(back.sync_back(), back.c)


@pytest.mark.parametrize("ncls,nreg", [(15, 3), (1, 45)])
def test_losses_and_decode(hip, ncls, nreg):
    from oracle import hrradarpose_ref as O
    n, dims = 2, (4, 8, 16)
    d, h, w = dims
    ex = O.synth_example(n, 1, dims, seed=5, one_hm=ncls == 1)["rdr"]
    m = ex["ind"][0].shape[1]
    hm_c, rg_c = pad_to(ncls, 16), pad_to(nreg, 16)
    hp, hc, hg = views(hip, rnd((n, d, h, w, hm_c), 50, torch.float32, 2.0) - 2.0, n, d, h, w)
    rp, rc, rg = views(hip, rnd((n, d, h, w, rg_c), 51, torch.float32), n, d, h, w)
    # duplicate a positive voxel (two joints in one voxel) and mask one out
    if m > 2:
        ex["ind"][0][0, 1] = ex["ind"][0][0, 0]
        ex["mask"][0][1, 2] = 0
    tgt, ind, mask, cat = Pair(hip, ex["hm"][0]), Pair(hip, ex["ind"][0]), Pair(hip, ex["mask"][0]), Pair(hip, ex["cat"][0])
    pose = Pair(hip, ex["anno_pose"][0].reshape(n, m, nreg).contiguous())
    cw = Pair(hip, torch.linspace(1, 2, nreg))
    gh_c, gr_c = pad_to(hm_c, 32), pad_to(rg_c, 32)
    ghp, ghc, ghg = views(hip, torch.ones(n, d, h, w, gh_c, dtype=torch.bfloat16), n, d, h, w)
    grp, grc, grg = views(hip, torch.ones(n, d, h, w, gr_c, dtype=torch.bfloat16), n, d, h, w)
    lh, lr = Pair(hip, torch.zeros(1)), Pair(hip, torch.zeros(nreg + 1))
    run(hip, EMU.focal_loss(hc, tgt.c, ind.c, mask.c, cat.c, ncls, 1.0, None, lh.c, ghc),
        hip.focal_loss(hg, tgt.g, ind.g, mask.g, cat.g, ncls, 1.0, hip.focal_scratch(n), lh.g, ghg))
    check(lh, 1e-4, "focal loss value")
    check(ghp, BF, "focal grad")
    # without the padding stores: identical class channels, padding channels left as they were
    g3 = torch.full((n, d, h, w, gh_c), 7.0, dtype=torch.bfloat16, device=hip.device)
    lh3 = torch.zeros(1, device=hip.device)
    hip.focal_loss(hg, tgt.g, ind.g, mask.g, cat.g, ncls, 1.0, hip.focal_scratch(n), lh3, View(g3, n, d, h, w, gh_c, 0, gh_c), False)(hip.stream())
    torch.cuda.synchronize()
    wr = (ncls + 7) // 8 * 8
    assert torch.equal(g3[..., :wr].cpu(), ghp.g[..., :wr].cpu()) and bool((g3[..., wr:] == 7.0).all()) and torch.equal(lh3.cpu(), lh.g.cpu())
    run(hip, EMU.reg_loss(rc, pose.c, ind.c, mask.c, cw.c, nreg, 0.2, lr.c, grc),
        hip.reg_loss(rg, pose.g, ind.g, mask.g, cw.g, nreg, 0.2, lr.g, grg))
    check(lr, 1e-4, "reg loss values")
    check(grp, BF, "reg grad")
    # the stateful variant (no zero fill: only the previous call's voxels are cleared), three calls with moving objects, against
    # the zero-filling kernel on the same inputs -- bit for bit, values and the whole gradient tensor
    g2 = torch.zeros(n, d, h, w, gr_c, dtype=torch.bfloat16, device=hip.device)
    g2v = View(g2, n, d, h, w, gr_c, 0, gr_c)
    prev = torch.full((n, m), -1, dtype=torch.int64, device=hip.device)
    lr2 = torch.zeros(nreg + 1, device=hip.device)
    gen = torch.Generator().manual_seed(7)
    for it in range(3):
        ind_i = ind.c.clone() if it == 0 else torch.randint(0, d * h * w, (n, m), generator=gen)
        if it == 2 and m > 1:
            ind_i[:, 1] = ind_i[:, 0]           # duplicate voxels within a frame (their gradients add, like gather's backward)
        ind_g = ind_i.to(hip.device)
        hip.reg_loss(rg, pose.g, ind_g, mask.g, cw.g, nreg, 0.2, lr.g, grg)(hip.stream())
        hip.reg_loss(rg, pose.g, ind_g, mask.g, cw.g, nreg, 0.2, lr2, g2v, prev)(hip.stream())
        torch.cuda.synchronize()
        assert torch.equal(lr2.cpu(), lr.g.cpu()), it
        assert torch.equal(g2.cpu(), grp.g.cpu()), "stateful reg-loss gradient tensor, call %d" % it
        assert torch.equal(prev.cpu(), ind_i)
    out = Pair(hip, torch.zeros(n, ncls, 2 + nreg))
    sc, og = (0.05, 0.15, 0.36), (0.77, -5.0, -1.1)
    run(hip, EMU.decode(hc, rc, ncls, nreg, sc, og, None, out.c),
        hip.decode(hg, rg, ncls, nreg, sc, og, hip.decode_scratch(n, ncls), out.g))
    got, ref = out.sync_back(), out.c
    assert torch.equal(got[..., 0], ref[..., 0]), "argmax indices"
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)


def test_adam_and_sqnorm(hip):
    nel = 100003
    p, g = Pair(hip, rnd((nel,), 60, torch.float32)), Pair(hip, rnd((nel,), 61, torch.float32, 3.0))
    m, v = Pair(hip, rnd((nel,), 62, torch.float32, 0.1)), Pair(hip, rnd((nel,), 63, torch.float32).abs() * 0.01)
    hyper = Pair(hip, torch.tensor([1e-3, 0.9, 0.99, 1e-8, 0.01, 35.0, 1 - 0.9 ** 3, 1 - 0.99 ** 3, 0.5, 0.0]))
    pc, pg = torch.zeros(1), hip.alloc((256,), "f32")
    nc, ng = torch.zeros(1), hip.alloc((1,), "f32")
    run(hip, EMU.sqnorm(g.c, nel, hyper.c, pc), hip.sqnorm(g.g, nel, hyper.g, pg))
    assert abs(float(pg.sum()) - float(pc.sum())) < 1e-4 * float(pc.sum())
    run(hip, EMU.adam_step(p.c, g.c, m.c, v.c, nel, hyper.c, pc, 0, nc), hip.adam_step(p.g, g.g, m.g, v.g, nel, hyper.g, pg, 0, ng))
    assert float(nc) > 35.0  # the clip engaged
    assert abs(float(ng.cpu()) - float(nc)) < 1e-4 * float(nc)
    check(p, 1e-5, "adam p")
    check(m, 1e-5, "adam m")
    check(v, 1e-5, "adam v")
    run(hip, EMU.adam_step(p.c, None, None, None, nel, hyper.c, pc, 1, None), hip.adam_step(p.g, None, None, None, nel, hyper.g, pg, 1, None))
    check(p, 1e-6, "decay only")


# ---------------------------------------------------------------------------------------------- fused backward chain
FUSED_CASES = [
    # n, dims, co_real, with GroupNorm, extras (plain / GN terms), mask
    (2, (4, 8, 32), 32, True, (), True),
    (2, (4, 8, 32), 32, True, ("plain",), True),
    (3, (2, 16, 64), 32, True, ("plain", "gn"), False),
    (2, (8, 32, 80), 32, True, ("gn", "plain", "plain"), True),     # ragged W (half-padding brick column), 3 extras
    (1, (2, 4, 16), 15, False, (), True),                            # head tower: no GroupNorm, 15 real output channels
    (2, (4, 8, 16), 3, False, ("plain",), True),
    (8, (4, 16, 64), 32, True, ("plain",), True),                    # one sample per XCD
]


@pytest.mark.parametrize("case", FUSED_CASES)
def test_fused_backward_chain(hip, case):
    """rtp_wgrad_q -> rtp_gn_bwd_coeffs_cls -> rtp_conv_dgrad_fused against the emulation, and against the UNFUSED chain on
    the GPU (data gradient + chan_stats(dxhat, x) + gn_bwd_coeffs + grad_combine): P and Q obtained from the class sums and
    the slab contraction equal the sums over dxhat."""
    n, dims, co_real, has_gn, extras, mask = case
    d, h, w = dims
    ci, co = 32, pad_to(co_real, 16)
    co32 = pad_to(co, 32)
    geom = Geom(n, d, h, w, d, h, w, ci, co, 3, 1, 1)
    xp, xc, xg = views(hip, rnd((n, d, h, w, ci), 1, relu=True), n, d, h, w)
    gyt = rnd((n, d, h, w, co32), 2)
    gyt[..., co_real:] = 0
    gp_, gyc, gyg = views(hip, gyt, n, d, h, w)
    wt = rnd((co_real, ci, 27), 3, torch.float32, scale=0.05)
    w32 = Pair(hip, wt)
    wd = Pair(hip, torch.zeros(27, ci, co32, dtype=torch.bfloat16))
    run(hip, EMU.pack_dgrad_w(w32.c, geom, ci, co_real, wd.c), hip.pack_dgrad_w(w32.g, geom, ci, co_real, wd.g))
    S = hip.wgrad_nsplit(geom)
    assert S > 0
    Sc = 2
    slab = Pair(hip, torch.zeros(n, S, 27, co32, ci))
    slab_e = torch.zeros(n, Sc, 27, co32, ci)
    qp = Pair(hip, torch.zeros(n, S, ci))
    qp_e = torch.zeros(n, Sc, ci)
    coeff = coeff_e = None
    groups = 8
    if has_gn:
        # slabs + Q partials
        EMU.wgrad_q(gyc, xc, geom, Sc, slab_e, wd.c, qp_e)(None)
        hip.wgrad_q(gyg, xg, geom, S, slab.g, wd.g, qp.g)(hip.stream())
        torch.cuda.synchronize()
        assert rel_err(slab.g.sum(1).cpu(), slab_e.sum(1)) < BF
        assert rel_err(qp.g.sum(1).cpu(), qp_e.sum(1)) < 2e-3, "Q partials"
        # plain rtp_wgrad writes the same slabs
        slab2 = hip.alloc((n, S, 27, co32, ci), "f32")
        hip.wgrad(gyg, xg, geom, S, slab2)(hip.stream())
        torch.cuda.synchronize()
        assert same_partials(slab2, slab.g)
        # coefficients
        cs_split = 3
        clsp = Pair(hip, torch.zeros(n, cs_split, 64, co32))
        run(hip, EMU.class_sums(gyc, cs_split, clsp.c, None), hip.class_sums(gyg, cs_split, clsp.g, None))
        mr = Pair(hip, torch.stack([rnd((n, groups), 5, torch.float32, 0.3), rnd((n, groups), 6, torch.float32, 0.2).abs() + 0.5], -1))
        gam = Pair(hip, rnd((ci,), 7, torch.float32) + 1.5)
        cf = Pair(hip, torch.zeros(n * ci * 5))
        cso = Pair(hip, torch.zeros(n, 64, co32))
        cf_e = torch.zeros(n * ci * 5)
        cso_e = torch.zeros(n, 64, co32)
        EMU.gn_bwd_coeffs_cls(qp_e, Sc, clsp.c, cs_split, cso_e, wd.c, mr.c, gam.c, geom, ci, co_real, groups, cf_e)(None)
        hip.gn_bwd_coeffs_cls(qp.g, S, clsp.g, cs_split, cso.g, wd.g, mr.g, gam.g, geom, ci, co_real, groups, cf.g)(hip.stream())
        torch.cuda.synchronize()
        assert rel_err(cso.g.cpu(), cso_e) < F32
        got, want = cf.g.cpu()[:n * ci * 3].view(n, ci, 3), cf_e[:n * ci * 3].view(n, ci, 3)
        for k, nm in enumerate("ABC"):
            assert rel_err(got[..., k], want[..., k]) < 3e-3, "coefficient " + nm
        assert rel_err(cf.g.cpu()[n * ci * 3:], cf_e[n * ci * 3:]) < 3e-3, "dgamma / dbeta partials"
        coeff, coeff_e = cf.g, cf_e
        # ... and they equal the coefficients of the unfused chain (sums over the stored dxhat), to bf16 level
        dxh_p, dxh_c, dxh_g = views(hip, torch.zeros(n, d, h, w, ci, dtype=torch.bfloat16), n, d, h, w)
        hip.conv(gyg, wd.g, False, None, None, dxh_g, geom, False, True, False)(hip.stream())
        pq = hip.alloc((n, 4, ci, 2), "f32")
        hip.chan_stats(dxh_g, xg, 4, pq)(hip.stream())
        cf_old = hip.alloc((n * ci * 5,), "f32")
        hip.gn_bwd_coeffs(pq, 4, mr.g, gam.g, n, ci, groups, d * h * w, cf_old, None, None, 0)(hip.stream())
        torch.cuda.synchronize()
        old = cf_old.cpu()[:n * ci * 3].view(n, ci, 3)
        for k, nm in enumerate("ABC"):
            assert rel_err(got[..., k], old[..., k]) < 2e-2, "coefficient %s vs the dxhat-pass chain" % nm
    # extras
    terms_c, terms_g = [], []
    for i, kind in enumerate(extras):
        ep, ec, eg = views(hip, rnd((n, d, h, w, ci), 20 + i), n, d, h, w)
        if kind == "gn":
            k = Pair(hip, torch.cat([rnd((n * ci * 3,), 30 + i, torch.float32, 0.5), torch.zeros(n * ci * 2)]))
            terms_c.append((ec, k.c)); terms_g.append((eg, k.g))
        else:
            terms_c.append((ec, None)); terms_g.append((eg, None))
    dxp, dxc, dxg = views(hip, torch.zeros(n, d, h, w, ci, dtype=torch.bfloat16), n, d, h, w)
    ts = hip.conv_stats_nsplit(gyg, geom, True)
    assert ts > 0
    tot = hip.alloc((n, ts, 32), "f32")
    EMU.conv_dgrad_fused(gyc, wd.c, xc, coeff_e, terms_c, mask, dxc, geom)(None)
    hip.conv_dgrad_fused(gyg, wd.g, xg, coeff, terms_g, mask, dxg, geom, tot)(hip.stream())
    torch.cuda.synchronize()
    check(dxp, BF, "fused data gradient %r" % (case,))
    if mask:
        assert float(dxp.sync_back().float()[xp.c.float() <= 0].abs().max()) == 0.0
    # per-channel totals of the STORED gradient, and the boundary-only class sums built on them == a full class-sum scan
    stored = dxg.buf.float().reshape(n, -1, ci)
    assert rel_err(tot.sum(1).cpu(), stored.sum(1).cpu()) < F32, "epilogue totals"
    full = hip.alloc((n, 64, ci), "f32")
    hip.class_sums(dxg, 5, hip.alloc((n, 5, 64, ci), "f32"), full)(hip.stream())
    bnd = hip.alloc((n, 64, ci), "f32")
    hip.class_sums_boundary(dxg, 4, hip.alloc((n, 4, 64, ci), "f32"), tot, ts, bnd)(hip.stream())
    torch.cuda.synchronize()
    assert torch.equal(bnd[:, 1:].cpu(), full[:, 1:].cpu()) or rel_err(bnd[:, 1:].cpu(), full[:, 1:].cpu()) < 1e-6
    assert (bnd[:, 0] - full[:, 0]).abs().max() <= 2e-4 * max(1.0, float(stored.abs().sum(1).max())), "interior class = total - boundary"
    if has_gn:
        # coefficients computed in the data gradient's own prologue (P from rtp_gn_bwd_p, Q from the slab contractions)
        pb = hip.alloc((n, ci), "f32")
        hip.gn_bwd_p(cso.g, 1, None, wd.g, geom, ci, co_real, pb)(hip.stream())
        cf2 = hip.alloc((n * ci * 5,), "f32")
        dx2p, dx2c, dx2g = views(hip, torch.zeros(n, d, h, w, ci, dtype=torch.bfloat16), n, d, h, w)
        gn = dict(qpart=qp.g, q_nsplit=S, p=pb, mr=mr.g, gamma=gam.g, groups=groups, coeff_out=cf2)
        hip.conv_dgrad_fused(gyg, wd.g, xg, None, terms_g, mask, dx2g, geom, None, gn)(hip.stream())
        torch.cuda.synchronize()
        assert rel_err(cf2.cpu(), cf.g.cpu()) < 1e-5, "in-kernel coefficients == rtp_gn_bwd_coeffs_cls"
        assert rel_err(dx2g.buf.float().cpu(), dxg.buf.float().cpu()) < 1e-3
        pe = torch.zeros(n, ci)
        EMU.gn_bwd_p(cso_e, 1, None, wd.c, geom, ci, co_real, pe)(None)
        assert rel_err(pb.cpu(), pe) < 2e-3
    if has_gn:
        # subset sums of gy from the weight-gradient kernel's loader waves -> P, class sums and coefficients in the data
        # gradient's prologue: no kernel between the two
        tg = hip.alloc((n, S, 27, 32), "f32")
        tg_e = torch.zeros(n, Sc, 27, 32)
        slab3 = hip.alloc((n, S, 27, co32, ci), "f32")
        qp3 = hip.alloc((n, S, ci), "f32")
        prev = None
        for rep in range(3):
            tg.fill_(float("nan"))   # every workgroup must write its whole table
            hip.wgrad_q(gyg, xg, geom, S, slab3, wd.g, qp3, tg)(hip.stream())
            torch.cuda.synchronize()
            assert same_partials(slab3, slab.g) and same_partials(qp3, qp.g), "the subset sums do not disturb the slabs"
            assert prev is None or same_partials(prev, tg), "subset sums are reproducible bit for bit"
            prev = tg.clone()
        EMU.wgrad_q(gyc, xc, geom, Sc, torch.zeros_like(slab_e), wd.c, torch.zeros_like(qp_e), tg_e)(None)
        assert rel_err(tg.sum(1).cpu(), tg_e.sum(1)) < 1e-5, "inclusive subset sums of gy"
        cf3, cs3 = hip.alloc((n * ci * 5,), "f32"), hip.alloc((n, 64, co32), "f32")
        dx3p, dx3c, dx3g = views(hip, torch.zeros(n, d, h, w, ci, dtype=torch.bfloat16), n, d, h, w)
        gn3 = dict(qpart=qp.g, q_nsplit=S, p=None, tg=tg, csum_out=cs3, mr=mr.g, gamma=gam.g, groups=groups, coeff_out=cf3)
        hip.conv_dgrad_fused(gyg, wd.g, xg, None, terms_g, mask, dx3g, geom, None, gn3)(hip.stream())
        torch.cuda.synchronize()
        assert rel_err(cs3.cpu(), cso.g.cpu()) < 1e-4, "class sums rebuilt from the subset sums"
        assert rel_err(cf3.cpu(), cf.g.cpu()) < 1e-4, "coefficients from the subset sums"
        assert rel_err(dx3g.buf.float().cpu(), dxg.buf.float().cpu()) < 1e-3
    # one-launch class sums + P (last-block finalisation), with and without the totals; repeated launches reuse the counters
    for use_tot in (True, False):
        cs1, p1 = hip.alloc((n, 64, ci), "f32"), hip.alloc((n, ci), "f32")
        f1 = hip.class_sums_p(dxg, 4, hip.alloc((n, 4, 64, ci), "f32"), tot if use_tot else None, ts if use_tot else 0, cs1, wd.g,
                              geom, ci, co_real, p1) if co32 == ci else None
        if f1 is None:
            break
        for rep in range(3):
            cs1.zero_(); p1.zero_()
            f1(hip.stream())
            torch.cuda.synchronize()
            assert rel_err(cs1[:, 1:].cpu(), full[:, 1:].cpu()) < 1e-6
            assert (cs1[:, 0] - full[:, 0]).abs().max() <= 2e-4 * max(1.0, float(stored.abs().sum(1).max()))
            pr = hip.alloc((n, ci), "f32")
            hip.gn_bwd_p(cs1, 1, None, wd.g, geom, ci, co_real, pr)(hip.stream())
            torch.cuda.synchronize()
            assert rel_err(p1.cpu(), pr.cpu()) < 1e-5, "P of the one-launch path (totals=%s, rep %d)" % (use_tot, rep)
    # the same without the second pass through the emulation
    bnd_e = torch.zeros(n, 64, ci)
    tot_e = torch.zeros(n, 2, 32)
    EMU.conv_dgrad_fused(gyc, wd.c, xc, coeff_e, terms_c, mask, dxc, geom, tot_e)(None)
    EMU.class_sums_boundary(dxc, 4, None, tot_e, 2, bnd_e)(None)
    assert rel_err(bnd.cpu(), bnd_e) < 2e-2


@pytest.mark.parametrize("case", [(2, (4, 8, 32), ("plain",), True), (3, (8, 16, 64), ("gn", "plain"), True),
                                  (1, (2, 4, 32), (), False), (8, (4, 16, 64), ("plain",), True)])
def test_fused_stride2_data_gradient(hip, case):
    """rtp_qpart_from_slabs + rtp_conv_dgrad_fused on the stride-2 parity-class kernel (csrc/dgrad_s2_tiled.hip): Q from the
    generic weight-gradient slabs, P from gy's boundary-class sums, the finished gradient of the full-resolution input."""
    n, dims, extras, mask = case
    d, h, w = dims
    do, ho, wo = d // 2, h // 2, w // 2
    ci = co = 32
    geom = Geom(n, d, h, w, do, ho, wo, ci, co, 3, 2, 1)
    xp, xc, xg = views(hip, rnd((n, d, h, w, ci), 1, relu=True), n, d, h, w)
    gp_, gyc, gyg = views(hip, rnd((n, do, ho, wo, co), 2), n, do, ho, wo)
    w32 = Pair(hip, rnd((co, ci, 27), 3, torch.float32, scale=0.05))
    wd = Pair(hip, torch.zeros(27, ci, co, dtype=torch.bfloat16))
    run(hip, EMU.pack_dgrad_w(w32.c, geom, ci, co, wd.c), hip.pack_dgrad_w(w32.g, geom, ci, co, wd.g))
    assert hip.conv_dgrad_fused_ok(gyg, geom)
    S = 3
    slab = Pair(hip, torch.zeros(n, S, 27, co, ci))
    run(hip, EMU.wgrad(gyc, xc, geom, S, slab.c), hip.wgrad(gyg, xg, geom, S, slab.g))
    qp = Pair(hip, torch.zeros(n, S, ci))
    run(hip, EMU.qpart_from_slabs(slab.c, n, S, 27, co, ci, wd.c, qp.c), hip.qpart_from_slabs(slab.g, n, S, 27, co, ci, wd.g, qp.g))
    assert rel_err(qp.g.sum(1).cpu(), qp.c.sum(1)) < 3e-3
    cs = Pair(hip, torch.zeros(n, 64, co))
    run(hip, EMU.class_sums(gyc, 2, torch.zeros(n, 2, 64, co), cs.c), hip.class_sums(gyg, 2, hip.alloc((n, 2, 64, co), "f32"), cs.g))
    groups = 8
    mr = Pair(hip, torch.stack([rnd((n, groups), 5, torch.float32, 0.3), rnd((n, groups), 6, torch.float32, 0.2).abs() + 0.5], -1))
    gam = Pair(hip, rnd((ci,), 7, torch.float32) + 1.5)
    terms_c, terms_g = [], []
    for i, kind in enumerate(extras):
        ep, ec, eg = views(hip, rnd((n, d, h, w, ci), 20 + i), n, d, h, w)
        if kind == "gn":
            k = Pair(hip, torch.cat([rnd((n * ci * 3,), 30 + i, torch.float32, 0.5), torch.zeros(n * ci * 2)]))
            terms_c.append((ec, k.c)); terms_g.append((eg, k.g))
        else:
            terms_c.append((ec, None)); terms_g.append((eg, None))
    cf = Pair(hip, torch.zeros(n * ci * 5))
    dxp, dxc, dxg = views(hip, torch.zeros(n, d, h, w, ci, dtype=torch.bfloat16), n, d, h, w)
    gn_c = dict(qpart=qp.c, q_nsplit=S, p=None, tg=None, csum=cs.c, mr=mr.c, gamma=gam.c, groups=groups, coeff_out=cf.c)
    gn_g = dict(qpart=qp.g, q_nsplit=S, p=None, tg=None, csum=cs.g, mr=mr.g, gamma=gam.g, groups=groups, coeff_out=cf.g)
    EMU.conv_dgrad_fused(gyc, wd.c, xc, None, terms_c, mask, dxc, geom, None, gn_c)(None)
    hip.conv_dgrad_fused(gyg, wd.g, xg, None, terms_g, mask, dxg, geom, None, gn_g)(hip.stream())
    torch.cuda.synchronize()
    assert rel_err(cf.g.cpu(), cf.c) < 5e-3, "coefficients"
    check(dxp, BF, "fused stride-2 data gradient %r" % (case,))
    # the coefficients equal those of the unfused chain (P, Q as sums over the stored dxhat)
    dxh = hip.alloc((n, d, h, w, ci), "bf16")
    dxv = View(dxh, n, d, h, w, ci, 0, ci)
    Sd = hip.conv_stats_nsplit(gyg, geom, True)
    pq = hip.alloc((n, Sd, ci, 2), "f32")
    hip.conv(gyg, wd.g, False, None, None, dxv, geom, False, True, False, (xg, pq))(hip.stream())
    cf_old = hip.alloc((n * ci * 5,), "f32")
    hip.gn_bwd_coeffs(pq, Sd, mr.g, gam.g, n, ci, groups, d * h * w, cf_old, None, None, 0)(hip.stream())
    torch.cuda.synchronize()
    got, old = cf.g.cpu()[:n * ci * 3].view(n, ci, 3), cf_old.cpu()[:n * ci * 3].view(n, ci, 3)
    for k, nm in enumerate("ABC"):
        assert rel_err(got[..., k], old[..., k]) < 2e-2, "coefficient %s vs the dxhat-pass chain" % nm
    # no-GroupNorm variant with totals
    ts = hip.conv_stats_nsplit(gyg, geom, True)
    tot = hip.alloc((n, ts, 32), "f32")
    hip.conv_dgrad_fused(gyg, wd.g, xg, None, terms_g, mask, dxg, geom, tot)(hip.stream())
    EMU.conv_dgrad_fused(gyc, wd.c, xc, None, terms_c, mask, dxc, geom)(None)
    torch.cuda.synchronize()
    check(dxp, BF, "fused stride-2 data gradient without GroupNorm %r" % (case,))
    assert rel_err(tot.sum(1).cpu(), dxg.buf.float().reshape(n, -1, ci).sum(1).cpu()) < F32


@pytest.mark.parametrize("case", [
    # n, dims, ci, co, per-sample weights + class bias (GroupNorm fold output), residual, relu
    (2, (4, 8, 32), 64, 64, True, True, True), (1, (2, 8, 48), 128, 64, True, False, False),
    (2, (2, 4, 16), 64, 128, False, True, True), (3, (4, 8, 80), 128, 128, True, False, True),
    (2, (4, 8, 32), 32, 64, True, False, True), (1, (2, 4, 32), 64, 32, False, True, False),
    # 64 -> 64: ONE launch of csrc/conv64_tiled.hip (weights in registers, no workspace traffic)
    (1, (2, 8, 48), 64, 64, False, False, False), (3, (4, 8, 80), 64, 64, True, True, True), (8, (4, 16, 64), 64, 64, True, False, True),
    (2, (2, 4, 16), 64, 64, True, True, False),
    # ... ragged W (the level-2 / level-3 tensors of the native shape: 40 and 20 columns; the last brick column partly outside)
    # (launches of fewer than 64 bricks stay on the slice / generic kernels: the cases are sized to reach the 64-wide kernel)
    (8, (4, 16, 40), 64, 64, True, True, True), (16, (2, 8, 20), 64, 64, True, False, True), (64, (2, 4, 7), 64, 64, False, True, False),
    (8, (4, 8, 48), 64, 64, True, True, True),
])
def test_wide_convs_as_channel_slices_of_the_tiled_kernel(hip, case):
    """rtp_conv_igemm_ws / rtp_wgrad on Cin = 32 K, Cout = 32 J (the feat64 backbone's 64- and 128-channel layers,
    hrnet3D_config.py:149-177): K x J launches of the LDS-tiled 32 -> 32 kernel over windows of ONE weight image, the partial
    sums in an fp32 workspace -- forward (+ class bias, residual, ReLU, statistics), data gradient (+ P / Q statistics) and
    weight gradient, each against the emulation of the whole wide conv."""
    n, dims, ci, co, per_sample, has_res, relu = case
    d, h, w = dims
    geom = Geom(n, d, h, w, d, h, w, ci, co, 3, 1, 1)
    xp, xc, xg = views(hip, rnd((n, d, h, w, ci), 401, relu=True), n, d, h, w)
    assert hip.conv_sliced_ok(xg, geom, False)
    nw = n if per_sample else 1
    wf = Pair(hip, rnd((nw, 27, co, ci), 402, scale=0.05))
    bt = Pair(hip, rnd((nw, 64, co), 403, torch.float32))
    yp, yc, yg = views(hip, torch.zeros(n, d, h, w, co, dtype=torch.bfloat16), n, d, h, w)
    rc = rg = None
    if has_res:
        rp, rc, rg = views(hip, rnd((n, d, h, w, co), 404), n, d, h, w)
    S = hip.conv_stats_nsplit(xg, geom, False, ws=True)
    assert S > 0
    st = hip.alloc((n, S, co, 2), "f32")
    ws = hip.alloc((n * d * h * w * 32,), "f32")
    run(hip, EMU.conv(xc, wf.c, per_sample, bt.c, rc, yc, geom, relu, False, False),
        hip.conv(xg, wf.g, per_sample, bt.g, rg, yg, geom, relu, False, False, (None, st), ws=ws))
    check(yp, BF, "sliced conv forward %r" % (case,))
    ref = hip.alloc((n, 3, co, 2), "f32")
    hip.chan_stats(yg, None, 3, ref)(hip.stream())
    torch.cuda.synchronize()
    assert rel_err(st.sum(1).cpu(), ref.sum(1).cpu()) < F32, "sliced conv statistics %r" % (case,)
    ragged = w % 16 != 0    # only the native 64 -> 64 kernel takes these (bf16 output; the slice kernels want W % 16 == 0)
    if not ragged:
        # the same launch without statistics / with an fp32 output
        y32p, y32c, y32g = views(hip, torch.zeros(n, d, h, w, co), n, d, h, w)
        run(hip, EMU.conv(xc, wf.c, per_sample, bt.c, rc, y32c, geom, relu, False, True),
            hip.conv(xg, wf.g, per_sample, bt.g, rg, y32g, geom, relu, False, True, ws=ws))
        check(y32p, F32 * 5, "sliced conv forward, fp32 output %r" % (case,))
    # data gradient: contraction over the conv's output channels, P / Q statistics against the conv's input
    gyp, gyc, gyg = views(hip, rnd((n, d, h, w, co), 405), n, d, h, w)
    assert hip.conv_sliced_ok(gyg, geom, True)
    wd = Pair(hip, rnd((27, ci, co), 406, scale=0.05))
    dxp, dxc, dxg = views(hip, torch.zeros(n, d, h, w, ci, dtype=torch.bfloat16), n, d, h, w)
    Sb = hip.conv_stats_nsplit(gyg, geom, True, ws=True)
    assert Sb > 0
    pq = hip.alloc((n, Sb, ci, 2), "f32")
    run(hip, EMU.conv(gyc, wd.c, False, None, None, dxc, geom, False, True, False),
        hip.conv(gyg, wd.g, False, None, None, dxg, geom, False, True, False, (xg, pq), ws=ws))
    check(dxp, BF, "sliced data gradient %r" % (case,))
    refq = hip.alloc((n, 3, ci, 2), "f32")
    hip.chan_stats(dxg, xg, 3, refq)(hip.stream())
    torch.cuda.synchronize()
    assert rel_err(pq.sum(1).cpu(), refq.sum(1).cpu()) < F32 * 5, "sliced data gradient P / Q %r" % (case,)
    if ragged:
        return
    # weight gradient: every (output slice, input slice) launch fills its window of the wide slabs
    Sw = hip.wgrad_nsplit(geom)
    assert Sw > 0
    gp = Pair(hip, torch.full((n, Sw, 27, co, ci), 7.0))
    run(hip, EMU.wgrad(gyc, xc, geom, Sw, gp.c), hip.wgrad(gyg, xg, geom, Sw, gp.g))
    assert rel_err(gp.sync_back().sum(1), gp.c.sum(1)) < F32 * 5, "sliced weight gradient %r" % (case,)


@pytest.mark.parametrize("case", [
    # n, dims (input), ci, co, per-sample weights + class bias, residual, relu
    (2, (4, 8, 32), 32, 32, True, False, True), (1, (8, 16, 64), 32, 32, False, True, False), (3, (4, 4, 80), 32, 32, True, False, True),
    (2, (4, 8, 40), 32, 32, True, False, True),       # Wo = 20: the second brick column is three quarters padding
    (2, (8, 32, 80), 32, 64, True, False, True),      # the level-1 -> level-2 down-sampling convs: two output slices
    (1, (4, 8, 32), 64, 64, True, False, True), (2, (4, 8, 32), 64, 128, False, True, False), (1, (4, 4, 32), 128, 128, True, False, True),
    (8, (16, 32, 128), 64, 64, True, False, True),   # input-channel slices through the fp32 workspace (>= 65536 output voxels)
])
def test_stride2_forward_on_the_tiled_kernel(hip, case):
    """csrc/conv_s2_tiled.hip (32 -> 32 per launch; wider convs as channel slices through rtp_conv_igemm_ws) against the emulation:
    output, statistics epilogue, fp32 output."""
    n, dims, ci, co, per_sample, has_res, relu = case
    d, h, w = dims
    do, ho, wo = d // 2, h // 2, w // 2
    geom = Geom(n, d, h, w, do, ho, wo, ci, co, 3, 2, 1)
    xp, xc, xg = views(hip, rnd((n, d, h, w, ci), 501, relu=True), n, d, h, w)
    wide = ci * co > 1024
    # (input-channel slices re-stream x per launch: only volumes that fill the chip take that route, the rest stay generic)
    sliced = hip.conv_sliced_ok(xg, geom, False)
    assert sliced == (wide and (ci == 32 or n * do * ho * wo >= 65536))
    nw = n if per_sample else 1
    wf = Pair(hip, rnd((nw, 27, co, ci), 502, scale=0.05))
    bt = Pair(hip, rnd((nw, 64, co), 503, torch.float32))
    yp, yc, yg = views(hip, torch.zeros(n, do, ho, wo, co, dtype=torch.bfloat16), n, do, ho, wo)
    rc = rg = None
    if has_res:
        rp, rc, rg = views(hip, rnd((n, do, ho, wo, co), 504), n, do, ho, wo)
    kw = dict(ws=hip.alloc((n * do * ho * wo * 32,), "f32")) if sliced else {}
    S = hip.conv_stats_nsplit(xg, geom, False, ws=sliced)
    if sliced or not wide:
        assert S == min(256 // n, do * (ho // 2) * ((wo + 15) // 16)), "expected the tiled stride-2 kernel's partial count"
    st = hip.alloc((n, S, co, 2), "f32")
    run(hip, EMU.conv(xc, wf.c, per_sample, bt.c, rc, yc, geom, relu, False, False),
        hip.conv(xg, wf.g, per_sample, bt.g, rg, yg, geom, relu, False, False, (None, st), **kw))
    check(yp, BF, "stride-2 forward %r" % (case,))
    ref = hip.alloc((n, 3, co, 2), "f32")
    hip.chan_stats(yg, None, 3, ref)(hip.stream())
    torch.cuda.synchronize()
    assert rel_err(st.sum(1).cpu(), ref.sum(1).cpu()) < F32, "stride-2 forward statistics %r" % (case,)
    y32p, y32c, y32g = views(hip, torch.zeros(n, do, ho, wo, co), n, do, ho, wo)
    run(hip, EMU.conv(xc, wf.c, per_sample, bt.c, rc, y32c, geom, relu, False, True),
        hip.conv(xg, wf.g, per_sample, bt.g, rg, y32g, geom, relu, False, True, **kw))
    check(y32p, F32 * 5, "stride-2 forward, fp32 output %r" % (case,))


def test_bias_gradient_from_weight_gradient_subset_sums(hip):
    """rtp_wgrad_tg + rtp_tail_desc_wgrad_fold_tg (bias gradient of a conv without GroupNorm read off the weight-gradient kernel's
    sums of gy) against the class-sum route (rtp_class_sums + rtp_tail_desc_wgrad_fold) and against plain sums of gy."""
    n, d, h, w, co_real = 3, 4, 8, 48, 15
    geom = Geom(n, d, h, w, d, h, w, 32, 16, 3, 1, 1)
    gt = rnd((n, d, h, w, 32), 601)
    gt[..., co_real:] = 0
    _, gc, gg = views(hip, gt, n, d, h, w)
    _, xc, xg = views(hip, rnd((n, d, h, w, 32), 602, relu=True), n, d, h, w)
    S = hip.wgrad_nsplit(geom)
    assert S > 0
    gp1, gp2 = hip.alloc((n, S, 27, 32, 32), "f32"), hip.alloc((n, S, 27, 32, 32), "f32")
    tg = hip.alloc((n, S, 27, 32), "f32")
    dw1, dw2 = hip.alloc((co_real, 32, 3, 3, 3), "f32"), hip.alloc((co_real, 32, 3, 3, 3), "f32")
    db1, db2 = hip.alloc((co_real,), "f32"), hip.alloc((co_real,), "f32")
    s = hip.stream()
    hip.wgrad_tg(gg, xg, geom, S, gp1, tg)(s)
    hip.tail([("wgrad_fold", gp1, S, None, None, None, None, 1, geom, 32, co_real, dw1, db1, 0, tg)])(s)
    cs = hip.alloc((n, 64, 32), "f32")
    hip.wgrad(gg, xg, geom, S, gp2)(s)
    hip.class_sums(gg, 3, hip.alloc((n, 3, 64, 32), "f32"), cs)(s)
    hip.tail([("wgrad_fold", gp2, S, cs, None, None, None, 1, geom, 32, co_real, dw2, db2, 0, None)])(s)
    torch.cuda.synchronize()
    assert same_partials(gp1, gp2) and torch.equal(dw1, dw2)
    want = gt.float().reshape(-1, 32)[:, :co_real].sum(0)
    assert rel_err(db1.cpu(), want) < F32 * 5 and rel_err(db2.cpu(), want) < F32 * 5
    # ... and the folded weight gradient is the slabs' sum, which is the emulation's
    tot = gp1.sum((0, 1))[:, :co_real, :].permute(1, 2, 0).reshape(co_real, 32, 3, 3, 3)
    assert rel_err(dw1.cpu(), tot.cpu()) < 1e-5
    gpe = torch.zeros(n, 2, 27, 32, 32)
    EMU.wgrad(gc, xc, geom, 2, gpe)(None)
    assert rel_err(dw1.cpu(), gpe.sum((0, 1))[:, :co_real, :].permute(1, 2, 0).reshape(co_real, 32, 3, 3, 3)) < F32 * 5


# ------------------------------------------------------------------------------------------------ shared launches (rtp_multi_*)
@pytest.mark.parametrize("n", [8, 4, 16])
def test_two_convs_and_two_weight_gradients_in_one_launch(hip, n):
    """HipBackend.multi: a 'full-resolution' and a 'level-1' problem (n samples each, the same kernel variant) as ONE launch --
    conv + GroupNorm-fold prologue + residual + ReLU + statistics epilogue, and the weight-gradient kernel.  Every problem must
    produce exactly what it produces alone (outputs bit for bit; the per-workgroup partials as sums: a problem runs on fewer
    workgroups per sample inside the shared launch), the untouched partial slots must stay zero, and launches that cannot share a
    kernel must be refused (None) without side effects."""
    ci, co = 32, 32
    probs = []
    # 256 and 32 bricks per sample (n = 8: 28 + 4 workgroups per XCD); n = 4: 512 bricks, so that the launch still counts as a large one
    for k, dims in enumerate((((8, 64, 128) if n >= 8 else (16, 64, 128)), (4, 32, 64))):
        d, h, w = dims
        geom = Geom(n, d, h, w, d, h, w, ci, co, 3, 1, 1)
        xp, xc, xg = views(hip, rnd((n, d, h, w, ci), 100 + k, relu=True), n, d, h, w)
        rp, rc, rg = views(hip, rnd((n, d, h, w, co), 110 + k), n, d, h, w)
        W = Pair(hip, rnd((co, ci, 3, 3, 3), 120 + k, torch.float32, scale=0.05))
        gamma, beta = Pair(hip, rnd((ci,), 130 + k, torch.float32) * 0.2 + 1.0), Pair(hip, rnd((ci,), 140 + k, torch.float32) * 0.2)
        xv = xp.c.float().reshape(n, -1, ci)
        st = Pair(hip, torch.stack([xv.sum(1), (xv * xv).sum(1)], -1)[:, None].contiguous())       # one statistics partial per sample
        wt = Pair(hip, torch.zeros(27, co, ci))
        hip.tail([("pack_wt", W.g, co, co, ci, 27, wt.g)])(hip.stream())
        S = hip.conv_stats_nsplit(xg, geom, False)
        assert S > 0
        ys, sos, mrs = [], [], []
        for rep in range(2):   # [0]: alone, [1]: inside the shared launch
            ys.append(views(hip, torch.zeros(n, d, h, w, co, dtype=torch.bfloat16), n, d, h, w))
            sos.append(hip.alloc((n, S, co, 2), "f32"))
            mrs.append(hip.alloc((n, 8, 2), "f32"))
        def mk(r, xg=xg, wt=wt, gamma=gamma, beta=beta, st=st, mrs=mrs, rg=rg, ys=ys, geom=geom, sos=sos):   # (bound now, not at call time)
            return hip.conv_gn_fused(xg, wt.g, None, gamma.g, beta.g, st.g, 1, 8, 1e-5, co, mrs[r], rg, ys[r][2], geom, True, sos[r])
        # weight gradient of the same geometry (gy = the residual tensor, any bf16 tensor will do)
        Sw = hip.wgrad_nsplit(geom)
        slabs = [hip.alloc((n, Sw, 27, co, ci), "f32") for _ in range(2)]
        def mkw(r, rg=rg, xg=xg, geom=geom, Sw=Sw, slabs=slabs):
            return hip.wgrad(rg, xg, geom, Sw, slabs[r])
        probs.append(dict(mk=mk, mkw=mkw, ys=ys, sos=sos, mrs=mrs, slabs=slabs, S=S, Sw=Sw))
    s = hip.stream()
    for pr in probs:
        pr["mk"](0)(s)
        pr["mkw"](0)(s)
    both = hip.multi([pr["mk"](1) for pr in probs])
    bothw = hip.multi([pr["mkw"](1) for pr in probs])
    assert both is not None and bothw is not None, "two %d-sample launches of one variant must be mergeable" % n
    for _ in range(2):      # replays
        both(s)
        bothw(s)
    torch.cuda.synchronize()
    for k, pr in enumerate(probs):
        assert torch.equal(pr["ys"][0][0].g, pr["ys"][1][0].g), "conv output of problem %d" % k
        assert torch.equal(pr["mrs"][0], pr["mrs"][1])
        assert rel_err(pr["sos"][1].sum(1).cpu(), pr["sos"][0].sum(1).cpu()) < 1e-5, "statistics of problem %d" % k
        assert rel_err(pr["slabs"][1].sum(1).cpu(), pr["slabs"][0].sum(1).cpu()) < 1e-5, "weight gradient of problem %d" % k
    # the level-1 problem ran on fewer workgroups than its buffers have slots: the upper slots were never touched
    small = probs[1]
    used = int((small["sos"][1].abs().sum((0, 2, 3)) > 0).sum())
    assert 0 < used < small["S"] and float(small["sos"][1][:, used:].abs().max()) == 0.0, (used, small["S"], small["sos"][1].abs().sum((0, 2, 3)).tolist())
    assert int((small["slabs"][1].abs().sum((0, 2, 3, 4)) > 0).sum()) < small["Sw"]
    # different variants (with / without the residual) cannot share a kernel: refused, and nothing was launched or left open
    odd = hip.multi([probs[0]["mk"](1), probs[0]["mkw"](1)])
    assert odd is None
    again = hip.multi([pr["mk"](1) for pr in probs])
    assert again is not None


# ------------------------------------------------------------------------------------------------ per-launch width hints
def test_width_hints_change_no_result(hip):
    """RtpConvGeom::wgs (include/rtp.h): an LDS-tiled launch told to run on another number of workgroups -- narrower than one per CU,
    or wider than the narrow default of a small launch -- gives the conv output bit for bit and the per-workgroup partials
    (statistics, weight-gradient slabs: stride 1 and stride 2) as sums; the slot-count queries report the width they are given, every
    reported slot is written, and wgs = 0 afterwards is the default launch exactly (the width is a parameter of the launch: nothing of
    it outlives the call)."""
    from dataclasses import replace
    n, ci, co = 8, 32, 32

    def problem(d, h, w, seed):
        geom = Geom(n, d, h, w, d, h, w, ci, co, 3, 1, 1)
        xp, xc, xg = views(hip, rnd((n, d, h, w, ci), seed, relu=True), n, d, h, w)
        rp, rc, rg = views(hip, rnd((n, d, h, w, co), seed + 1), n, d, h, w)
        xv = xp.c.float().reshape(n, -1, ci)
        st = Pair(hip, torch.stack([xv.sum(1), (xv * xv).sum(1)], -1)[:, None].contiguous())
        # stride 2 beside it: [n, 2d, h, w] -> [n, d, h/2, w/2]
        g2 = Geom(n, d * 2, h, w, d, h // 2, w // 2, ci, co, 3, 2, 1)
        x2p, x2c, x2g = views(hip, rnd((n, d * 2, h, w, ci), seed + 2, relu=True), n, d * 2, h, w)
        gy2p, gy2c, gy2g = views(hip, rnd((n, d, h // 2, w // 2, co), seed + 3), n, d, h // 2, w // 2)
        return dict(geom=geom, g2=g2, xg=xg, rg=rg, st=st, x2g=x2g, gy2g=gy2g, dims=(d, h, w))

    Wt = Pair(hip, rnd((co, ci, 3, 3, 3), 702, torch.float32, scale=0.05))
    gamma, beta = Pair(hip, rnd((ci,), 703, torch.float32) * 0.2 + 1.0), Pair(hip, rnd((ci,), 704, torch.float32) * 0.2)
    wt = Pair(hip, torch.zeros(27, co, ci))
    hip.tail([("pack_wt", Wt.g, co, co, ci, 27, wt.g)])(hip.stream())
    s = hip.stream()

    def once(pr, total_wgs):
        d, h, w = pr["dims"]
        gw, g2w = replace(pr["geom"], wgs=total_wgs), replace(pr["g2"], wgs=total_wgs)
        S, Sw, S2 = hip.conv_stats_nsplit(pr["xg"], gw, False), hip.wgrad_nsplit(gw), hip.wgrad_nsplit(g2w)
        y = views(hip, torch.zeros(n, d, h, w, co, dtype=torch.bfloat16), n, d, h, w)
        so, mr = hip.alloc((n, S, co, 2), "f32"), hip.alloc((n, 8, 2), "f32")
        slab, slab2 = hip.alloc((n, Sw, 27, co, ci), "f32"), hip.alloc((n, S2, 27, co, ci), "f32")
        hip.conv_gn_fused(pr["xg"], wt.g, None, gamma.g, beta.g, pr["st"].g, 1, 8, 1e-5, co, mr, pr["rg"], y[2], gw, True, so)(s)
        hip.wgrad(pr["rg"], pr["xg"], gw, Sw, slab)(s)
        hip.wgrad(pr["gy2g"], pr["x2g"], g2w, S2, slab2)(s)
        torch.cuda.synchronize()
        return (y[0].g.clone(), so, mr, slab, slab2), (S, Sw, S2)

    # a large launch (256 bricks per sample: one workgroup per CU by default), narrowed
    big = problem(8, 64, 128, 700)
    full, slots0 = once(big, 0)
    assert slots0[0] == 32 and slots0[1] == 32
    for total in (192, 64):
        got, slots = once(big, total)
        per = total // n
        assert slots == (per, per, min(per, slots0[2])), (total, slots)
        assert torch.equal(got[0], full[0]) and torch.equal(got[2], full[2]), "conv output / group statistics on %d workgroups" % total
        for k in (1, 3, 4):
            a, b = got[k].flatten(2), full[k].flatten(2)
            assert rel_err(a.sum(1).cpu(), b.sum(1).cpu()) < 1e-5, (total, k)
            assert int((a.abs().sum((0, 2)) > 0).sum()) == a.shape[1], "every reported slot is written (%d workgroups, buffer %d)" % (total, k)
    again, _ = once(big, 0)
    assert all(torch.equal(u, v) for u, v in zip(again, full)), "wgs = 0 again: the default launch, bit for bit"
    # a small launch (32 bricks per sample: narrow by default -- 64 / 128 workgroups in all), widened
    small = problem(4, 32, 64, 710)
    base, sl0 = once(small, 0)
    wide, sl1 = once(small, 128)
    assert sl0[0] == 8 and sl1[0] == 16 and sl1[1] >= sl0[1], (sl0, sl1)
    assert torch.equal(wide[0], base[0]) and torch.equal(wide[2], base[2])
    for k in (1, 3, 4):
        assert rel_err(wide[k].flatten(2).sum(1).cpu(), base[k].flatten(2).sum(1).cpu()) < 1e-5, k


@pytest.mark.parametrize("n", [8, 4, 16])
def test_head_last_convs_in_one_launch(hip, n):
    """The head towers' last convs (<= 16 output channels, class-bias table, fp32 output: conv_tiled variant 200) as ONE shared
    launch -- what the default plan does with conv:head.reg.2 + conv:head.hm.2 -- against the same two convs launched alone.
    n = 8: a sample per XCD; n = 4 / 16 (round 5: any sample count that divides the chip's 256 workgroups): a sample on two XCDs /
    two samples per XCD."""
    ci, d, h, w = 32, (16 if n < 8 else 8), 64, (128 if n <= 8 else 64)   # (n = 4: 512 bricks per sample -- still a "large" launch)
    outs, mk = [], []
    for k, co_real in enumerate((15, 3)):
        co = pad_to(co_real, 16)
        geom = Geom(n, d, h, w, d, h, w, ci, co, 3, 1, 1)
        xp, xc, xg = views(hip, rnd((n, d, h, w, ci), 800 + k, relu=True), n, d, h, w)
        wf = Pair(hip, rnd((1, 27, co, ci), 810 + k, scale=0.05))
        bt = Pair(hip, rnd((1, 64, co), 820 + k, torch.float32))
        ys = [views(hip, torch.zeros(n, d, h, w, co, dtype=torch.float32), n, d, h, w) for _ in range(2)]
        outs.append(ys)
        mk.append(lambda r, xg=xg, wf=wf, bt=bt, ys=ys, geom=geom: hip.conv(xg, wf.g, False, bt.g, None, ys[r][2], geom, False, False, True))
    s = hip.stream()
    for f in mk:
        f(0)(s)
    both = hip.multi([f(1) for f in mk])
    assert both is not None, "the two last convs of the head towers must share a launch"
    both(s)
    both(s)
    torch.cuda.synchronize()
    for k, ys in enumerate(outs):
        assert float(ys[0][0].g.abs().max()) > 0
        assert torch.equal(ys[0][0].g, ys[1][0].g), "problem %d" % k


@pytest.mark.parametrize("case", [(2, (4, 8, 32), 128), (1, (2, 4, 16), 64), (8, (2, 8, 32), 256)])
def test_conv64_blocks_two_towers_in_one_launch(hip, case):
    """rtp_conv64_blocks (csrc/conv64_tiled.hip): SepHead's two towers' first convs, Conv3d(C, 32, 3x3x3, bias) + ReLU each
    (center_head.py:86-93), as ONE 64-wide launch per 64-channel slice of the feature -- chained through fp32 partial sums for
    C = 128 / 256 -- and their data gradients as one launch per 64 input channels that writes the SUM over both towers; against
    torch's conv3d / autograd in fp32 on the same bf16-representable operands."""
    import torch.nn.functional as F
    n, dims, C = case
    d, h, w = dims
    geom = Geom(n, d, h, w, d, h, w, 64, 64, 3, 1, 1)
    x = rnd((n, d, h, w, C), 700, relu=True)
    W = [rnd((32, C, 3, 3, 3), 701 + t, torch.float32, 0.05).to(torch.bfloat16).float() for t in range(2)]
    b = [rnd((32,), 703 + t, torch.float32) for t in range(2)]
    dev = hip.device
    xg = x.to(dev)
    K = C // 32
    wf = [[W[t][:, 32 * k:32 * k + 32].reshape(32, 32, 27).permute(2, 0, 1).contiguous().to(torch.bfloat16).to(dev) for k in range(K)]
          for t in range(2)]                                                        # [tower][slice]: [27][32 co][32 ci]
    wd = [[W[t][:, 32 * k:32 * k + 32].reshape(32, 32, 27).permute(2, 1, 0).contiguous().to(torch.bfloat16).to(dev) for k in range(K)]
          for t in range(2)]                                                        # [27][32 ci][32 co]
    bt = [b[t][None].expand(64, 32).contiguous().to(dev) for t in range(2)]
    y = [torch.zeros(n, d, h, w, 32, dtype=torch.bfloat16, device=dev) for _ in range(2)]
    yv = [View(t, n, d, h, w, 32, 0, 32) for t in y]
    acc = hip.alloc((n, d * h * w, 64), "f32")
    s = hip.stream()
    M = C // 64
    for m in range(M):
        xs = [View(xg, n, d, h, w, C, 64 * m + 32 * k, 32) for k in range(2)]
        blocks = [[(wf[t][2 * m + k], 0) for k in range(2)] for t in range(2)]
        last = m == M - 1
        hip.conv64_blocks(xs, blocks, 32, 1024, [(bt[0], 0), (bt[1], 0)] if last else None, 32, None, yv if last else None, geom,
                          last, False, acc if M > 1 else None, m > 0, not last)(s)
    torch.cuda.synchronize()
    xr = x.float().permute(0, 4, 1, 2, 3).contiguous().requires_grad_(True)
    want = [F.relu(F.conv3d(xr, W[t], b[t], padding=1)) for t in range(2)]
    for t in range(2):
        assert rel_err(y[t].float().cpu().permute(0, 4, 1, 2, 3), want[t]) < BF, "tower %d forward %r" % (t, case)
    # data gradients: one tensor for both towers
    gy = [rnd((n, d, h, w, 32), 710 + t) for t in range(2)]
    gyg = [View(t.to(dev), n, d, h, w, 32, 0, 32) for t in gy]
    dx = torch.zeros(n, d, h, w, C, dtype=torch.bfloat16, device=dev)
    for j in range(C // 64):
        blocks = [[(wd[t][2 * j + hh], 0) for t in range(2)] for hh in range(2)]   # [output half = slice 2j + hh][input half = tower]
        ys = [View(dx, n, d, h, w, C, 64 * j + 32 * hh, 32) for hh in range(2)]
        hip.conv64_blocks(gyg, blocks, 32, 1024, None, 32, None, ys, geom, False, True)(s)
    torch.cuda.synchronize()
    lin = sum((F.conv3d(xr, W[t], None, padding=1) * gy[t].float().permute(0, 4, 1, 2, 3)).sum() for t in range(2))
    lin.backward()
    assert rel_err(dx.float().cpu().permute(0, 4, 1, 2, 3), xr.grad) < BF, "data gradient of both towers %r" % (case,)


def test_conv64_blocks_refuses_a_ragged_width(hip):
    """include/rtp.h documents W % 16 == 0 for rtp_conv64_blocks: the chain's fp32 partial sums are in brick layout, so W = 40 (three
    brick columns = 48 voxel columns) would write past an `acc` of the documented n * voxels * 64 floats.  The entry
    point returns RTP_ERR_UNSUPPORTED before anything is launched."""
    from rt_pose_amd._lib import RtpError
    n, d, h, w, C = 1, 2, 4, 40, 64
    geom = Geom(n, d, h, w, d, h, w, 64, 64, 3, 1, 1)
    dev = hip.device
    xg = torch.zeros(n, d, h, w, C, dtype=torch.bfloat16, device=dev)
    wf = torch.zeros(27, 32, 32, dtype=torch.bfloat16, device=dev)
    y = [torch.zeros(n, d, h, w, 32, dtype=torch.bfloat16, device=dev) for _ in range(2)]
    xs = [View(xg, n, d, h, w, C, 32 * k, 32) for k in range(2)]
    yv = [View(t, n, d, h, w, 32, 0, 32) for t in y]
    acc = hip.alloc((n, d * h * w, 64), "f32")
    blocks = [[(wf, 0) for _ in range(2)] for _ in range(2)]
    for kw in (dict(), dict(acc=acc, acc_in=False, acc_out=True)):
        with pytest.raises(RtpError, match="UNSUPPORTED"):
            hip.conv64_blocks(xs, blocks, 32, 1024, None, 32, None, None if kw else yv, geom, False, False, **kw)(hip.stream())
    torch.cuda.synchronize()
