"""Deformable convolution on a real MI355X: the C-ABI kernels (through rt_pose_amd.dcn) against oracle/dcn_ref.py
(forward and all gradients), plus the reference wrapper's error behaviour.  fp32; tolerance 2e-4 norm-wise (the
gradients w.r.t. the input use fp32 atomics, so their summation order varies run to run)."""
import pytest
import torch

from oracle.dcn_ref import deform_conv2d
from tests.util import rel_err

pytestmark = pytest.mark.gpu
TOL = 2e-4


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


CASES = [
    # n, c, h, w, co, k, stride, pad, dil, groups, dg, im2col_step
    (2, 8, 9, 11, 6, 3, 1, 1, 1, 1, 1, 64),
    (4, 16, 12, 10, 8, 3, 2, 1, 1, 2, 4, 2),
    (2, 64, 16, 20, 64, 3, 1, 1, 1, 1, 4, 64),     # FeatureAdaption shape: 3x3, pad 1, deformable_groups=4 (center_head.py:24-62)
    (1, 4, 7, 7, 4, 3, 1, 2, 2, 1, 2, 64),
    (2, 8, 16, 16, 8, 3, 1, 1, 1, 1, 2, 2),        # P = 256: the MFMA weight-gradient GEMM (needs P % 64 == 0, <= 32 output channels)
    (2, 40, 8, 8, 24, 3, 1, 1, 1, 1, 4, 1),        # 360 column rows: two row chunks in both MFMA GEMMs of the backward
    (3, 12, 33, 41, 40, 3, 1, 1, 1, 1, 2, 3),      # ragged: P = 1353 is no multiple of the 256-position block, co = 40 -> two MFMA row tiles
    # the fused weight gradient (dcn_gradw_fused_kernel: 32 channels in 4 deformable groups, 3x3, <= 32 output channels, P % 64 == 0)
    (3, 32, 8, 16, 32, 3, 1, 1, 1, 1, 4, 1),
    (2, 32, 16, 24, 20, 3, 1, 1, 1, 1, 4, 2),
    (2, 32, 16, 16, 32, 3, 2, 1, 1, 1, 4, 64),     # stride 2: P = 64
    (1, 32, 12, 20, 32, 3, 1, 2, 2, 1, 4, 64),     # dilation 2: P = 240 -> not fused (P % 64), the column route on the same channels
]


MODES = {"window": {}, "gather": {"RTP_DCN_NOWIN": "1"}, "unfused": {"RTP_DCN_UNFUSED": "1"}}


def set_mode(monkeypatch, mode):
    """Forward paths of csrc/dcn.hip: LDS-window fused kernel (default), global-gather fused kernel, im2col + GEMM (the
    only path for groups > 1 or an odd number of channels per deformable group)."""
    for k in ("RTP_DCN_NOWIN", "RTP_DCN_UNFUSED"):
        monkeypatch.delenv(k, raising=False)
    for k, v in MODES[mode].items():
        monkeypatch.setenv(k, v)


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("case", CASES)
def test_dcn_v1_forward_backward(case, mode, monkeypatch):
    set_mode(monkeypatch, mode)
    from rt_pose_amd.dcn import deform_conv
    n, c, h, w, co, k, stride, pad, dil, groups, dg, step = case
    x = rnd(n, c, h, w, seed=1).requires_grad_(True)
    wt = rnd(co, c // groups, k, k, seed=2, scale=0.2).requires_grad_(True)
    ho = (h + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    wo = (w + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    off = (rnd(n, dg * 2 * k * k, ho, wo, seed=3, scale=1.5)).requires_grad_(True)
    ref = deform_conv2d(x, off, wt, stride, pad, dil, groups, dg)
    gy = rnd(*ref.shape, seed=4)
    ref.backward(gy)
    xg, wg, og = [t.detach().cuda().requires_grad_(True) for t in (x, wt, off)]
    out = deform_conv(xg, og, wg, stride, pad, dil, groups, dg, step)
    out.backward(gy.cuda())
    torch.cuda.synchronize()
    assert rel_err(out.cpu(), ref) < TOL
    assert rel_err(xg.grad.cpu(), x.grad) < TOL
    assert rel_err(wg.grad.cpu(), wt.grad) < TOL
    assert rel_err(og.grad.cpu(), off.grad) < TOL


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("off_scale,halo", [(6.0, "2"), (3.0, "0"), (40.0, "1")])
def test_dcn_forward_offsets_leaving_the_window(off_scale, halo, mode, monkeypatch):
    """Offsets far larger than the staged halo (and past the image): the window kernel fetches those samples from global
    memory per lane; every forward path must still give the oracle's result.  Several 512-position tiles per image."""
    from rt_pose_amd.dcn import deform_conv
    set_mode(monkeypatch, mode)
    monkeypatch.setenv("RTP_DCN_HALO", halo)
    n, c, h, w, co, dg = 2, 8, 48, 40, 8, 2
    x = rnd(n, c, h, w, seed=11)
    wt = rnd(co, c, 3, 3, seed=12, scale=0.2)
    off = rnd(n, dg * 18, h, w, seed=13, scale=off_scale)
    ref = deform_conv2d(x, off, wt, 1, 1, 1, 1, dg)
    out = deform_conv(x.cuda(), off.cuda(), wt.cuda(), 1, 1, 1, 1, dg, 2)
    torch.cuda.synchronize()
    assert rel_err(out.cpu(), ref) < TOL


@pytest.mark.parametrize("R", ["0", "1", "3"])
@pytest.mark.parametrize("off_scale", [0.4, 2.5])
def test_dcn_grad_input_gather_radius(R, off_scale, monkeypatch):
    """grad_input paths of csrc/dcn.hip: atomic-free gather for offsets <= R pixels + atomic scatter of the outliers
    (R = 0: everything through the LDS-plane scatter).  Stride 2 / dilation 2 / modulated cases included."""
    from rt_pose_amd.dcn import deform_conv, modulated_deform_conv
    monkeypatch.setenv("RTP_DCN_GATHER_R", R)
    for (n, c, h, w, co, stride, pad, dil, dg) in [(2, 8, 20, 24, 8, 1, 1, 1, 2), (2, 6, 21, 17, 4, 2, 2, 2, 3)]:
        x = rnd(n, c, h, w, seed=21).requires_grad_(True)
        wt = rnd(co, c, 3, 3, seed=22, scale=0.2).requires_grad_(True)
        ho = (h + 2 * pad - (dil * 2 + 1)) // stride + 1
        wo = (w + 2 * pad - (dil * 2 + 1)) // stride + 1
        off = rnd(n, dg * 18, ho, wo, seed=23, scale=off_scale).requires_grad_(True)
        ref = deform_conv2d(x, off, wt, stride, pad, dil, 1, dg)
        gy = rnd(*ref.shape, seed=24)
        ref.backward(gy)
        xg, wg, og = [t.detach().cuda().requires_grad_(True) for t in (x, wt, off)]
        deform_conv(xg, og, wg, stride, pad, dil, 1, dg, n).backward(gy.cuda())
        torch.cuda.synchronize()
        assert rel_err(xg.grad.cpu(), x.grad) < TOL
        assert rel_err(og.grad.cpu(), off.grad) < TOL
    # modulated (mask multiplies the column gradient before the scatter)
    n, c, h, w, co, dg = 2, 8, 14, 18, 4, 2
    x = rnd(n, c, h, w, seed=31).requires_grad_(True)
    wt = rnd(co, c, 3, 3, seed=32, scale=0.2).requires_grad_(True)
    off = rnd(n, dg * 18, h, w, seed=33, scale=off_scale).requires_grad_(True)
    m = torch.sigmoid(rnd(n, dg * 9, h, w, seed=34)).requires_grad_(True)
    ref = deform_conv2d(x, off, wt, 1, 1, 1, 1, dg, mask=m)
    gy = rnd(*ref.shape, seed=35)
    ref.backward(gy)
    ts = [t.detach().cuda().requires_grad_(True) for t in (x, off, m, wt)]
    modulated_deform_conv(ts[0], ts[1], ts[2], ts[3], None, 1, 1, 1, 1, dg).backward(gy.cuda())
    torch.cuda.synchronize()
    for got, want in zip(ts, (x, off, m, wt)):
        assert rel_err(got.grad.cpu(), want.grad) < TOL


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("with_bias", [True, False])
def test_dcn_v2_forward_backward(with_bias, mode, monkeypatch):
    set_mode(monkeypatch, mode)
    from rt_pose_amd.dcn import modulated_deform_conv
    n, c, h, w, co, k, dg = 2, 16, 10, 12, 8, 3, 2
    x = rnd(n, c, h, w, seed=5).requires_grad_(True)
    wt = rnd(co, c, k, k, seed=6, scale=0.2).requires_grad_(True)
    off = rnd(n, dg * 18, h, w, seed=7, scale=1.2).requires_grad_(True)
    m = torch.sigmoid(rnd(n, dg * 9, h, w, seed=8)).requires_grad_(True)
    b = rnd(co, seed=9).requires_grad_(True) if with_bias else None
    ref = deform_conv2d(x, off, wt, 1, 1, 1, 1, dg, mask=m, bias=b)
    gy = rnd(*ref.shape, seed=10)
    ref.backward(gy)
    ts = [t.detach().cuda().requires_grad_(True) for t in (x, off, m, wt)]
    bg = b.detach().cuda().requires_grad_(True) if with_bias else None
    out = modulated_deform_conv(ts[0], ts[1], ts[2], ts[3], bg, 1, 1, 1, 1, dg)
    out.backward(gy.cuda())
    torch.cuda.synchronize()
    assert rel_err(out.cpu(), ref) < TOL
    for got, want in zip(ts, (x, off, m, wt)):
        assert rel_err(got.grad.cpu(), want.grad) < TOL
    if with_bias:
        assert rel_err(bg.grad.cpu(), b.grad) < TOL


def test_zero_offset_pack_is_conv_and_error_behaviour():
    import torch.nn.functional as F
    from rt_pose_amd.dcn import DeformConv, DeformConvPack, deform_conv
    torch.manual_seed(0)
    m = DeformConvPack(8, 8, 3, padding=1, deformable_groups=4).cuda()   # conv_offset is zero-initialised
    x = torch.randn(2, 8, 12, 12, device="cuda")
    assert rel_err(m(x).cpu(), F.conv2d(x, m.weight, None, 1, 1).cpu()) < 1e-5
    with pytest.raises(ValueError):
        deform_conv(x[0], x[0], m.weight)                                 # non 4-D input
    with pytest.raises(NotImplementedError):
        deform_conv(x.cpu(), torch.zeros(2, 18, 12, 12), m.weight.cpu())  # CPU tensors
    with pytest.raises(AssertionError):
        deform_conv(torch.randn(3, 8, 12, 12, device="cuda"), torch.zeros(3, 72, 12, 12, device="cuda"), m.weight, 1, 1, 1, 1, 4, 2)
    with pytest.raises(AssertionError):
        DeformConv(8, 8, 3, bias=True)
    # input smaller than the kernel is padded and cropped (deform_conv.py:230-244)
    small = DeformConv(4, 4, 3, padding=1).cuda()
    y = small(torch.randn(1, 4, 2, 2, device="cuda"), torch.zeros(1, 18, 2, 2, device="cuda"))
    assert tuple(y.shape) == (1, 4, 2, 2)


def test_full_size_properties():
    """BASELINE config 4 shape ([128,32,64,160], deformable_groups 4, im2col_step 64), where the oracle is too slow:
    size-independent properties.  (a) zero offsets: DCN == conv2d (forward and all three gradients, torch fp32 as the
    comparator); (b) integer offsets (every tap shifted by (+1, -2)): DCN == conv2d of the shifted image (away from the border);
    (c) linearity in the input at random fractional offsets; (d) the gather and the LDS-plane / scatter grad_input paths
    agree at random offsets."""
    import os
    import torch.nn.functional as F
    from rt_pose_amd.dcn import deform_conv
    n, c, h, w, co, dg = 128, 32, 64, 160, 32, 4
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(n, c, h, w, device="cuda", generator=g)
    wt = torch.randn(co, c, 3, 3, device="cuda", generator=g) * 0.05
    gy = torch.randn(n, co, h, w, device="cuda", generator=g)
    zero = torch.zeros(n, dg * 18, h, w, device="cuda")
    # (a)
    xa, wa, oa = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), zero.clone().requires_grad_(True)
    ya = deform_conv(xa, oa, wa, 1, 1, 1, 1, dg, 64)
    ya.backward(gy)
    xr, wr = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, 1, 1)
    yr.backward(gy)
    assert rel_err(ya, yr) < 1e-5
    assert rel_err(xa.grad, xr.grad) < 1e-5
    assert rel_err(wa.grad, wr.grad) < 1e-4   # 1.3 M-term fp32 sums in different orders
    # (b)
    off = zero.clone()
    off[:, 0::2] = 1.0
    off[:, 1::2] = -2.0
    yb = deform_conv(x, off, wt, 1, 1, 1, 1, dg, 64)
    xs = torch.zeros_like(x)
    xs[:, :, :h - 1, 2:] = x[:, :, 1:, :w - 2]           # xs[y, x] = x[y + 1, x - 2]
    # interior only: at the border the shifted taps read real pixels where the plain conv sees its zero padding
    assert rel_err(yb[:, :, 2:h - 2, 4:w - 4], F.conv2d(xs, wt, None, 1, 1)[:, :, 2:h - 2, 4:w - 4]) < 1e-5
    # (c) + (d)
    offr = torch.randn(n, dg * 18, h, w, device="cuda", generator=g) * 0.8
    x2 = torch.randn(n, c, h, w, device="cuda", generator=g)
    y1, y2 = deform_conv(x, offr, wt, 1, 1, 1, 1, dg, 64), deform_conv(x2, offr, wt, 1, 1, 1, 1, dg, 64)
    y12 = deform_conv(x + 2.0 * x2, offr, wt, 1, 1, 1, 1, dg, 64)
    assert rel_err(y12, y1 + 2.0 * y2) < 1e-5
    grads = {}
    for R in ("2", "0"):
        os.environ["RTP_DCN_GATHER_R"] = R
        try:
            xg = x.clone().requires_grad_(True)
            og = offr.clone().requires_grad_(True)
            deform_conv(xg, og, wt, 1, 1, 1, 1, dg, 64).backward(gy)
            grads[R] = (xg.grad.clone(), og.grad.clone())
        finally:
            del os.environ["RTP_DCN_GATHER_R"]
    assert rel_err(grads["2"][0], grads["0"][0]) < 1e-5
    assert rel_err(grads["2"][1], grads["0"][1]) < 1e-6


def _random_case(rs):
    k = int(rs.choice([1, 3, 3, 3]))
    stride, dil = int(rs.choice([1, 1, 2])), int(rs.choice([1, 1, 2]))
    pad = int(rs.randint(0, 3))
    groups = int(rs.choice([1, 1, 1, 2]))
    dg = int(rs.choice([1, 2, 4]))
    cmul = int(rs.randint(1, 5))
    c = groups * dg * cmul * int(rs.choice([1, 2]))
    co = groups * int(rs.randint(1, 21))
    h, w = int(rs.randint(5, 40)), int(rs.randint(5, 40))
    if rs.rand() < 0.5:
        w = (w + 3) // 4 * 4          # the padded-row window kernel needs W % 4 == 0
    n = int(rs.choice([1, 2, 3, 4]))
    step = int(rs.choice([d for d in (1, 2, 3, 4) if n % d == 0]))
    return n, c, h, w, co, k, stride, pad, dil, groups, dg, step


@pytest.mark.parametrize("seed", range(24))
def test_dcn_random_geometries(seed):
    """Random kernel size / stride / padding / dilation / groups / deformable groups / image sizes (ragged tiles, windows
    that hang over the image, single-row outputs ...) and offset scales against the oracle: forward + all gradients."""
    import numpy as np
    from rt_pose_amd.dcn import deform_conv
    rs = np.random.RandomState(1000 + seed)
    for _ in range(20):
        n, c, h, w, co, k, stride, pad, dil, groups, dg, step = _random_case(rs)
        ho = (h + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
        wo = (w + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
        if ho >= 1 and wo >= 1:
            break
    else:
        pytest.skip("no valid geometry drawn")
    scale = float(rs.choice([0.0, 0.3, 1.0, 3.0, 12.0]))
    x = rnd(n, c, h, w, seed=seed * 7 + 1).requires_grad_(True)
    wt = rnd(co, c // groups, k, k, seed=seed * 7 + 2, scale=0.3).requires_grad_(True)
    off = rnd(n, dg * 2 * k * k, ho, wo, seed=seed * 7 + 3, scale=scale).requires_grad_(True)
    ref = deform_conv2d(x, off, wt, stride, pad, dil, groups, dg)
    gy = rnd(*ref.shape, seed=seed * 7 + 4)
    ref.backward(gy)
    xg, wg, og = [t.detach().cuda().requires_grad_(True) for t in (x, wt, off)]
    out = deform_conv(xg, og, wg, stride, pad, dil, groups, dg, step)
    out.backward(gy.cuda())
    torch.cuda.synchronize()
    case = (n, c, h, w, co, k, stride, pad, dil, groups, dg, step, scale)
    assert rel_err(out.cpu(), ref) < TOL, case
    assert rel_err(xg.grad.cpu(), x.grad) < TOL, case
    assert rel_err(wg.grad.cpu(), wt.grad) < TOL, case
    if scale > 0 or float(off.grad.abs().max()) > 0:
        assert rel_err(og.grad.cpu(), off.grad) < TOL, case


FUSED_BWD = [
    # n, h, w, co, offset scale, row segments (RTP_DCN_SEGS; 0 = the launcher's choice), weight gradient in the same call
    (2, 8, 16, 32, 0.5, 0, True),
    (1, 12, 60, 32, 0.5, 1, True),       # two strips (56 owned columns each), the second one ragged
    (2, 9, 160, 32, 0.5, 0, True),       # the head's row width: three strips
    (1, 20, 64, 20, 1.0, 3, True),       # co < 32, three row segments with halo rows, some samples beyond the 5 x 5 patch
    (2, 7, 8, 32, 0.0, 1, True),         # zero offsets: a plain convolution
    (1, 16, 120, 32, 3.0, 2, True),      # most samples are outliers: the listed-row scatter kernel
    (1, 33, 36, 8, 12.0, 4, True),       # nearly every sample leaves the image
    (2, 10, 116, 32, 0.7, 2, False),     # input / offset gradients alone (rtp_deform_conv_backward_input), weights by their own entry
    (3, 1, 4, 32, 0.6, 1, True),         # a single row, the narrowest row the path takes: the ring only ever flushes
    (2, 2, 8, 32, 1.2, 2, True),         # two rows in two segments (each segment is all halo but one row)
    (1, 3, 12, 5, 0.3, 3, True),         # three rows, three segments, 5 output channels
]


@pytest.mark.parametrize("case", FUSED_BWD)
def test_dcn_one_pass_backward(case, monkeypatch):
    """dcn_bwd_fused_kernel (+ dcn_bwd_outlier_rows_kernel): the geometry of the DCN head -- 3x3, stride 1, pad 1, 32 channels in
    4 deformable groups -- takes the one-pass backward; strips, ragged last strip, row segments with halo rows, co < 32, samples
    outside the register patch and outside the image, against the oracle and against the column route of the same library."""
    from rt_pose_amd.dcn import deform_conv
    n, h, w, co, scale, segs, together = case
    monkeypatch.delenv("RTP_DCN_NO_FUSED_BWD", raising=False)
    if segs:
        monkeypatch.setenv("RTP_DCN_SEGS", str(segs))
    else:
        monkeypatch.delenv("RTP_DCN_SEGS", raising=False)
    x = rnd(n, 32, h, w, seed=11).requires_grad_(True)
    wt = rnd(co, 32, 3, 3, seed=12, scale=0.2).requires_grad_(True)
    off = rnd(n, 72, h, w, seed=13, scale=scale).requires_grad_(True)
    ref = deform_conv2d(x, off, wt, 1, 1, 1, 1, 4)
    gy = rnd(*ref.shape, seed=14)
    ref.backward(gy)

    def run():
        xg, wg, og = [t.detach().cuda().requires_grad_(True) for t in (x, wt, off)]
        out = deform_conv(xg, og, wg, 1, 1, 1, 1, 4, 64)
        if together:
            out.backward(gy.cuda())
        else:
            gx, go = torch.autograd.grad(out, (xg, og), gy.cuda(), retain_graph=True)
            gw, = torch.autograd.grad(out, (wg,), gy.cuda())
            xg.grad, og.grad, wg.grad = gx, go, gw
        torch.cuda.synchronize()
        return xg.grad.cpu(), wg.grad.cpu(), og.grad.cpu()

    gx, gw, go = run()
    assert rel_err(gx, x.grad) < TOL, case
    assert rel_err(gw, wt.grad) < TOL, case
    if scale > 0:
        assert rel_err(go, off.grad) < TOL, case
    else:
        assert rel_err(go, off.grad) < TOL or float(off.grad.abs().max()) == 0
    monkeypatch.setenv("RTP_DCN_FP32_MFMA", "1")         # the two products on v_mfma_f32_32x32x2_f32 instead of the bf16 (hi, lo) split
    fx, fw, fo = run()
    monkeypatch.delenv("RTP_DCN_FP32_MFMA")
    assert rel_err(fx, x.grad) < TOL and rel_err(fw, wt.grad) < TOL and rel_err(fo, go) < TOL, case
    monkeypatch.setenv("RTP_DCN_NO_FUSED_BWD", "1")      # the column route (im2col / GEMM / gather kernels) on the same inputs
    cx, cw, co_ = run()
    assert rel_err(gx, cx) < TOL and rel_err(gw, cw) < TOL and rel_err(go, co_) < TOL, case
    # the accumulating entry (the reference's contract: the caller's buffers are added to) on buffers that hold ones
    if together:
        import ctypes as C
        from rt_pose_amd import _lib
        monkeypatch.delenv("RTP_DCN_NO_FUSED_BWD", raising=False)
        lib = _lib.load()
        xg, wg, og, gyg = [t.detach().cuda().contiguous() for t in (x, wt, off, gy)]
        bi, bo, bw = torch.ones_like(xg), torch.ones_like(og), torch.ones_like(wg)
        ws = torch.empty(lib.rtp_dcn_workspace_bytes(n, 32, h, w, co, 3, 3, h, w) // 4, device="cuda")
        pv = lambda t: C.c_void_p(t.data_ptr())
        rc = lib.rtp_deform_conv_backward(pv(xg), pv(og), pv(gyg), pv(bi), pv(bo), pv(wg), pv(bw), pv(ws), n, 32, h, w, co, 3, 3,
                                          1, 1, 1, 1, 1, 1, 1, 4, 1.0, n, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert rc == 0
        assert rel_err(bi.cpu() - 1, x.grad) < 5 * TOL and rel_err(bw.cpu() - 1, wt.grad) < 5 * TOL, case
        # grad_offset is assigned by both routes (deformable_col2im_coord writes, it does not add)
        assert rel_err(bo.cpu(), off.grad) < TOL or (scale == 0 and float(off.grad.abs().max()) == 0), case
    # offsets within the patch everywhere: the input gradient has no atomics -> bit-reproducible
    if scale <= 0.5:
        monkeypatch.delenv("RTP_DCN_NO_FUSED_BWD", raising=False)
        rx, _, ro = run()
        assert torch.equal(rx, gx) and torch.equal(ro, go), case


@pytest.mark.parametrize("seed", range(16))
def test_dcn_one_pass_backward_random_shapes(seed, monkeypatch):
    """Random image sizes (rows 1 .. 40, widths 4 .. 200 in steps of 4: one to four strips, ragged last strips), output channel
    counts, offset scales and row-segment counts through the one-pass backward: all three gradients against the oracle."""
    import numpy as np
    from rt_pose_amd.dcn import deform_conv
    rs = np.random.RandomState(4000 + seed)
    n, h, w = int(rs.randint(1, 4)), int(rs.randint(1, 41)), 4 * int(rs.randint(1, 51))
    co = int(rs.choice([1, 7, 16, 31, 32]))
    scale = float(rs.choice([0.0, 0.2, 0.5, 0.9, 1.5, 2.5, 6.0]))
    segs = int(rs.randint(0, 6))
    monkeypatch.delenv("RTP_DCN_NO_FUSED_BWD", raising=False)
    if segs:
        monkeypatch.setenv("RTP_DCN_SEGS", str(segs))
    else:
        monkeypatch.delenv("RTP_DCN_SEGS", raising=False)
    x = rnd(n, 32, h, w, seed=seed * 5 + 1).requires_grad_(True)
    wt = rnd(co, 32, 3, 3, seed=seed * 5 + 2, scale=0.2).requires_grad_(True)
    off = rnd(n, 72, h, w, seed=seed * 5 + 3, scale=scale).requires_grad_(True)
    ref = deform_conv2d(x, off, wt, 1, 1, 1, 1, 4)
    gy = rnd(*ref.shape, seed=seed * 5 + 4)
    ref.backward(gy)
    xg, wg, og = [t.detach().cuda().requires_grad_(True) for t in (x, wt, off)]
    deform_conv(xg, og, wg, 1, 1, 1, 1, 4, 64).backward(gy.cuda())
    torch.cuda.synchronize()
    case = (n, h, w, co, scale, segs)
    assert rel_err(xg.grad.cpu(), x.grad) < TOL, case
    assert rel_err(wg.grad.cpu(), wt.grad) < TOL, case
    if float(off.grad.abs().max()) > 0:
        assert rel_err(og.grad.cpu(), off.grad) < TOL, case


@pytest.mark.parametrize("mode", list(MODES))
def test_small_fixed_case_on_the_gpu(mode, monkeypatch):
    """The HIP operator (through the C ABI) on the small fixed case of tests/golden/dcn_known_answer.json, whose expected
    values were generated by the oracle (self-consistency, not an external pin): forward and all gradients."""
    from rt_pose_amd.dcn import deform_conv
    from tests.test_dcn_oracle import run_known_answer
    set_mode(monkeypatch, mode)

    def dcn(x, off, w):
        return deform_conv(x.cuda(), off.cuda(), w.cuda(), 1, 0, 1, 1, 1, 1).cpu()
    run_known_answer(dcn)


@pytest.mark.parametrize("case", [
    # frames, z, h, w, offset sigma (pixels), relu
    (1, 2, 8, 16, 0.5, True), (2, 3, 16, 32, 1.5, False), (1, 1, 4, 4, 0.0, True), (2, 2, 12, 20, 6.0, True),   # 6 px: samples leave the image
    (1, 4, 64, 160, 0.7, True),                                                                                    # the native slice shape
])
def test_dcn_forward_on_the_plans_layout(case):
    """rtp_dcn_cl_forward (csrc/dcn_cl.hip): bf16 channels-last feature + fp32 channels-last offsets -> bf16 channels-last output, the
    deformable half of FeatureAdaption (center_head.py:24-62) without the hand-off to the fp32 NCHW operator -- against the oracle
    (oracle/dcn_ref.py, parity unpinned) on the same bf16-representable feature and weights, and against the section-D operator on
    the device.  Tolerance: the samples are rounded to bf16 before the product (one more bf16 rounding than the operator's output
    rounding): norm-wise 6e-3."""
    from oracle import dcn_ref
    from rt_pose_amd.backend import HipBackend
    from rt_pose_amd.graph import View
    from tests.util import rel_err
    hip = HipBackend("cuda:0")
    nf, z, h, w, sigma, relu = case
    g = torch.Generator().manual_seed(99)
    x = (torch.randn(nf, z, h, w, 32, generator=g)).to(torch.bfloat16)
    off = torch.randn(nf, z, h, w, 80, generator=g) * sigma          # 72 used of an 80-channel row
    wt = (torch.randn(32, 32, 3, 3, generator=g) * 0.1).to(torch.bfloat16).float()
    dev = hip.device
    xg, og, wg = x.to(dev), off.to(dev), wt.to(dev)
    y = torch.zeros(nf, z, h, w, 32, dtype=torch.bfloat16, device=dev)
    hip.dcn_cl_forward(View(xg, nf, z, h, w, 32, 0, 32), View(og, nf, z, h, w, 80, 0, 72), wg, View(y, nf, z, h, w, 32, 0, 32), relu)(hip.stream())
    torch.cuda.synchronize()
    x2 = x.float().permute(0, 1, 4, 2, 3).reshape(nf * z, 32, h, w)
    o2 = off[..., :72].permute(0, 1, 4, 2, 3).reshape(nf * z, 72, h, w).contiguous()
    want = dcn_ref.deform_conv2d(x2, o2, wt, 1, 1, 1, 1, 4)
    if relu:
        want = torch.relu(want)
    got = y.float().cpu().permute(0, 1, 4, 2, 3).reshape(nf * z, 32, h, w)
    assert torch.isfinite(got).all()
    assert rel_err(got, want) < 6e-3, (case, rel_err(got, want))
    # the fp32 NCHW operator on the same inputs
    from rt_pose_amd import deform_conv_cuda as dcc
    out = torch.zeros(nf * z, 32, h, w, device=dev)
    cols, ones = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    dcc.deform_conv_forward_cuda(x2.to(dev).contiguous(), wg, o2.to(dev), out, cols, ones, 3, 3, 1, 1, 1, 1, 1, 1, 1, 4, min(64, nf * z))
    ref = torch.relu(out) if relu else out
    assert rel_err(got, ref.cpu()) < 6e-3
