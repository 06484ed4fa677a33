"""Property tests for oracle/dcn_ref.py.  The reference ships no tests for this operator and its CUDA sources cannot be built
here, so the oracle is PARITY UNPINNED: the properties below (and a small fixed case whose expected values the oracle itself
produced, tests/golden/dcn_known_answer.json -- read its provenance) are all that holds it."""
import pytest
import torch
import torch.nn.functional as F

from oracle.dcn_ref import deform_conv2d


def rnd(*shape, seed=0, dtype=torch.float64):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed), dtype=dtype)


@pytest.mark.parametrize("stride,pad,dil,groups,dg", [(1, 1, 1, 1, 1), (2, 1, 1, 2, 4), (1, 2, 2, 1, 2)])
def test_zero_offsets_is_plain_conv(stride, pad, dil, groups, dg):
    x, w = rnd(2, 8, 9, 11, seed=1), rnd(6, 8 // groups, 3, 3, seed=2)
    ref = F.conv2d(x, w, None, stride, pad, dil, groups)
    off = torch.zeros(2, dg * 18, ref.shape[2], ref.shape[3], dtype=torch.float64)
    torch.testing.assert_close(deform_conv2d(x, off, w, stride, pad, dil, groups, dg), ref, rtol=1e-10, atol=1e-10)


def test_integer_offsets_shift_the_input():
    x, w = rnd(1, 4, 10, 10, seed=3), rnd(5, 4, 3, 3, seed=4)
    off = torch.zeros(1, 18, 10, 10, dtype=torch.float64)
    off[:, 0::2] = 1.0   # every tap samples one row lower
    off[:, 1::2] = -2.0  # and two columns to the left
    shifted = torch.zeros_like(x)
    shifted[:, :, :-1, 2:] = x[:, :, 1:, :-2]  # shifted[h,w] = x[h+1, w-2], zero outside
    # interior only: at the border the deformable window (-1,H) differs from zero padding of the shifted image
    got = deform_conv2d(x, off, w, 1, 1)[:, :, 2:-2, 3:-2]
    ref = F.conv2d(shifted, w, None, 1, 1)[:, :, 2:-2, 3:-2]
    torch.testing.assert_close(got, ref, rtol=1e-10, atol=1e-10)


def test_taps_pushed_outside_give_zero():
    x, w = rnd(1, 2, 6, 6, seed=5), rnd(3, 2, 3, 3, seed=6)
    off = torch.full((1, 18, 6, 6), 100.0, dtype=torch.float64)
    assert float(deform_conv2d(x, off, w, 1, 1).abs().max()) == 0.0
    # exactly on the open window edge h = -1 is outside, h = -0.5 is half of row 0
    off = torch.zeros(1, 2, 1, 1, dtype=torch.float64)
    x1 = torch.ones(1, 1, 3, 3, dtype=torch.float64)
    w1 = torch.ones(1, 1, 1, 1, dtype=torch.float64)
    off[0, 0] = -1.0
    assert float(deform_conv2d(x1[:, :, :1, :1].expand(1, 1, 1, 1).contiguous(), off, w1)) == 0.0
    off[0, 0] = -0.5
    assert float(deform_conv2d(x1[:, :, :1, :1].contiguous(), off, w1)) == pytest.approx(0.5)


def test_unit_mask_reduces_v2_to_v1_and_bias_adds():
    x, w = rnd(2, 8, 7, 7, seed=7), rnd(4, 8, 3, 3, seed=8)
    off = rnd(2, 2 * 18, 7, 7, seed=9) * 0.7
    b = rnd(4, seed=10)
    v1 = deform_conv2d(x, off, w, 1, 1, 1, 1, 2)
    v2 = deform_conv2d(x, off, w, 1, 1, 1, 1, 2, mask=torch.ones(2, 2 * 9, 7, 7, dtype=torch.float64), bias=b)
    torch.testing.assert_close(v2, v1 + b.view(1, -1, 1, 1), rtol=1e-12, atol=1e-12)


def test_gradcheck_fp64():
    x = rnd(1, 4, 5, 5, seed=11).requires_grad_(True)
    w = rnd(2, 2, 3, 3, seed=12).requires_grad_(True)
    # keep sample points away from integer coordinates (the bilinear kinks) so the numeric Jacobian is well defined
    off = (torch.rand(1, 2 * 18, 5, 5, generator=torch.Generator().manual_seed(13), dtype=torch.float64) * 0.6 + 0.2).requires_grad_(True)
    m = torch.rand(1, 2 * 9, 5, 5, generator=torch.Generator().manual_seed(14), dtype=torch.float64).requires_grad_(True)
    fn = lambda x, off, w, m: deform_conv2d(x, off, w, 1, 1, 1, 2, 2, mask=m)
    assert torch.autograd.gradcheck(fn, (x, off, w, m), eps=1e-6, atol=1e-5)


def _known_answer():
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "dcn_known_answer.json")) as f:
        v = json.load(f)
    t = lambda k, shape=None: torch.tensor(v[k], dtype=torch.float32).reshape(shape) if shape else torch.tensor(v[k], dtype=torch.float32)
    return v, t


def run_known_answer(dcn):
    """dcn(x, offset, weight) -> output, checked against the fixture's oracle-generated values (2-3 decimals)."""
    v, t = _known_answer()
    x = t("input").requires_grad_(True)
    ow = t("offset_weight", (8, 1, 2, 2)).requires_grad_(True)
    ob = t("offset_bias").requires_grad_(True)
    dw = t("deform_weight", (1, 1, 2, 2)).requires_grad_(True)
    off = torch.nn.functional.conv2d(x, ow, ob)          # DeformConv2dPack: offsets from a plain conv on the same input
    out = dcn(x, off, dw)
    out.backward(torch.ones_like(out))
    got = dict(gt_out=out.detach(), gt_x_grad=x.grad, gt_offset_weight_grad=ow.grad.reshape(8, 1, 2, 2), gt_offset_bias_grad=ob.grad,
               gt_deform_weight_grad=dw.grad)
    for k, g in got.items():
        want = torch.tensor(v[k], dtype=torch.float32)
        assert torch.allclose(g.cpu().reshape(want.shape), want, atol=2e-3 if k in ("gt_out", "gt_x_grad") else 6e-3), (k, g, want)


def test_small_fixed_case_regression():
    run_known_answer(lambda x, off, w: deform_conv2d(x, off, w, 1, 0, 1, 1, 1))


def _deform_conv2d_via_grid_sample(x, offset, weight, stride, padding, dilation, groups, dg, mask=None):
    """The same operator written a second, independent way: every (deformable group, tap) is ONE call of ATen's
    F.grid_sample(mode='bilinear', padding_mode='zeros', align_corners=True) at the positions  base + tap * dilation + offset
    -- ATen's bilinear sampler treats each out-of-image corner as zero, which is the per-corner rule of
    deform_conv_cuda_kernel.cu:85-115 -- followed by an einsum with the weights.  No code shared with oracle/dcn_ref.py."""
    B, C, H, W = x.shape
    co, cig, kh, kw = weight.shape
    Ho = (H + 2 * padding - (dilation * (kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * padding - (dilation * (kw - 1) + 1)) // stride + 1
    cpg = C // dg
    ys = (torch.arange(Ho, dtype=x.dtype) * stride - padding).view(1, Ho, 1)
    xs = (torch.arange(Wo, dtype=x.dtype) * stride - padding).view(1, 1, Wo)
    cols = x.new_zeros(B, C, kh * kw, Ho, Wo)
    for g in range(dg):
        for i in range(kh):
            for j in range(kw):
                k = i * kw + j
                oh = offset[:, (g * kh * kw + k) * 2]
                ow = offset[:, (g * kh * kw + k) * 2 + 1]
                py, px = ys + i * dilation + oh, xs + j * dilation + ow            # [B, Ho, Wo] pixel coordinates
                grid = torch.stack([2 * px / max(W - 1, 1) - 1, 2 * py / max(H - 1, 1) - 1], dim=-1)   # (x, y) in [-1, 1], align_corners
                v = F.grid_sample(x[:, g * cpg:(g + 1) * cpg], grid, mode="bilinear", padding_mode="zeros", align_corners=True)
                if mask is not None:
                    v = v * mask[:, g * kh * kw + k].unsqueeze(1)
                cols[:, g * cpg:(g + 1) * cpg, k] = v
    cols = cols.view(B, groups, C // groups, kh * kw, Ho, Wo)
    w = weight.view(groups, co // groups, cig, kh * kw)
    return torch.einsum("bgckhw,gock->bgohw", cols, w).reshape(B, co, Ho, Wo)


@pytest.mark.parametrize("stride,padding,dilation,groups,dg,modulated", [(1, 1, 1, 1, 4, False), (2, 1, 1, 1, 2, False), (1, 2, 2, 2, 2, True),
                                                                          (1, 1, 1, 1, 1, True)])
def test_dcn_ref_agrees_with_an_independent_grid_sample_formulation(stride, padding, dilation, groups, dg, modulated):
    """A second opinion on the unpinned oracle (the reference ships no vectors and its CUDA does not build here): ATen's own bilinear
    sampler, a different code path by different authors, gives the same forward values and the same gradients with respect to input,
    offsets, mask and weights -- offsets up to +-2.5 px so that samples leave the image on every side."""
    torch.manual_seed(3)
    B, C, H, W, co, k = 2, 8, 7, 9, 6, 3
    x = torch.randn(B, C, H, W, dtype=torch.float64, requires_grad=True)
    Ho = (H + 2 * padding - (dilation * (k - 1) + 1)) // stride + 1
    Wo = (W + 2 * padding - (dilation * (k - 1) + 1)) // stride + 1
    off = (torch.randn(B, dg * 2 * k * k, Ho, Wo, dtype=torch.float64) * 1.3).requires_grad_(True)
    msk = torch.rand(B, dg * k * k, Ho, Wo, dtype=torch.float64).requires_grad_(True) if modulated else None
    wt = torch.randn(co, C // groups, k, k, dtype=torch.float64, requires_grad=True)
    a = deform_conv2d(x, off, wt, stride, padding, dilation, groups, dg, mask=msk)
    b = _deform_conv2d_via_grid_sample(x, off, wt, stride, padding, dilation, groups, dg, mask=msk)
    assert a.shape == b.shape and torch.allclose(a, b, rtol=1e-10, atol=1e-10), float((a - b).abs().max())
    gy = torch.randn_like(a)
    wrt = [t for t in (x, off, wt, msk) if t is not None]
    ga = torch.autograd.grad(a, wrt, gy, retain_graph=True)
    gb = torch.autograd.grad(b, wrt, gy)
    for u, v, name in zip(ga, gb, ("input", "offset", "weight", "mask")):
        # offsets that land exactly on an integer coordinate have a one-sided derivative: none do with random fp64 offsets
        assert torch.allclose(u, v, rtol=1e-8, atol=1e-9), (name, float((u - v).abs().max()))
