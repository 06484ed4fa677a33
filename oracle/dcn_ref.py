"""CPU restatement of the reference's deformable convolution (DCNv1 / modulated DCNv2) in differentiable PyTorch.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED by the reference: it ships no tests for this operator
and its CUDA sources (det3d/ops/dcn/src/*.cu, THC headers, nvcc) cannot be built or imported in this image, so this
file restates the published algorithm from the reference's own kernels and is held only by property tests
(tests/test_dcn_oracle.py: zero offsets == F.conv2d, integer offsets == shifted conv, out-of-window taps == 0,
mask == 1 reduces v2 to v1, fp64 gradcheck) and a small fixed case whose expected values this file itself produced
(tests/golden/dcn_known_answer.json -- not an external pin; see its provenance).  Round 6 adds a second opinion that shares no
code with this file: the same operator written on ATen's F.grid_sample (bilinear, zeros padding, align_corners) agrees in values and
in every gradient to 1e-10 (tests/test_dcn_oracle.py::test_dcn_ref_agrees_with_an_independent_grid_sample_formulation) -- a
cross-check by a third party's sampler, still not a pin by the reference.

Restated from det3d/ops/dcn/src/deform_conv_cuda_kernel.cu:
  bilinear sample with per-corner bounds ................. :85-115  (dmcn_im2col_bilinear :467-495)
  im2col indexing, offset channel order (dh,dw per tap),
  validity window  -1 < h < H, -1 < w < W ................ :190-243 (modulated :570-633)
  gradient weights (== d bilinear / d input, d coords) .... :117-188 -- obtained here by autograd of the same function
and det3d/ops/dcn/src/deform_conv_cuda.cpp:196-246 (group GEMM; im2col_step only batches the same arithmetic).
"""
import torch


def _bilinear(img, h, w):
    """img [C,H,W]; h,w [...] float sample coords.  Per-corner zero outside, and the whole sample is zero outside the
    open window (-1,H)x(-1,W) (kernel.cu:230-238)."""
    C, H, W = img.shape
    inside = (h > -1) & (w > -1) & (h < H) & (w < W)
    h_low = torch.floor(h)
    w_low = torch.floor(w)
    lh, lw = h - h_low, w - w_low
    hh, hw = 1 - lh, 1 - lw
    h_low, w_low = h_low.long(), w_low.long()
    h_high, w_high = h_low + 1, w_low + 1

    def corner(hi, wi, ok):
        ok = ok & inside
        v = img[:, hi.clamp(0, H - 1), wi.clamp(0, W - 1)]
        return v * ok.to(img.dtype)

    v1 = corner(h_low, w_low, (h_low >= 0) & (w_low >= 0))
    v2 = corner(h_low, w_high, (h_low >= 0) & (w_high <= W - 1))
    v3 = corner(h_high, w_low, (h_high <= H - 1) & (w_low >= 0))
    v4 = corner(h_high, w_high, (h_high <= H - 1) & (w_high <= W - 1))
    return hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4


def deform_im2col(x, offset, kh, kw, stride, padding, dilation, deformable_groups, mask=None):
    """x [B,C,H,W]; offset [B, dg*2*kh*kw, Ho, Wo] ordered (dh,dw) per tap; mask [B, dg*kh*kw, Ho, Wo] or None.
    Returns columns [B, C, kh*kw, Ho, Wo]."""
    B, C, H, W = x.shape
    sh, sw = stride
    ph, pw = padding
    dh, dw = dilation
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    cpg = C // deformable_groups
    hs = (torch.arange(Ho, dtype=x.dtype) * sh - ph).view(Ho, 1)
    ws = (torch.arange(Wo, dtype=x.dtype) * sw - pw).view(1, Wo)
    cols = []
    for b in range(B):
        per_g = []
        for g in range(deformable_groups):
            taps = []
            for i in range(kh):
                for j in range(kw):
                    t = i * kw + j
                    oh = offset[b, g * 2 * kh * kw + 2 * t]
                    ow = offset[b, g * 2 * kh * kw + 2 * t + 1]
                    v = _bilinear(x[b, g * cpg:(g + 1) * cpg], hs + i * dh + oh, ws + j * dw + ow)
                    if mask is not None:
                        v = v * mask[b, g * kh * kw + t]
                    taps.append(v)
            per_g.append(torch.stack(taps, 1))  # [cpg, kh*kw, Ho, Wo]
        cols.append(torch.cat(per_g, 0))
    return torch.stack(cols)


def deform_conv2d(x, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1, mask=None, bias=None):
    """DeformConvFunction.forward / ModulatedDeformConvFunction.forward (det3d/ops/dcn/deform_conv.py:16-59, 121-150)."""
    pair = lambda v: (v, v) if isinstance(v, int) else tuple(v)
    stride, padding, dilation = pair(stride), pair(padding), pair(dilation)
    B, C, H, W = x.shape
    Co, Cg, kh, kw = weight.shape
    cols = deform_im2col(x, offset, kh, kw, stride, padding, dilation, deformable_groups, mask)
    Ho, Wo = cols.shape[-2:]
    cols = cols.view(B, groups, Cg * kh * kw, Ho * Wo)
    wg = weight.view(groups, Co // groups, Cg * kh * kw)
    out = torch.einsum("gok,bgkp->bgop", wg, cols).reshape(B, Co, Ho, Wo)
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out
