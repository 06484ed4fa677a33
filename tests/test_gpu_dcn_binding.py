"""The five reference-named native functions (rt_pose_amd/deform_conv_cuda.py = the reference's pybind module
`deform_conv_cuda`, det3d/ops/dcn/src/deform_conv_cuda.cpp:687-701) called the way det3d/ops/dcn/deform_conv.py calls them
(:52-58, 77-93, 145-149, 163-168), against the oracle (oracle/dcn_ref.py -- parity unpinned by the reference, see its
header), plus the dtypes the reference dispatches (fp64 / fp32 / fp16; bf16 as this build's compute dtype)."""
import pytest
import torch

from oracle import dcn_ref as R
from rt_pose_amd import dcn, deform_conv_cuda as M
from tests.util import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def case(seed, n=4, c=8, h=12, w=20, co=6, dg=2, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, c, h, w, generator=g)
    off = torch.randn(n, dg * 18, h, w, generator=g) * 0.7
    mask = torch.rand(n, dg * 9, h, w, generator=g)
    wt = torch.randn(co, c, 3, 3, generator=g) * 0.2
    b = torch.randn(co, generator=g)
    gy = torch.randn(n, co, h, w, generator=g)
    return [t.to(DEV).to(dtype) for t in (x, off, mask, wt, b, gy)]


def test_v1_three_functions_like_the_reference_wrapper():
    x, off, _, wt, _, gy = case(1)
    n, dg, step = x.shape[0], 2, 2
    empty = x.new_empty(0)
    out = x.new_empty(n, wt.shape[0], x.shape[2], x.shape[3])
    assert M.deform_conv_forward_cuda(x, wt, off, out, empty, empty, 3, 3, 1, 1, 1, 1, 1, 1, 1, dg, step) == 1
    ref = R.deform_conv2d(x.cpu().double(), off.cpu().double(), wt.cpu().double(), 1, 1, 1, 1, dg)
    assert rel_err(out.cpu(), ref) < 1e-5
    gi, goff = torch.zeros_like(x), torch.zeros_like(off)
    assert M.deform_conv_backward_input_cuda(x, off, gy, gi, goff, wt, empty, 3, 3, 1, 1, 1, 1, 1, 1, 1, dg, step) == 1
    gw = torch.zeros_like(wt)
    assert M.deform_conv_backward_parameters_cuda(x, off, gy, gw, empty, empty, 3, 3, 1, 1, 1, 1, 1, 1, 1, dg, 1.0, step) == 1
    xs = [t.cpu().double().requires_grad_(True) for t in (x, off, wt)]
    R.deform_conv2d(xs[0], xs[1], xs[2], 1, 1, 1, 1, dg).backward(gy.cpu().double())
    for got, want, nm in ((gi, xs[0].grad, "grad_input"), (goff, xs[1].grad, "grad_offset"), (gw, xs[2].grad, "grad_weight")):
        assert rel_err(got.cpu(), want) < 2e-5, nm
    # gradWeight accumulates scale * ... (deform_conv_cuda.cpp:460-466)
    assert M.deform_conv_backward_parameters_cuda(x, off, gy, gw, empty, empty, 3, 3, 1, 1, 1, 1, 1, 1, 1, dg, 0.5, step) == 1
    assert rel_err(gw.cpu(), 1.5 * xs[2].grad) < 2e-5


def test_modulated_two_functions_like_the_reference_wrapper():
    x, off, mask, wt, b, gy = case(2)
    dg = 2
    empty = x.new_empty(0)
    out = x.new_empty(x.shape[0], wt.shape[0], x.shape[2], x.shape[3])
    assert M.modulated_deform_conv_cuda_forward(x, wt, b, empty, off, mask, out, empty, 3, 3, 1, 1, 1, 1, 1, 1, 1, dg, True) is None
    xs = [t.cpu().double().requires_grad_(True) for t in (x, off, mask, wt, b)]
    ref = R.deform_conv2d(xs[0], xs[1], xs[3], 1, 1, 1, 1, dg, mask=xs[2], bias=xs[4])
    assert rel_err(out.cpu(), ref.detach()) < 1e-5
    gi, gw, gb, goff, gm = [torch.zeros_like(t) for t in (x, wt, b, off, mask)]
    M.modulated_deform_conv_cuda_backward(x, wt, b, empty, off, mask, empty, gi, gw, gb, goff, gm, gy, 3, 3, 1, 1, 1, 1, 1, 1, 1, dg, True)
    ref.backward(gy.cpu().double())
    for got, want, nm in ((gi, xs[0].grad, "input"), (goff, xs[1].grad, "offset"), (gm, xs[2].grad, "mask"), (gw, xs[3].grad, "weight"),
                          (gb, xs[4].grad, "bias")):
        assert rel_err(got.cpu(), want) < 2e-5, nm


def test_errors_surface_as_runtime_error_like_at_check():
    x, off, _, wt, _, _ = case(3)
    out = x.new_empty(x.shape[0], wt.shape[0], x.shape[2], x.shape[3])
    with pytest.raises(RuntimeError):   # CPU tensor: AT_CHECK(input.is_cuda()) in the reference
        M.deform_conv_forward_cuda(x.cpu(), wt, off, out, None, None, 3, 3, 1, 1, 1, 1, 1, 1, 1, 2, 2)
    with pytest.raises(RuntimeError):   # batch not divisible by im2col_step (shape_check, deform_conv_cuda.cpp:62-150)
        M.deform_conv_forward_cuda(x, wt, off, out, None, None, 3, 3, 1, 1, 1, 1, 1, 1, 1, 2, 3)


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 4e-3), (torch.bfloat16, 3e-2), (torch.float64, 1e-5)])
def test_other_floating_dtypes_through_wrapper_and_binding(dtype, tol):
    x, off, mask, wt, b, gy = case(4, dtype=dtype)
    # autograd wrapper: outputs and gradients come back in the caller's dtype
    xs = [t.clone().requires_grad_(True) for t in (x, off, wt)]
    y = dcn.deform_conv(xs[0], xs[1], xs[2], 1, 1, 1, 1, 2, 2)
    assert y.dtype == dtype
    y.backward(gy)
    assert all(t.grad.dtype == dtype for t in xs)
    ref_in = [t.detach().cpu().double().requires_grad_(True) for t in (x, off, wt)]
    ref = R.deform_conv2d(ref_in[0], ref_in[1], ref_in[2], 1, 1, 1, 1, 2)
    ref.backward(gy.cpu().double())
    assert rel_err(y.detach().cpu(), ref.detach()) < tol
    for t, r in zip(xs, ref_in):
        assert rel_err(t.grad.cpu(), r.grad) < 2 * tol
    # binding: the caller's tensors are written in place, in their dtype
    out = x.new_empty(x.shape[0], wt.shape[0], x.shape[2], x.shape[3])
    M.deform_conv_forward_cuda(x, wt, off, out, None, None, 3, 3, 1, 1, 1, 1, 1, 1, 1, 2, 2)
    assert out.dtype == dtype and rel_err(out.cpu(), ref.detach()) < tol
    ym = dcn.modulated_deform_conv(x, off, mask, wt, b, 1, 1, 1, 1, 2)
    assert ym.dtype == dtype and torch.isfinite(ym.float()).all()
