"""Drop-in for the reference's pybind module `deform_conv_cuda` (det3d/ops/dcn/src/deform_conv_cuda.cpp:687-701): the same
five function names, positional argument order and return conventions, on the C ABI of include/rtp.h section D.

    det3d/ops/dcn/deform_conv.py:11   `from . import deform_conv_cuda`
can point at this module unchanged -- copy it next to deform_conv.py (or alias it in sys.modules, which is what
rt_pose_amd.registry.install_det3d_shim() does for `det3d.ops.dcn.deform_conv_cuda`).

Conventions kept from the reference (SURVEY.md 8b "Native op ABI"): W-before-H argument order (kW, kH, dW, dH, ...); the
caller allocates `output` and the zero-filled gradients and the callee writes in place; `columns` / `ones` are accepted and
ignored (the reference re-allocates them inside every call; here the scratch is an explicit workspace sized by
rtp_dcn_workspace_bytes); the three DCNv1 functions return 1, the modulated ones None; a failing call raises RuntimeError
(AT_CHECK / AT_ERROR in the reference); work is queued on the current stream and never synchronises.
Tensors are contiguous CUDA tensors; fp32 runs natively, fp16 / bf16 / fp64 (the reference dispatches
AT_DISPATCH_FLOATING_TYPES_AND_HALF, deform_conv_cuda_kernel.cu:259,353,451) are converted to fp32 around the call and the
results cast back into the caller's tensors.
"""
import ctypes as C

import torch

from . import _lib


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _ws(x, weight, ho, wo, step):
    nbytes = _lib.load().rtp_dcn_workspace_bytes(step, x.size(1), x.size(2), x.size(3), weight.size(0), weight.size(2),
                                                 weight.size(3), ho, wo)
    return torch.empty(max(1, nbytes // 4), dtype=torch.float32, device=x.device)


def _f32(*ts):
    """fp32 contiguous views/copies of the inputs (fp32 contiguous tensors pass through untouched)."""
    out = []
    for t in ts:
        if t is None:
            out.append(None)
            continue
        if not t.is_cuda:
            raise RuntimeError("deform_conv_cuda: expected a CUDA tensor")   # the reference's AT_CHECK(input.is_cuda())
        out.append(t.contiguous() if t.dtype == torch.float32 else t.float().contiguous())
    return out


def _back(dst, src):
    """Write a fp32 result into the caller's tensor when that one is not the fp32 buffer the kernel wrote."""
    if dst is not None and dst.data_ptr() != src.data_ptr():
        dst.copy_(src.to(dst.dtype))


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s (%d)" % (what, _lib._ERR.get(rc, "?"), rc))


def deform_conv_forward_cuda(input, weight, offset, output, columns, ones, kW, kH, dW, dH, padW, padH, dilationW, dilationH,
                             group, deformable_group, im2col_step):
    """deform_conv_cuda.cpp:152-260."""
    x, w, off, out = _f32(input, weight, offset, output)
    n, c, h, wd = x.shape
    rc = _lib.load().rtp_deform_conv_forward(_p(x), _p(w), _p(off), _p(out), _p(_ws(x, w, out.size(2), out.size(3), im2col_step)),
                                             n, c, h, wd, w.size(0), kW, kH, dW, dH, padW, padH, dilationW, dilationH, group,
                                             deformable_group, im2col_step, _stream(x))
    _check(rc, "deform_conv_forward_cuda")
    _back(output, out)
    return 1


def deform_conv_backward_input_cuda(input, offset, gradOutput, gradInput, gradOffset, weight, columns, kW, kH, dW, dH, padW, padH,
                                    dilationW, dilationH, group, deformable_group, im2col_step):
    """deform_conv_cuda.cpp:262-374."""
    x, off, go, gi, goff, w = _f32(input, offset, gradOutput, gradInput, gradOffset, weight)
    n, c, h, wd = x.shape
    rc = _lib.load().rtp_deform_conv_backward_input(_p(x), _p(off), _p(go), _p(gi), _p(goff), _p(w),
                                                    _p(_ws(x, w, go.size(2), go.size(3), im2col_step)), n, c, h, wd, w.size(0), kW,
                                                    kH, dW, dH, padW, padH, dilationW, dilationH, group, deformable_group,
                                                    im2col_step, _stream(x))
    _check(rc, "deform_conv_backward_input_cuda")
    _back(gradInput, gi)
    _back(gradOffset, goff)
    return 1


def deform_conv_backward_parameters_cuda(input, offset, gradOutput, gradWeight, columns, ones, kW, kH, dW, dH, padW, padH,
                                         dilationW, dilationH, group, deformable_group, scale, im2col_step):
    """deform_conv_cuda.cpp:376-488 (gradWeight += scale * ...)."""
    x, off, go, gw = _f32(input, offset, gradOutput, gradWeight)
    n, c, h, wd = x.shape
    rc = _lib.load().rtp_deform_conv_backward_parameters(_p(x), _p(off), _p(go), _p(gw),
                                                         _p(_ws(x, gw, go.size(2), go.size(3), im2col_step)), n, c, h, wd,
                                                         gw.size(0), kW, kH, dW, dH, padW, padH, dilationW, dilationH, group,
                                                         deformable_group, float(scale), im2col_step, _stream(x))
    _check(rc, "deform_conv_backward_parameters_cuda")
    _back(gradWeight, gw)
    return 1


def modulated_deform_conv_cuda_forward(input, weight, bias, ones, offset, mask, output, columns, kernel_h, kernel_w, stride_h,
                                       stride_w, pad_h, pad_w, dilation_h, dilation_w, group, deformable_group, with_bias):
    """deform_conv_cuda.cpp:490-569."""
    x, w, b, off, m, out = _f32(input, weight, bias, offset, mask, output)
    n, c, h, wd = x.shape
    rc = _lib.load().rtp_modulated_deform_conv_forward(_p(x), _p(w), _p(b), _p(off), _p(m), _p(out),
                                                       _p(_ws(x, w, out.size(2), out.size(3), 1)), n, c, h, wd, w.size(0), kernel_h,
                                                       kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group,
                                                       deformable_group, int(bool(with_bias)), _stream(x))
    _check(rc, "modulated_deform_conv_cuda_forward")
    _back(output, out)


def modulated_deform_conv_cuda_backward(input, weight, bias, ones, offset, mask, columns, grad_input, grad_weight, grad_bias,
                                        grad_offset, grad_mask, grad_output, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w,
                                        dilation_h, dilation_w, group, deformable_group, with_bias):
    """deform_conv_cuda.cpp:571-685."""
    x, w, b, off, m, gi, gw, gb, goff, gm, go = _f32(input, weight, bias, offset, mask, grad_input, grad_weight, grad_bias,
                                                      grad_offset, grad_mask, grad_output)
    n, c, h, wd = x.shape
    rc = _lib.load().rtp_modulated_deform_conv_backward(_p(x), _p(w), _p(b), _p(off), _p(m), _p(gi), _p(gw), _p(gb), _p(goff),
                                                        _p(gm), _p(go), _p(_ws(x, w, go.size(2), go.size(3), 1)), n, c, h, wd,
                                                        w.size(0), kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h,
                                                        dilation_w, group, deformable_group, int(bool(with_bias)), _stream(x))
    _check(rc, "modulated_deform_conv_cuda_backward")
    for dst, src in ((grad_input, gi), (grad_weight, gw), (grad_bias, gb), (grad_offset, goff), (grad_mask, gm)):
        _back(dst, src)
