"""Deformable convolution operators on the HIP kernels -- the Python face of det3d/ops/dcn/deform_conv.py.

Same names, argument meaning and error behaviour as the reference wrapper (deform_conv.py:14-112, 115-190, 192-255,
323-350): DeformConvFunction / ModulatedDeformConvFunction (autograd), deform_conv / modulated_deform_conv,
DeformConv, DeformConvPack, ModulatedDeformConv, ModulatedDeformConvPack.  The five native calls go through the C ABI
of include/rtp.h section D instead of the pybind module `deform_conv_cuda`.

Error behaviour kept: non-4-D input -> ValueError; CPU tensors -> NotImplementedError (there is no CPU path here
either); batch not divisible by im2col_step -> AssertionError('im2col step must divide batchsize'); a failing native
call -> RuntimeError.  The kernels compute in fp32; fp16 / bf16 / fp64 tensors (the reference dispatches fp64/fp32/fp16) are
converted around the native call and results come back in the caller's dtype.
"""
import ctypes as C
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair, _single

from . import _lib


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _workspace(x, weight, out_hw, step):
    lib = _lib.load()
    n, c, h, w = x.shape
    nbytes = lib.rtp_dcn_workspace_bytes(step, c, h, w, weight.shape[0], weight.shape[2], weight.shape[3], out_hw[0], out_hw[1])
    return torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)


_FLOATS = (torch.float32, torch.float16, torch.bfloat16, torch.float64)


def _check(x, *others):
    if not x.is_cuda:
        raise NotImplementedError
    for t in (x, *others):
        if t is not None and t.dtype not in _FLOATS:
            raise NotImplementedError("rt_pose_amd.dcn: floating-point tensors only (got %s)" % t.dtype)


def _conv_out(size, kernel, stride, padding, dilation):
    """Output extent of one convolution axis."""
    return (size + 2 * padding - dilation * (kernel - 1) - 1) // stride + 1


def _uniform_fan_in_(weight, in_channels, kernel_size):
    """U(-1/sqrt(fan), 1/sqrt(fan)) with fan = in_channels * prod(kernel) -- the reference's initialisation of both DeformConv and
    ModulatedDeformConv weights (deform_conv.py:221-227, 355-363; note: in_channels, not in_channels / groups)."""
    bound = 1.0 / math.sqrt(in_channels * math.prod(kernel_size))
    with torch.no_grad():
        weight.uniform_(-bound, bound)


def _f32(t):
    """The kernels compute in fp32; half / bfloat16 / double tensors (the reference dispatches fp64/fp32/fp16,
    deform_conv_cuda_kernel.cu:259) are converted around the call and results returned in the caller's dtype."""
    return t if t is None or t.dtype == torch.float32 else t.float()


class DeformConvFunction(Function):
    @staticmethod
    def forward(ctx, input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1, im2col_step=64):
        if input is not None and input.dim() != 4:
            raise ValueError("Expected 4D tensor as input, got {}D tensor instead.".format(input.dim()))
        ctx.stride, ctx.padding, ctx.dilation = _pair(stride), _pair(padding), _pair(dilation)
        ctx.groups, ctx.deformable_groups, ctx.im2col_step = groups, deformable_groups, im2col_step
        ctx.save_for_backward(input, offset, weight)
        _check(input, offset, weight)
        out_size = DeformConvFunction._output_size(input, weight, ctx.padding, ctx.dilation, ctx.stride)
        out_dtype = input.dtype
        step = min(ctx.im2col_step, input.shape[0])
        assert (input.shape[0] % step) == 0, "im2col step must divide batchsize"
        input, offset, weight = _f32(input).contiguous(), _f32(offset).contiguous(), _f32(weight).contiguous()
        output = input.new_empty(out_size)
        ws = _workspace(input, weight, out_size[2:], step)
        n, c, h, w = input.shape
        rc = _lib.load().rtp_deform_conv_forward(
            _p(input), _p(weight), _p(offset), _p(output), _p(ws), n, c, h, w, weight.size(0), weight.size(3), weight.size(2),
            ctx.stride[1], ctx.stride[0], ctx.padding[1], ctx.padding[0], ctx.dilation[1], ctx.dilation[0], ctx.groups,
            ctx.deformable_groups, step, _stream(input))
        if rc != 0:
            raise RuntimeError("rtp_deform_conv_forward failed (%d)" % rc)
        return output if out_dtype == torch.float32 else output.to(out_dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input, offset, weight = ctx.saved_tensors
        grad_input = grad_offset = grad_weight = None
        if not grad_output.is_cuda:
            raise NotImplementedError
        step = min(ctx.im2col_step, input.shape[0])
        assert (input.shape[0] % step) == 0, "im2col step must divide batchsize"
        lib = _lib.load()
        dts = (input.dtype, offset.dtype, weight.dtype)
        input, offset, weight, grad_output = (_f32(input).contiguous(), _f32(offset).contiguous(), _f32(weight).contiguous(),
                                              _f32(grad_output).contiguous())
        ws = _workspace(input, weight, grad_output.shape[2:], step)
        n, c, h, w = input.shape
        geo = (n, c, h, w, weight.size(0), weight.size(3), weight.size(2), ctx.stride[1], ctx.stride[0], ctx.padding[1],
               ctx.padding[0], ctx.dilation[1], ctx.dilation[0], ctx.groups, ctx.deformable_groups)
        if (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) and ctx.needs_input_grad[2]:
            # fresh buffers: the overwriting entry needs no zeros (deform_conv.py:75-76 allocates zeros and accumulates)
            grad_input, grad_offset, grad_weight = torch.empty_like(input), torch.empty_like(offset), torch.empty_like(weight)
            rc = lib.rtp_deform_conv_backward_overwrite(_p(input), _p(offset), _p(grad_output), _p(grad_input), _p(grad_offset),
                                                        _p(weight), _p(grad_weight), _p(ws), *geo, 1.0, step, _stream(input))
            if rc != 0:
                raise RuntimeError("rtp_deform_conv_backward_overwrite failed (%d)" % rc)
            cast = lambda g, dt: g if g is None or g.dtype == dt else g.to(dt)
            return (cast(grad_input, dts[0]), cast(grad_offset, dts[1]), cast(grad_weight, dts[2]), None, None, None, None, None, None)
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            grad_input = torch.zeros_like(input)
            grad_offset = torch.zeros_like(offset)
            rc = lib.rtp_deform_conv_backward_input(_p(input), _p(offset), _p(grad_output), _p(grad_input), _p(grad_offset),
                                                    _p(weight), _p(ws), *geo, step, _stream(input))
            if rc != 0:
                raise RuntimeError("rtp_deform_conv_backward_input failed (%d)" % rc)
        if ctx.needs_input_grad[2]:
            grad_weight = torch.zeros_like(weight)
            rc = lib.rtp_deform_conv_backward_parameters(_p(input), _p(offset), _p(grad_output), _p(grad_weight), _p(ws), *geo,
                                                         1.0, step, _stream(input))
            if rc != 0:
                raise RuntimeError("rtp_deform_conv_backward_parameters failed (%d)" % rc)
        cast = lambda g, dt: g if g is None or g.dtype == dt else g.to(dt)
        return (cast(grad_input, dts[0]), cast(grad_offset, dts[1]), cast(grad_weight, dts[2]), None, None, None, None, None, None)

    @staticmethod
    def _output_size(input, weight, padding, dilation, stride):
        """(N, Cout, Ho, Wo) of the convolution (deform_conv.py:97-112); ValueError when a spatial size would not be positive."""
        spatial = tuple(_conv_out(input.shape[2 + i], weight.shape[2 + i], stride[i], padding[i], dilation[i])
                        for i in range(input.dim() - 2))
        if min(spatial, default=1) <= 0:
            raise ValueError("convolution input is too small (output would be %s)"
                             % "x".join(str(v) for v in (input.shape[0], weight.shape[0]) + spatial))
        return (input.shape[0], weight.shape[0]) + spatial


class ModulatedDeformConvFunction(Function):
    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1):
        ctx.stride, ctx.padding, ctx.dilation = stride, padding, dilation
        ctx.groups, ctx.deformable_groups = groups, deformable_groups
        ctx.with_bias = bias is not None
        if not ctx.with_bias:
            bias = input.new_empty(1)  # fake tensor
        _check(input, offset, mask, weight)
        if weight.requires_grad or mask.requires_grad or offset.requires_grad or input.requires_grad:
            ctx.save_for_backward(input, offset, mask, weight, bias)
        out_shape = ModulatedDeformConvFunction._infer_shape(ctx, input, weight)
        out_dtype = input.dtype
        input, offset, mask, weight, bias = (_f32(input).contiguous(), _f32(offset).contiguous(), _f32(mask).contiguous(),
                                             _f32(weight).contiguous(), _f32(bias).contiguous())
        output = input.new_empty(out_shape)
        ws = _workspace(input, weight, out_shape[2:], 1)
        n, c, h, w = input.shape
        rc = _lib.load().rtp_modulated_deform_conv_forward(
            _p(input), _p(weight), _p(bias), _p(offset), _p(mask), _p(output), _p(ws), n, c, h, w, weight.shape[0],
            weight.shape[2], weight.shape[3], stride, stride, padding, padding, dilation, dilation, groups, deformable_groups,
            int(ctx.with_bias), _stream(input))
        if rc != 0:
            raise RuntimeError("rtp_modulated_deform_conv_forward failed (%d)" % rc)
        return output if out_dtype == torch.float32 else output.to(out_dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        if not grad_output.is_cuda:
            raise NotImplementedError
        input, offset, mask, weight, bias = ctx.saved_tensors
        dts = (input.dtype, offset.dtype, mask.dtype, weight.dtype, bias.dtype)
        input, offset, mask, weight, bias = (_f32(input).contiguous(), _f32(offset).contiguous(), _f32(mask).contiguous(),
                                             _f32(weight).contiguous(), _f32(bias).contiguous())
        grad_input, grad_offset = torch.zeros_like(input), torch.zeros_like(offset)
        grad_mask, grad_weight, grad_bias = torch.zeros_like(mask), torch.zeros_like(weight), torch.zeros_like(bias)
        grad_output = _f32(grad_output).contiguous()
        ws = _workspace(input, weight, grad_output.shape[2:], 1)
        n, c, h, w = input.shape
        rc = _lib.load().rtp_modulated_deform_conv_backward(
            _p(input), _p(weight), _p(bias), _p(offset), _p(mask), _p(grad_input), _p(grad_weight), _p(grad_bias),
            _p(grad_offset), _p(grad_mask), _p(grad_output), _p(ws), n, c, h, w, weight.shape[0], weight.shape[2],
            weight.shape[3], ctx.stride, ctx.stride, ctx.padding, ctx.padding, ctx.dilation, ctx.dilation, ctx.groups,
            ctx.deformable_groups, int(ctx.with_bias), _stream(input))
        if rc != 0:
            raise RuntimeError("rtp_modulated_deform_conv_backward failed (%d)" % rc)
        if not ctx.with_bias:
            grad_bias = None
        cast = lambda g, dt: g if g is None or g.dtype == dt else g.to(dt)
        return (cast(grad_input, dts[0]), cast(grad_offset, dts[1]), cast(grad_mask, dts[2]), cast(grad_weight, dts[3]),
                cast(grad_bias, dts[4]), None, None, None, None, None)

    @staticmethod
    def _infer_shape(ctx, input, weight):
        """(N, Cout, Ho, Wo) for the scalar stride / padding / dilation this function takes (deform_conv.py:176-189)."""
        ho, wo = (_conv_out(input.shape[2 + i], weight.shape[2 + i], ctx.stride, ctx.padding, ctx.dilation) for i in (0, 1))
        return input.shape[0], weight.shape[0], ho, wo


deform_conv = DeformConvFunction.apply
modulated_deform_conv = ModulatedDeformConvFunction.apply


class DeformConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=False):
        super().__init__()
        assert not bias
        assert in_channels % groups == 0, "in_channels {} cannot be divisible by groups {}".format(in_channels, groups)
        assert out_channels % groups == 0, "out_channels {} cannot be divisible by groups {}".format(out_channels, groups)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        self.groups, self.deformable_groups = groups, deformable_groups
        self.transposed, self.output_padding = False, _single(0)
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // self.groups, *self.kernel_size))
        self.reset_parameters()

    def reset_parameters(self):
        _uniform_fan_in_(self.weight, self.in_channels, self.kernel_size)

    def forward(self, x, offset):
        # an input smaller than the kernel is zero-padded up to it (bottom / right), the offsets likewise, and the output cropped
        # back by the same amounts (deform_conv.py:230-247)
        grow = [max(k - s, 0) for k, s in zip(self.kernel_size, x.shape[2:4])]
        if any(grow):
            x = F.pad(x, (0, grow[1], 0, grow[0])).contiguous()
            offset = F.pad(offset, (0, grow[1], 0, grow[0])).contiguous()
        out = deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups, self.deformable_groups)
        if any(grow):
            out = out[:, :, :out.shape[2] - grow[0], :out.shape[3] - grow[1]].contiguous()
        return out


class DeformConvPack(DeformConv):
    _version = 2

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.conv_offset = nn.Conv2d(self.in_channels, self.deformable_groups * 2 * self.kernel_size[0] * self.kernel_size[1],
                                     kernel_size=self.kernel_size, stride=_pair(self.stride), padding=_pair(self.padding), bias=True)
        self.conv_offset.weight.data.zero_()
        self.conv_offset.bias.data.zero_()

    def forward(self, x):
        return deform_conv(x, self.conv_offset(x), self.weight, self.stride, self.padding, self.dilation, self.groups,
                           self.deformable_groups)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        """Checkpoints written before version 2 name the offset conv '<name>_offset.*' instead of '<name>.conv_offset.*'
        (deform_conv.py:298-321 / 420-444): move those keys, then load as usual."""
        version = local_metadata.get("version", None)
        if version is None or version < 2:
            for leaf in ("weight", "bias"):
                new_key, old_key = prefix + "conv_offset." + leaf, prefix[:-1] + "_offset." + leaf
                if new_key not in state_dict and old_key in state_dict:
                    state_dict[new_key] = state_dict.pop(old_key)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)


class ModulatedDeformConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.groups, self.deformable_groups, self.with_bias = groups, deformable_groups, bias
        self.transposed, self.output_padding = False, _single(0)
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // groups, *self.kernel_size))
        if bias:
            self.bias = nn.Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        _uniform_fan_in_(self.weight, self.in_channels, self.kernel_size)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def forward(self, x, offset, mask):
        return modulated_deform_conv(x, offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation,
                                     self.groups, self.deformable_groups)


class ModulatedDeformConvPack(ModulatedDeformConv):
    _version = 2

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.conv_offset = nn.Conv2d(self.in_channels, self.deformable_groups * 3 * self.kernel_size[0] * self.kernel_size[1],
                                     kernel_size=self.kernel_size, stride=_pair(self.stride), padding=_pair(self.padding), bias=True)
        self.conv_offset.weight.data.zero_()
        self.conv_offset.bias.data.zero_()

    def forward(self, x):
        # conv_offset's 3 * dg * kh * kw channels: the first two thirds are the offsets (in place, the channel order the operator
        # expects), the last third the modulation logits (deform_conv.py:409-416)
        raw = self.conv_offset(x)
        k = 2 * raw.shape[1] // 3
        return modulated_deform_conv(x, raw[:, :k].contiguous(), raw[:, k:].sigmoid(), self.weight, self.bias, self.stride,
                                     self.padding, self.dilation, self.groups, self.deformable_groups)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        """Checkpoints written before version 2 name the offset conv '<name>_offset.*' instead of '<name>.conv_offset.*'
        (deform_conv.py:298-321 / 420-444): move those keys, then load as usual."""
        version = local_metadata.get("version", None)
        if version is None or version < 2:
            for leaf in ("weight", "bias"):
                new_key, old_key = prefix + "conv_offset." + leaf, prefix[:-1] + "_offset." + leaf
                if new_key not in state_dict and old_key in state_dict:
                    state_dict[new_key] = state_dict.pop(old_key)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)


def migrate_pre_v2_keys(state_dict, module=None, target_keys=None):
    """Rename '<name>_offset.{weight,bias}' to '<name>.conv_offset.{weight,bias}' in a whole checkpoint state dict.

    The reference does this inside DeformConvPack._load_from_state_dict (deform_conv.py:298-321, 420-444), which current
    PyTorch never reaches with such a key: Module.load_state_dict hands a child only the keys under its own prefix
    ('conv2.'), and 'conv2_offset.weight' is not one of them.  Call this on the dict before load_state_dict instead
    (rt_pose_amd.checkpoint.load_checkpoint does).  With `module` given, only names that are Pack modules are touched;
    with `target_keys` (the destination's parameter names), only renames whose NEW key the destination has.
    Current-format keys -- '<pack>.conv_offset.weight', which also END in '_offset.weight' -- are never touched."""
    packs = None
    if module is not None:
        packs = {n for n, m in module.named_modules() if isinstance(m, (DeformConvPack, ModulatedDeformConvPack))}
    for k in list(state_dict.keys()):
        for leaf in ("weight", "bias"):
            tail = "_offset." + leaf
            if k.endswith(tail) and not k.endswith(".conv_offset." + leaf):
                name = k[:-len(tail)]
                new = name + ".conv_offset." + leaf
                if (new not in state_dict and (packs is None or name in packs)
                        and (target_keys is None or (new in target_keys and k not in target_keys))):
                    state_dict[new] = state_dict.pop(k)
    return state_dict
