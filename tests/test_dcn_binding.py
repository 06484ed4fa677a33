"""Drop-in boundary of the native DCN seam, CPU side: the five reference-named functions exist with the reference's
positional parameters (det3d/ops/dcn/src/deform_conv_cuda.cpp:152-157, 262-268, 376-381, 490-496, 571-578 and the call
sites det3d/ops/dcn/deform_conv.py:52-58, 77-93, 145-149, 163-168), the det3d shim resolves the module under the name the
reference imports, and the Pack modules migrate pre-version-2 checkpoint keys (deform_conv.py:298-321, 420-444)."""
import inspect

import torch

from rt_pose_amd import dcn, deform_conv_cuda, registry

REF_SIGNATURES = {
    "deform_conv_forward_cuda": ["input", "weight", "offset", "output", "columns", "ones", "kW", "kH", "dW", "dH", "padW", "padH",
                                 "dilationW", "dilationH", "group", "deformable_group", "im2col_step"],
    "deform_conv_backward_input_cuda": ["input", "offset", "gradOutput", "gradInput", "gradOffset", "weight", "columns", "kW", "kH",
                                        "dW", "dH", "padW", "padH", "dilationW", "dilationH", "group", "deformable_group",
                                        "im2col_step"],
    "deform_conv_backward_parameters_cuda": ["input", "offset", "gradOutput", "gradWeight", "columns", "ones", "kW", "kH", "dW",
                                             "dH", "padW", "padH", "dilationW", "dilationH", "group", "deformable_group", "scale",
                                             "im2col_step"],
    "modulated_deform_conv_cuda_forward": ["input", "weight", "bias", "ones", "offset", "mask", "output", "columns", "kernel_h",
                                           "kernel_w", "stride_h", "stride_w", "pad_h", "pad_w", "dilation_h", "dilation_w", "group",
                                           "deformable_group", "with_bias"],
    "modulated_deform_conv_cuda_backward": ["input", "weight", "bias", "ones", "offset", "mask", "columns", "grad_input",
                                            "grad_weight", "grad_bias", "grad_offset", "grad_mask", "grad_output", "kernel_h",
                                            "kernel_w", "stride_h", "stride_w", "pad_h", "pad_w", "dilation_h", "dilation_w",
                                            "group", "deformable_group", "with_bias"],
}


def test_five_reference_functions_with_reference_parameter_order():
    for name, params in REF_SIGNATURES.items():
        fn = getattr(deform_conv_cuda, name)
        assert list(inspect.signature(fn).parameters) == params, name


def test_shim_resolves_the_module_name_the_reference_imports():
    registry.install_det3d_shim(force=True)
    import importlib
    m = importlib.import_module("det3d.ops.dcn.deform_conv_cuda")
    assert m is deform_conv_cuda
    from det3d.ops.dcn import DeformConvPack   # noqa: F401  (the reference's own import path for the modules)


def test_pack_modules_migrate_old_checkpoint_keys():
    for cls, noff in ((dcn.DeformConvPack, 2), (dcn.ModulatedDeformConvPack, 3)):
        holder = torch.nn.Module()
        holder.conv2 = cls(4, 6, 3, padding=1, deformable_groups=1)
        sd = holder.state_dict()
        assert "conv2.conv_offset.weight" in sd and tuple(sd["conv2.conv_offset.weight"].shape) == (noff * 9, 4, 3, 3)
        old = {}
        for k, v in sd.items():
            old[k.replace("conv2.conv_offset.", "conv2_offset.")] = torch.randn_like(v)
        assert "conv2_offset.weight" in old and "conv2.conv_offset.weight" not in old
        fresh = torch.nn.Module()
        fresh.conv2 = cls(4, 6, 3, padding=1, deformable_groups=1)
        # (1) the hook itself, driven the way PyTorch versions that passed the whole dict to children did
        d = {"conv2." + k: v for k, v in fresh.conv2.state_dict().items() if not k.startswith("conv_offset.")}
        d.update({k: v for k, v in old.items() if "_offset." in k})
        fresh.conv2._load_from_state_dict(d, "conv2.", {}, True, [], [], [])
        assert "conv2.conv_offset.weight" in d and "conv2_offset.weight" not in d
        # (2) current PyTorch filters a child's keys by prefix before the hook runs, so a whole checkpoint goes through
        # dcn.migrate_pre_v2_keys first (rt_pose_amd.checkpoint does this)
        res = fresh.load_state_dict(dcn.migrate_pre_v2_keys(dict(old), fresh), strict=True)
        assert not res.missing_keys and not res.unexpected_keys
        assert torch.equal(fresh.conv2.conv_offset.weight, old["conv2_offset.weight"])
        assert torch.equal(fresh.conv2.conv_offset.bias, old["conv2_offset.bias"])
        # a current checkpoint (with metadata) loads unchanged
        fresh.load_state_dict(holder.state_dict(), strict=True)
        assert torch.equal(fresh.conv2.conv_offset.weight, holder.conv2.conv_offset.weight)
