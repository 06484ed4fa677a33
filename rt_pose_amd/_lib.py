"""ctypes binding of librtp_hip.so (the C ABI declared in include/rtp.h).

Fails loudly: a missing library raises at import of the symbols, a non-zero return code raises RtpError.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTP_LIB") or os.path.join(HERE, "lib", "librtp_hip.so")  # RTP_LIB: A/B-test another build

RTP_MAX_TERMS = 6

# kernel families for rtp_prof_* (csrc/rtp_prof.h)
(FAM_CONV, FAM_CONV_TILED, FAM_WGRAD, FAM_POINTWISE, FAM_NORM, FAM_LOSS, FAM_OPTIM, FAM_DCN, FAM_WGRAD_TILED, FAM_CONV_TILED_FULL,
 FAM_CONV_TILED_FULL_BWD, FAM_CONV64) = range(12)

_ERR = {-1: "RTP_ERR_SHAPE", -2: "RTP_ERR_UNSUPPORTED", -3: "RTP_ERR_LAUNCH", -4: "RTP_ERR_ALIGN"}


class RtpError(RuntimeError):
    pass


class RtpAct(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("cs", C.c_int), ("co", C.c_int), ("c", C.c_int)]


class RtpConvGeom(C.Structure):
    _fields_ = [("n", C.c_int), ("di", C.c_int), ("hi", C.c_int), ("wi", C.c_int), ("dov", C.c_int), ("ho", C.c_int),
                ("wo", C.c_int), ("ci", C.c_int), ("co", C.c_int), ("ks", C.c_int), ("stride", C.c_int),
                ("pad", C.c_int), ("w_ci_total", C.c_int), ("w_ci_off", C.c_int), ("wgs", C.c_int)]


class RtpGnBwd(C.Structure):
    _fields_ = [("qpart", C.c_void_p), ("q_nsplit", C.c_int), ("p", C.c_void_p), ("tg", C.c_void_p), ("csum_out", C.c_void_p),
                ("csum", C.c_void_p), ("mr", C.c_void_p), ("gamma", C.c_void_p), ("groups", C.c_int), ("coeff_out", C.c_void_p)]


class RtpGnFold(C.Structure):
    _fields_ = [("w", C.c_void_p), ("bias", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("stats", C.c_void_p),
                ("nsplit", C.c_int), ("groups", C.c_int), ("co_real", C.c_int), ("eps", C.c_float), ("mr", C.c_void_p)]


class RtpConv64(C.Structure):
    """include/rtp.h: operands of rtp_conv64_blocks, per 32-channel half."""
    _fields_ = [("x", C.c_void_p * 2), ("x_cs", C.c_int * 2),
                ("w", (C.c_void_p * 2) * 2), ("w_row_stride", C.c_int), ("w_tap_stride", C.c_int), ("w_sample_stride", C.c_long),
                ("w_per_sample", C.c_int),
                ("btab", C.c_void_p * 2), ("bt_cs", C.c_int),
                ("res", C.c_void_p * 2), ("r_cs", C.c_int * 2),
                ("y", C.c_void_p * 2), ("y_cs", C.c_int * 2),
                ("acc", C.c_void_p), ("acc_in", C.c_int), ("acc_out", C.c_int),
                ("relu", C.c_int), ("transposed", C.c_int)]


class RtpGnLazy(C.Structure):
    _fields_ = [("pq", C.c_void_p), ("nsplit", C.c_int), ("mr", C.c_void_p), ("gamma", C.c_void_p), ("groups", C.c_int)]


class RtpTerm(C.Structure):
    _fields_ = [("t", RtpAct), ("coeff", C.c_void_p), ("d", C.c_int), ("h", C.c_int), ("w", C.c_int)]


_P = C.c_void_p
_I = C.c_int
_L = C.c_long
_F = C.c_float
_A = C.POINTER(RtpAct)
_G = C.POINTER(RtpConvGeom)
_T = C.POINTER(RtpTerm)

# name -> argtypes (restype int unless listed in _RESTYPE)
PROTOTYPES = {
    "rtp_chan_stats": [_A, _A, _I, _L, _I, _P, _P],
    "rtp_fold_fwd": [_P, _P, _P, _P, _P, _I, _I, _F, _G, _I, _I, _P, _P, _P, _P, _P],
    "rtp_pack_dgrad_w": [_P, _G, _I, _I, _P, _P],
    "rtp_conv_igemm": [_A, _P, _I, _P, _A, _A, _G, _I, _I, _I, _P],
    "rtp_conv_igemm_stats": [_A, _P, _I, _P, _A, _A, _G, _I, _I, _I, _A, _P, _P],
    "rtp_conv_stats_nsplit": [_A, _G, _I],
    "rtp_conv_sliced_ok": [_A, _G, _I],
    "rtp_conv64_blocks": [_P, _G, _P],
    "rtp_dcn_cl_forward": [_A, _A, _P, _A, _I, _I, _I, _I, _P],
    "rtp_conv_stats_nsplit_ws": [_A, _G, _I],
    "rtp_conv_igemm_ws": [_A, _P, _I, _P, _A, _A, _G, _I, _I, _I, _A, _P, _P, _P],
    "rtp_conv_igemm_acc": [_A, _P, _I, _P, _A, _A, _G, _I, _I, _I, _P, _I, _P],
    "rtp_conv_tiled_ok": [_A, _G, _I],
    "rtp_wgrad": [_A, _A, _G, _I, _P, _P],
    "rtp_wgrad_nsplit": [_G],
    "rtp_wgrad_q": [_A, _A, _G, _I, _P, _P, _P, _P, _P],
    "rtp_wgrad_tg": [_A, _A, _G, _I, _P, _P, _P],
    "rtp_zero_f32": [_P, _L, _P],
    "rtp_qpart_from_slabs": [_P, _I, _I, _I, _I, _I, _P, _P, _P],
    "rtp_gn_bwd_coeffs_cls": [_P, _I, _P, _I, _P, _P, _P, _P, _G, _I, _I, _I, _P, _P],
    "rtp_conv_dgrad_fused": [_A, _P, _A, _P, _P, _T, _I, _I, _A, _G, _P, _P],
    "rtp_conv_gn_fused": [_A, _P, _A, _A, _G, _I, _P, _P],
    "rtp_gn_bwd_p": [_P, _I, _P, _P, _G, _I, _I, _P, _P],
    "rtp_conv_dgrad_fused_ok": [_A, _G],
    "rtp_class_sums_boundary": [_A, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P],
    "rtp_class_sums_p": [_A, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P, _G, _I, _I, _P, _P, _P],
    "rtp_class_sums": [_A, _I, _I, _I, _I, _I, _P, _P, _P],
    "rtp_class_sums_reduce": [_P, _I, _I, _I, _P, _P],
    "rtp_wgrad_fold": [_P, _I, _P, _P, _P, _P, _I, _G, _I, _I, _P, _P, _I, _P],
    "rtp_gn_bwd_coeffs": [_P, _I, _P, _P, _I, _I, _I, _L, _P, _P, _P, _I, _P],
    "rtp_tail_desc_bytes": [],
    "rtp_tail_desc_class_reduce": [_P, _I, _I, _I, _P, _P, C.POINTER(_I), C.POINTER(_I)],
    "rtp_tail_desc_wgrad_fold": [_P, _I, _P, _P, _P, _P, _I, _G, _I, _I, _P, _P, _I, _P, C.POINTER(_I), C.POINTER(_I)],
    "rtp_tail_desc_wgrad_fold_tg": [_P, _I, _P, _G, _I, _I, _P, _P, _I, _P, C.POINTER(_I), C.POINTER(_I)],
    "rtp_tail_desc_gn_param": [_P, _I, _I, _P, _P, _I, _P, C.POINTER(_I), C.POINTER(_I)],
    "rtp_tail_desc_pack_wt": [_P, _I, _I, _I, _I, _P, _P, C.POINTER(_I), C.POINTER(_I)],
    "rtp_tail_desc_fold_fwd": [_P, _P, _P, _P, _P, _I, _I, _F, _G, _I, _I, _P, _P, _P, _P, _P, C.POINTER(_I), C.POINTER(_I)],
    "rtp_tail_launch": [_P, _P, _I, _I, _I, _P],
    "rtp_grad_combine": [_T, _I, _A, _A, _A, _I, _L, _P],
    "rtp_grad_combine_cls": [_T, _I, _A, _A, _A, _I, _I, _I, _I, _I, _P, _P],
    "rtp_grad_combine_cls_lazy": [_T, _I, _P, _A, _A, _A, _I, _I, _I, _I, _I, _P, _P],
    "rtp_fuse_sum": [_T, _I, _P, _A, _I, _I, _I, _I, _I, _P],
    "rtp_fuse_sum_stats": [_T, _I, _P, _A, _I, _I, _I, _I, _I, _P, _I, _P],
    "rtp_fuse_stats_nsplit": [_I, _I, _I, _I, _I],
    "rtp_upsample_bwd": [_A, _I, _I, _I, _A, _I, _I, _I, _I, _P, _P],
    "rtp_upsample_bwd_scratch_floats": [_I] * 8,
    "rtp_stem_fwd": [_P, _P, _P, _A, _I, _L, _P],
    "rtp_stem_stats_nsplit": [_I, _I, _L],
    "rtp_stem_fwd_stats": [_P, _P, _P, _A, _I, _L, _P, _I, _P],
    "rtp_stem_bwd": [_P, _A, _I, _L, _P, _P, _P, _I, _P],
    "rtp_stem_bwd_blocks": [],
    "rtp_pack_ncdhw": [_P, _A, _I, _I, _L, _P],
    "rtp_unpack_ncdhw": [_A, _P, _I, _I, _L, _P],
    "rtp_pack_ncdhw_ex": [_P, _P, _A, _I, _I, _L, _I, _P],
    "rtp_unpack_ncdhw_f32": [_P, _I, _I, _P, _I, _I, _L, _P],
    "rtp_focal_loss": [_P, _I, _P, _P, _P, _P, _I, _I, _L, _I, _F, _P, _P, _A, _P],
    "rtp_focal_loss_ex": [_P, _I, _P, _P, _P, _P, _I, _I, _L, _I, _F, _P, _P, _A, _I, _P],
    "rtp_focal_blocks": [],
    "rtp_reg_loss": [_P, _I, _P, _P, _P, _P, _I, _I, _L, _I, _F, _P, _A, _P],
    "rtp_reg_loss_sparse": [_P, _I, _P, _P, _P, _P, _I, _I, _L, _I, _F, _P, _A, _P, _P],
    "rtp_decode": [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, C.POINTER(_F), C.POINTER(_F), _P, _P, _P],
    "rtp_decode_scratch_floats": [_I, _I],
    "rtp_cube_prep": [_P, _L, _I, _I, _I, C.POINTER(_I), _F, _F, _I, _P, _P],
    "rtp_gaussian_table": [_I, C.POINTER(_F)],
    "rtp_assign_labels": [_P, _P, _I, _I, _I, _I, _I, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_I), _I, _I, _I,
                          _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "rtp_lidar_transform": [_P, _I, _I, C.POINTER(C.c_double), _P],
    "rtp_voxelize_workspace_bytes": [_I],
    "rtp_dynamic_voxelize": [_P, _I, _I, C.POINTER(_F), C.POINTER(_F), _P, _P, _P, _P, _L, _P],
    "rtp_voxels_to_dense": [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P],
    "rtp_sqnorm": [_P, _L, _P, _P, _P],
    "rtp_sqnorm_blocks": [],
    "rtp_adam_step": [_P, _P, _P, _P, _L, _P, _P, _I, _P, _P],
    "rtp_dcn_workspace_bytes": [_I] * 9,
    "rtp_deform_conv_forward": [_P, _P, _P, _P, _P] + [_I] * 16 + [_P],
    "rtp_deform_conv_backward_input": [_P, _P, _P, _P, _P, _P, _P] + [_I] * 16 + [_P],
    "rtp_deform_conv_backward_parameters": [_P, _P, _P, _P, _P] + [_I] * 15 + [_F, _I, _P],
    "rtp_deform_conv_backward": [_P] * 8 + [_I] * 15 + [_F, _I, _P],
    "rtp_deform_conv_backward_overwrite": [_P] * 8 + [_I] * 15 + [_F, _I, _P],
    "rtp_modulated_deform_conv_forward": [_P, _P, _P, _P, _P, _P, _P] + [_I] * 16 + [_P],
    "rtp_modulated_deform_conv_backward": [_P] * 12 + [_I] * 16 + [_P],
    "rtp_prof_enable": [_I, _I],
    "rtp_prof_collect": [_I, C.POINTER(_F), C.POINTER(_I)],
    "rtp_version": [],
    "rtp_event_create": [C.POINTER(_P), _I], "rtp_event_destroy": [_P], "rtp_event_record": [_P, _P], "rtp_stream_wait_event": [_P, _P],
    "rtp_multi_begin": [], "rtp_multi_param_bytes": [], "rtp_multi_end": [_P, _L, C.POINTER(_I)], "rtp_multi_abort": [],
    "rtp_multi_launch": [_I, _P], "rtp_multi_free": [_I],
}
_RESTYPE = {"rtp_version": C.c_char_p, "rtp_multi_param_bytes": C.c_long, "rtp_dcn_workspace_bytes": C.c_long, "rtp_voxelize_workspace_bytes": C.c_long,
            "rtp_upsample_bwd_scratch_floats": C.c_long}

_lib = None


def load():
    """Load the shared library (once).  Raises if it has not been built -- there is no fallback path."""
    global _lib
    if _lib is None:
        # PyTorch-ROCm ships its own libamdhip64; it must be in the process before this library is dlopen'ed so
        # both resolve to ONE HIP runtime (streams and device pointers are shared with torch).
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise RtpError("%s not found: run `python -m rt_pose_amd.build` (hipcc, gfx950). "
                           "rt_pose_amd has no CPU fallback." % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, argtypes in PROTOTYPES.items():
            try:
                fn = getattr(lib, name)  # AttributeError if the symbol is missing
            except AttributeError:
                if os.environ.get("RTP_LIB"):   # A/B runs against an older build: its missing entry points simply stay unbound
                    continue
                raise
            fn.argtypes = argtypes
            fn.restype = _RESTYPE.get(name, C.c_int)
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        raise RtpError("%s failed: %s (%d)" % (what, _ERR.get(rc, "?"), rc))


def act(ptr, cs, co, c):
    return RtpAct(ptr, cs, co, c)
