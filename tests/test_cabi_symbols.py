"""The C-ABI library loads on a machine without a GPU and exports every entry point include/rtp.h declares; the ctypes
table binds all of them; compute calls fail loudly (there is no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rtp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from rt_pose_amd import _lib
    from rt_pose_amd import build
    if not os.path.exists(_lib.LIB_PATH):
        build.build(verbose=False)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), "include/rtp.h declares %s but librtp_hip.so does not export it" % s
    assert set(_lib.PROTOTYPES) == set(syms), set(_lib.PROTOTYPES) ^ set(syms)
    assert _lib.load().rtp_version().startswith(b"rt_pose_amd")


def test_no_cpu_fallback():
    import torch
    from rt_pose_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from rt_pose_amd.backend import HipBackend
    with pytest.raises(_lib.RtpError):
        HipBackend()
    from rt_pose_amd.dcn import deform_conv
    with pytest.raises(NotImplementedError):
        deform_conv(torch.zeros(1, 4, 5, 5), torch.zeros(1, 18, 5, 5), torch.zeros(4, 4, 3, 3))
    from rt_pose_amd import configs
    from rt_pose_amd.registry import build_detector
    model = build_detector(configs.model_dict("hr3d"), None, configs.test_cfg())
    with pytest.raises(_lib.RtpError):
        model({"rdr": {"rdr_tensor": torch.zeros(1, 1, 8, 16, 16)}, "meta": [{}]}, return_loss=False)
