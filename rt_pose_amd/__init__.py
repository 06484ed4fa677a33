"""rt_pose_amd -- MI355X (gfx950) native HRRadarPose hot path.

Hand-written HIP kernels behind a C ABI (include/rtp.h, librtp_hip.so), driven by a Python host that mirrors
the reference's det3d registry interface (RadarPoseNet / HRNet3D / CenterHead / RadarFeatureNet).
There is NO CPU or eager-PyTorch fallback: if librtp_hip.so is missing or no GPU is present, compute calls raise.
"""
__version__ = "0.1.0"
