"""rt_pose_amd -- MI355X (gfx950) native HRRadarPose hot path.

Hand-written HIP kernels behind a C ABI (include/rtp.h, librtp_hip.so), driven by a Python host that mirrors
the reference's det3d registry interface (RadarPoseNet / HRNet3D / CenterHead / RadarFeatureNet).
There is NO CPU or eager-PyTorch fallback: if librtp_hip.so is missing or no GPU is present, compute calls raise.
"""
__version__ = "0.1.0"

# The lane plan (rt_pose_amd/lanes.py) is tuned for HIP's default of FOUR hardware queues: with five or more, the lanes' streams
# stop sharing queues and their kernels time-share the CUs with the main lane's persistent kernels all the time (measured on
# MI355X: 6.1 ms/step with 4 queues, 6.4 with 3, 9.0 with 5-16).  Pin the default before the HIP runtime initialises; an
# explicit GPU_MAX_HW_QUEUES in the environment still wins.
import os as _os
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
