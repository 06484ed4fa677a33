"""rt_pose_amd -- MI355X (gfx950) native HRRadarPose hot path.

Hand-written HIP kernels behind a C ABI (include/rtp.h, librtp_hip.so), driven by a Python host that mirrors
the reference's det3d registry interface (RadarPoseNet / HRNet3D / CenterHead / RadarFeatureNet).
There is NO CPU or eager-PyTorch fallback: if librtp_hip.so is missing or no GPU is present, compute calls raise.
"""
__version__ = "0.1.0"



def pin_hw_queues(default="4"):
    """The lane plan (rt_pose_amd/lanes.py) is tuned for HIP's default of FOUR hardware queues: with five or more the lanes' streams
    stop sharing queues and their kernels time-share the CUs with the main lane's persistent kernels all the time (measured on
    MI355X: 6.1 ms/step with 4 queues, 6.4 with 3, 9.0 with 5-16).  Called by the entry points that own the process (the trainer,
    bench.py) -- importing the package no longer touches the environment.  An explicit GPU_MAX_HW_QUEUES wins; the variable is
    only read when the HIP runtime initialises, so a process that has already done so gets a warning instead of a silent no-op.
    Returns the value in effect for this process as far as it can be known."""
    import os
    import warnings
    cur = os.environ.get("GPU_MAX_HW_QUEUES")
    if cur is not None:
        return cur
    try:
        import torch
        late = torch.cuda.is_initialized()
    except Exception:
        late = False
    if late:
        warnings.warn("rt_pose_amd: HIP was initialised before GPU_MAX_HW_QUEUES could be pinned to %s; the lane plan is tuned for "
                      "that value (export it before starting the process)" % default, RuntimeWarning, stacklevel=2)
        return "unset (runtime default)"
    os.environ["GPU_MAX_HW_QUEUES"] = default
    return default
