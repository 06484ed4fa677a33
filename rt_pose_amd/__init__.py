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
    only read when the HIP runtime initialises, so a process that may already have done so gets a warning instead of a silent no-op.
    Launchers should export GPU_MAX_HW_QUEUES=4 themselves (bench.py and the test children do it before anything touches the GPU).
    Returns what this process KNOWS: the value when it was in the environment at the call or could still be pinned; otherwise
    "unverified ..." -- torch.cuda.is_initialized() stays False after torch.cuda.set_device() and after HIP calls made through ctypes
    (librtp), so "probably not initialised yet" is not reported as a fact."""
    import os
    import sys
    import warnings
    cur = os.environ.get("GPU_MAX_HW_QUEUES")
    if cur is not None:
        return cur
    late = False
    torch = sys.modules.get("torch")
    if torch is not None:
        try:
            late = bool(torch.cuda.is_initialized())
        except Exception:
            late = False
    lib = sys.modules.get("rt_pose_amd._lib")
    if lib is not None and getattr(lib, "_lib", None) is not None:
        late = True   # librtp_hip.so is loaded: its entry points may have initialised HIP already
    if late:
        warnings.warn("rt_pose_amd: HIP was initialised before GPU_MAX_HW_QUEUES could be pinned to %s; the lane plan is tuned for "
                      "that value (export it before starting the process)" % default, RuntimeWarning, stacklevel=2)
        return "unverified (not in the environment when HIP initialised; the runtime's default is 4)"
    os.environ["GPU_MAX_HW_QUEUES"] = default
    # torch may already be imported and a device selected (set_device initialises HIP without flipping is_initialized): then the
    # variable was set too late and nothing here can tell
    return default if torch is None else "%s (set by this process; unverified if a device was selected earlier)" % default
