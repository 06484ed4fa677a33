"""Device-side input pipeline for the radar stream.

The reference prepares every frame in NumPy inside DataLoader workers and ships finished fp32 tensors to the GPU with
blocking copies (det3d/datasets/cruw_pose/cruw_pose.py:167-194, det3d/datasets/pipelines/pose.py:146-451,
det3d/datasets/loader/build_loader.py:46-57, det3d/torchie/trainer/trainer.py:378-380).  Here the host only hands over
what is on disk -- the raw fp16 cube(s) of a batch and the key-point list -- through a ring of pinned staging buffers
and ONE asynchronous H2D copy on a side stream; crop / normalise / clamp and the CenterNet label assignment run as two
small kernels (csrc/input_pipe.hip) that write straight into the training plan's input and label buffers.  A batch of
hr3d costs 16.8 MB over PCIe instead of 84 MB of fp32 input plus 79 MB of heat-maps.

Everything arithmetic happens in the kernels; this file is plumbing (buffers, streams, events) plus the reference's
configuration arithmetic (ROI index ranges).  There is no CPU fallback.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check

# stored cube grid of the dataset (cruw_pose.py:38-40)
CUBE_AXES = (("z", -5.8, 5.8, 32), ("y", -10.05, 10.05, 128), ("x", 0.0, 11.6, 256))


def roi_indices(roi):
    """Inclusive (z0,z1,y0,y1,x0,x1) of a Cartesian ROI on the stored grid -- CRUW_POSE_Dataset.consider_roi_cube /
    get_arr_in_roi (cruw_pose.py:125-146): nearest grid point to each bound, upper bound exclusive unless it lies beyond
    the axis."""
    out = []
    for key, lo, hi, n in CUBE_AXES:
        arr = np.arange(lo, hi, (hi - lo) / n)
        mn, mx = roi[key]
        i0 = int(np.argmin(abs(arr - mn)))
        i1 = int(np.argmin(abs(arr - mx)))
        out += [i0, i1 if mx > arr[-1] else i1 - 1]
    return out


class DeviceInputPipeline:
    """Feeds a PoseEngine (rt_pose_amd.engine) from raw cubes + key-points.

        pipe = DeviceInputPipeline(engine, roi, voxel_size_xyz, norm=(150000, 200000), rdr_type="zyx_real",
                                   max_poses=1, min_radius=1)
        ev = pipe.submit(cubes_f16, poses)        # async: H2D + two kernels on the pipeline's stream
        torch.cuda.current_stream().wait_event(ev)  # before the step that consumes the batch

    cubes_f16: numpy fp16 [B, 32,128,256] ('zyx_real'), [B, D, 32,128,256] ('dzyx_real') or [B, 2, D, 32,128,256] (phase)
    poses:     per frame a list of poses, each [15][3] (x, y, z) metres (the dataset's `poses` entry)
    """

    def __init__(self, engine, roi, voxel_size_xyz, norm, rdr_type="zyx_real", max_poses=1, min_radius=1,
                 out_size_factor=(1, 1, 1), stored_dims=(32, 128, 256), ring=2, max_in=None, numpy_legacy=False):
        self.eng, self.be = engine, engine.be
        # voxel-coordinate arithmetic: fp32 throughout (NumPy >= 2, the captured vectors) or NumPy 1.x's float64 intermediate
        self.numpy_legacy = bool(numpy_legacy)
        self.lib = _lib.load()
        dev = self.be.device
        self.roi_idx = roi_indices(roi)
        z, y, x = [self.roi_idx[2 * i + 1] - self.roi_idx[2 * i] + 1 for i in range(3)]
        if (z, y, x) != tuple(engine.dims):
            raise ValueError("ROI %r gives %r, the engine was planned for %r" % (roi, (z, y, x), engine.dims))
        # the reference keeps the ROI bounds in an np.float32 array (pipelines/pose.py:190): use the fp32-rounded minima
        self.range_min = (C.c_double * 3)(*[float(np.float32(roi[k][0])) for k in ("z", "y", "x")])
        self.vsize = (C.c_double * 3)(*[float(v) for v in voxel_size_xyz])
        self.osf = (C.c_int * 3)(*[int(v) for v in out_size_factor])
        self.roi_c = (C.c_int * 6)(*self.roi_idx)
        self.norm = (float(norm[0]), float(norm[1]))
        self.phase = "complex" in rdr_type
        self.stored = tuple(stored_dims)
        self.b = engine.n
        self.cin = engine.x_in.shape[1]
        self.one_hm = engine.ncls == 1
        self.max_poses = int(max_poses)
        # AssignLabelPose hard-codes radius 1 (>= min_radius), AssignLabelPose2 uses min_radius (pose.py:213-214, 405)
        self.radius = int(min_radius) if self.one_hm else max(int(min_radius), 1)
        self.m = engine.m
        want_m = self.max_poses if self.one_hm else 15 * self.max_poses
        if want_m != self.m:
            raise ValueError("engine has %d label slots, max_poses=%d needs %d" % (self.m, self.max_poses, want_m))
        self.max_in = int(max_in or self.max_poses)
        tab = (C.c_float * ((2 * self.radius + 1) ** 3))()
        check(self.lib.rtp_gaussian_table(self.radius, tab), "rtp_gaussian_table")
        self.table = torch.tensor(list(tab), dtype=torch.float32, device=dev)
        self.prev = torch.zeros(self.b, self.m, 4, dtype=torch.int32, device=dev)
        engine.tgt_hm.zero_()   # once: from here on only the previously written boxes are cleared
        n_cube = self.b * self.cin * int(np.prod(self.stored))
        self.stream = torch.cuda.Stream(dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))   # the zero fill above precedes the first splat
        self.ring = []
        for _ in range(ring):
            self.ring.append(dict(
                cube_h=torch.empty(n_cube, dtype=torch.float16).pin_memory(),
                pose_h=torch.zeros(self.b, self.max_in, 15, 3, dtype=torch.float64).pin_memory(),
                np_h=torch.zeros(self.b, dtype=torch.int32).pin_memory(),
                cube_d=torch.empty(n_cube, dtype=torch.float16, device=dev),
                pose_d=torch.zeros(self.b, self.max_in, 15, 3, dtype=torch.float64, device=dev),
                np_d=torch.zeros(self.b, dtype=torch.int32, device=dev),
                done=torch.cuda.Event()))
        self.k = 0

    def submit(self, cubes_f16, poses, after=None):
        """Stage one batch and launch its preparation; returns the event that marks the engine's buffers ready.
        after: event recorded at the end of the step that still reads the plan's (single) input / label buffers -- the
        H2D copies into the ring run at once, beside that step; only the two preparation kernels wait for it.
        The caller must not run a step that reads those buffers before waiting on the returned event."""
        slot = self.ring[self.k % len(self.ring)]
        self.k += 1
        slot["done"].synchronize()   # the slot's previous H2D has left the pinned buffer
        cubes = np.ascontiguousarray(cubes_f16, dtype=np.float16)
        if cubes.size != slot["cube_h"].numel():
            raise ValueError("expected %d fp16 values (B=%d, Cin=%d, stored %r), got %r" % (
                slot["cube_h"].numel(), self.b, self.cin, self.stored, cubes.shape))
        slot["cube_h"].numpy()[:] = cubes.reshape(-1)
        ph, nh = slot["pose_h"].numpy(), slot["np_h"].numpy()
        ph[:] = 0.0
        for f, plist in enumerate(poses):
            n = len(plist)
            if not self.one_hm and 0 < n < self.max_poses:
                # pipelines/pose.py:210-212 indexes gt points up to 15*max_poses: the reference raises here
                raise IndexError("frame %d has %d pose(s), max_poses=%d (AssignLabelPose would index past its key-point list)"
                                 % (f, n, self.max_poses))
            nh[f] = n
            for j in range(min(n, self.max_in)):
                ph[f, j] = np.asarray(plist[j], dtype=np.float64).reshape(15, 3)
        eng = self.eng
        with torch.cuda.stream(self.stream):
            slot["cube_d"].copy_(slot["cube_h"], non_blocking=True)
            slot["pose_d"].copy_(slot["pose_h"], non_blocking=True)
            slot["np_d"].copy_(slot["np_h"], non_blocking=True)
            if after is not None:
                self.stream.wait_event(after)
            s = C.c_void_p(self.stream.cuda_stream)
            zs, ys, xs = self.stored
            check(self.lib.rtp_cube_prep(C.c_void_p(slot["cube_d"].data_ptr()), self.b * self.cin, zs, ys, xs, self.roi_c,
                                         self.norm[0], self.norm[1], 0 if self.phase else 1,
                                         C.c_void_p(eng.x_in.data_ptr()), s), "rtp_cube_prep")
            d, h, w = eng.dims
            check(self.lib.rtp_assign_labels(C.c_void_p(slot["pose_d"].data_ptr()), C.c_void_p(slot["np_d"].data_ptr()),
                                             self.b, self.max_in, self.max_poses, int(self.one_hm), self.radius,
                                             self.range_min, self.vsize, self.osf, d, h, w,
                                             C.c_void_p(self.table.data_ptr()), C.c_void_p(eng.tgt_hm.data_ptr()),
                                             C.c_void_p(eng.tgt_pose.data_ptr()), C.c_void_p(eng.tgt_ind.data_ptr()),
                                             C.c_void_p(eng.tgt_mask.data_ptr()), C.c_void_p(eng.tgt_cat.data_ptr()),
                                             C.c_void_p(self.prev.data_ptr()), int(self.numpy_legacy), s), "rtp_assign_labels")
            slot["done"].record(self.stream)
        return slot["done"]
