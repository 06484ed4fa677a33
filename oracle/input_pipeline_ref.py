"""CPU restatement (numpy) of the reference's input pipeline for the radar stream -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product path
(rt_pose_amd.input_pipeline -> csrc/input_pipe.hip) never does.

What is restated, all paths relative to /root/reference:
  cube axes + ROI index ranges   det3d/datasets/cruw_pose/cruw_pose.py:38-40 (axes), :125-146 (consider_roi_cube /
                                 get_arr_in_roi)
  cube crop + normalise          cruw_pose.py:167-185 (get_cube), :188-194 (get_cube_phase)
  channel axis                   det3d/datasets/pipelines/pose.py:163-170 (none -> 1, (2,D,...) -> 2D)
  label assignment, 15 heat-maps det3d/datasets/pipelines/pose.py:186-254 (AssignLabelPose, radar branch)
  label assignment, 1 heat-map   det3d/datasets/pipelines/pose.py:386-451 (AssignLabelPose2, radar branch)
  gaussian splat                 det3d/core/utils/center_utils.py:67-91 (gaussian3D / draw_gaussian3D; note the
                                 exponent (2 sigma^2)^(3/2), not a true Gaussian)
Pinned by tests/golden/input_pipeline_golden.npz, captured by importing those reference files in the authoring
container (tests/golden/gen_golden_input.py).

Arithmetic notes kept from the reference: the cube is cast fp16 -> fp32 first and normalised in fp32
((x - lo) / (hi - lo), negatives clamped to 0); voxel coordinates are computed per key-point as
(p - fp32(range_min)) / voxel_size / out_size_factor in fp32 (NumPy >= 2 promotion, what the captured vectors pin; NumPy 1.x's
float64 intermediate is available as numpy_legacy=True) and truncated toward zero for the integer voxel; a
key-point whose voxel falls outside the feature map is skipped but keeps its slot (ind = 0, mask = 0).
"""
import numpy as np


def cube_axes():
    """cruw_pose.py:38-40: the stored cube's z / y / x coordinates."""
    return (np.arange(-5.8, 5.8, 11.6 / 32), np.arange(-10.05, 10.05, 20.1 / 128), np.arange(0, 11.6, 11.6 / 256))


def arr_in_roi(arr, min_max):
    """cruw_pose.py:140-146 -> (idx_min, idx_max) inclusive."""
    lo, hi = min_max
    i0 = int(np.argmin(abs(arr - lo)))
    i1 = int(np.argmin(abs(arr - hi)))
    if hi > arr[-1]:
        return i0, i1
    return i0, i1 - 1


def roi_indices(roi):
    """consider_roi_cube (cruw_pose.py:125-138): [z0, z1, y0, y1, x0, x1], inclusive."""
    az, ay, ax = cube_axes()
    out = []
    for arr, key in ((az, "z"), (ay, "y"), (ax, "x")):
        out += list(arr_in_roi(arr, roi[key]))
    return out


def prep_cube(cube_f16, roi_idx, norm, doppler):
    """get_cube (cruw_pose.py:167-185) + the channel-axis rule of AssignLabelPose (pose.py:163-167) -> [C,Z,Y,X] fp32."""
    a = cube_f16.astype(np.float32)
    z0, z1, y0, y1, x0, x1 = roi_idx
    a = a[:, z0:z1 + 1, y0:y1 + 1, x0:x1 + 1] if doppler else a[z0:z1 + 1, y0:y1 + 1, x0:x1 + 1]
    lo, scale = float(norm[0]), float(norm[1]) - float(norm[0])
    a = (a - lo) / scale
    a[a < 0.0] = 0.0
    return a if doppler else a[None]


def prep_cube_phase(cube_f16, roi_idx):
    """get_cube_phase (cruw_pose.py:188-194) + (2,D,Z,Y,X) -> (2D,Z,Y,X) (pose.py:169-170); already normalised."""
    a = cube_f16.astype(np.float32)
    z0, z1, y0, y1, x0, x1 = roi_idx
    a = a[:, :, z0:z1 + 1, y0:y1 + 1, x0:x1 + 1]
    return a.reshape(-1, *a.shape[2:])


def gaussian3d(diameter):
    """center_utils.py:67-72 with shape (d,d,d), sigma = d/6 (draw_gaussian3D :75-76) -> float64 [d,d,d]."""
    m = (diameter - 1.0) / 2.0
    z, y, x = np.ogrid[-m:m + 1, -m:m + 1, -m:m + 1]
    sigma = diameter / 6
    h = np.exp(-(x * x + y * y + z * z) / (2 * sigma * sigma) ** (3 / 2))
    h[h < np.finfo(h.dtype).eps * h.max()] = 0
    return h


def draw_gaussian3d(hm, center, radius):
    """center_utils.py:74-91 (k = 1): element-wise max of the clipped gaussian into hm [Z,Y,X] fp32."""
    g = gaussian3d(2 * radius + 1)
    x, y, z = int(center[0]), int(center[1]), int(center[2])
    height, width, length = hm.shape
    front, rear = min(x, radius), min(length - x, radius + 1)
    right, left = min(y, radius), min(width - y, radius + 1)
    bottom, top = min(z, radius), min(height - z, radius + 1)
    mh = hm[z - bottom:z + top, y - right:y + left, x - front:x + rear]
    mg = g[radius - bottom:radius + top, radius - right:radius + left, radius - front:radius + rear]
    if min(mg.shape) > 0 and min(mh.shape) > 0:
        np.maximum(mh, mg, out=mh)
    return hm


def _voxel_coords(p_xyz, range_zyx_min, voxel_size_xyz, osf_zyx, numpy_legacy=False):
    """(x,y,z) metres -> fp32 voxel coordinates (pose.py:222-227): (x - radar_range[k]) / voxel_size[k] / osf[k] with x a
    Python float, radar_range an np.float32 array (pose.py:190: the bound is the fp32-ROUNDED value), voxel_size / osf Python
    scalars.  NumPy >= 2 (NEP 50, weak Python scalars) evaluates every step in fp32 -- the captured vectors come from 2.2 and
    pin exactly that; numpy_legacy=True restates NumPy 1.x (value-based promotion: float64 intermediate, one rounding)."""
    x, y, z = [float(v) for v in p_xyz]
    rz, ry, rx = [np.float32(v) for v in range_zyx_min]
    if numpy_legacy:
        c = [(x - float(rx)) / float(voxel_size_xyz[0]) / float(osf_zyx[2]),
             (y - float(ry)) / float(voxel_size_xyz[1]) / float(osf_zyx[1]),
             (z - float(rz)) / float(voxel_size_xyz[2]) / float(osf_zyx[0])]
        return np.array(c, dtype=np.float32)
    f = np.float32
    c = [(f(x) - rx) / f(voxel_size_xyz[0]) / f(osf_zyx[2]),
         (f(y) - ry) / f(voxel_size_xyz[1]) / f(osf_zyx[1]),
         (f(z) - rz) / f(voxel_size_xyz[2]) / f(osf_zyx[0])]
    return np.array(c, dtype=np.float32)


def assign_labels(poses, fmap_zyx, range_zyx_min, voxel_size_xyz, osf_zyx=(1, 1, 1), max_poses=1, min_radius=1,
                  one_hm=False, numpy_legacy=False):
    """One frame.  poses: list of [15][3] (x,y,z metres).  Returns dict(hm, anno_pose, ind, mask, cat) as the reference's
    per-task arrays (single task).
      one_hm=False  AssignLabelPose  (pose.py:186-254): 15 classes, slot k = key-point k of the FIRST pose(s) in list order
                    (num_points = min(15 * len(points), 15 * max_poses)), radius = max(min_radius, 1), anno_pose [M,3]
      one_hm=True   AssignLabelPose2 (pose.py:386-451): 1 class ('Pelvis' = key-point 0 is the centre), slot k = pose k,
                    radius = min_radius, anno_pose [M,45] = every key-point's offset from the centre's integer voxel
    """
    fz, fy, fx = [int(v) for v in fmap_zyx]
    if not one_hm:
        ncls, m, width = 15, 15 * max_poses, 3
        pts = [(k, pose[k]) for pose in poses for k in range(15)]     # gt_points_by_task[0]: [class_idx, x, y, z]
        num = min(len(pts) * 15, m)
        radius = max(min_radius, 1)
    else:
        ncls, m, width = 1, max_poses, 45
        pts = [(0, pose) for pose in poses]
        num = min(len(pts), m)
        radius = min_radius
    hm = np.zeros((ncls, fz, fy, fx), np.float32)
    anno = np.zeros((m, width), np.float32)
    ind = np.zeros((m,), np.int64)
    mask = np.zeros((m,), np.uint8)
    cat = np.zeros((m,), np.int64)
    for k in range(num):
        cls_id, payload = pts[k]   # IndexError when max_poses exceeds the poses present, exactly as the reference
        if not one_hm:
            ct = _voxel_coords(payload, range_zyx_min, voxel_size_xyz, osf_zyx, numpy_legacy)
            ct_int = ct.astype(np.int32)
        else:
            ct = np.concatenate([_voxel_coords(p, range_zyx_min, voxel_size_xyz, osf_zyx, numpy_legacy) for p in payload])
            ct_int = ct.astype(np.int32)[:3]
        if not (0 <= ct_int[0] < fx and 0 <= ct_int[1] < fy and 0 <= ct_int[2] < fz):
            continue
        draw_gaussian3d(hm[cls_id], ct_int, radius)
        x, y, z = int(ct_int[0]), int(ct_int[1]), int(ct_int[2])
        cat[k] = cls_id
        ind[k] = z * fy * fx + y * fx + x
        mask[k] = 1
        if not one_hm:
            anno[k] = ct - np.array([x, y, z], np.float32)
        else:
            anno[k] = (ct.reshape(-1, 3) - ct_int[None, :].astype(np.float32)).flatten()
    return dict(hm=hm, anno_pose=anno, ind=ind, mask=mask, cat=cat)
