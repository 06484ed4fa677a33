"""Golden vectors for the input pipeline, captured by importing the REFERENCE's own files.

Runs only in the authoring container (needs /root/reference).  det3d cannot be imported as a package, so
det3d/datasets/pipelines/pose.py, det3d/core/utils/center_utils.py and det3d/datasets/cruw_pose/cruw_pose.py are loaded
one by one into synthetic packages; their unrelated imports (box ops, samplers, voxel generator, munch, eval_util, numba)
are empty stand-in modules.  CRUW_POSE_Dataset's methods are called UNBOUND on a bare namespace object carrying only
the attributes they read, with np.load patched to hand back a seeded synthetic fp16 cube (the reference reads
/mnt/ssd3/...).  Only numeric inputs' seeds and the reference's outputs are written (tests/golden/input_pipeline_golden.npz).

    python tests/golden/gen_golden_input.py
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def _pkg(name):
    if name not in sys.modules:
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
        if "." in name:
            parent, child = name.rsplit(".", 1)
            setattr(_pkg(parent), child, m)
    return sys.modules[name]


def _load(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    if "." in modname:
        parent, child = modname.rsplit(".", 1)
        setattr(_pkg(parent), child, m)
    spec.loader.exec_module(m)
    return m


class AttrDict(dict):
    """attribute dict whose missing keys raise AttributeError (what getattr(cfg, name, default) expects)"""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = dict.__setitem__


def import_reference():
    numba = types.ModuleType("numba")
    numba.jit = lambda *a, **k: (lambda f: f)
    sys.modules["numba"] = numba
    for p in ["det3d", "det3d.core", "det3d.core.bbox", "det3d.core.sampler", "det3d.core.input", "det3d.core.utils",
              "det3d.datasets", "det3d.datasets.pipelines", "det3d.datasets.cruw_pose", "det3d.utils", "det3d.torchie"]:
        _pkg(p)
    sys.modules["det3d.torchie"].is_str = lambda x: isinstance(x, str)
    sys.modules["det3d.core.bbox"].box_np_ops = None
    sys.modules["det3d.core.sampler"].preprocess = None
    b = types.ModuleType("det3d.builder"); b.build_dbsampler = None
    sys.modules["det3d.builder"] = b; sys.modules["det3d"].builder = b
    vg = types.ModuleType("det3d.core.input.voxel_generator"); vg.VoxelGenerator = None
    sys.modules["det3d.core.input.voxel_generator"] = vg
    reg = _load("det3d.utils.registry", "det3d/utils/registry.py")
    sys.modules["det3d.utils"].Registry = reg.Registry
    sys.modules["det3d.utils"].build_from_cfg = reg.build_from_cfg
    dreg = types.ModuleType("det3d.datasets.registry")
    dreg.PIPELINES, dreg.DATASETS = reg.Registry("pipeline"), reg.Registry("dataset")
    sys.modules["det3d.datasets.registry"] = dreg; sys.modules["det3d.datasets"].registry = dreg
    sys.modules["det3d.datasets.pipelines"].Compose = None
    _load("det3d.core.utils.circle_nms_jit", "det3d/core/utils/circle_nms_jit.py")
    _load("det3d.core.utils.center_utils", "det3d/core/utils/center_utils.py")
    pose = _load("det3d.datasets.pipelines.pose", "det3d/datasets/pipelines/pose.py")
    munch = types.ModuleType("munch"); munch.DefaultMunch = None; sys.modules["munch"] = munch
    sys.modules["eval_util"] = types.ModuleType("eval_util")
    ds = _load("det3d.datasets.cruw_pose.cruw_pose", "det3d/datasets/cruw_pose/cruw_pose.py")
    return pose, ds


# values of configs/cruw_pose/hr3d.py:28-45, 97-108 and hr3d_one_hm.py:102-107 (numbers, not code)
ROI1 = {"z": [-1.0875000000000021, 4.7125], "y": [-5.0250000000000234, 5.024999999999931], "x": [0.7703125, 8.0203125]}
GRID_SIZE = [0.0453125, 0.15703125, 0.3625]
NORM_ZYX, NORM_DZYX = (150000, 200000), (100000, 9000000)


def synth_cube(seed, doppler=0, lo=100000.0, hi=260000.0):
    """Seeded fp16-representable cube on the stored grid (32,128,256) [x doppler]."""
    rng = np.random.default_rng(seed)
    shape = (doppler, 32, 128, 256) if doppler else (32, 128, 256)
    return np.minimum(rng.uniform(lo, hi, size=shape), 65000.0 if hi < 1e5 else np.inf).astype(np.float32)


def synth_cube_f16(seed, doppler=0, phase=False):
    rng = np.random.default_rng(seed)
    if phase:
        return rng.standard_normal((2, doppler, 32, 128, 256)).astype(np.float16)
    shape = (doppler, 32, 128, 256) if doppler else (32, 128, 256)
    return rng.uniform(0.0, 60000.0, size=shape).astype(np.float16)   # fp16 max is 65504: the on-disk dtype bounds the values


def synth_poses(seed, n):
    """n poses of 15 key-points around a random in-range pelvis; a few key-points land outside the ROI on purpose."""
    rng = np.random.default_rng(seed)
    poses = []
    for _ in range(n):
        c = np.array([rng.uniform(1.5, 7.2), rng.uniform(-4.0, 4.0), rng.uniform(-0.5, 4.0)])
        p = c + rng.normal(0, [0.25, 0.25, 0.45], size=(15, 3))
        p[rng.integers(0, 15)] += np.array([9.0, 0.0, 0.0])      # one key-point beyond the x range
        poses.append(p.tolist())
    return poses


def edge_poses():
    """One pose whose key-points sit exactly on voxel boundaries of the EXACT (double) ROI minimum: p = min + k * voxel.
    The reference subtracts the fp32-ROUNDED minimum (pose.py:190), so each of these lands a hair above or below an
    integer voxel coordinate -- the case where rounding the bound differently changes ind / mask / the heat-map."""
    p = []
    for k in range(15):
        p.append([ROI1["x"][0] + (3 + 9 * k) * GRID_SIZE[0], ROI1["y"][0] + (2 + 4 * k) * GRID_SIZE[1],
                  ROI1["z"][0] + (k % 16) * GRID_SIZE[2]])
    return [p]


def main():
    pose_mod, ds_mod = import_reference()
    DS = ds_mod.CRUW_POSE_Dataset
    out = {}
    # ---- ROI indices: consider_roi_cube on the reference axes
    self = types.SimpleNamespace()
    self.arr_z_cb = np.arange(-5.8, 5.8, 11.6 / 32)
    self.arr_y_cb = np.arange(-10.05, 10.05, 20.1 / 128)
    self.arr_x_cb = np.arange(0, 11.6, 11.6 / 256)
    self.get_arr_in_roi = lambda arr, mm: DS.get_arr_in_roi(self, arr, mm)
    DS.consider_roi_cube(self, ROI1)
    out["roi_idx"] = np.array(self.list_roi_idx_cb, np.int64)
    out["roi_len"] = np.array([len(self.arr_z_cb), len(self.arr_y_cb), len(self.arr_x_cb)], np.int64)
    # ---- get_cube for zyx_real and dzyx_real, get_cube_phase
    real_load = np.load
    for tag, rdr_type, doppler, norm, seed in (("zyx", "zyx_real", 0, (20000, 45000), 11), ("dzyx", "dzyx_real", 3, (0, 10), 12)):
        cube = synth_cube_f16(seed, doppler)
        if tag == "dzyx":
            cube = (cube / 4000.0).astype(np.float16)
        self.seq_id_to_name = {0: "seq"}
        self.rad_normalize_values = norm
        self.cfg = AttrDict(DATASET=AttrDict(RDR_TYPE=rdr_type))
        np.load = lambda *a, **k: cube
        try:
            ref = DS.get_cube(self, 0, "000000")
        finally:
            np.load = real_load
        # channel-axis rule of AssignLabelPose.__call__ (test mode: no labels)
        res, _ = pose_mod.AssignLabelPose(cfg=AttrDict(out_size_factor=[1, 1, 1], target_assigner=AttrDict(tasks=[AttrDict(class_names=["a"])]),
                                                       gaussian_overlap=0.1, max_poses=1, min_radius=1))(
            {"rdr_cube": ref, "mode": "val", "meta": {}}, None)
        t = res["rdr"]["rdr_tensor"]
        out["cube_%s_seed" % tag] = np.array([seed, doppler], np.int64)
        out["cube_%s_norm" % tag] = np.array(norm, np.float64)
        out["cube_%s_shape" % tag] = np.array(t.shape, np.int64)
        out["cube_%s_sample" % tag] = t.reshape(-1)[::997].astype(np.float32)
        out["cube_%s_sums" % tag] = np.array([t.astype(np.float64).sum(), (t.astype(np.float64) ** 2).sum(), float((t == 0).sum())])
    cube = synth_cube_f16(13, 2, phase=True)
    np.load = lambda *a, **k: cube
    try:
        ref = DS.get_cube_phase(self, 0, "000000")
    finally:
        np.load = real_load
    res, _ = pose_mod.AssignLabelPose2(cfg=AttrDict(out_size_factor=[1, 1, 1], target_assigner=AttrDict(tasks=[AttrDict(class_names=["a"])]),
                                                    gaussian_overlap=0.1, max_poses=1, min_radius=2))(
        {"rdr_cube": ref, "mode": "val", "meta": {}}, None)
    t = res["rdr"]["rdr_tensor"]
    out["cube_phase_seed"] = np.array([13, 2], np.int64)
    out["cube_phase_shape"] = np.array(t.shape, np.int64)
    out["cube_phase_sample"] = t.reshape(-1)[::997].astype(np.float32)
    out["cube_phase_sums"] = np.array([t.astype(np.float64).sum(), (t.astype(np.float64) ** 2).sum()])
    # ---- label assignment, both assigners, several frames (incl. no pose, two poses with max_poses 2)
    info = AttrDict(DATASET=AttrDict(ROI=AttrDict(roi1=ROI1), LABEL=AttrDict(ROI_TYPE="roi1"), RDR_CUBE=AttrDict(GRID_SIZE=GRID_SIZE)))
    names15 = ["k%d" % i for i in range(15)]
    cases = [("a15", pose_mod.AssignLabelPose, names15, 1, 1, 1, 21), ("a15_none", pose_mod.AssignLabelPose, names15, 1, 1, 0, 22),
             ("a15_two", pose_mod.AssignLabelPose, names15, 2, 1, 2, 23), ("a1", pose_mod.AssignLabelPose2, ["Pelvis"], 1, 2, 1, 24),
             ("a1_two", pose_mod.AssignLabelPose2, ["Pelvis"], 2, 2, 2, 25), ("a1_none", pose_mod.AssignLabelPose2, ["Pelvis"], 1, 2, 0, 26),
             ("a15_edge", pose_mod.AssignLabelPose, names15, 1, 1, 1, -1), ("a1_edge", pose_mod.AssignLabelPose2, ["Pelvis"], 1, 2, 1, -1)]
    dummy = np.zeros((16, 64, 160), np.float32)
    for tag, cls, names, max_poses, min_radius, nposes, seed in cases:
        poses = synth_poses(seed, nposes) if seed >= 0 else edge_poses()
        cfg = AttrDict(out_size_factor=[1, 1, 1], target_assigner=AttrDict(tasks=[AttrDict(class_names=names)]),
                       gaussian_overlap=0.1, max_poses=max_poses, min_radius=min_radius)
        res, _ = cls(cfg=cfg)({"rdr_cube": dummy, "mode": "train", "meta": {}, "poses": poses, "hm_size": (16, 64, 160)}, info)
        r = res["rdr"]
        hm = r["hm"][0]
        nz = np.flatnonzero(hm)
        out["lab_%s_cfg" % tag] = np.array([seed, nposes, max_poses, min_radius, len(names)], np.int64)
        out["lab_%s_poses" % tag] = np.array(poses, np.float64).reshape(nposes, 15, 3)
        out["lab_%s_hm_idx" % tag] = nz.astype(np.int64)
        out["lab_%s_hm_val" % tag] = hm.reshape(-1)[nz].astype(np.float32)
        out["lab_%s_anno" % tag] = r["anno_pose"][0]
        out["lab_%s_ind" % tag] = r["ind"][0]
        out["lab_%s_mask" % tag] = r["mask"][0]
        out["lab_%s_cat" % tag] = r["cat"][0]
    np.savez_compressed(os.path.join(HERE, "input_pipeline_golden.npz"), **out)
    print("wrote", len(out), "arrays;", os.path.getsize(os.path.join(HERE, "input_pipeline_golden.npz")), "bytes")
    print("roi idx", out["roi_idx"], "lens", out["roi_len"])


if __name__ == "__main__":
    main()
