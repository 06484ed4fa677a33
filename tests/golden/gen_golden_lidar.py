"""Golden vectors for the LiDAR-stream row, captured by importing the REFERENCE's own files:
det3d/models/readers/dynamic_voxel_encoder.py (voxelization, DynamicVoxelEncoder.forward) with det3d/core/utils/scatter.py,
and Preprocess.__call__ of det3d/datasets/pipelines/pose.py for the extrinsic transform.  Authoring container only.

    python tests/golden/gen_golden_lidar.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.golden.gen_golden_input import AttrDict, _load, _pkg, import_reference  # noqa: E402

# radar ROI as the LiDAR voxel grid (configs/cruw_pose/hr3d.py:32,39): x, y, z order
PC_RANGE = [0.7703125, -5.0250000000000234, -1.0875000000000021, 8.0203125, 5.024999999999931, 4.7125]
VOXEL_SIZE = [0.0453125, 0.15703125, 0.3625]


def synth_points(seed, n, c=5):
    rng = np.random.default_rng(seed)
    p = np.empty((n, c), np.float32)
    p[:, 0] = rng.uniform(-0.5, 9.0, n)
    p[:, 1] = rng.uniform(-6.0, 6.0, n)
    p[:, 2] = rng.uniform(-1.5, 5.0, n)
    p[:, 3:] = rng.uniform(0, 1, (n, c - 3))
    # clusters so that many voxels hold several points
    k = n // 3
    p[:k, :3] = np.array([3.0, 0.5, 1.0], np.float32) + rng.normal(0, 0.15, (k, 3)).astype(np.float32)
    p[-3:, 0] = np.float32(PC_RANGE[3])      # points exactly on the upper x bound: kept, coordinate == grid size
    return p


def main():
    pose_mod, _ = import_reference()
    for p in ["det3d.models", "det3d.models.readers"]:
        _pkg(p)
    reg = sys.modules["det3d.utils.registry"]
    mreg = types.ModuleType("det3d.models.registry")
    mreg.READERS = reg.Registry("reader")
    sys.modules["det3d.models.registry"] = mreg
    sys.modules["det3d.models"].registry = mreg
    _load("det3d.core.utils.scatter", "det3d/core/utils/scatter.py")
    dve = _load("det3d.models.readers.dynamic_voxel_encoder", "det3d/models/readers/dynamic_voxel_encoder.py")
    out = {}
    enc = dve.DynamicVoxelEncoder(PC_RANGE, VOXEL_SIZE)
    out["grid_shape_xyz"] = np.asarray(enc.shape_np, np.int64)
    pts = [torch.from_numpy(synth_points(31, 6000)), torch.from_numpy(synth_points(32, 1500)), torch.from_numpy(synth_points(33, 40) + np.float32(100))]
    vox, coors, shape = enc(pts)
    out["seeds"] = np.array([[31, 6000], [32, 1500], [33, 40]], np.int64)
    out["voxels"] = vox.numpy()
    out["coors"] = coors.numpy()
    # extrinsic transform through Preprocess.__call__
    rng = np.random.default_rng(5)
    a = 0.1
    P = np.eye(4)
    P[:3, :3] = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]) @ np.array([[1, 0, 0], [0, np.cos(0.05), -np.sin(0.05)], [0, np.sin(0.05), np.cos(0.05)]])
    P[:3, 3] = [0.12, -0.3, 1.7]
    raw = synth_points(34, 2000)
    pre = pose_mod.Preprocess(cfg=AttrDict(shuffle_points=False, pc_type="lidar_pc", mode="val", no_augmentation=True))
    res, _ = pre({"lidar_pc": raw.copy(), "P_L2R": P}, None)
    out["P_L2R"] = P
    out["xform_seed"] = np.array([34, 2000], np.int64)
    out["xform_points"] = res["lidar"]["points"]
    np.savez_compressed(os.path.join(HERE, "lidar_golden.npz"), **out)
    print("wrote", {k: v.shape for k, v in out.items()}, os.path.getsize(os.path.join(HERE, "lidar_golden.npz")), "bytes")


if __name__ == "__main__":
    main()
