"""CPU check of the synthetic task the key-point agreement harness trains on (tests/keypoint_agreement.py): targets follow
the reference's label layout (AssignLabelPose, det3d/datasets/pipelines/pose.py:206-254) and agree with the metric
ground truth through the reference's decode formula (center_head.py:304-311)."""
import numpy as np

from rt_pose_amd import configs
from tests.keypoint_agreement import make_pose_batch


def test_pose_batch_layout_and_decode_roundtrip():
    dims = (16, 64, 160)
    ex, gt = make_pose_batch(2, dims, 7)
    r = ex["rdr"]
    hm, ind, anno = r["hm"][0], r["ind"][0], r["anno_pose"][0]
    assert tuple(r["rdr_tensor"].shape) == (2, 1, *dims) and float(r["rdr_tensor"].min()) >= 0
    assert (hm.reshape(2, 15, -1).argmax(2) == ind).all() and float(hm.max()) == 1.0
    assert float(anno.min()) >= 0 and float(anno.max()) < 1
    vs, org = np.array(configs.VOXEL_SIZE), np.array(configs.test_cfg()["pc_range"])
    Z, Y, X = dims
    i = ind.numpy()
    vox = np.stack([i % X, (i // X) % Y, i // (X * Y)], -1)          # x, y, z
    dec = (vox + anno.numpy()) * vs + org                            # the reference's decode of a perfect prediction
    assert np.abs(dec - gt).max() < 1e-5


def test_one_heat_map_pose_batch_decodes_to_the_ground_truth():
    """One-heat-map layout (pose.py:407-451): one peak at the root joint's voxel, 45 offsets relative to it; the reference's decode of
    a perfect prediction (center_head.py:348-355: every joint = peak voxel + its three regressed offsets) returns the metric truth."""
    dims = (16, 64, 160)
    ex, gt = make_pose_batch(2, dims, 9, "hr3d_one_hm_doppler")
    r = ex["rdr"]
    hm, ind, anno = r["hm"][0], r["ind"][0], r["anno_pose"][0]
    assert tuple(r["rdr_tensor"].shape) == (2, 32, *dims) and tuple(hm.shape) == (2, 1, *dims) and tuple(anno.shape) == (2, 1, 45)
    assert (hm.reshape(2, 1, -1).argmax(2) == ind).all() and float(hm.max()) == 1.0
    vs, org = np.array(configs.VOXEL_SIZE), np.array(configs.test_cfg()["pc_range"])
    Z, Y, X = dims
    i = ind.numpy()[:, 0]
    vox = np.stack([i % X, (i // X) % Y, i // (X * Y)], -1)[:, None, :]                 # [B, 1, (x, y, z)]
    dec = (vox + anno.numpy().reshape(2, 15, 3)) * vs + org
    assert np.abs(dec - gt).max() < 1e-5
