"""The input-pipeline oracle (oracle/input_pipeline_ref.py) against vectors captured from the reference's own
cruw_pose.py / pipelines/pose.py / center_utils.py (tests/golden/gen_golden_input.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import input_pipeline_ref as R
from tests.golden.gen_golden_input import GRID_SIZE, ROI1, synth_cube_f16

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "input_pipeline_golden.npz"))
RANGE_ZYX_MIN = (ROI1["z"][0], ROI1["y"][0], ROI1["x"][0])


def test_roi_indices_and_native_shape():
    idx = R.roi_indices(ROI1)
    assert idx == G["roi_idx"].tolist() == [13, 28, 32, 95, 17, 176]
    assert [idx[1] - idx[0] + 1, idx[3] - idx[2] + 1, idx[5] - idx[4] + 1] == G["roi_len"].tolist() == [16, 64, 160]


@pytest.mark.parametrize("tag", ["zyx", "dzyx"])
def test_cube_crop_normalise(tag):
    seed, doppler = [int(v) for v in G["cube_%s_seed" % tag]]
    cube = synth_cube_f16(seed, doppler)
    if tag == "dzyx":
        cube = (cube / 4000.0).astype(np.float16)
    t = R.prep_cube(cube, G["roi_idx"].tolist(), G["cube_%s_norm" % tag], doppler > 0)
    assert list(t.shape) == G["cube_%s_shape" % tag].tolist()
    assert np.array_equal(t.reshape(-1)[::997], G["cube_%s_sample" % tag])          # bit-exact fp32
    s = np.array([t.astype(np.float64).sum(), (t.astype(np.float64) ** 2).sum(), float((t == 0).sum())])
    assert np.allclose(s, G["cube_%s_sums" % tag], rtol=1e-12, atol=0)


def test_cube_phase():
    seed, doppler = [int(v) for v in G["cube_phase_seed"]]
    t = R.prep_cube_phase(synth_cube_f16(seed, doppler, phase=True), G["roi_idx"].tolist())
    assert list(t.shape) == G["cube_phase_shape"].tolist() == [2 * doppler, 16, 64, 160]
    assert np.array_equal(t.reshape(-1)[::997], G["cube_phase_sample"])


@pytest.mark.parametrize("tag", ["a15", "a15_none", "a15_two", "a1", "a1_two", "a1_none", "a15_edge", "a1_edge"])
def test_label_assignment(tag):
    seed, nposes, max_poses, min_radius, ncls = [int(v) for v in G["lab_%s_cfg" % tag]]
    poses = G["lab_%s_poses" % tag].tolist()
    r = R.assign_labels(poses, (16, 64, 160), RANGE_ZYX_MIN, GRID_SIZE, (1, 1, 1), max_poses, min_radius, one_hm=(ncls == 1))
    nz = np.flatnonzero(r["hm"])
    assert np.array_equal(nz, G["lab_%s_hm_idx" % tag])
    assert np.array_equal(r["hm"].reshape(-1)[nz], G["lab_%s_hm_val" % tag])      # exact: the table is float64 -> fp32 once
    assert np.array_equal(r["ind"], G["lab_%s_ind" % tag])
    assert np.array_equal(r["mask"], G["lab_%s_mask" % tag])
    assert np.array_equal(r["cat"], G["lab_%s_cat" % tag])
    assert np.array_equal(r["anno_pose"], G["lab_%s_anno" % tag])     # bit-exact: same fp32 arithmetic as NumPy >= 2
    # NumPy 1.x promotion (float64 intermediate, one rounding) stays within one fp32 ulp of a coordinate < 160
    r1 = R.assign_labels(poses, (16, 64, 160), RANGE_ZYX_MIN, GRID_SIZE, (1, 1, 1), max_poses, min_radius, one_hm=(ncls == 1),
                         numpy_legacy=True)
    if "edge" not in tag:    # off the voxel boundaries the two promotions pick the same voxels
        assert np.array_equal(r1["ind"], r["ind"]) and np.abs(r1["anno_pose"] - r["anno_pose"]).max() <= 2e-5
    if nposes:
        assert r["mask"].sum() > 0 and r["mask"].sum() < r["mask"].size + 1


def test_max_poses_beyond_poses_present_raises_like_the_reference():
    with pytest.raises(IndexError):
        R.assign_labels(G["lab_a15_poses"].tolist(), (16, 64, 160), RANGE_ZYX_MIN, GRID_SIZE, max_poses=2)
