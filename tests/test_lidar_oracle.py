"""LiDAR-stream oracle (oracle/lidar_ref.py) against vectors captured from the reference's dynamic_voxel_encoder.py /
scatter.py / Preprocess (tests/golden/gen_golden_lidar.py).  CPU only."""
import os

import numpy as np
import torch

from oracle import lidar_ref as L
from tests.golden.gen_golden_lidar import PC_RANGE, VOXEL_SIZE, synth_points

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "lidar_golden.npz"))


def test_dynamic_voxelization_matches_reference_bitwise():
    assert G["grid_shape_xyz"].tolist() == [160, 64, 16]
    vs, cs = [], []
    for seed, n in G["seeds"]:
        p = synth_points(int(seed), int(n))
        if seed == 33:
            p = p + np.float32(100)          # a frame whose points all fall outside the range
        v, c = L.voxelization(p, PC_RANGE, VOXEL_SIZE)
        vs.append(v)
        cs.append(c)
    assert len(vs[2]) == 0
    coors = L.batch_coords(cs).numpy()
    assert np.array_equal(coors, G["coors"])
    assert np.array_equal(torch.cat(vs).numpy(), G["voxels"])          # same summation order: bit-exact
    assert (coors[:, 3] == 160).any()                                   # points on the upper x bound keep coordinate == size
    c0 = coors[coors[:, 0] == 0][:, 1:]
    keys = (c0[:, 0] * 100000 + c0[:, 1]) * 100000 + c0[:, 2]
    assert np.all(np.diff(keys) > 0)                                    # sorted unique (z, y, x)


def test_extrinsic_transform():
    seed, n = [int(v) for v in G["xform_seed"]]
    got = L.l2r_transform(synth_points(seed, n), G["P_L2R"])
    assert np.array_equal(got, G["xform_points"])


def test_dense_scatter_drops_out_of_grid_voxels():
    v, c = L.voxelization(synth_points(31, 6000), PC_RANGE, VOXEL_SIZE)
    grid, occ = L.voxels_to_dense(v, c, (16, 64, 160))
    inside = (c[:, 2] < 160)
    assert int(occ.sum()) == int(inside.sum()) < len(c)
    k = int(np.flatnonzero(inside.numpy())[5])
    assert torch.equal(grid[c[k, 0], c[k, 1], c[k, 2]], v[k])
