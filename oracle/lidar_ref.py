"""CPU restatement (numpy / torch CPU) of the LiDAR-stream pieces next to the radar path -- TEST INFRASTRUCTURE ONLY
(only tests/, smoke() and bench.py's cpu_baseline leg may import oracle/).

  l2r_transform      det3d/datasets/pipelines/pose.py:34-38 (Preprocess.__call__): homogeneous LiDAR points through P_L2R,
                     float64 matmul, written back into the float32 point array
  voxelization       det3d/models/readers/dynamic_voxel_encoder.py:8-19: inclusive range filter, coords = ((p - min) /
                     voxel_size).to(int64) on the (z, y, x) columns, torch.unique(dim=0) (lexicographic ascending), scatter_mean of
                     every point feature (det3d/core/utils/scatter.py:21-60: sum in index order, divided by the clamped count)
  batch_coords       DynamicVoxelEncoder.forward :86-100: per-sample results concatenated, batch index padded in front
Pinned by tests/golden/lidar_golden.npz, captured by importing the reference files (tests/golden/gen_golden_lidar.py).
The reference defines no fusion of these voxels with the radar feature (voxelnet.py:47-49 calls a missing backbone); the
dense scatter below is this repo's own step (parity unpinned by construction)."""
import numpy as np
import torch


def l2r_transform(points, P_L2R):
    pts = np.array(points, dtype=np.float32, copy=True)
    homo = np.hstack((pts[:, :3], np.ones(len(pts)).reshape(-1, 1)))
    pts[:, :3] = (np.asarray(P_L2R, np.float64) @ homo.T).T[:, :3]
    return pts


def voxelization(points, pc_range, voxel_size):
    points = torch.as_tensor(points, dtype=torch.float32)
    pc_range = torch.as_tensor(pc_range, dtype=torch.float32)
    voxel_size = torch.as_tensor(voxel_size, dtype=torch.float32)
    keep = ((points[:, 0] >= pc_range[0]) & (points[:, 0] <= pc_range[3]) & (points[:, 1] >= pc_range[1])
            & (points[:, 1] <= pc_range[4]) & (points[:, 2] >= pc_range[2]) & (points[:, 2] <= pc_range[5]))
    points = points[keep, :]
    coords = ((points[:, [2, 1, 0]] - pc_range[[2, 1, 0]]) / voxel_size[[2, 1, 0]]).to(torch.int64)
    uniq, inv = coords.unique(return_inverse=True, dim=0)
    sums = torch.zeros(len(uniq), points.shape[1], dtype=points.dtype)
    sums.index_add_(0, inv, points)                      # sequential in point order, like scatter_add_ on CPU
    cnt = torch.zeros(len(uniq), dtype=points.dtype)
    cnt.index_add_(0, inv, torch.ones(len(points), dtype=points.dtype))
    return sums / cnt.clamp(min=1).unsqueeze(1), uniq


def batch_coords(coords_list):
    return torch.cat([torch.nn.functional.pad(c, (1, 0), value=i) for i, c in enumerate(coords_list)], 0)


def voxels_to_dense(voxels, coords, dims_zyx):
    """This repo's own step: voxel means scattered into a dense [Z,Y,X,C] grid + occupancy (voxels outside the grid --
    a point exactly on the upper range bound -- are dropped)."""
    z, y, x = dims_zyx
    grid = torch.zeros(z, y, x, voxels.shape[1])
    occ = torch.zeros(z, y, x, dtype=torch.uint8)
    ok = (coords[:, 0] < z) & (coords[:, 1] < y) & (coords[:, 2] < x)
    c = coords[ok]
    grid[c[:, 0], c[:, 1], c[:, 2]] = voxels[ok]
    occ[c[:, 0], c[:, 1], c[:, 2]] = 1
    return grid, occ
