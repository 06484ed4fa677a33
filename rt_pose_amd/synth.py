"""Synthetic radar frames + CenterNet-style targets with the dataset's native layout.

Input recipe: relu(randn*0.5+0.1) -- non-negative with ~42 % zeros, like the normalised/clamped cube
(datasets/cruw_pose.py:173-183).  Targets follow AssignLabelPose / AssignLabelPose2
(datasets/pipelines/pose.py:206-254, 407-451): one pose per frame, a radius-r "gaussian" splat per key-point
with the reference's exp(-r^2/(2 sigma^2)^1.5) profile (core/utils/center_utils.py:67-72, sic), flat voxel index,
mask=1, cat=arange.
"""
import torch


def splat_profile(radius):
    d = 2 * radius + 1
    sigma = d / 6.0
    r = torch.arange(-radius, radius + 1, dtype=torch.float64)
    rr = r.view(-1, 1, 1) ** 2 + r.view(1, -1, 1) ** 2 + r.view(1, 1, -1) ** 2
    g = torch.exp(-rr / (2 * sigma * sigma) ** 1.5)
    g[g < torch.finfo(torch.float64).eps * g.max()] = 0
    return g.float()


def draw_splat(hm, cz, cy, cx, radius, prof=None):
    """element-wise max of the heat-map [Z,Y,X] with the splat centred at an integer voxel, clipped at the borders."""
    prof = splat_profile(radius) if prof is None else prof
    Z, Y, X = hm.shape
    z0, z1 = min(cz, radius), min(Z - cz, radius + 1)
    y0, y1 = min(cy, radius), min(Y - cy, radius + 1)
    x0, x1 = min(cx, radius), min(X - cx, radius + 1)
    if min(z0 + z1, y0 + y1, x0 + x1) <= 0:
        return hm
    view = hm[cz - z0:cz + z1, cy - y0:cy + y1, cx - x0:cx + x1]
    torch.maximum(view, prof[radius - z0:radius + z1, radius - y0:radius + y1, radius - x0:radius + x1], out=view)
    return hm


def lidar_grid(batch, channels, dims, seed, occupancy=0.06):
    """A dense LiDAR voxel grid [B, C_l, Z, Y, X] of the sparsity a real sweep leaves in the radar ROI: `occupancy` of the voxels
    hold mean point features (metric x, y, z inside the voxel's range + an intensity), the rest are zeros."""
    Z, Y, X = dims
    g = torch.Generator().manual_seed(seed + 77)
    occ = (torch.rand(batch, 1, Z, Y, X, generator=g) < occupancy).float()
    feat = torch.rand(batch, channels, Z, Y, X, generator=g) * 2 - 1
    return feat * occ


def make_batch(batch, cin, dims, seed, one_hm=False, rank=0, lidar_channels=0):
    """Returns the reference's collated `example` dict (datasets/cruw_pose.py:225-275) on CPU."""
    Z, Y, X = dims
    g = torch.Generator().manual_seed(seed + rank)
    rdr = torch.relu(torch.randn(batch, cin, Z, Y, X, generator=g) * 0.5 + 0.1)
    ncls, nreg, radius = (1, 45, 2) if one_hm else (15, 3, 1)
    prof = splat_profile(radius)
    hm = torch.zeros(batch, ncls, Z, Y, X)
    ind = torch.zeros(batch, ncls, dtype=torch.int64)
    for b in range(batch):
        for c in range(ncls):
            cz = int(torch.randint(0, Z, (1,), generator=g))
            cy = int(torch.randint(0, Y, (1,), generator=g))
            cx = int(torch.randint(0, X, (1,), generator=g))
            draw_splat(hm[b, c], cz, cy, cx, radius, prof)
            ind[b, c] = (cz * Y + cy) * X + cx
    pose = (torch.rand(batch, 1, 45, generator=g) * 16 - 8) if one_hm else torch.rand(batch, 15, 3, generator=g)
    ex = dict(rdr_tensor=rdr, hm=[hm], ind=[ind], mask=[torch.ones(batch, ncls, dtype=torch.uint8)],
              cat=[torch.arange(ncls).repeat(batch, 1)], anno_pose=[pose])
    if lidar_channels:
        ex["lidar_grid"] = lidar_grid(batch, lidar_channels, dims, seed + rank)
    return {"rdr": ex, "meta": [{"seq": "synth", "frame": b, "rdr_frame": b} for b in range(batch)]}
