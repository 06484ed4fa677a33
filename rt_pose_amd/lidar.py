"""LiDAR stream on the device: extrinsic transform, dynamic voxelisation, dense scatter.

Mirrors `DynamicVoxelEncoder` (det3d/models/readers/dynamic_voxel_encoder.py:69-101: pc_range / voxel_size, forward(points:
list of [N_i, C] tensors) -> (voxels_batch, coors_batch [V,4] = (batch, z, y, x), grid shape (x, y, z))) on top of
csrc/voxelize.hip; registered under the same READERS name by rt_pose_amd.modules.  No CPU fallback."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check


class DynamicVoxelEncoder(torch.nn.Module):
    def __init__(self, pc_range, voxel_size, virtual=False):
        super().__init__()
        if virtual:
            raise NotImplementedError("voxelization_virtual (painted / virtual nuScenes points) is not part of the radar-pose path")
        self.pc_range = torch.tensor(pc_range)
        self.voxel_size = torch.tensor(voxel_size)
        self.shape = torch.round((self.pc_range[3:] - self.pc_range[:3]) / self.voxel_size)
        self.shape_np = self.shape.numpy().astype(np.int32)
        self._pr = (C.c_float * 6)(*[float(v) for v in pc_range])
        self._vs = (C.c_float * 3)(*[float(v) for v in voxel_size])
        self._ws = None

    def voxelize(self, points):
        """One frame: points [N, C] fp32 on the GPU -> (voxels [V, C], coords [V, 3] int64 (z,y,x))."""
        if not points.is_cuda:
            raise NotImplementedError("rt_pose_amd has no CPU path")
        lib = _lib.load()
        pts = points.contiguous().float()
        n, c = pts.shape
        if n == 0:
            return pts.new_zeros((0, c)), torch.zeros(0, 3, dtype=torch.int64, device=pts.device)
        need = int(lib.rtp_voxelize_workspace_bytes(max(n, 1)))
        if need < 0:
            raise _lib.RtpError("rtp_voxelize_workspace_bytes failed")
        if self._ws is None or self._ws.numel() < need or self._ws.device != pts.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=pts.device)
        voxels = torch.empty(max(n, 1), c, dtype=torch.float32, device=pts.device)
        coords = torch.empty(max(n, 1), 3, dtype=torch.int64, device=pts.device)
        nv = torch.zeros(1, dtype=torch.int32, device=pts.device)
        s = C.c_void_p(torch.cuda.current_stream(pts.device).cuda_stream)
        check(lib.rtp_dynamic_voxelize(C.c_void_p(pts.data_ptr()), n, c, self._pr, self._vs, C.c_void_p(voxels.data_ptr()),
                                       C.c_void_p(coords.data_ptr()), C.c_void_p(nv.data_ptr()), C.c_void_p(self._ws.data_ptr()),
                                       need, s), "rtp_dynamic_voxelize")
        k = int(nv.item())     # the reference returns exactly-sized tensors, which needs the count on the host
        return voxels[:k], coords[:k]

    @torch.no_grad()
    def forward(self, points):
        voxels, coors = [], []
        for i, res in enumerate(points):
            v, c = self.voxelize(res)
            voxels.append(v)
            coors.append(torch.nn.functional.pad(c, (1, 0), mode="constant", value=i))
        return torch.cat(voxels, 0), torch.cat(coors, 0), self.shape_np

    def to_dense(self, voxels, coords):
        """Voxel means of one frame scattered into [Z, Y, X, C] + occupancy (this repo's fusion input)."""
        lib = _lib.load()
        x, y, z = [int(v) for v in self.shape_np]
        c = voxels.shape[1]
        if voxels.shape[0] == 0:   # nothing to scatter (and nothing to point the kernel at)
            return (torch.zeros(z, y, x, c, dtype=torch.float32, device=voxels.device),
                    torch.zeros(z, y, x, dtype=torch.uint8, device=voxels.device))
        grid = torch.empty(z, y, x, c, dtype=torch.float32, device=voxels.device)
        occ = torch.empty(z, y, x, dtype=torch.uint8, device=voxels.device)
        nv = torch.tensor([voxels.shape[0]], dtype=torch.int32, device=voxels.device)
        v, co = voxels.contiguous(), coords.contiguous()
        s = C.c_void_p(torch.cuda.current_stream(voxels.device).cuda_stream)
        check(lib.rtp_voxels_to_dense(C.c_void_p(v.data_ptr()), C.c_void_p(co.data_ptr()), C.c_void_p(nv.data_ptr()),
                                      voxels.shape[0], c, z, y, x, C.c_void_p(grid.data_ptr()), C.c_void_p(occ.data_ptr()), s),
              "rtp_voxels_to_dense")
        return grid, occ


def lidar_to_radar(points, P_L2R):
    """In place on the GPU: points[:, :3] <- (P_L2R @ [x, y, z, 1])[:3] (pipelines/pose.py:34-38)."""
    if not points.is_cuda or points.dtype != torch.float32 or not points.is_contiguous():
        raise NotImplementedError("lidar_to_radar needs a contiguous fp32 CUDA tensor (no CPU path)")
    P = (C.c_double * 12)(*[float(v) for v in np.asarray(P_L2R, np.float64)[:3].reshape(-1)])
    s = C.c_void_p(torch.cuda.current_stream(points.device).cuda_stream)
    check(_lib.load().rtp_lidar_transform(C.c_void_p(points.data_ptr()), points.shape[0], points.shape[1], P, s), "rtp_lidar_transform")
    return points
