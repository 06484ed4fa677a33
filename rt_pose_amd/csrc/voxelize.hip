// LiDAR stream, device side: the LiDAR -> radar extrinsic transform of
// det3d/datasets/pipelines/pose.py:34-38 (Preprocess) and the dynamic voxelisation of
// det3d/models/readers/dynamic_voxel_encoder.py:8-19 (`voxelization`: range filter, voxel coordinates, torch.unique over
// (z,y,x) and scatter_mean of every point feature), plus a scatter of the voxel means into the dense radar grid for fusion.
//
// The reference's unique + scatter_mean is atomics-based on a GPU; here points are radix-sorted by their voxel key (stable,
// so every voxel's points stay in input order) and each voxel's mean is a sequential sum over its run -- the result is the
// sorted unique order torch.unique produces, bit-identical to the reference's CPU arithmetic, and reproducible.
// HBM-bound integer/byte work: one key pass, one sort (rocPRIM), one scan, one gather pass.
#include "rtp_common.h"
#include "rtp_prof.h"

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

// ------------------------------------------------------------------------------------------------
// rtp_lidar_transform : points[:, :3] = (P_L2R @ [x, y, z, 1])[:3], float64 arithmetic, one rounding to fp32
// ------------------------------------------------------------------------------------------------
struct XformParams { float* pts; int n, c; double P[12]; };

__global__ __launch_bounds__(256) void lidar_transform_kernel(XformParams p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.n) return;
  float* q = p.pts + (long)i * p.c;
  const double x = q[0], y = q[1], z = q[2];
  const double rx = p.P[0] * x + p.P[1] * y + p.P[2] * z + p.P[3];
  const double ry = p.P[4] * x + p.P[5] * y + p.P[6] * z + p.P[7];
  const double rz = p.P[8] * x + p.P[9] * y + p.P[10] * z + p.P[11];
  q[0] = (float)rx; q[1] = (float)ry; q[2] = (float)rz;
}

extern "C" int rtp_lidar_transform(float* points, int n, int c, const double* P_L2R /*host, first 3 rows of the 4x4*/,
                                   void* stream) {
  if (!points || !P_L2R || n < 0 || c < 3) return RTP_ERR_SHAPE;
  if (n == 0) return RTP_OK;
  XformParams p;
  p.pts = points; p.n = n; p.c = c;
  for (int i = 0; i < 12; ++i) p.P[i] = P_L2R[i];
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_POINTWISE, s);
  hipLaunchKernelGGL(lidar_transform_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

// ------------------------------------------------------------------------------------------------
// rtp_dynamic_voxelize
// ------------------------------------------------------------------------------------------------
struct VoxParams {
  const float* pts; int n, c;
  float rmin[3], rmax[3], vs[3];   // x, y, z
};

#define VOX_INVALID 0xFFFFFFFFFFFFFFFFull

// key = (z << 42) | (y << 21) | x of the truncated voxel coordinate: sorting by key is the (z,y,x) lexicographic order of
// torch.unique(dim=0) on the reference's coords[:, (z,y,x)]
__global__ __launch_bounds__(256) void vox_key_kernel(VoxParams p, unsigned long long* keys, unsigned* idx) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.n) return;
  const float* q = p.pts + (long)i * p.c;
  const float x = q[0], y = q[1], z = q[2];
  const bool keep = x >= p.rmin[0] && x <= p.rmax[0] && y >= p.rmin[1] && y <= p.rmax[1] && z >= p.rmin[2] && z <= p.rmax[2];
  unsigned long long k = VOX_INVALID;
  if (keep) {
    // ((p - min) / voxel_size).to(int64): fp32 subtract, correctly rounded fp32 divide, truncation
    const long long cx = (long long)((x - p.rmin[0]) / p.vs[0]);
    const long long cy = (long long)((y - p.rmin[1]) / p.vs[1]);
    const long long cz = (long long)((z - p.rmin[2]) / p.vs[2]);
    k = ((unsigned long long)cz << 42) | ((unsigned long long)cy << 21) | (unsigned long long)cx;
  }
  keys[i] = k;
  idx[i] = (unsigned)i;
}

__global__ __launch_bounds__(256) void vox_head_kernel(const unsigned long long* keys, int n, unsigned* head) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const unsigned long long k = keys[i];
  head[i] = (k != VOX_INVALID && (i == 0 || keys[i - 1] != k)) ? 1u : 0u;
}

// seg[v] = first sorted position of voxel v; seg[nv] = number of kept points; *num_voxels = nv
__global__ __launch_bounds__(256) void vox_seg_kernel(const unsigned long long* keys, const unsigned* head, const unsigned* vid,
                                                      int n, unsigned* seg, int* num_voxels) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (head[i]) seg[vid[i]] = (unsigned)i;
  const bool valid = keys[i] != VOX_INVALID;
  const bool last_valid = valid && (i == n - 1 || keys[i + 1] == VOX_INVALID);
  if (last_valid) {
    const unsigned nv = vid[i] + head[i];   // exclusive scan value + own flag = voxels up to and including this point
    seg[nv] = (unsigned)(i + 1);
    *num_voxels = (int)nv;
  }
  if (i == 0 && !valid) *num_voxels = 0;    // no point survived the range filter
}

// one thread per (voxel, channel): sequential sum over the voxel's run (input order), then / count
__global__ __launch_bounds__(256) void vox_mean_kernel(const float* pts, int c, const unsigned long long* keys,
                                                       const unsigned* idx, const unsigned* seg, const int* num_voxels,
                                                       float* voxels, long long* coords) {
  const int nv = *num_voxels;
  const long t = blockIdx.x * 256L + threadIdx.x;
  const int v = (int)(t / c), ch = (int)(t % c);
  if (v >= nv) return;
  const unsigned a = seg[v], b = seg[v + 1];
  float acc = 0.f;
  for (unsigned i = a; i < b; ++i) acc += pts[(long)idx[i] * c + ch];
  voxels[(long)v * c + ch] = acc / (float)(b - a);
  if (ch == 0) {
    const unsigned long long k = keys[a];
    coords[(long)v * 3 + 0] = (long long)(k >> 42);
    coords[(long)v * 3 + 1] = (long long)((k >> 21) & 0x1FFFFF);
    coords[(long)v * 3 + 2] = (long long)(k & 0x1FFFFF);
  }
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct VoxWs {
  unsigned long long *k0, *k1; unsigned *i0, *i1, *head, *vid, *seg; void* tmp; size_t tmp_bytes; size_t total;
};

static int vox_layout(int n, void* ws, VoxWs& w) {
  size_t sort_tmp = 0, scan_tmp = 0;
  unsigned long long* kd = nullptr;
  unsigned* vd = nullptr;
  if (rocprim::radix_sort_pairs(nullptr, sort_tmp, kd, kd, vd, vd, (size_t)n, 0, 64, (hipStream_t)0) != hipSuccess) return RTP_ERR_LAUNCH;
  if (rocprim::exclusive_scan(nullptr, scan_tmp, vd, vd, 0u, (size_t)n, rocprim::plus<unsigned>(), (hipStream_t)0) != hipSuccess)
    return RTP_ERR_LAUNCH;
  w.tmp_bytes = sort_tmp > scan_tmp ? sort_tmp : scan_tmp;
  char* b = (char*)ws;
  size_t o = 0;
  w.k0 = (unsigned long long*)(b + o); o += align256(sizeof(unsigned long long) * n);
  w.k1 = (unsigned long long*)(b + o); o += align256(sizeof(unsigned long long) * n);
  w.i0 = (unsigned*)(b + o); o += align256(sizeof(unsigned) * n);
  w.i1 = (unsigned*)(b + o); o += align256(sizeof(unsigned) * n);
  w.head = (unsigned*)(b + o); o += align256(sizeof(unsigned) * n);
  w.vid = (unsigned*)(b + o); o += align256(sizeof(unsigned) * n);
  w.seg = (unsigned*)(b + o); o += align256(sizeof(unsigned) * ((size_t)n + 1));
  w.tmp = b + o; o += align256(w.tmp_bytes);
  w.total = o;
  return RTP_OK;
}

extern "C" long rtp_voxelize_workspace_bytes(int n) {
  if (n < 1) return 256;
  VoxWs w;
  if (vox_layout(n, nullptr, w) != RTP_OK) return -1;
  return (long)w.total;
}

extern "C" int rtp_dynamic_voxelize(const float* points, int n, int c, const float* pc_range /*host [6]*/,
                                    const float* voxel_size /*host [3]*/, float* voxels, long long* coords, int* num_voxels,
                                    void* workspace, long ws_bytes, void* stream) {
  if (!points || !pc_range || !voxel_size || !voxels || !coords || !num_voxels || n < 0 || c < 3) return RTP_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_POINTWISE, s);
  if (n == 0) {
    if (hipMemsetAsync(num_voxels, 0, sizeof(int), s) != hipSuccess) return RTP_ERR_LAUNCH;
    return RTP_OK;
  }
  VoxParams p;
  p.pts = points; p.n = n; p.c = c;
  for (int i = 0; i < 3; ++i) {
    p.rmin[i] = pc_range[i]; p.rmax[i] = pc_range[3 + i]; p.vs[i] = voxel_size[i];
    if (!(p.vs[i] > 0.f)) return RTP_ERR_SHAPE;
    if ((double)(p.rmax[i] - p.rmin[i]) / p.vs[i] >= (double)(1 << 21)) return RTP_ERR_UNSUPPORTED;   // 21 bits per axis in the key
  }
  VoxWs w;
  int rc = vox_layout(n, workspace, w);
  if (rc) return rc;
  if (!workspace || (size_t)ws_bytes < w.total) return RTP_ERR_SHAPE;
  const int blocks = (n + 255) / 256;
  hipLaunchKernelGGL(vox_key_kernel, dim3(blocks), dim3(256), 0, s, p, w.k0, w.i0);
  size_t tb = w.tmp_bytes;
  if (rocprim::radix_sort_pairs(w.tmp, tb, w.k0, w.k1, w.i0, w.i1, (size_t)n, 0, 64, s) != hipSuccess) return RTP_ERR_LAUNCH;
  hipLaunchKernelGGL(vox_head_kernel, dim3(blocks), dim3(256), 0, s, w.k1, n, w.head);
  tb = w.tmp_bytes;
  if (rocprim::exclusive_scan(w.tmp, tb, w.head, w.vid, 0u, (size_t)n, rocprim::plus<unsigned>(), s) != hipSuccess) return RTP_ERR_LAUNCH;
  hipLaunchKernelGGL(vox_seg_kernel, dim3(blocks), dim3(256), 0, s, w.k1, w.head, w.vid, n, w.seg, num_voxels);
  const long threads = (long)n * c;   // upper bound on voxels * c; the kernel reads the real count
  hipLaunchKernelGGL(vox_mean_kernel, dim3((int)((threads + 255) / 256)), dim3(256), 0, s, points, c, w.k1, w.i1, w.seg,
                     num_voxels, voxels, coords);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

// ------------------------------------------------------------------------------------------------
// rtp_voxels_to_dense : grid[z][y][x][c] = voxel mean (0 where empty), occ[z][y][x] = 1/0 ; voxels outside the grid skipped
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vox_zero_kernel(float* grid, unsigned char* occ, long cells, int c) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < cells * c; i += (long)gridDim.x * 256) grid[i] = 0.f;
  if (occ)
    for (long i = blockIdx.x * 256L + threadIdx.x; i < cells; i += (long)gridDim.x * 256) occ[i] = 0;
}

__global__ __launch_bounds__(256) void vox_dense_kernel(const float* voxels, const long long* coords, const int* num_voxels,
                                                        int c, int Z, int Y, int X, float* grid, unsigned char* occ) {
  const int nv = *num_voxels;
  const long t = blockIdx.x * 256L + threadIdx.x;
  const int v = (int)(t / c), ch = (int)(t % c);
  if (v >= nv) return;
  const long long z = coords[(long)v * 3], y = coords[(long)v * 3 + 1], x = coords[(long)v * 3 + 2];
  if (z < 0 || z >= Z || y < 0 || y >= Y || x < 0 || x >= X) return;   // a point exactly on the upper range bound
  const long cell = ((long)z * Y + y) * X + x;
  grid[cell * c + ch] = voxels[(long)v * c + ch];
  if (occ && ch == 0) occ[cell] = 1;
}

extern "C" int rtp_voxels_to_dense(const float* voxels, const long long* coords, const int* num_voxels, int max_voxels, int c,
                                   int Z, int Y, int X, float* grid, unsigned char* occ, void* stream) {
  if (!voxels || !coords || !num_voxels || !grid || max_voxels < 0 || c < 1 || Z < 1 || Y < 1 || X < 1) return RTP_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_POINTWISE, s);
  const long cells = (long)Z * Y * X;
  long zb = (cells * c + 255) / 256;
  if (zb > 2048) zb = 2048;
  hipLaunchKernelGGL(vox_zero_kernel, dim3((int)zb), dim3(256), 0, s, grid, occ, cells, c);
  if (max_voxels > 0) {
    const long threads = (long)max_voxels * c;
    hipLaunchKernelGGL(vox_dense_kernel, dim3((int)((threads + 255) / 256)), dim3(256), 0, s, voxels, coords, num_voxels, c, Z, Y,
                       X, grid, occ);
  }
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}
