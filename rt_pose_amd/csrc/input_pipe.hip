// Device-side input pipeline of the radar stream: what the reference does in NumPy inside DataLoader
// workers -- det3d/datasets/cruw_pose/cruw_pose.py:167-194 (fp16 cube -> fp32, ROI crop, normalise, clamp) and
// det3d/datasets/pipelines/pose.py:186-254 / 386-451 (AssignLabelPose / AssignLabelPose2: voxel coordinates, gaussian
// max-splat of det3d/core/utils/center_utils.py:67-91, ind / mask / cat / offsets) -- as two tiny HBM-bound kernels fed by
// one async H2D copy of the raw fp16 cubes and the key-point list.  The network input buffer and the label buffers of the
// training plan are written in place; heat-maps are never memset (79 MB per batch for hr3d): the boxes splatted for the
// previous batch are cleared first, from a small record kept on the device.
#include "rtp_common.h"
#include "rtp_prof.h"

#include <hip/hip_fp16.h>
#include <math.h>

// ------------------------------------------------------------------------------------------------
// rtp_cube_prep
// ------------------------------------------------------------------------------------------------
struct CubeParams {
  const __half* cube; float* out;
  long lead;             // frames * leading channels (1, D or 2*D): every "plane stack" is an independent [Zs][Ys][Xs] cube
  int Zs, Ys, Xs;        // stored cube dims
  int z0, y0, x0, Z, Y, X;  // ROI origin and cropped dims
  float lo, scale; int normalise;
};

__global__ __launch_bounds__(256) void cube_prep_kernel(CubeParams p) {
  const long total = p.lead * p.Z * p.Y * p.X;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int x = (int)(i % p.X);
    long r = i / p.X;
    const int y = (int)(r % p.Y);
    r /= p.Y;
    const int z = (int)(r % p.Z);
    const long l = r / p.Z;
    float v = __half2float(p.cube[((l * p.Zs + (p.z0 + z)) * p.Ys + (p.y0 + y)) * p.Xs + (p.x0 + x)]);
    if (p.normalise) {
      v = (v - p.lo) / p.scale;   // fp32, correctly rounded division: bit-identical to numpy's (a - lo) / scale
      v = v < 0.f ? 0.f : v;
    }
    p.out[i] = v;
  }
}

extern "C" int rtp_cube_prep(const void* cube_f16, long lead, int zs, int ys, int xs, const int* roi_zyx /*host [6]*/,
                             float norm_lo, float norm_hi, int normalise, float* out, void* stream) {
  if (!cube_f16 || !out || !roi_zyx || lead < 1) return RTP_ERR_SHAPE;
  CubeParams p;
  p.cube = (const __half*)cube_f16; p.out = out; p.lead = lead; p.Zs = zs; p.Ys = ys; p.Xs = xs;
  p.z0 = roi_zyx[0]; p.y0 = roi_zyx[2]; p.x0 = roi_zyx[4];
  p.Z = roi_zyx[1] - roi_zyx[0] + 1; p.Y = roi_zyx[3] - roi_zyx[2] + 1; p.X = roi_zyx[5] - roi_zyx[4] + 1;
  if (p.z0 < 0 || p.y0 < 0 || p.x0 < 0 || p.Z < 1 || p.Y < 1 || p.X < 1 || roi_zyx[1] >= zs || roi_zyx[3] >= ys || roi_zyx[5] >= xs)
    return RTP_ERR_SHAPE;
  p.lo = norm_lo; p.scale = norm_hi - norm_lo; p.normalise = normalise;
  if (normalise && p.scale == 0.f) return RTP_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_POINTWISE, s);
  const long total = lead * p.Z * p.Y * p.X;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(cube_prep_kernel, dim3((int)blocks), dim3(256), 0, s, p);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}

// ------------------------------------------------------------------------------------------------
// rtp_gaussian_table : host helper, center_utils.py:67-72 with shape (2r+1)^3, sigma = (2r+1)/6, float64 -> fp32
// ------------------------------------------------------------------------------------------------
extern "C" int rtp_gaussian_table(int radius, float* out_host) {
  if (radius < 0 || radius > 8 || !out_host) return RTP_ERR_SHAPE;
  const int d = 2 * radius + 1;
  const double sigma = (double)d / 6.0;
  const double den = pow(2.0 * sigma * sigma, 1.5);
  double mx = 0.0;
  for (int i = 0; i < d * d * d; ++i) {
    const int z = i / (d * d) - radius, y = (i / d) % d - radius, x = i % d - radius;
    const double h = exp(-(double)(x * x + y * y + z * z) / den);
    if (h > mx) mx = h;
  }
  const double eps = 2.220446049250313e-16 * mx;   // np.finfo(float64).eps * h.max()
  for (int i = 0; i < d * d * d; ++i) {
    const int z = i / (d * d) - radius, y = (i / d) % d - radius, x = i % d - radius;
    double h = exp(-(double)(x * x + y * y + z * z) / den);
    if (h < eps) h = 0.0;
    out_host[i] = (float)h;
  }
  return RTP_OK;
}

// ------------------------------------------------------------------------------------------------
// rtp_assign_labels
// ------------------------------------------------------------------------------------------------
struct LabelParams {
  const double* poses;     // [frames][max_in][15][3]  (x, y, z) metres
  const int* nposes;       // [frames]
  int frames, max_in, max_poses, one_hm, radius;
  double rmin[3];          // range minimum, (z, y, x)
  double vsize[3];         // voxel size, (x, y, z)
  double osf[3];           // out_size_factor, (z, y, x)
  int fz, fy, fx, ncls, m, width, legacy;
  const float* table;      // [(2r+1)^3] device
  float* hm; float* anno; long long* ind; unsigned char* mask; long long* cat;
  int* prev;               // [frames][m][4] = (valid, class, centre voxel index, -): boxes splatted by the previous call
};

// pose.py:222-227: (x - radar_range[k]) / voxel_size[k] / out_size_factor[k] with x a Python float, radar_range an
// np.float32 array, voxel_size Python floats.  NumPy >= 2 (NEP 50: Python scalars are weakly typed) evaluates every step
// in fp32 -- that is what the captured vectors pin.  legacy: NumPy 1.x value-based promotion, float64 intermediate with
// one rounding to fp32.  The two differ by <= 1 ulp, which flips the integer voxel of a key-point on a voxel boundary.
__device__ __forceinline__ float vox_coord(double p, double rmin, double vs, double osf, int legacy) {
  if (legacy) return (float)(((p - rmin) / vs) / osf);
  float c = (float)p - (float)rmin;
  c = c / (float)vs;
  return c / (float)osf;
}

// pass 0: clear the boxes of the previous call; pass 1: this call's slots.  One block per (frame, slot).
__global__ __launch_bounds__(128) void assign_labels_kernel(LabelParams p, int pass) {
  const int f = blockIdx.x / p.m, k = blockIdx.x - f * p.m, tid = threadIdx.x;
  int* prev = p.prev + ((long)f * p.m + k) * 4;
  const int d = 2 * p.radius + 1;
  const long vox = (long)p.fz * p.fy * p.fx;
  if (pass == 0) {
    if (!prev[0]) return;
    const int cls = prev[1], c = prev[2];
    const int cz = c / (p.fy * p.fx), cy = (c / p.fx) % p.fy, cx = c % p.fx;
    float* hm = p.hm + ((long)f * p.ncls + cls) * vox;
    for (int i = tid; i < d * d * d; i += 128) {
      const int z = cz + i / (d * d) - p.radius, y = cy + (i / d) % d - p.radius, x = cx + i % d - p.radius;
      if (z >= 0 && z < p.fz && y >= 0 && y < p.fy && x >= 0 && x < p.fx) hm[((long)z * p.fy + y) * p.fx + x] = 0.f;
    }
    return;
  }
  // slot k: AssignLabelPose -> key-point (k % 15) of pose (k / 15); AssignLabelPose2 -> pose k, centre = key-point 0
  const int np_ = p.nposes[f];
  const int pose = p.one_hm ? k : k / 15, kp = p.one_hm ? 0 : k % 15;
  const int cls = p.one_hm ? 0 : kp;
  const bool present = pose < np_ && pose < p.max_poses && pose < p.max_in;
  float* anno = p.anno + ((long)f * p.m + k) * p.width;
  bool ok = false;
  int ix = 0, iy = 0, iz = 0;
  float cx = 0.f, cy = 0.f, cz = 0.f;
  const double* q = p.poses + (((long)f * p.max_in + (present ? pose : 0)) * 15 + kp) * 3;
  if (present) {
    cx = vox_coord(q[0], p.rmin[2], p.vsize[0], p.osf[2], p.legacy);
    cy = vox_coord(q[1], p.rmin[1], p.vsize[1], p.osf[1], p.legacy);
    cz = vox_coord(q[2], p.rmin[0], p.vsize[2], p.osf[0], p.legacy);
    ix = (int)cx; iy = (int)cy; iz = (int)cz;   // astype(int32): truncation toward zero
    ok = ix >= 0 && ix < p.fx && iy >= 0 && iy < p.fy && iz >= 0 && iz < p.fz;
  }
  if (tid == 0) {
    p.ind[(long)f * p.m + k] = ok ? ((long long)iz * p.fy + iy) * p.fx + ix : 0;
    p.mask[(long)f * p.m + k] = ok ? 1 : 0;
    p.cat[(long)f * p.m + k] = ok ? cls : 0;
    prev[0] = ok; prev[1] = cls; prev[2] = (iz * p.fy + iy) * p.fx + ix; prev[3] = 0;
  }
  if (!p.one_hm) {
    if (tid < 3) anno[tid] = ok ? (tid == 0 ? cx - (float)ix : tid == 1 ? cy - (float)iy : cz - (float)iz) : 0.f;
  } else {
    // every key-point's offset from the CENTRE's integer voxel
    for (int i = tid; i < 45; i += 128) {
      float v = 0.f;
      if (ok) {
        const int j = i / 3, a = i % 3;
        const double* qq = p.poses + (((long)f * p.max_in + pose) * 15 + j) * 3;
        const float c = a == 0 ? vox_coord(qq[0], p.rmin[2], p.vsize[0], p.osf[2], p.legacy)
                      : a == 1 ? vox_coord(qq[1], p.rmin[1], p.vsize[1], p.osf[1], p.legacy)
                               : vox_coord(qq[2], p.rmin[0], p.vsize[2], p.osf[0], p.legacy);
        v = c - (float)(a == 0 ? ix : a == 1 ? iy : iz);
      }
      anno[i] = v;
    }
  }
  if (!ok) return;
  float* hm = p.hm + ((long)f * p.ncls + cls) * vox;
  for (int i = tid; i < d * d * d; i += 128) {
    const int z = iz + i / (d * d) - p.radius, y = iy + (i / d) % d - p.radius, x = ix + i % d - p.radius;
    if (z >= 0 && z < p.fz && y >= 0 && y < p.fy && x >= 0 && x < p.fx)
      // np.maximum: values are >= 0, so the integer order of the bit patterns is the float order
      atomicMax(reinterpret_cast<int*>(hm + ((long)z * p.fy + y) * p.fx + x), __float_as_int(p.table[i]));
  }
}

extern "C" int rtp_assign_labels(const double* poses, const int* nposes, int frames, int max_in, int max_poses, int one_hm,
                                 int radius, const double* range_min_zyx /*host [3]*/, const double* voxel_size_xyz /*host [3]*/,
                                 const int* out_size_factor_zyx /*host [3]*/, int fz, int fy, int fx, const float* table,
                                 float* hm, float* anno, long long* ind, unsigned char* mask, long long* cat, int* prev,
                                 int numpy_legacy, void* stream) {
  if (!poses || !nposes || !table || !hm || !anno || !ind || !mask || !cat || !prev || !range_min_zyx || !voxel_size_xyz ||
      !out_size_factor_zyx)
    return RTP_ERR_SHAPE;
  if (frames < 1 || max_in < 1 || max_poses < 1 || radius < 0 || radius > 2 || fz < 1 || fy < 1 || fx < 1) return RTP_ERR_SHAPE;
  LabelParams p;
  p.poses = poses; p.nposes = nposes; p.frames = frames; p.max_in = max_in; p.max_poses = max_poses; p.one_hm = one_hm;
  p.radius = radius;
  for (int i = 0; i < 3; ++i) {
    p.rmin[i] = range_min_zyx[i]; p.vsize[i] = voxel_size_xyz[i]; p.osf[i] = (double)out_size_factor_zyx[i];
    if (p.vsize[i] == 0.0 || p.osf[i] == 0.0) return RTP_ERR_SHAPE;
  }
  p.fz = fz; p.fy = fy; p.fx = fx; p.legacy = numpy_legacy;
  p.ncls = one_hm ? 1 : 15; p.m = one_hm ? max_poses : 15 * max_poses; p.width = one_hm ? 45 : 3;
  p.table = table; p.hm = hm; p.anno = anno; p.ind = ind; p.mask = mask; p.cat = cat; p.prev = prev;
  hipStream_t s = (hipStream_t)stream;
  RtpProfScope prof(RTP_FAM_LOSS, s);
  hipLaunchKernelGGL(assign_labels_kernel, dim3(frames * p.m), dim3(128), 0, s, p, 0);
  hipLaunchKernelGGL(assign_labels_kernel, dim3(frames * p.m), dim3(128), 0, s, p, 1);
  RTP_CHECK_LAUNCH();
  return RTP_OK;
}
