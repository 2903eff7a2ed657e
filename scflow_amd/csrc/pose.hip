// Pose-induced flow ("shape constraint" re-projection), dense un-projection and the pose
// update of the SCFlow decoder loop, for gfx950.
//
// The reference compacts foreground pixels with torch.nonzero (a device->host sync and a
// variable-length python list per sample, models/utils/pose.py:44-64) and scatters the
// projected flow back (:66-88).  Scatter targets are exactly the compacted pixels, so the
// computation is an independent per-pixel map: this file keeps it dense (depth > 0 mask),
// which removes the sync, the index tensors and the python loop over the batch.
#include "scf_common.h"

// 3x3 inverse in fp64 (adjugate), rounded to fp32.  The reference uses torch.inverse (fp32
// LU); both are within a few fp32 ulp of the true inverse.
__device__ inline void inv3x3(const float* m, float* o) {
  const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7],
               i = m[8];
  const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  const double det = a * A + b * B + c * C;
  const double id = 1.0 / det;
  o[0] = (float)(A * id);
  o[1] = (float)(-(b * i - c * h) * id);
  o[2] = (float)((b * f - c * e) * id);
  o[3] = (float)(B * id);
  o[4] = (float)((a * i - c * g) * id);
  o[5] = (float)(-(a * f - c * d) * id);
  o[6] = (float)(C * id);
  o[7] = (float)(-(a * h - b * g) * id);
  o[8] = (float)((a * e - b * d) * id);
}

struct PoseMats {
  float Kinv[9], R0inv[9], t0[3], K[9], R[9], t[3];
};

__device__ inline void load_mats(PoseMats* s, const float* K, const float* R0, const float* t0,
                                 const float* R, const float* t, int n) {
  const int tid = threadIdx.x;
  if (tid == 0) inv3x3(K + 9 * n, s->Kinv);
  if (tid == 64) inv3x3(R0 + 9 * n, s->R0inv);
  if (tid >= 128 && tid < 137) {
    s->K[tid - 128] = K[9 * n + tid - 128];
    if (R) s->R[tid - 128] = R[9 * n + tid - 128];
  }
  if (tid >= 192 && tid < 195) {
    s->t0[tid - 192] = t0[3 * n + tid - 192];
    if (t) s->t[tid - 192] = t[3 * n + tid - 192];
  }
  __syncthreads();
}

// object-frame point of pixel (x, y) with depth d: lift_2d_to_3d, pose.py:26-41
__device__ __forceinline__ void unproject(const PoseMats& s, float x, float y, float d, float& X,
                                          float& Y, float& Z) {
  const float hx = x * d, hy = y * d, hz = d;
  const float cx = s.Kinv[0] * hx + s.Kinv[1] * hy + s.Kinv[2] * hz - s.t0[0];
  const float cy = s.Kinv[3] * hx + s.Kinv[4] * hy + s.Kinv[5] * hz - s.t0[1];
  const float cz = s.Kinv[6] * hx + s.Kinv[7] * hy + s.Kinv[8] * hz - s.t0[2];
  X = s.R0inv[0] * cx + s.R0inv[1] * cy + s.R0inv[2] * cz;
  Y = s.R0inv[3] * cx + s.R0inv[4] * cy + s.R0inv[5] * cz;
  Z = s.R0inv[6] * cx + s.R0inv[7] * cy + s.R0inv[8] * cz;
}

// r5: the pose update of the iteration (one thread per sample: a launch of its own, 4.8 us + a launch boundary per
// iteration) can ride in this kernel: with `pu.rot_all` set, one thread of EVERY block of sample n runs the update of that
// sample (pose_update_one: the function the stand-alone kernel calls) into LDS, the x = 0 block also stores the results.
struct PoseOut { float d_rot[6], d_trans[3], R[9], t[3]; };
struct PoseUpdateArgs {
  const float* rot_all; const float* trans_all; const long long* label; int num_class, label_mode;
  const float* R_in; const float* t_in; float* d_rot; float* d_trans; float* R_out; float* t_out;
};
__device__ __noinline__ void pose_update_one(const float* __restrict__ rot_all, const float* __restrict__ trans_all,
                                             const long long* __restrict__ label, int num_class, int label_mode,
                                             const float* R_in, const float* t_in, int n, PoseOut* o);
__device__ __forceinline__ void pose_store(const PoseOut& o, int n, float* __restrict__ d_rot, float* __restrict__ d_trans,
                                           float* R_out, float* t_out);

__global__ __launch_bounds__(256) void reproject_flow_kernel(
    const float* __restrict__ depth, const float* __restrict__ K, const float* __restrict__ R0,
    const float* __restrict__ t0, const float* R, const float* t,
    float* __restrict__ flow, int H, int W, float invalid, PoseUpdateArgs pu) {
  __shared__ PoseMats s;
  __shared__ PoseOut po;
  const int n = blockIdx.y;
  const bool fused = pu.rot_all != nullptr;
  if (fused && threadIdx.x == 224) {
    PoseOut o;
    pose_update_one(pu.rot_all, pu.trans_all, pu.label, pu.num_class, pu.label_mode, pu.R_in, pu.t_in, n, &o);
    po = o;
    if (blockIdx.x == 0) pose_store(o, n, pu.d_rot, pu.d_trans, pu.R_out, pu.t_out);
  }
  load_mats(&s, K, R0, t0, fused ? nullptr : R, fused ? nullptr : t, n);
  if (fused) {
    if (threadIdx.x < 9) s.R[threadIdx.x] = po.R[threadIdx.x];
    if (threadIdx.x >= 64 && threadIdx.x < 67) s.t[threadIdx.x - 64] = po.t[threadIdx.x - 64];
    __syncthreads();
  }
  const int hw = H * W;
  const float* dp = depth + (long long)n * hw;
  float* fx = flow + (long long)n * 2 * hw;
  float* fy = fx + hw;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < hw; idx += gridDim.x * blockDim.x) {
    const float d = dp[idx];
    float u = invalid, v = invalid;
    if (d > 0.f) {
      const int yi = idx / W, xi = idx - yi * W;
      const float x = (float)xi, y = (float)yi;
      float X, Y, Z;
      unproject(s, x, y, d, X, Y, Z);
      // p = K (R P + t): get_flow_from_delta_pose_and_points, pose.py:82-86
      const float px = s.R[0] * X + s.R[1] * Y + s.R[2] * Z + s.t[0];
      const float py = s.R[3] * X + s.R[4] * Y + s.R[5] * Z + s.t[1];
      const float pz = s.R[6] * X + s.R[7] * Y + s.R[8] * Z + s.t[2];
      const float qx = s.K[0] * px + s.K[1] * py + s.K[2] * pz;
      const float qy = s.K[3] * px + s.K[4] * py + s.K[5] * pz;
      const float qz = s.K[6] * px + s.K[7] * py + s.K[8] * pz;
      u = qx / qz - x;
      v = qy / qz - y;
    }
    fx[idx] = u;
    fy[idx] = v;
  }
}

extern "C" int scf_reproject_flow(const float* depth, const float* K, const float* R0,
                                  const float* t0, const float* R, const float* t, float* flow,
                                  int N, int H, int W, float invalid_num, scf_stream_t stream) {
  if (!depth || !K || !R0 || !t0 || !R || !t || !flow || N <= 0 || H <= 0 || W <= 0) return SCF_EINVAL;
  if (N > 65535) return SCF_EUNSUPPORTED;
  const int bx = (int)(scf_cdiv((int64_t)H * W, 256) < 64 ? scf_cdiv((int64_t)H * W, 256) : 64);
  const PoseUpdateArgs none = {};
  scf_launch(reproject_flow_kernel, dim3(bx, N), dim3(256), 0, scf_stream(stream), depth, K,
                     R0, t0, R, t, flow, H, W, invalid_num, none);
  return scf_launch_status();
}

int scf_pose_update_reproject(const float* rot_all, const float* trans_all, const int64_t* label, int num_class,
                              int label_mode, const float* R_in, const float* t_in, float* d_rot, float* d_trans,
                              float* R_out, float* t_out, const float* depth, const float* K, const float* R0,
                              const float* t0, float* flow, int N, int H, int W, float invalid_num, scf_stream_t stream) {
  if (!rot_all || !trans_all || !label || !R_in || !t_in || !d_rot || !d_trans || !R_out || !t_out || num_class <= 0 ||
      (label_mode & ~(SCF_POSE_LABEL_PER_SAMPLE | SCF_POSE_DEPTH_LINEAR)))
    return SCF_EINVAL;
  if (!depth || !K || !R0 || !t0 || !flow || N <= 0 || H <= 0 || W <= 0) return SCF_EINVAL;
  if (N > 65535) return SCF_EUNSUPPORTED;
  // an in-place update (R_out == R_in) would be read by the other blocks of the sample while block 0 writes it
  if (R_in == R_out || t_in == t_out) return SCF_EUNSUPPORTED;
  // blocks per sample: 64 at large batches (each walks its share of the pixels); a small batch spreads a sample over up to
  // two blocks per CU so that a pass of few samples is not 64 blocks of four pixels per thread behind one serial pose update
  int64_t cap = 2LL * scf_cu_count() / N;
  cap = cap < 64 ? 64 : cap;
  const int bx = (int)(scf_cdiv((int64_t)H * W, 256) < cap ? scf_cdiv((int64_t)H * W, 256) : cap);
  const PoseUpdateArgs pu = {rot_all, trans_all, (const long long*)label, num_class, label_mode, R_in, t_in,
                             d_rot, d_trans, R_out, t_out};
  scf_launch(reproject_flow_kernel, dim3(bx, N), dim3(256), 0, scf_stream(stream), depth, K,
                     R0, t0, (const float*)nullptr, (const float*)nullptr, flow, H, W, invalid_num, pu);
  return scf_launch_status();
}

__global__ __launch_bounds__(256) void unproject_depth_kernel(
    const float* __restrict__ depth, const float* __restrict__ K, const float* __restrict__ R0,
    const float* __restrict__ t0, float* __restrict__ pts, int H, int W) {
  __shared__ PoseMats s;
  const int n = blockIdx.y;
  load_mats(&s, K, R0, t0, nullptr, nullptr, n);
  const int hw = H * W;
  const float* dp = depth + (long long)n * hw;
  float* o = pts + (long long)n * 3 * hw;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < hw; idx += gridDim.x * blockDim.x) {
    const float d = dp[idx];
    float X = 0.f, Y = 0.f, Z = 0.f;
    if (d > 0.f) {
      const int yi = idx / W, xi = idx - yi * W;
      unproject(s, (float)xi, (float)yi, d, X, Y, Z);
    }
    o[idx] = X;
    o[hw + idx] = Y;
    o[2 * hw + idx] = Z;
  }
}

extern "C" int scf_unproject_depth(const float* depth, const float* K, const float* R0,
                                   const float* t0, float* pts, int N, int H, int W,
                                   scf_stream_t stream) {
  if (!depth || !K || !R0 || !t0 || !pts || N <= 0 || H <= 0 || W <= 0) return SCF_EINVAL;
  if (N > 65535) return SCF_EUNSUPPORTED;
  const int bx = (int)(scf_cdiv((int64_t)H * W, 256) < 64 ? scf_cdiv((int64_t)H * W, 256) : 64);
  scf_launch(unproject_depth_kernel, dim3(bx, N), dim3(256), 0, scf_stream(stream), depth,
                     K, R0, t0, pts, H, W);
  return scf_launch_status();
}

// class select (pose_head.py:207-210) + ortho6d -> R (pose.py:153-169) + pose compose
// (pose.py:124-149, both depth_transform branches, weight=10) of ONE sample.  One function (never inlined) for the stand-alone
// kernel and for the fused pose-update + re-projection launch: the same instructions, so the same bits.
__device__ __noinline__ void pose_update_one(const float* __restrict__ rot_all, const float* __restrict__ trans_all,
                                             const long long* __restrict__ label, int num_class, int label_mode,
                                             const float* R_in, const float* t_in, int n, PoseOut* o) {
  long long cls = (label_mode & SCF_POSE_LABEL_PER_SAMPLE) ? label[n] : label[0];
  if (cls < 0) cls += num_class;                 // torch.index_select rejects these; stay in range
  if (cls < 0) cls = 0;
  if (cls >= num_class) cls = num_class - 1;
  const float* a = rot_all + ((long long)n * num_class + cls) * 6;
  const float* dt = trans_all + ((long long)n * num_class + cls) * 3;
  float o6[6], dtr[3];
  for (int i = 0; i < 6; ++i) { o6[i] = a[i]; o->d_rot[i] = o6[i]; }
  for (int i = 0; i < 3; ++i) { dtr[i] = dt[i]; o->d_trans[i] = dtr[i]; }
  // x = normalize(a); z = normalize(x X b); y = z X x
  float nx = sqrtf(o6[0] * o6[0] + o6[1] * o6[1] + o6[2] * o6[2]);
  nx = fmaxf(nx, 1e-12f);
  const float x0 = o6[0] / nx, x1 = o6[1] / nx, x2 = o6[2] / nx;
  float z0 = x1 * o6[5] - x2 * o6[4], z1 = x2 * o6[3] - x0 * o6[5], z2 = x0 * o6[4] - x1 * o6[3];
  float nz = fmaxf(sqrtf(z0 * z0 + z1 * z1 + z2 * z2), 1e-12f);
  z0 /= nz; z1 /= nz; z2 /= nz;
  const float y0 = z1 * x2 - z2 * x1, y1 = z2 * x0 - z0 * x2, y2 = z0 * x1 - z1 * x0;
  const float Rd[9] = {x0, y0, z0, x1, y1, z1, x2, y2, z2};   // columns [x y z]
  float Rs[9], ts[3];
  for (int i = 0; i < 9; ++i) Rs[i] = R_in[n * 9 + i];
  for (int i = 0; i < 3; ++i) ts[i] = t_in[n * 3 + i];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      o->R[r * 3 + c] = Rd[r * 3 + 0] * Rs[0 * 3 + c] + Rd[r * 3 + 1] * Rs[1 * 3 + c] +
                        Rd[r * 3 + 2] * Rs[2 * 3 + c];
  // pose.py:137-141: 'exp' -> tz / exp(dz); any other depth_transform -> tz * (dz + 1)
  const float vz = (label_mode & SCF_POSE_DEPTH_LINEAR) ? ts[2] * (dtr[2] + 1.f) : ts[2] / expf(dtr[2]);
  const float vx = vz * (dtr[0] / 10.f + ts[0] / ts[2]);
  const float vy = vz * (dtr[1] / 10.f + ts[1] / ts[2]);
  o->t[0] = vx;
  o->t[1] = vy;
  o->t[2] = vz;
}
__device__ __forceinline__ void pose_store(const PoseOut& o, int n, float* __restrict__ d_rot, float* __restrict__ d_trans,
                                           float* R_out, float* t_out) {
  for (int i = 0; i < 6; ++i) d_rot[n * 6 + i] = o.d_rot[i];
  for (int i = 0; i < 3; ++i) d_trans[n * 3 + i] = o.d_trans[i];
  for (int i = 0; i < 9; ++i) R_out[n * 9 + i] = o.R[i];
  for (int i = 0; i < 3; ++i) t_out[n * 3 + i] = o.t[i];
}

// One thread per sample.
__global__ void pose_update_kernel(const float* __restrict__ rot_all,
                                   const float* __restrict__ trans_all,
                                   const long long* __restrict__ label, int num_class,
                                   int label_mode, const float* R_in, const float* t_in,
                                   float* __restrict__ d_rot, float* __restrict__ d_trans,
                                   float* R_out, float* t_out, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  PoseOut o;
  pose_update_one(rot_all, trans_all, label, num_class, label_mode, R_in, t_in, n, &o);
  pose_store(o, n, d_rot, d_trans, R_out, t_out);
}

extern "C" int scf_pose_update(const float* rot_all, const float* trans_all, const int64_t* label,
                               int num_class, int label_mode, const float* R_in, const float* t_in,
                               float* d_rot, float* d_trans, float* R_out, float* t_out, int N,
                               scf_stream_t stream) {
  if (!rot_all || !trans_all || !label || !R_in || !t_in || !d_rot || !d_trans || !R_out || !t_out ||
      N <= 0 || num_class <= 0 || (label_mode & ~(SCF_POSE_LABEL_PER_SAMPLE | SCF_POSE_DEPTH_LINEAR)))
    return SCF_EINVAL;
  scf_launch(pose_update_kernel, dim3((N + 63) / 64), dim3(64), 0, scf_stream(stream),
                     rot_all, trans_all, (const long long*)label, num_class, label_mode, R_in, t_in,
                     d_rot, d_trans, R_out, t_out, N);
  return scf_launch_status();
}


// ---------------------------------------------------------------------------------
// filter_flow_by_mask (models/utils/flow.py:6-26): ground-truth flow vectors whose end point
// leaves the target-image mask are marked invalid.  Per pixel, in place:
//   grid = ((x + fx) * 2 / max(W-1,1) - 1, ...)                (coords_grid, warp.py:9-29)
//   m    = grid_sample(mask, grid, bilinear, zeros, align_corners)
//   invalid <- m < 0.9  ||  (fx >= invalid_num && fy >= invalid_num)
// The de-normalisation follows ATen's grid_sampler (align_corners=False: ((g+1)*size-1)/2).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void filter_flow_by_mask_kernel(float* __restrict__ flow,
                                                                  const float* __restrict__ mask,
                                                                  int N, int H, int W,
                                                                  float invalid_num, int align_corners) {
  const long long total = (long long)N * H * W;
  const int HW = H * W;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(idx / HW), q = (int)(idx - (long long)n * HW);
    const int y = q / W, x = q - y * W;
    float* f = flow + (long long)n * 2 * HW + q;
    const float fx = f[0], fy = f[HW];
    const bool both = fx >= invalid_num && fy >= invalid_num;
    const float gx = ((float)x + fx) * 2.f / (float)max(W - 1, 1) - 1.f;
    const float gy = ((float)y + fy) * 2.f / (float)max(H - 1, 1) - 1.f;
    const float ix = align_corners ? (gx + 1.f) / 2.f * (float)(W - 1) : ((gx + 1.f) * (float)W - 1.f) / 2.f;
    const float iy = align_corners ? (gy + 1.f) / 2.f * (float)(H - 1) : ((gy + 1.f) * (float)H - 1.f) / 2.f;
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float x1f = x0f + 1.f, y1f = y0f + 1.f;
    const float nw = (x1f - ix) * (y1f - iy), ne = (ix - x0f) * (y1f - iy);
    const float sw = (x1f - ix) * (iy - y0f), se = (ix - x0f) * (iy - y0f);
    const float* m = mask + (long long)n * HW;
    float acc = 0.f;
    // |ix| can be huge (invalid_num = 400 on a 64-px image): range-test in float first
    const bool fin = ix > -2.f && ix < (float)W + 1.f && iy > -2.f && iy < (float)H + 1.f;
    if (fin) {
      const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
      const bool xa = x0 >= 0 && x0 < W, xb = x1 >= 0 && x1 < W;
      const bool ya = y0 >= 0 && y0 < H, yb = y1 >= 0 && y1 < H;
      if (ya && xa) acc += m[y0 * W + x0] * nw;
      if (ya && xb) acc += m[y0 * W + x1] * ne;
      if (yb && xa) acc += m[y1 * W + x0] * sw;
      if (yb && xb) acc += m[y1 * W + x1] * se;
    }
    if (acc < 0.9f || both) {
      f[0] = invalid_num;
      f[HW] = invalid_num;
    }
  }
}

extern "C" int scf_filter_flow_by_mask(float* flow, const float* mask, int N, int H, int W,
                                       float invalid_num, int align_corners, scf_stream_t stream) {
  if (!flow || !mask || N <= 0 || H <= 0 || W <= 0) return SCF_EINVAL;
  const long long total = (long long)N * H * W;
  const int grid = (int)(scf_cdiv(total, 256) < 262144 ? scf_cdiv(total, 256) : 262144);
  scf_launch(filter_flow_by_mask_kernel, dim3(grid), dim3(256), 0, scf_stream(stream), flow, mask,
                     N, H, W, invalid_num, align_corners);
  return scf_launch_status();
}


// ---------------------------------------------------------------------------------
// Pose-error evaluation (BaseDataset.eval_pose_error, datasets/base_dataset.py:378-424 +
// project_3d_point, datasets/pose.py:18-78), float64 like the reference's numpy arrays.
// One block per sample; all samples of a launch share one vertex set (one class):
//   gt_i = R_gt v_i + t_gt, pr_i = R_pr v_i + t_pr
//   err3d = mean_i |gt_i - pr_i|                          (ADD)
//         = mean_i |gt_i - pr_{argmin_j |gt_i - pr_j|}|   (ADD-S, symmetric != 0)
//   err2d = mean_i |proj(gt_i) - proj(pr_i)|,  proj = K p, (x, y) / (z + 1e-8)
// Sums are combined in a fixed order (per-thread strided partials, then an LDS tree).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pose_error_kernel(const double* __restrict__ verts, int nv,
                                                         const double* __restrict__ gt_r,
                                                         const double* __restrict__ gt_t,
                                                         const double* __restrict__ pr_r,
                                                         const double* __restrict__ pr_t,
                                                         const double* __restrict__ kmat,
                                                         const int* __restrict__ sample_idx,
                                                         int symmetric, double* __restrict__ err3d,
                                                         double* __restrict__ err2d) {
  __shared__ double red3[256], red2[256];
  __shared__ double tile[256 * 3];
  const int s = sample_idx[blockIdx.x];
  const int tid = threadIdx.x;
  double Rg[9], Rp[9], K[9], tg[3], tp[3];
  for (int i = 0; i < 9; ++i) { Rg[i] = gt_r[s * 9 + i]; Rp[i] = pr_r[s * 9 + i]; K[i] = kmat[s * 9 + i]; }
  for (int i = 0; i < 3; ++i) { tg[i] = gt_t[s * 3 + i]; tp[i] = pr_t[s * 3 + i]; }
  auto xform = [](const double (&R)[9], const double (&t)[3], const double* v, double (&o)[3]) {
    for (int a = 0; a < 3; ++a) o[a] = (R[3 * a] * v[0] + R[3 * a + 1] * v[1]) + R[3 * a + 2] * v[2] + t[a];
  };
  double acc3 = 0.0, acc2 = 0.0;
  const int rounds = (nv + 255) / 256;
  for (int rd = 0; rd < rounds; ++rd) {
    const int i = rd * 256 + tid;
    const bool live = i < nv;
    double g[3] = {0, 0, 0}, q[3] = {0, 0, 0};
    if (live) {
      xform(Rg, tg, verts + 3 * i, g);
      xform(Rp, tp, verts + 3 * i, q);
      double pg[3], pp[3];
      for (int a = 0; a < 3; ++a) {
        pg[a] = (K[3 * a] * g[0] + K[3 * a + 1] * g[1]) + K[3 * a + 2] * g[2];
        pp[a] = (K[3 * a] * q[0] + K[3 * a + 1] * q[1]) + K[3 * a + 2] * q[2];
      }
      const double dx = pg[0] / (pg[2] + 1e-8) - pp[0] / (pp[2] + 1e-8);
      const double dy = pg[1] / (pg[2] + 1e-8) - pp[1] / (pp[2] + 1e-8);
      acc2 += sqrt(dx * dx + dy * dy);
    }
    if (!symmetric) {
      if (live) {
        const double a = g[0] - q[0], b = g[1] - q[1], c = g[2] - q[2];
        acc3 += sqrt(a * a + b * b + c * c);
      }
    } else {                                   // nearest predicted point, tiles of 256 through LDS
      double best = 1e300;
      for (int j0 = 0; j0 < nv; j0 += 256) {
        __syncthreads();
        if (j0 + tid < nv) {
          double o[3];
          xform(Rp, tp, verts + 3 * (j0 + tid), o);
          tile[tid * 3] = o[0]; tile[tid * 3 + 1] = o[1]; tile[tid * 3 + 2] = o[2];
        }
        __syncthreads();
        const int m = min(256, nv - j0);
        if (live)
          for (int j = 0; j < m; ++j) {
            const double a = g[0] - tile[3 * j], b = g[1] - tile[3 * j + 1], c = g[2] - tile[3 * j + 2];
            const double d = a * a + b * b + c * c;
            best = d < best ? d : best;
          }
      }
      if (live) acc3 += sqrt(best);
    }
  }
  red3[tid] = acc3;
  red2[tid] = acc2;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) { red3[tid] += red3[tid + w]; red2[tid] += red2[tid + w]; }
    __syncthreads();
  }
  if (tid == 0) {
    err3d[s] = red3[0] / (double)nv;
    err2d[s] = red2[0] / (double)nv;
  }
}

extern "C" int scf_pose_error(const double* verts, int nv, const double* gt_r, const double* gt_t,
                              const double* pred_r, const double* pred_t, const double* K,
                              const int* sample_idx, int nsel, int symmetric, double* err3d,
                              double* err2d, scf_stream_t stream) {
  if (!verts || !gt_r || !gt_t || !pred_r || !pred_t || !K || !sample_idx || !err3d || !err2d)
    return SCF_EINVAL;
  if (nv <= 0 || nsel <= 0) return SCF_EINVAL;
  scf_launch(pose_error_kernel, dim3(nsel), dim3(256), 0, scf_stream(stream), verts, nv, gt_r,
                     gt_t, pred_r, pred_t, K, sample_idx, symmetric, err3d, err2d);
  return scf_launch_status();
}
