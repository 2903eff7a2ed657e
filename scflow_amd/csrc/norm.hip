// InstanceNorm (+residual, +ReLU), GroupNorm+ReLU and the small linear layers of the pose
// head, for gfx950.  All are memory-bound: one workgroup per normalisation group keeps the
// group in registers (single HBM read, single write), two-pass variance in fp32.
#include "scf_common.h"

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// sum over an NT-thread workgroup (fixed order), result broadcast to every thread
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < NT / 64; w += 4) s += (red[w] + red[w + 1]) + (red[w + 2] + red[w + 3]);
  return s;
}

// sum over a 256-thread workgroup, result broadcast to every thread
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// ---------------------------------------------------------------------------------
// InstanceNorm2d(eps, affine=False, no running stats): F.instance_norm on the feature
// encoder (norm_cfg IN; resnet.py:75-86, raft_encoder.py:300-302).  One block per (n, c)
// plane; VEC4 float4 per thread cached in registers.  NT = 256 threads for planes up to 128 x 128,
// 1024 threads (16 waves, <= 128 VGPRs each) up to 80 K elements (the 240 x 320 planes of a
// 480 x 640 crop): still one read and one write of the plane.
// ---------------------------------------------------------------------------------
template <int VEC4, int NT = 256>
__global__ __launch_bounds__(NT) void instance_norm_kernel(const float* x,      // x / res may alias out (in place):
                                                           const float* res,    // no __restrict__
                                                           float* out, int HW,
                                                           float eps, int relu) {
  __shared__ float red[NT / 64];
  const long long pl = blockIdx.x;
  const float4* xp = reinterpret_cast<const float4*>(x + pl * HW);
  const int n4 = HW >> 2;
  float4 v[VEC4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VEC4; ++i) {
    const int idx = i * NT + threadIdx.x;
    v[i] = idx < n4 ? xp[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = block_sum<NT>(s, red) / (float)HW;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VEC4; ++i) {
    const int idx = i * NT + threadIdx.x;
    if (idx < n4) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float var = block_sum<NT>(q, red) / (float)HW;
  const float rstd = 1.0f / sqrtf(var + eps);
  float4* op = reinterpret_cast<float4*>(out + pl * HW);
  const float4* rp = res ? reinterpret_cast<const float4*>(res + pl * HW) : nullptr;
#pragma unroll
  for (int i = 0; i < VEC4; ++i) {
    const int idx = i * NT + threadIdx.x;
    if (idx < n4) {
      float4 o;
      o.x = (v[i].x - mean) * rstd;
      o.y = (v[i].y - mean) * rstd;
      o.z = (v[i].z - mean) * rstd;
      o.w = (v[i].w - mean) * rstd;
      if (rp) {
        const float4 r = rp[idx];
        o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
      }
      if (relu) {        // NaN -> 0 (v_max), like the convolution epilogue's compare+select
        o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
      }
      scf_store4<(SCF_ST_SC1 & 2) != 0>(reinterpret_cast<float*>(op + idx), o.x, o.y, o.z, o.w);
    }
  }
}

// generic fallback (any HW): three sweeps, the 2nd/3rd hit L2
__global__ __launch_bounds__(256) void instance_norm_generic_kernel(const float* x,     // may alias out
                                                                    const float* res,
                                                                    float* out, int HW,
                                                                    float eps, int relu) {
  __shared__ float red[4];
  const long long pl = blockIdx.x;
  const float* xp = x + pl * HW;
  float s = 0.f;
  for (int i = threadIdx.x; i < HW; i += 256) s += xp[i];
  const float mean = block_sum_256(s, red) / (float)HW;
  float q = 0.f;
  for (int i = threadIdx.x; i < HW; i += 256) {
    const float a = xp[i] - mean;
    q += a * a;
  }
  const float var = block_sum_256(q, red) / (float)HW;
  const float rstd = 1.0f / sqrtf(var + eps);
  for (int i = threadIdx.x; i < HW; i += 256) {
    float o = (xp[i] - mean) * rstd;
    if (res) o += res[pl * HW + i];
    if (relu) o = fmaxf(o, 0.f);
    out[pl * HW + i] = o;
  }
}

extern "C" int scf_instance_norm(const float* x, const float* res, float* out, int64_t planes,
                                 int HW, float eps, int relu, scf_stream_t stream) {
  if (!x || !out || planes <= 0 || HW <= 0) return SCF_EINVAL;
  if (planes > 0x7fffffffLL) return SCF_EUNSUPPORTED;
  hipStream_t st = scf_stream(stream);
  const dim3 grid((unsigned)planes), blk(256);
  const bool vec = (HW % 4 == 0) && ((((uintptr_t)x | (uintptr_t)out | (uintptr_t)res) & 15) == 0);
  const int n4 = HW / 4;
  if (vec && n4 <= 256) scf_launch(instance_norm_kernel<1>, grid, blk, 0, st, x, res, out, HW, eps, relu);
  else if (vec && n4 <= 1024) scf_launch(instance_norm_kernel<4>, grid, blk, 0, st, x, res, out, HW, eps, relu);
  else if (vec && n4 <= 4096) scf_launch(instance_norm_kernel<16>, grid, blk, 0, st, x, res, out, HW, eps, relu);
  else if (vec && n4 <= 8192) scf_launch((instance_norm_kernel<8, 1024>), grid, dim3(1024), 0, st, x, res, out, HW, eps, relu);
  else if (vec && n4 <= 20480) scf_launch((instance_norm_kernel<20, 1024>), grid, dim3(1024), 0, st, x, res, out, HW, eps, relu);
  else scf_launch(instance_norm_generic_kernel, grid, blk, 0, st, x, res, out, HW, eps, relu);
  return scf_launch_status();
}

// ---------------------------------------------------------------------------------
// GroupNorm(G, eps) with affine + ReLU: pose_head.py:151-159 (norm_cfg GN, 32 groups).
// Channels of a group are contiguous in NCHW, so a group is one contiguous run of
// (C/G)*HW floats.  One block per (n, g).
// ---------------------------------------------------------------------------------
// x may arrive as `parts` partial tensors (a convolution whose K was split across blocks, scf_conv_desc.k_slices):
// the value of an element is the sum of its parts IN PART ORDER, formed the same way in all three passes.
__global__ __launch_bounds__(256) void group_norm_relu_kernel(const float* __restrict__ x, int parts,
                                                              long long part_stride,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta,
                                                              float* __restrict__ out, int C, int HW,
                                                              int G, float eps) {
  __shared__ float red[4];
  const int n = blockIdx.x / G, g = blockIdx.x % G;
  const int cpg = C / G;
  const int cnt = cpg * HW;
  const long long base = ((long long)n * C + (long long)g * cpg) * HW;
  const float* xp = x + base;
  auto at = [&](int i) {
    float v = xp[i];
    for (int sl = 1; sl < parts; ++sl) v += xp[(long long)sl * part_stride + i];
    return v;
  };
  // groups of up to 8 elements per thread (every group of the pose head: 4 channels x 256 / 64 / 16 pixels) are read ONCE
  // -- parts added in order -- and kept in registers through the three passes; larger groups re-read (L2)
  constexpr int KEEP = 8;
  const bool keep = cnt <= KEEP * 256;
  float kv[KEEP];
  if (keep) {
    // r6: every load of the (up to four) partial tensors is requested before the first add -- one memory round trip per
    // launch instead of one per (element, part): a launch of this kernel at batch 1 is nothing but that chain (6.8 us).
    // The adds keep their order (part 0 + part 1 + ...), so the sums keep their bits.
    float u[4][KEEP];
#pragma unroll
    for (int j = 0; j < KEEP; ++j) {
      const int i = threadIdx.x + j * 256;
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) u[sl][j] = (i < cnt && sl < parts) ? xp[(long long)sl * part_stride + i] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < KEEP; ++j) {
      kv[j] = u[0][j];
#pragma unroll
      for (int sl = 1; sl < 4; ++sl)
        if (sl < parts) kv[j] += u[sl][j];
    }
    for (int sl = 4; sl < parts; ++sl) {
#pragma unroll
      for (int j = 0; j < KEEP; ++j) {
        const int i = threadIdx.x + j * 256;
        if (i < cnt) kv[j] += xp[(long long)sl * part_stride + i];
      }
    }
  }
  float s = 0.f;
  if (keep) {
#pragma unroll
    for (int j = 0; j < KEEP; ++j) s += kv[j];
  } else {
    for (int i = threadIdx.x; i < cnt; i += 256) s += at(i);
  }
  const float mean = block_sum_256(s, red) / (float)cnt;
  float q = 0.f;
  if (keep) {
#pragma unroll
    for (int j = 0; j < KEEP; ++j) {
      const float a = kv[j] - mean;
      q += (threadIdx.x + j * 256 < cnt) ? a * a : 0.f;
    }
  } else {
    for (int i = threadIdx.x; i < cnt; i += 256) {
      const float a = at(i) - mean;
      q += a * a;
    }
  }
  const float var = block_sum_256(q, red) / (float)cnt;
  const float rstd = 1.0f / sqrtf(var + eps);
  if (keep) {
#pragma unroll
    for (int j = 0; j < KEEP; ++j) {
      const int i = threadIdx.x + j * 256;
      if (i < cnt) {
        const int c = g * cpg + i / HW;
        out[base + i] = fmaxf((kv[j] - mean) * rstd * gamma[c] + beta[c], 0.f);
      }
    }
  } else {
    for (int i = threadIdx.x; i < cnt; i += 256) {
      const int c = g * cpg + i / HW;
      const float o = (at(i) - mean) * rstd * gamma[c] + beta[c];
      out[base + i] = fmaxf(o, 0.f);
    }
  }
}

extern "C" int scf_group_norm_relu_parts(const float* x, int parts, int64_t part_stride, const float* gamma,
                                         const float* beta, float* out, int N, int C, int HW, int G, float eps,
                                         scf_stream_t stream) {
  if (!x || !gamma || !beta || !out || N <= 0 || C <= 0 || HW <= 0 || G <= 0 || parts < 1) return SCF_EINVAL;
  if (parts > 1 && part_stride < (int64_t)N * C * HW) return SCF_EINVAL;
  if (C % G != 0) return SCF_EUNSUPPORTED;
  scf_launch(group_norm_relu_kernel, dim3(N * G), dim3(256), 0, scf_stream(stream), x, parts, (long long)part_stride,
             gamma, beta, out, C, HW, G, eps);
  return scf_launch_status();
}

extern "C" int scf_group_norm_relu(const float* x, const float* gamma, const float* beta, float* out,
                                   int N, int C, int HW, int G, float eps, scf_stream_t stream) {
  return scf_group_norm_relu_parts(x, 1, 0, gamma, beta, out, N, C, HW, G, eps, stream);
}

// ---------------------------------------------------------------------------------
// nn.Linear (+ReLU): pose_head.py:166-172, 203-206.  Weight-streaming GEMV batch: one wave
// per output feature streams its weight row once (float4, four 1 KiB pieces in flight) and dots
// it with up to NB sample rows (activations are L2 resident), wave-level shuffle reduction.
// Two layers that read the same input (rotation_pred / translation_pred) share one launch: the
// feature index runs over O + O2.
// ---------------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x,
                                                     const float* __restrict__ W,
                                                     const float* __restrict__ b,
                                                     float* __restrict__ y, int N, int K, int O,
                                                     int act, const float* __restrict__ W2,
                                                     const float* __restrict__ b2,
                                                     float* __restrict__ y2, int O2) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int o = blockIdx.x * 4 + wave;
  const int n0 = blockIdx.y * NB;
  if (o >= O + O2) return;
  if (o >= O) { o -= O; W = W2; b = b2; y = y2; O = O2; }
  float acc[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) acc[i] = 0.f;
  const float* wrow = W + (long long)o * K;
  if ((K & 3) == 0) {
    for (int k0 = lane * 4; k0 < K; k0 += 1024) {
      float4 w[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + j * 256;
        w[j] = k < K ? *reinterpret_cast<const float4*>(wrow + k) : float4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + j * 256;
        if (k < K) {
#pragma unroll
          for (int i = 0; i < NB; ++i) {
            if (n0 + i < N) {
              const float4 xv = *reinterpret_cast<const float4*>(x + (long long)(n0 + i) * K + k);
              acc[i] += (w[j].x * xv.x + w[j].y * xv.y) + (w[j].z * xv.z + w[j].w * xv.w);
            }
          }
        }
      }
    }
  } else {
    for (int k = lane; k < K; k += 64) {
      const float w = wrow[k];
#pragma unroll
      for (int i = 0; i < NB; ++i)
        if (n0 + i < N) acc[i] += w * x[(long long)(n0 + i) * K + k];
    }
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const float s = wave_sum(acc[i]);
    if (lane == 0 && n0 + i < N) {
      float v = s + (b ? b[o] : 0.f);
      y[(long long)(n0 + i) * O + o] = scf_apply_act(v, act);
    }
  }
}

static int linear_launch(const float* x, const float* W, const float* b, float* y, int N, int K, int O,
                         int act, const float* W2, const float* b2, float* y2, int O2,
                         scf_stream_t stream) {
  if ((((uintptr_t)x | (uintptr_t)W | (uintptr_t)W2) & 15) != 0 && (K & 3) == 0) return SCF_EUNSUPPORTED;
  constexpr int NB = 8;
  const dim3 grid((O + O2 + 3) / 4, (N + NB - 1) / NB);
  if (grid.y > 65535) return SCF_EUNSUPPORTED;
  scf_launch(linear_kernel<NB>, grid, dim3(256), 0, scf_stream(stream), x, W, b, y, N, K, O, act, W2,
             b2, y2, O2);
  return scf_launch_status();
}

extern "C" int scf_linear(const float* x, const float* W, const float* b, float* y, int N, int K,
                          int O, int act, scf_stream_t stream) {
  if (!x || !W || !y || N <= 0 || K <= 0 || O <= 0) return SCF_EINVAL;
  return linear_launch(x, W, b, y, N, K, O, act, nullptr, nullptr, nullptr, 0, stream);
}

extern "C" int scf_linear_pair(const float* x, const float* W1, const float* b1, float* y1, int O1,
                               const float* W2, const float* b2, float* y2, int O2, int N, int K,
                               int act, scf_stream_t stream) {
  if (!x || !W1 || !y1 || !W2 || !y2 || N <= 0 || K <= 0 || O1 <= 0 || O2 <= 0) return SCF_EINVAL;
  return linear_launch(x, W1, b1, y1, N, K, O1, act, W2, b2, y2, O2, stream);
}
