// Bilinear resize (align_corners=True), 2x2 average pooling and strided copies.
// All three are pure HBM streaming kernels: one output element per thread, coalesced along
// the innermost (x) dimension, grid capped and grid-strided.
#include "scf_common.h"

// F.interpolate(mode='bilinear', align_corners=True): scflow_decoder.py:188-197 (1/8 flow
// down-sampling, mixed with the 1/scale factor) and :222-227 (x8 up-sampling of flow+dflow
// and of the mask).  src = dst * (in-1)/(out-1); the +1 neighbour is clamped at the edge.
// One thread = four consecutive output pixels of one row (one 16-byte store when VEC), a block = 4 rows x 256 columns,
// grid = (column segments, row groups, planes): 32-bit index arithmetic only (r4: the one-pixel-per-thread form spent
// its time in three 64-bit divisions per pixel: 10 us for the 16.8 MB of a x8 up-sampled batch-32 flow).
// r5: a launch carries up to TWO jobs of the same geometry (planes [0, planes) = job 0, the rest job 1: the x8
// up-sampling of flow + delta flow and of the mask were two launches per iteration), and job 0 may write a second,
// sample-strided copy of its result (the 1/8 flow also goes into the GRU input buffer: a copy launch per iteration).
struct ResizeJob {
  const float* a; const float* b; float* out; float mul;
};
template <bool VEC>
__global__ __launch_bounds__(256) void resize_bilinear_kernel(ResizeJob j0, ResizeJob j1, int planes0,
                                                              int planes, int Hin, int Win,
                                                              int Hout, int Wout, float sh, float sw,
                                                              float* __restrict__ out2, int out2_group, long long out2_gs) {
  // no fma contraction: torch rounds the source coordinate scale * index before it takes the fraction; a fused
  // sw * ox - x0 is the exact product minus x0 and moves the weights by up to half an ulp of the coordinate (2.5e-5 in the
  // result at coordinate 255)
#pragma clang fp contract(off)
  const int ox0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
  const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (ox0 >= Wout || oy >= Hout) return;
  const float fy = sh * (float)oy;
  const int y0 = (int)fy;
  const int y1 = y0 + (y0 < Hin - 1 ? 1 : 0);
  const float ly = fy - (float)y0, hy = 1.f - ly;
  int x0[4], x1[4];
  float lx[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float fx = sw * (float)(ox0 + i);
    x0[i] = (int)fx;
    if (x0[i] > Win - 1) x0[i] = Win - 1;          // columns past Wout (ragged last quadruple): any valid address
    x1[i] = x0[i] + (x0[i] < Win - 1 ? 1 : 0);
    lx[i] = fx - (float)x0[i];
  }
  for (int plg = blockIdx.z; plg < planes; plg += gridDim.z) {
    const bool second = plg >= planes0;            // block-uniform
    const int pl = second ? plg - planes0 : plg;
    const float* a = second ? j1.a : j0.a;
    const float* b = second ? j1.b : j0.b;
    float* out = second ? j1.out : j0.out;
    const float mul = second ? j1.mul : j0.mul;
    const long long base = (long long)pl * Hin * Win;
    const float* r0 = a + base + (long long)y0 * Win;
    const float* r1 = a + base + (long long)y1 * Win;
    const float* s0 = b ? b + base + (long long)y0 * Win : nullptr;
    const float* s1 = b ? b + base + (long long)y1 * Win : nullptr;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v00 = r0[x0[i]], v01 = r0[x1[i]], v10 = r1[x0[i]], v11 = r1[x1[i]];
      if (b) {
        v00 += s0[x0[i]];
        v01 += s0[x1[i]];
        v10 += s1[x0[i]];
        v11 += s1[x1[i]];
      }
      const float hx = 1.f - lx[i];
      v[i] = mul * (hy * (hx * v00 + lx[i] * v01) + ly * (hx * v10 + lx[i] * v11));
    }
    float* o = out + ((long long)pl * Hout + oy) * Wout + ox0;
    float* o2 = (out2 && !second) ? out2 + (long long)(pl / out2_group) * out2_gs +
                                        ((long long)(pl % out2_group) * Hout + oy) * Wout + ox0
                                  : nullptr;
    if (VEC) {
      scf_store4<(SCF_ST_SC1 & 1) != 0>(o, v[0], v[1], v[2], v[3]);
      if (o2) scf_store4<(SCF_ST_SC1 & 1) != 0>(o2, v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (ox0 + i < Wout) {
          o[i] = v[i];
          if (o2) o2[i] = v[i];
        }
    }
  }
}

// internal form (scf_internal.h): job 1 optional (planes1 = 0), second destination of job 0 optional (out2 = nullptr;
// samples of out2_group planes each, out2_gstride floats apart)
int scf_resize_bilinear_jobs(const float* a, const float* b, float* out, int64_t planes, float mul,
                             const float* a1, const float* b1, float* out1, int64_t planes1, float mul1,
                             float* out2, int out2_group, int64_t out2_gstride,
                             int Hin, int Win, int Hout, int Wout, scf_stream_t stream) {
  if (!a || !out || planes <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0) return SCF_EINVAL;
  if (planes1 < 0 || (planes1 > 0 && (!a1 || !out1)) || (out2 && out2_group <= 0)) return SCF_EINVAL;
  const long long total = (long long)planes + planes1;
  if (total > 0x7fffffffLL) return SCF_EINVAL;
  const float sh = Hout > 1 ? (float)(Hin - 1) / (float)(Hout - 1) : 0.f;
  const float sw = Wout > 1 ? (float)(Win - 1) / (float)(Wout - 1) : 0.f;
  // about four workgroups per CU, each walking several planes: the x8 flow up-sampling of a batch of 32 was 4096
  // workgroups of one plane-row-group each -- a launch the workgroup dispatcher (~1.4 ns per workgroup), not the
  // memory system, paced (r5: 8.8 -> see DESIGN)
  const long long gxy = scf_cdiv(Wout, 256) * scf_cdiv(Hout, 4);
  long long gz = (4LL * scf_cu_count() + gxy - 1) / gxy;
  gz = gz < 1 ? 1 : gz > total ? total : gz > 65535 ? 65535 : gz;
  const dim3 grid((unsigned)scf_cdiv(Wout, 256), (unsigned)scf_cdiv(Hout, 4), (unsigned)gz);
  if (grid.y > 65535u) return SCF_EUNSUPPORTED;
  const ResizeJob j0 = {a, b, out, mul}, j1 = {a1, b1, out1, mul1};
  const bool vec = (Wout & 3) == 0 && ((uintptr_t)out & 15) == 0 && (planes1 == 0 || ((uintptr_t)out1 & 15) == 0) &&
                   (!out2 || (((uintptr_t)out2 & 15) == 0 && (out2_gstride & 3) == 0));
  if (vec)
    scf_launch(resize_bilinear_kernel<true>, grid, dim3(256), 0, scf_stream(stream), j0, j1, (int)planes, (int)total,
               Hin, Win, Hout, Wout, sh, sw, out2, out2_group, (long long)out2_gstride);
  else
    scf_launch(resize_bilinear_kernel<false>, grid, dim3(256), 0, scf_stream(stream), j0, j1, (int)planes, (int)total,
               Hin, Win, Hout, Wout, sh, sw, out2, out2_group, (long long)out2_gstride);
  return scf_launch_status();
}

extern "C" int scf_resize_bilinear(const float* a, const float* b, float* out, int64_t planes,
                                   int Hin, int Win, int Hout, int Wout, float mul,
                                   scf_stream_t stream) {
  return scf_resize_bilinear_jobs(a, b, out, planes, mul, nullptr, nullptr, nullptr, 0, 1.0f, nullptr, 1, 0,
                                  Hin, Win, Hout, Wout, stream);
}

// nn.AvgPool2d(2, 2) of CorrelationPyramid (raft_decoder.py:32, 54-56): window summed in
// row-major order, then divided by 4.
__global__ __launch_bounds__(256) void avgpool2x2_kernel(const float* __restrict__ x,
                                                         float* __restrict__ out, long long planes,
                                                         int Hin, int Win, int Ho, int Wo) {
  const long long total = planes * Ho * Wo;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(idx % Wo);
    const long long t = idx / Wo;
    const int oy = (int)(t % Ho);
    const long long pl = t / Ho;
    const float* s = x + pl * Hin * Win + (long long)(2 * oy) * Win + 2 * ox;
    out[idx] = (((s[0] + s[1]) + s[Win]) + s[Win + 1]) * 0.25f;
  }
}

extern "C" int scf_avgpool2x2(const float* x, float* out, int64_t planes, int Hin, int Win,
                              scf_stream_t stream) {
  if (!x || !out || planes <= 0 || Hin < 2 || Win < 2) return SCF_EINVAL;
  const int Ho = Hin / 2, Wo = Win / 2;
  const long long total = (long long)planes * Ho * Wo;
  const int grid = (int)(scf_cdiv(total, 256) < 262144 ? scf_cdiv(total, 256) : 262144);
  scf_launch(avgpool2x2_kernel, dim3(grid), dim3(256), 0, scf_stream(stream), x, out,
                     (long long)planes, Hin, Win, Ho, Wo);
  return scf_launch_status();
}

// The same pool between two pyramid levels in either layout (scf_corr_build_ex): a level is
// row-major (pw4 = 0) or stored in 8x4-float tiles of a map padded to 4 rows x 8 columns (pw4 = 4 x
// padded width = floats per row of tiles); msz = floats per query map.  Same arithmetic as above;
// tile padding is neither read nor written.
__device__ __forceinline__ int pyr_offset(int y, int x, int pw4, int lw) {
  return pw4 ? (y >> 2) * pw4 + (x >> 3) * 32 + (y & 3) * 8 + (x & 7) : y * lw + x;
}
__global__ __launch_bounds__(256) void avgpool2x2_layout_kernel(const float* __restrict__ x,
                                                                float* __restrict__ out,
                                                                long long planes, int Win, int in_pw4,
                                                                int in_msz, int Ho, int Wo, int out_pw4,
                                                                int out_msz) {
  const long long total = planes * Ho * Wo;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(idx % Wo);
    const long long t = idx / Wo;
    const int oy = (int)(t % Ho);
    const long long pl = t / Ho;
    const float* s = x + pl * in_msz;
    const int y = 2 * oy, xx = 2 * ox;
    const float a = s[pyr_offset(y, xx, in_pw4, Win)], b = s[pyr_offset(y, xx + 1, in_pw4, Win)];
    const float c = s[pyr_offset(y + 1, xx, in_pw4, Win)], d = s[pyr_offset(y + 1, xx + 1, in_pw4, Win)];
    out[pl * out_msz + pyr_offset(oy, ox, out_pw4, Wo)] = (((a + b) + c) + d) * 0.25f;
  }
}

int scf_avgpool2x2_layout(const float* x, float* out, int64_t planes, int Hin, int Win, int in_tiled,
                          int out_tiled, hipStream_t st) {
  if (!x || !out || planes <= 0 || Hin < 2 || Win < 2) return SCF_EINVAL;
  const int Ho = Hin / 2, Wo = Win / 2;
  const int in_pw = (Win + 7) / 8 * 8, in_ph = (Hin + 3) / 4 * 4;
  const int out_pw = (Wo + 7) / 8 * 8, out_ph = (Ho + 3) / 4 * 4;
  const long long total = (long long)planes * Ho * Wo;
  const int grid = (int)(scf_cdiv(total, 256) < 262144 ? scf_cdiv(total, 256) : 262144);
  scf_launch(avgpool2x2_layout_kernel, dim3(grid), dim3(256), 0, st, x, out, (long long)planes, Win,
             in_tiled ? in_pw * 4 : 0, in_tiled ? in_ph * in_pw : Hin * Win, Ho, Wo,
             out_tiled ? out_pw * 4 : 0, out_tiled ? out_ph * out_pw : Ho * Wo);
  return scf_launch_status();
}

// out[n, c, p] = x[n, c, p] * mask[n, 0, p]: the decoder's optional occlusion masking of the looked-up
// correlation and of the flow (scflow_decoder.py:199-205, mask_corr / mask_flow).  x / out may be
// sample-strided NCHW views.
__global__ __launch_bounds__(256) void mul_mask_kernel(const float* __restrict__ x, long long xns,
                                                       const float* __restrict__ mask,
                                                       float* __restrict__ out, long long ons, int N,
                                                       int C, int HW) {
  const long long per = (long long)C * HW, total = (long long)N * per;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const long long n = idx / per, r = idx - n * per;
    const int px = (int)(r % HW);
    out[n * ons + r] = x[n * xns + r] * mask[n * HW + px];
  }
}

extern "C" int scf_mul_mask(const float* x, int64_t x_nstride, const float* mask, float* out,
                            int64_t out_nstride, int N, int C, int HW, scf_stream_t stream) {
  if (!x || !mask || !out || N <= 0 || C <= 0 || HW <= 0) return SCF_EINVAL;
  const long long total = (long long)N * C * HW;
  const int grid = (int)(scf_cdiv(total, 256) < 262144 ? scf_cdiv(total, 256) : 262144);
  scf_launch(mul_mask_kernel, dim3(grid), dim3(256), 0, scf_stream(stream), x, (long long)x_nstride, mask,
             out, (long long)out_nstride, N, C, HW);
  return scf_launch_status();
}

// VEC: count, both sample strides and both base pointers are multiples of 4 floats: 16 bytes per lane.  The grid is a
// few workgroups per CU walking the tensor with a grid stride (one workgroup per 256 elements was tens of thousands of
// workgroups for the stacked encoder input: the dispatcher, not the memory system, paced the launch).
template <bool VEC>
__global__ __launch_bounds__(256) void copy_strided_kernel(const float* __restrict__ src,
                                                           long long sns, float* __restrict__ dst,
                                                           long long dns, int N, long long count) {
  if (VEC) {
    const long long c4 = count >> 2, total = (long long)N * c4;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
      const long long n = idx / c4, r = idx - n * c4;
      reinterpret_cast<float4*>(dst + n * dns)[r] = reinterpret_cast<const float4*>(src + n * sns)[r];
    }
  } else {
    const long long total = (long long)N * count;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
      const long long n = idx / count, r = idx - n * count;
      dst[n * dns + r] = src[n * sns + r];
    }
  }
}

extern "C" int scf_copy_strided(const float* src, int64_t src_nstride, float* dst,
                                int64_t dst_nstride, int N, int64_t count, scf_stream_t stream) {
  if (!src || !dst || N <= 0 || count <= 0) return SCF_EINVAL;
  const bool vec = ((count | src_nstride | dst_nstride) & 3) == 0 && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0;
  const long long units = (long long)N * (vec ? count >> 2 : count);
  const long long cap = 8LL * scf_cu_count();
  const int grid = (int)(scf_cdiv(units, 256) < cap ? scf_cdiv(units, 256) : cap);
  if (vec)
    scf_launch(copy_strided_kernel<true>, dim3(grid), dim3(256), 0, scf_stream(stream), src, (long long)src_nstride, dst,
               (long long)dst_nstride, N, (long long)count);
  else
    scf_launch(copy_strided_kernel<false>, dim3(grid), dim3(256), 0, scf_stream(stream), src, (long long)src_nstride, dst,
               (long long)dst_nstride, N, (long long)count);
  return scf_launch_status();
}

// ---------------------------------------------------------------------------------
// RAFT convex up-sampling (x`scale`, 3x3 neighbourhood): RAFTDecoder._upsample,
// models/decoder/raft_decoder.py:381-416 (same arithmetic in RAFTDecoderMask.upsample_flow /
// upsample_mask, raft_decoder_mask.py:104-160):
//   w[k] = softmax_k(mask_mul * mask[n, k*s*s + sy*s + sx, y, x]),  k = ky*3 + kx
//   out[n, c, s*y+sy, s*x+sx] = sum_k w[k] * x_mul * x[n, c, y+ky-1, x+kx-1]   (zero padded)
// HBM-bound on the mask read (9*s*s*4 B per low-res pixel).  One block = one low-res row
// segment of 32 pixels: lanes run along x (coalesced mask reads, one 128-B line per channel),
// results are transposed through LDS so every (c, sy) output row is written as s*32
// contiguous floats.
// ---------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(256) void convex_upsample_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ mask,
                                                              float* __restrict__ out, int C, int h,
                                                              int w, float x_mul, float mask_mul) {
  constexpr int SS = S * S;
  extern __shared__ float tile[];              // [C][S(sy)][32*S] floats
  const int xt = blockIdx.x, y = blockIdx.y, n = blockIdx.z;
  const int xl = threadIdx.x & 31, sub0 = threadIdx.x >> 5;   // 8 sub-pixel rows of lanes
  const int px = xt * 32 + xl;
  const long long hw = (long long)h * w;
  const bool ok = px < w;
  // all SS / 8 * 9 mask values of this thread are requested before the first is used (the loop
  // below was a chain of SS / 8 dependent memory round trips)
  float wall[SS / 8][9];
#pragma unroll
  for (int it = 0; it < SS / 8; ++it) {
    const int sub = sub0 + it * 8;
#pragma unroll
    for (int k = 0; k < 9; ++k)
      wall[it][k] = ok ? mask[((long long)n * 9 * SS + k * SS + sub) * hw + (long long)y * w + px] : 0.f;
  }
#pragma unroll
  for (int it = 0; it < SS / 8; ++it) {
    const int sub = sub0 + it * 8;
    const int sy = sub / S, sx = sub - sy * S;
    float wk[9];
    float mx = -3.4e38f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      wk[k] = ok ? mask_mul * wall[it][k] : 0.f;
      mx = fmaxf(mx, wk[k]);
    }
    float den = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) { wk[k] = expf(wk[k] - mx); den += wk[k]; }
    const float inv = 1.f / den;
    for (int c = 0; c < C; ++c) {
      const float* xp = x + ((long long)n * C + c) * hw;
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int yy = y + k / 3 - 1, xx = px + k % 3 - 1;
        float v = 0.f;
        if (ok && yy >= 0 && yy < h && xx >= 0 && xx < w) v = x_mul * xp[(long long)yy * w + xx];
        acc += (wk[k] * inv) * v;
      }
      tile[(c * S + sy) * (32 * S) + xl * S + sx] = acc;
    }
  }
  __syncthreads();
  const int W8 = w * S;
  const int rowlen = 32 * S;
  for (int e = threadIdx.x; e < C * S * rowlen; e += 256) {
    const int col = e % rowlen, r = e / rowlen;      // r = c*S + sy
    const int c = r / S, sy = r - c * S;
    const int ox = xt * rowlen + col;
    if (ox < W8) out[(((long long)n * C + c) * h * S + (long long)y * S + sy) * W8 + ox] = tile[e];
  }
}

extern "C" int scf_convex_upsample(const float* x, const float* mask, float* out, int N, int C,
                                   int h, int w, int scale, float x_mul, float mask_mul,
                                   scf_stream_t stream) {
  if (!x || !mask || !out || N <= 0 || C <= 0 || h <= 0 || w <= 0) return SCF_EINVAL;
  if (scale != 8 || N > 65535 || h > 65535) return SCF_EUNSUPPORTED;
  const size_t lds = (size_t)C * 8 * 32 * 8 * sizeof(float);
  if (lds > 64 * 1024) return SCF_EUNSUPPORTED;
  const dim3 grid((w + 31) / 32, h, N);
  scf_launch(convex_upsample_kernel<8>, grid, dim3(256), lds, scf_stream(stream), x, mask,
                     out, C, h, w, x_mul, mask_mul);
  return scf_launch_status();
}
