// Direct 2-D convolution as an implicit GEMM on the fp32 matrix cores of gfx950
// (v_mfma_f32_32x32x2_f32: exact fp32 fma chain, 157 TFLOP/s chip peak), NCHW in / NCHW
// out, with the whole post-conv epilogue of the SCFlow hot path fused in.  The same kernel
// builds the 4-D correlation volume (per-sample "weights" = feat1, 1x1, out_div = sqrt(C)).
//
// GEMM view (D = A * B, "weights-stationary transposed" so that NCHW is the natural layout
// of every operand -- no im2col, no layout conversion anywhere on the path):
//     D[co, pix] = sum_k  A[co, k] * B[k, pix]
//     A[co, k]   = packed weight wp[k][co]          (lane <-> co, contiguous in memory)
//     B[k, pix]  = input[ci][oy*s+ky-p][ox*s+kx-p]  (lane <-> pixel, contiguous along x)
//     k          = (channel chunk, tap, channel in chunk)
// For the 32x32x2 MFMA a lane holds ONE A and ONE B scalar per instruction (A[l&31][l>>5],
// B[l>>5][l&31]), so both operands are read from LDS with conflict-free ds_read_b32 and the
// 32x32 result tile (col = lane&31 = pixel) stores as 128-B rows straight into NCHW.
//
// Block = 256 threads = 4 waves.  Block tile = (WM*32 output channels) x (WN*128 pixels);
// every wave owns WN pixel fragments of 32 pixels (FR rows x FC cols, FC = min(32, pow2(Wo)))
// and all WM channel fragments -> WM*WN accumulators of 16 VGPRs.
// Per channel chunk (KC channels, all taps) the block stages
//     weights  [KC*T][BM]      (straight copy of the packed rows, float4)
//     patch    [KC][PH][PW]    (input window incl. halo, zero padded)
// in LDS, then issues T*KC/2 k-steps of WM*WN MFMAs.  2-3 blocks are resident per CU, so
// one block's staging overlaps another's MFMA phase.
#include <atomic>
#include <mutex>
#include <vector>
#include "scf_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#include "conv_kernels.h"

// all taps of one staged channel chunk: NCP = KC/2 k-steps (channel pairs) per tap
template <int WM, int WN, int NCP>
__device__ __forceinline__ void mfma_taps(f32x16 (&acc)[WM][WN], const float* wl, const float* pl,
                                          const int (&boff)[WN], int T, int KW, int KC, int BM,
                                          int PW, int PHW, int half, int l32) {
  for (int t = 0; t < T; ++t) {
    const int ky = t / KW, kx = t - ky * KW;
    const float* wt = wl + (t * KC + half) * BM + l32;
    const float* pt = pl + ky * PW + kx;
#pragma unroll
    for (int c = 0; c < NCP; ++c) {
      const int cp = 2 * c;
      float a[WM], b[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) a[i] = wt[cp * BM + i * 32];
#pragma unroll
      for (int j = 0; j < WN; ++j) b[j] = pt[cp * PHW + boff[j]];
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
}

// per-thread prefetch registers by channel-chunk size: PU patch floats, WU weight float4
__host__ __device__ constexpr int pu_max(int kc, int wm) { return kc == 32 ? (wm * 1 >= 2 ? 16 : 26) : kc == 8 ? 20 : 12; }
__host__ __device__ constexpr int wu_max(int kc) { return kc == 32 ? 10 : kc == 8 ? 9 : 13; }

template <int WM, int WN, int KC>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(ConvK p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __builtin_amdgcn_s_setprio(3);
  constexpr int BM = WM * 32;
  constexpr int B4 = BM / 4;
  constexpr int NFRAG = WN * 4;
  constexpr int PU_MAX = pu_max(KC, WM * WN), WU_MAX = wu_max(KC);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l32 = lane & 31, half = lane >> 5;

  const int lb = scf_xcd_remap(blockIdx.x, gridDim.x);
  const int mblk = lb % p.mblocks;
  const int tile = lb / p.mblocks;
  const int m0 = mblk * BM;

  const int FC = 1 << p.fc_log2, FR = 32 >> p.fc_log2, TR = NFRAG * FR;
  const int txi = tile % p.tiles_x;
  const int t2 = tile / p.tiles_x;
  const int tyi = t2 % p.tiles_y;
  const int n = t2 / p.tiles_y;
  const int ty0 = tyi * TR, tx0 = txi * FC;
  const int s = p.stride;
  const int iy0 = ty0 * s - p.pad_h, ix0 = tx0 * s - p.pad_w;
  const int PH = p.PH, PW = p.PW, PHW = PH * PW;
  const int T = p.T;

  float* wl = lds;
  float* pl = lds + KC * T * BM;

  const int fr = l32 >> p.fc_log2, fc = l32 & (FC - 1);
  int boff[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j) boff[j] = (((wave * WN + j) * FR + fr) * s) * PW + fc * s + half * PHW;

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int HWin = p.H * p.W;
  const float* in0n = p.in0 + (long long)n * p.in0_ns;
  const float* in1n = p.in1 ? p.in1 + (long long)n * p.in1_ns : nullptr;
  const float* wpn = p.wp + (long long)n * p.w_ns;
  const int rows = KC * T;
  const int PE = KC * PHW;      // patch elements per chunk
  const int WE = rows * B4;     // weight float4 per chunk

  // Gather table: element e = tid + 256*u of the staged window [KC][PH][PW] lives at offset
  // toff[u] (floats) from the chunk's first channel plane, or is zero padding (-1).  The
  // mapping is chunk-invariant, so the two integer divisions are paid once per block.
  int toff[PU_MAX];
#pragma unroll
  for (int u = 0; u < PU_MAX; ++u) {
    const int e = tid + u * 256;
    int o = -1;
    if (e < PE) {
      const int cl = e / PHW, r = e - cl * PHW;
      const int py = r / PW, px = r - py * PW;
      const int iy = iy0 + py, ix = ix0 + px;
      if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) o = cl * HWin + iy * p.W + ix;
    }
    toff[u] = o;
  }

  // Software pipeline (issue-early / write-late): the global loads of chunk c+1 are issued
  // into registers BEFORE the MFMA phase of chunk c and written to LDS after it, so HBM/L2
  // latency hides under ~KC*T/2*WM*WN MFMAs even when only one block is resident per CU.
  float preg[PU_MAX];
  f32x4 wreg[WU_MAX];
  bool wfast = false;
  const float* wsrc = nullptr;
  long long krow0 = 0;

  for (int chunk = -1; chunk < p.nchunk; ++chunk) {
    if (chunk >= 0) {
      __syncthreads();                     // every wave is done reading the previous chunk
#pragma unroll
      for (int u = 0; u < PU_MAX; ++u) {
        const int e = tid + u * 256;
        if (e < PE) pl[e] = preg[u];
      }
      if (wfast) {
#pragma unroll
        for (int u = 0; u < WU_MAX; ++u) {
          const int e = tid + u * 256;
          if (e < WE) *reinterpret_cast<f32x4*>(wl + e * 4) = wreg[u];
        }
      } else {                              // ragged edge (partial M block / K tail / unaligned)
        for (int e = tid; e < rows * BM; e += 256) {
          const int r = e / BM, c = e - r * BM;
          float v = 0.f;
          if (m0 + c < p.Mld && krow0 + r < p.Krows) v = wsrc[(long long)r * p.Mld + c];
          wl[e] = v;
        }
      }
      __syncthreads();
    }
    if (chunk + 1 < p.nchunk) {            // issue the next chunk's loads (registers)
      const int c0 = (chunk + 1) * KC;
      const float* base;
      int nvalid;
      if (c0 < p.C0) { base = in0n + (long long)c0 * HWin; nvalid = p.C0 - c0; }
      else { base = in1n + (long long)(c0 - p.C0) * HWin; nvalid = p.Cin - c0; }
      const unsigned limit = (unsigned)(nvalid < KC ? nvalid : KC) * (unsigned)HWin;
#pragma unroll
      for (int u = 0; u < PU_MAX; ++u) {
        float v = 0.f;
        if ((unsigned)toff[u] < limit) v = base[toff[u]];
        preg[u] = v;
      }
      krow0 = (long long)(chunk + 1) * rows;
      wsrc = wpn + krow0 * p.Mld + m0;
      wfast = p.wvec && (m0 + BM <= p.Mld) && (krow0 + rows <= p.Krows);
      if (wfast) {
#pragma unroll
        for (int u = 0; u < WU_MAX; ++u) {
          const int e = tid + u * 256;
          if (e < WE) {
            const int r = e / B4, c4 = e - r * B4;
            wreg[u] = *reinterpret_cast<const f32x4*>(wsrc + (long long)r * p.Mld + c4 * 4);
          }
        }
      }
    }
    if (chunk >= 0) {
      // the MFMA stream yields issue priority to waves that are staging / in their epilogue:
      // their vector instructions otherwise crawl behind it
      __builtin_amdgcn_s_setprio(0);
      mfma_taps<WM, WN, KC / 2>(acc, wl, pl, boff, T, p.KW, KC, BM, PW, PHW, half, l32);
      __builtin_amdgcn_s_setprio(3);
    }
  }

  // ---- epilogue: C/D layout col = lane&31 (pixel), row = (reg&3) + 8*(reg>>2) + 4*half ----
  const ConvEpi epi = scf_conv_epi(p, n);
  const bool use_div = p.out_div != 1.0f;
  int pix[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int oy = ty0 + (wave * WN + j) * FR + fr, ox = tx0 + fc;
    const bool pok = oy < p.Ho && ox < p.Wo;
    const int lin = p.out_tile ? (((oy >> 2) * (p.Wo >> 3) + (ox >> 3)) * 32 + (oy & 3) * 8 + (ox & 7))
                               : oy * p.Wo + ox;
    pix[j] = pok ? lin : -1;
  }
  scf_conv_epilogue_tile<WM, WN>(p, epi, acc, m0, half, pix, use_div);
}

// ---------------------------------------------------------------------------------
// K-split variant for small grids (pose-head layers with 16^2..4^2 outputs, thin layers,
// batch 1): block tile = 32 channels x ONE 32-pixel fragment, and the four waves split the
// k-steps of every staged chunk instead of the pixels.  4x more blocks than the smallest
// pixel-split tile and 4x less serial MFMA work per wave; partial sums are combined through
// LDS in a fixed order (deterministic), each wave finalising 4 of the 16 accumulator rows.
// ---------------------------------------------------------------------------------
template <int KC>
__global__ __launch_bounds__(256, 2) void conv_mfma_ksplit_kernel(ConvK p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int BM = 32, B4 = 8;
  constexpr int PU_MAX = pu_max(KC, 1), WU_MAX = wu_max(KC);
  constexpr int NCP = KC / 2;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, half = lane >> 5;

  const int lb = scf_xcd_remap(blockIdx.x, gridDim.x);
  const int mblk = lb % p.mblocks;
  const int tile = lb / p.mblocks;
  const int m0 = mblk * BM;

  const int FC = 1 << p.fc_log2, FR = 32 >> p.fc_log2;
  const int txi = tile % p.tiles_x;
  const int t2 = tile / p.tiles_x;
  const int tyi = t2 % p.tiles_y;
  const int n = t2 / p.tiles_y;
  const int ty0 = tyi * FR, tx0 = txi * FC;
  const int s = p.stride;
  const int iy0 = ty0 * s - p.pad_h, ix0 = tx0 * s - p.pad_w;
  const int PH = p.PH, PW = p.PW, PHW = PH * PW;
  const int T = p.T;

  float* wl = lds;
  float* pl = lds + KC * T * BM;
  const int fr = l32 >> p.fc_log2, fc = l32 & (FC - 1);
  const int boff = (fr * s) * PW + fc * s + half * PHW;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  const int HWin = p.H * p.W;
  const float* in0n = p.in0 + (long long)n * p.in0_ns;
  const float* in1n = p.in1 ? p.in1 + (long long)n * p.in1_ns : nullptr;
  const float* wpn = p.wp + (long long)n * p.w_ns;
  const int rows = KC * T;
  const int PE = KC * PHW;
  const int WE = rows * B4;

  int toff[PU_MAX];
#pragma unroll
  for (int u = 0; u < PU_MAX; ++u) {
    const int e = tid + u * 256;
    int o = -1;
    if (e < PE) {
      const int cl = e / PHW, r = e - cl * PHW;
      const int py = r / PW, px = r - py * PW;
      const int iy = iy0 + py, ix = ix0 + px;
      if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) o = cl * HWin + iy * p.W + ix;
    }
    toff[u] = o;
  }
  float preg[PU_MAX];
  f32x4 wreg[WU_MAX];
  bool wfast = false;
  const float* wsrc = nullptr;
  long long krow0 = 0;

  for (int chunk = -1; chunk < p.nchunk; ++chunk) {
    if (chunk >= 0) {
      __syncthreads();
#pragma unroll
      for (int u = 0; u < PU_MAX; ++u) {
        const int e = tid + u * 256;
        if (e < PE) pl[e] = preg[u];
      }
      if (wfast) {
#pragma unroll
        for (int u = 0; u < WU_MAX; ++u) {
          const int e = tid + u * 256;
          if (e < WE) *reinterpret_cast<f32x4*>(wl + e * 4) = wreg[u];
        }
      } else {
        for (int e = tid; e < rows * BM; e += 256) {
          const int r = e / BM, c = e - r * BM;
          float v = 0.f;
          if (m0 + c < p.Mld && krow0 + r < p.Krows) v = wsrc[(long long)r * p.Mld + c];
          wl[e] = v;
        }
      }
      __syncthreads();
    }
    if (chunk + 1 < p.nchunk) {
      const int c0 = (chunk + 1) * KC;
      const float* base;
      int nvalid;
      if (c0 < p.C0) { base = in0n + (long long)c0 * HWin; nvalid = p.C0 - c0; }
      else { base = in1n + (long long)(c0 - p.C0) * HWin; nvalid = p.Cin - c0; }
      const unsigned limit = (unsigned)(nvalid < KC ? nvalid : KC) * (unsigned)HWin;
#pragma unroll
      for (int u = 0; u < PU_MAX; ++u) {
        float v = 0.f;
        if ((unsigned)toff[u] < limit) v = base[toff[u]];
        preg[u] = v;
      }
      krow0 = (long long)(chunk + 1) * rows;
      wsrc = wpn + krow0 * p.Mld + m0;
      wfast = p.wvec && (m0 + BM <= p.Mld) && (krow0 + rows <= p.Krows);
      if (wfast) {
#pragma unroll
        for (int u = 0; u < WU_MAX; ++u) {
          const int e = tid + u * 256;
          if (e < WE) {
            const int r = e / B4, c4 = e - r * B4;
            wreg[u] = *reinterpret_cast<const f32x4*>(wsrc + (long long)r * p.Mld + c4 * 4);
          }
        }
      }
    }
    if (chunk >= 0) {
      // wave w takes the k-steps  (tap*NCP + c) % 4 == w
      for (int t = 0; t < T; ++t) {
        const int ky = t / p.KW, kx = t - ky * p.KW;
        const float* wt = wl + (t * KC + half) * BM + l32;
        const float* pt = pl + ky * PW + kx + boff;
#pragma unroll
        for (int c = 0; c < NCP; ++c) {
          if (((t * NCP + c) & 3) == wave)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wt[2 * c * BM], pt[2 * c * PHW], acc, 0, 0, 0);
        }
      }
    }
  }

  // ---- cross-wave reduction (fixed order) + epilogue: wave w finalises rows 4w..4w+3 ----
  __syncthreads();
  float* red = lds;                         // [4 waves][16 regs][64 lanes]
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
  float v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = 4 * wave + q;
    v[q] = ((red[(0 * 16 + r) * 64 + lane] + red[(1 * 16 + r) * 64 + lane]) +
            red[(2 * 16 + r) * 64 + lane]) + red[(3 * 16 + r) * 64 + lane];
  }
  const int oy = ty0 + fr, ox = tx0 + fc;
  if (oy < p.Ho && ox < p.Wo) {
    const ConvEpi epi = scf_conv_epi(p, n);
    const int pixk = p.out_tile ? (((oy >> 2) * (p.Wo >> 3) + (ox >> 3)) * 32 + (oy & 3) * 8 + (ox & 7))
                                : oy * p.Wo + ox;
    scf_conv_epilogue_group(p, epi, v, m0 + 8 * wave + 4 * half, pixk, p.out_div != 1.0f);
  }
}

template <int KC>
static int launch_ksplit(const ConvK& k, int nblk, size_t lds_bytes, hipStream_t st) {
  scf_launch((conv_mfma_ksplit_kernel<KC>), dim3(nblk), dim3(256), lds_bytes, st, k);
  return scf_launch_status();
}

template <int WM, int WN>
static int launch_conv(const ConvK& k, int nblk, size_t lds_bytes, hipStream_t st) {
  if (k.KC == 32)
    scf_launch((conv_mfma_kernel<WM, WN, 32>), dim3(nblk), dim3(256), lds_bytes, st, k);
  else if (k.KC == 8)
    scf_launch((conv_mfma_kernel<WM, WN, 8>), dim3(nblk), dim3(256), lds_bytes, st, k);
  else
    scf_launch((conv_mfma_kernel<WM, WN, 2>), dim3(nblk), dim3(256), lds_bytes, st, k);
  return scf_launch_status();
}

static int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

struct ConvPlan {
  ConvK k;
  int WM, WN;
  long long nblk;
  size_t lds_bytes;
};

static int conv_plan(const scf_conv_desc* d, ConvPlan* plan) {
  if (!d || !d->in0 || !d->wp || !d->out) return SCF_EINVAL;
  if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->C0 <= 0 || d->C1 < 0 || d->Cout <= 0) return SCF_EINVAL;
  if (d->KH <= 0 || d->KW <= 0 || d->stride <= 0 || d->pad_h < 0 || d->pad_w < 0) return SCF_EINVAL;
  if (d->C1 > 0 && !d->in1) return SCF_EINVAL;
  if (d->KC != 2 && d->KC != 8 && d->KC != 32) return SCF_EUNSUPPORTED;
  if (d->C1 > 0 && (d->C0 % d->KC) != 0) return SCF_EUNSUPPORTED;  // a chunk never straddles segments
  if (d->Mld < d->Cout) return SCF_EINVAL;
  if (d->mode == SCF_CONV_GRU_ZR && (!d->gru_h || !d->gru_aux || (d->Cout & 1))) return SCF_EINVAL;
  if (d->mode == SCF_CONV_GRU_Q && (!d->gru_h || !d->gru_z)) return SCF_EINVAL;
  if ((d->scale == nullptr) != (d->shift == nullptr)) return SCF_EINVAL;
  if (d->out_tile8x4 && (d->mode != SCF_CONV_PLAIN || d->res)) return SCF_EUNSUPPORTED;
  if (d->k_slices > 1) {      // partial tensors: raw sums only, the consumer finishes them
    if (d->k_slices > 16 || d->out_slice_stride < (int64_t)d->N * d->out_nstride) return SCF_EINVAL;
    if (d->bias || d->scale || d->res || d->act != SCF_ACT_NONE || d->act_split > 0 || d->mode != SCF_CONV_PLAIN ||
        d->out_tile8x4 || (d->out_div != 0.f && d->out_div != 1.f) || d->w_nstride != 0)
      return SCF_EINVAL;
  }

  ConvK& k = plan->k;
  k.in0 = d->in0; k.in1 = d->C1 > 0 ? d->in1 : nullptr;
  k.C0 = d->C0; k.Cin = d->C0 + d->C1;
  k.in0_ns = d->in0_nstride; k.in1_ns = d->in1_nstride;
  k.H = d->H; k.W = d->W;
  k.Ho = (d->H + 2 * d->pad_h - d->KH) / d->stride + 1;
  k.Wo = (d->W + 2 * d->pad_w - d->KW) / d->stride + 1;
  if (k.Ho <= 0 || k.Wo <= 0) return SCF_EINVAL;
  k.wp = d->wp; k.w_ns = d->w_nstride; k.Mld = d->Mld; k.Cout = d->Cout;
  k.wp16 = d->wp_f16;
  k.wthin = d->wp_thin;
  k.wp4 = d->wp_a4; k.G4 = d->a4_groups; k.Mld4 = d->a4_mld;
  k.wp4s = d->wp_a4s; k.G4s = d->a4s_groups;
  k.wp4t = d->wp_a4t; k.G4t = d->a4t_groups;
  k.out_tile = d->out_tile8x4;
  k.KH = d->KH; k.KW = d->KW; k.T = d->KH * d->KW; k.stride = d->stride; k.in_step = 1;
  k.pad_h = d->pad_h; k.pad_w = d->pad_w; k.KC = d->KC;
  k.nchunk = (k.Cin + k.KC - 1) / k.KC;
  // shared packed weights carry zero rows up to nchunk*KC*T; per-sample "weights" (correlation
  // build: a raw feature map) only have Cin*T rows.
  k.Krows = d->w_nstride ? k.Cin * k.T : k.nchunk * k.KC * k.T;
  k.wvec = ((d->Mld & 3) == 0) && ((d->w_nstride & 3) == 0) && (((uintptr_t)d->wp & 15) == 0);
  k.out = d->out; k.out_ns = d->out_nstride;
  k.bias = d->bias; k.scale = d->scale; k.shift = d->shift;
  k.res = d->res; k.res_ns = d->res_nstride;
  k.out_div = d->out_div == 0.f ? 1.f : d->out_div;
  {
    int ex = 0;
    k.out_div_pow2 = (k.out_div > 0.f && frexpf(k.out_div, &ex) == 0.5f) ? 1 : 0;
  }
  k.act = d->act; k.act2 = d->act2; k.act_split = d->act_split; k.mode = d->mode;
  k.gru_h = d->gru_h; k.gru_h_ns = d->gru_h_nstride;
  k.gru_aux = d->gru_aux; k.gru_aux_ns = d->gru_aux_nstride;
  k.gru_z = d->gru_z; k.gru_z_ns = d->gru_z_nstride;
  k.kslices = d->k_slices > 1 ? d->k_slices : 1; k.slice_ns = d->out_slice_stride; k.slice_blocks = 0;

  // fragment = FR rows x FC columns of the output (FR * FC = 32).  FC = 32 unless a narrower
  // fragment wastes clearly fewer columns of the last tile of each row (Wo = 80: 3 x 32 covers 96
  // columns, 5 x 16 covers 80 -- 17 % fewer MFMAs on the 60 x 80 maps of a 480 x 640 crop)
  int FC = next_pow2(k.Wo) < 32 ? next_pow2(k.Wo) : 32;
  if (k.Wo > 32) {
    int best_w = (k.Wo + 31) / 32 * 32;
    for (int c = 16; c >= 8; c >>= 1) {
      const int wpad = (k.Wo + c - 1) / c * c;
      if (wpad * 10 <= best_w * 9) { best_w = wpad; FC = c; }      // at least 10 % fewer columns
    }
  }
  int fl = 0;
  while ((1 << fl) < FC) ++fl;
  k.fc_log2 = fl;
  const int FR = 32 / FC;

  // ---- tile shape selection -------------------------------------------------------------
  // WM channel fragments x WN pixel fragments per wave (<= 4 accumulators = 64 VGPRs).  Start
  // from the largest tile and shrink WM while the grid would leave CUs idle (< 2 blocks per
  // CU); use the double-width pixel tile only when the grid is large anyway.  Every choice
  // must fit the per-thread prefetch registers and 64 KiB of LDS.
  const int frags_m = (k.Cout + 31) / 32;
  auto tiles = [&](int WN) {
    const int TR = WN * 4 * FR;
    return (long long)d->N * ((k.Ho + TR - 1) / TR) * ((k.Wo + FC - 1) / FC);
  };
  auto fits = [&](int WM, int WN, size_t* lds_out) {
    const int TR = WN * 4 * FR;
    const int PH = (TR - 1) * k.stride + k.KH, PW = (FC - 1) * k.stride + k.KW;
    const long long PE = (long long)k.KC * PH * PW, WE = (long long)k.KC * k.T * WM * 8;
    const size_t lds = ((size_t)k.KC * k.T * WM * 32 + (size_t)PE) * sizeof(float);
    if (lds_out) *lds_out = lds;
    return (PE + 255) / 256 <= pu_max(k.KC, WM * WN) && (WE + 255) / 256 <= wu_max(k.KC) && lds <= 64 * 1024;
  };
  int WM = frags_m >= 4 ? ((frags_m % 4 == 0 || frags_m % 3 != 0) ? 4 : 3) : frags_m;
  int WN = 1;
  auto nblocks = [&](int wm, int wn) { return tiles(wn) * ((frags_m + wm - 1) / wm); };
  while (WM > 1 && (nblocks(WM, 1) < 512 || !fits(WM, 1, nullptr))) {
    int next = WM - 1;
    while (next > 1 && frags_m % next != 0) --next;
    WM = next;
  }
  if (WM <= 2 && nblocks(WM, 2) >= 2048 && fits(WM, 2, nullptr)) WN = 2;
  size_t lds_bytes = 0;
  const bool normal_fits = fits(WM, WN, &lds_bytes);
  k.mblocks = (frags_m + WM - 1) / WM;
  {
    const int TR = WN * 4 * FR;
    k.PH = (TR - 1) * k.stride + k.KH;
    k.PW = (FC - 1) * k.stride + k.KW;
    k.tiles_y = (k.Ho + TR - 1) / TR;
    k.tiles_x = (k.Wo + FC - 1) / FC;
  }
  plan->nblk = (long long)d->N * k.tiles_y * k.tiles_x * k.mblocks;
  if (plan->nblk > 0x7fffffffLL) return SCF_EUNSUPPORTED;
  // Still a small grid with the smallest pixel-split tile: split K across the 4 waves instead
  // (32-pixel tiles, 4x the blocks).
  // (a) truly small grids (fewer blocks than half the CUs: batch 1), or (b) the pixel-split
  // tile does not fit this chunk size at all (KC = 32 on a strided layer) while the 32-pixel
  // tile does.  Not for grids that already fill the chip: 4x the blocks means 4x the staging.
  k.ksplit = 0;
  if (!normal_fits && !(WM == 1 && WN == 1)) return SCF_EUNSUPPORTED;
  if (WM == 1 && WN == 1 && (plan->nblk < 128 || !normal_fits)) {
    const int PHk = (FR - 1) * k.stride + k.KH, PWk = (FC - 1) * k.stride + k.KW;
    const long long PEk = (long long)k.KC * PHk * PWk, WEk = (long long)k.KC * k.T * 8;
    size_t ldsk = ((size_t)k.KC * k.T * 32 + (size_t)PEk) * sizeof(float);
    if (ldsk < 4 * 16 * 64 * sizeof(float)) ldsk = 4 * 16 * 64 * sizeof(float);
    if ((PEk + 255) / 256 <= pu_max(k.KC, 1) && (WEk + 255) / 256 <= wu_max(k.KC) && ldsk <= 64 * 1024) {
      k.ksplit = 1;
      k.PH = PHk; k.PW = PWk;
      k.tiles_y = (k.Ho + FR - 1) / FR;
      k.tiles_x = (k.Wo + FC - 1) / FC;
      k.mblocks = frags_m;
      lds_bytes = ldsk;
      plan->nblk = (long long)d->N * k.tiles_y * k.tiles_x * k.mblocks;
    }
  }
  if (!normal_fits && !k.ksplit) return SCF_EUNSUPPORTED;
  plan->WM = WM; plan->WN = WN; plan->lds_bytes = lds_bytes;
  return SCF_OK;
}

// split-fp16 eligibility: shared weights, >= 16 input channels, spatial kernel (dense 1x1
// layers stay on the fp32 KC=32 kernel), shape fits the fp16 kernel's staging budget.
static bool want_f16x3(const scf_conv_desc* d) {
  return d->wp_f16 != nullptr && !d->out_tile8x4 && d->w_nstride == 0 && d->KH * d->KW > 1 && d->C0 + d->C1 >= 16;
}

static std::atomic<int> g_pair_mode{0};  // SCF_TUNE_CONV_PAIR
static std::atomic<int> g_wino1d4{1};    // SCF_TUNE_WINO1D4: 0 = F(2, 5) also where an F(4, 5) packing is given (A/B measurements), 2 = F(4, 5) on every grid

static bool want_dma(const scf_conv_desc* d) {      // a layer opts in by carrying the LDS-DMA packing
  return (d->wp_a4 != nullptr || d->wp_a4s != nullptr) && (d->stride == 1 || d->stride == 2) &&
         d->w_nstride == 0 && d->a4_mld >= d->Cout;
}

// ---- workspace for automatically K-sliced small-grid launches: one per launch stream (two streams never share one:
//      the side-stream branches of a step run concurrently), registered by the caller, borrowed ----
int scf_conv_kcombine_launch(const ConvK& k, const float* parts, int S, long long slice_ns, int N, hipStream_t st);   // conv_dma.hip
int scf_conv_dma_autoslice(const ConvK& k, int N);
namespace {
struct KWorkspace { hipStream_t st; float* ptr; int64_t floats; };
std::mutex g_kws_mu;
std::vector<KWorkspace> g_kws;
std::atomic<int> g_autoslice{1};      // SCF_TUNE_CONV_AUTOSLICE
}
extern "C" int scf_conv_workspace(scf_stream_t stream, float* ptr, int64_t floats) {
  if (floats < 0 || (ptr == nullptr) != (floats == 0)) return SCF_EINVAL;
  std::lock_guard<std::mutex> lk(g_kws_mu);
  hipStream_t st = scf_stream(stream);
  for (size_t i = 0; i < g_kws.size(); ++i)
    if (g_kws[i].st == st) {
      if (ptr) { g_kws[i].ptr = ptr; g_kws[i].floats = floats; }
      else g_kws.erase(g_kws.begin() + (long)i);
      return SCF_OK;
    }
  if (ptr) g_kws.push_back({st, ptr, floats});
  return SCF_OK;
}
static bool kws_lookup(hipStream_t st, float** ptr, int64_t* floats) {
  std::lock_guard<std::mutex> lk(g_kws_mu);
  for (const KWorkspace& w : g_kws)
    if (w.st == st) { *ptr = w.ptr; *floats = w.floats; return true; }
  return false;
}

// which kernel family took the launch (SCF_KERNEL_* of scflow_hip_prof.h)
static int conv2d_launch(const scf_conv_desc* d, scf_stream_t stream, int* which) {
  ConvPlan pl;
  const int rc = conv_plan(d, &pl);
  if (rc != SCF_OK) return rc;
  if (d->k_slices > 1) {        // only the LDS-DMA kernel splits K across blocks
    *which = SCF_KERNEL_DMA;
    if (!want_dma(d)) return SCF_EUNSUPPORTED;
    return scf_conv_dma_dispatch(pl.k, d->N, false, nullptr, scf_stream(stream));
  }
  {
    *which = SCF_KERNEL_THIN;
    const int rt = scf_conv_thin_dispatch(pl.k, d->N, false, scf_stream(stream));
    if (rt != SCF_EUNSUPPORTED) return rt;
  }
  if (d->wp_taps) {
    *which = SCF_KERNEL_TAPS;
    const int rp = scf_conv_taps_dispatch(pl.k, d->wp_taps, d->N, false, nullptr, scf_stream(stream));
    if (rp != SCF_EUNSUPPORTED) return rp;
  }
  if (d->wp_wino) {
    int quarter = 0;
    const int rw = scf_conv_wino_dispatch(pl.k, d->wp_wino, d->N, false, nullptr, scf_stream(stream), &quarter);
    *which = quarter ? SCF_KERNEL_WINO_Q : SCF_KERNEL_WINO;
    if (rw != SCF_EUNSUPPORTED) return rw;
  }
  if (d->wp_wino1d4 && g_wino1d4.load(std::memory_order_relaxed)) {
    *which = SCF_KERNEL_WINO1D4;
    const int rw = scf_conv_wino1d4_dispatch(pl.k, d->wp_wino1d4, d->N, g_wino1d4.load(std::memory_order_relaxed) == 2, false, nullptr, scf_stream(stream));
    if (rw != SCF_EUNSUPPORTED) return rw;
  }
  if (d->wp_wino1d) {
    *which = SCF_KERNEL_WINO1D;
    const int rw = scf_conv_wino1d_dispatch(pl.k, d->wp_wino1d, d->N, false, nullptr, scf_stream(stream));
    if (rw != SCF_EUNSUPPORTED) return rw;
  }
  if (want_f16x3(d)) {
    *which = SCF_KERNEL_F16X3;
    const int r16 = scf_conv_f16x3_dispatch(pl.k, d->N, false, nullptr, scf_stream(stream));
    if (r16 != SCF_EUNSUPPORTED) return r16;
  }
  if (want_dma(d)) {
    *which = SCF_KERNEL_DMA;
    // small-grid launch with a workspace on its stream: S slices into the workspace + the combine launch (conv_dma.hip)
    float* ws = nullptr;
    int64_t ws_floats = 0;
    if (g_autoslice.load(std::memory_order_relaxed) && d->k_slices <= 1 && kws_lookup(scf_stream(stream), &ws, &ws_floats)) {
      const int S = scf_conv_dma_autoslice(pl.k, d->N);
      const int64_t per_slice = (int64_t)d->N * d->Cout * pl.k.Ho * pl.k.Wo;
      if (S > 1 && per_slice * S <= ws_floats) {
        ConvK part = pl.k;
        part.out = ws; part.out_ns = (long long)d->Cout * pl.k.Ho * pl.k.Wo;
        part.bias = nullptr; part.scale = nullptr; part.shift = nullptr; part.res = nullptr; part.res_ns = 0;
        part.out_div = 1.f; part.out_div_pow2 = 1; part.act = SCF_ACT_NONE; part.act2 = SCF_ACT_NONE; part.act_split = 0;
        part.mode = SCF_CONV_PLAIN; part.gru_h = nullptr; part.gru_aux = nullptr; part.gru_z = nullptr;
        part.kslices = S; part.slice_ns = per_slice;
        const int rs = scf_conv_dma_dispatch(part, d->N, false, nullptr, scf_stream(stream));
        if (rs == SCF_OK) return scf_conv_kcombine_launch(pl.k, ws, S, per_slice, d->N, scf_stream(stream));
        if (rs != SCF_EUNSUPPORTED) return rs;
      }
    }
    const int rd = scf_conv_dma_dispatch(pl.k, d->N, false, nullptr, scf_stream(stream));
    if (rd != SCF_EUNSUPPORTED) return rd;
  }
  const ConvK& k = pl.k;
  const int WM = pl.WM, WN = pl.WN;
  const long long nblk = pl.nblk;
  const size_t lds_bytes = pl.lds_bytes;
  hipStream_t st = scf_stream(stream);
  if (k.ksplit) {
    *which = SCF_KERNEL_MFMA_KSPLIT;
    if (k.KC == 32) return launch_ksplit<32>(k, (int)nblk, lds_bytes, st);
    if (k.KC == 8) return launch_ksplit<8>(k, (int)nblk, lds_bytes, st);
    return launch_ksplit<2>(k, (int)nblk, lds_bytes, st);
  }
  *which = SCF_KERNEL_MFMA;
#define SCF_CASE(M, Nn) if (WM == M && WN == Nn) return launch_conv<M, Nn>(k, (int)nblk, lds_bytes, st);
  SCF_CASE(1, 1) SCF_CASE(1, 2) SCF_CASE(2, 1) SCF_CASE(2, 2)
  SCF_CASE(3, 1) SCF_CASE(4, 1)
#undef SCF_CASE
  return SCF_EUNSUPPORTED;
}

// ---- dispatch log (scflow_hip_prof.h): which kernel family ran each convolution launch of this process,
//      including the launches issued inside scf_sepconv_gru* / scf_scflow_iteration ----
namespace {
std::mutex g_log_mu;
std::vector<scf_conv_log_entry> g_log;
std::atomic<int> g_log_cap{0};
}

int scf_wino_variant_set(int v);      // conv_wino.hip
int scf_dma_force_ksplit_set(int v);  // conv_dma.hip
int scf_dma_ksplit_groups_set(int v);
int scf_lookup_pipe_set(int v);       // corr_lookup.hip
int scf_lookup_store_set(int v);
int scf_iter_merge_set(int v);        // scflow_iter.hip
int scf_wino1d4_half_set(int v);    // conv_wino1d4.hip

extern "C" int scf_tune(int key, int value) {
  if (key == SCF_TUNE_WINO_VARIANT) return scf_wino_variant_set(value);
  if (key == SCF_TUNE_DMA_FORCE_KSPLIT) return scf_dma_force_ksplit_set(value);
  if (key == SCF_TUNE_DMA_KSPLIT_GROUPS) return scf_dma_ksplit_groups_set(value);
  if (key == SCF_TUNE_LOOKUP_PIPE) return scf_lookup_pipe_set(value);
  if (key == SCF_TUNE_LOOKUP_STORE) return scf_lookup_store_set(value);
  if (key == SCF_TUNE_ITER_MERGE) return scf_iter_merge_set(value);
  if (key == SCF_TUNE_WINO1D4_HALF) return scf_wino1d4_half_set(value);
  if (key == SCF_TUNE_CONV_AUTOSLICE) {
    if (value < 0 || value > 1) return SCF_EINVAL;
    return g_autoslice.exchange(value);
  }
  if (key == SCF_TUNE_CONV_PAIR) {
    if (value < 0 || value > 1) return SCF_EINVAL;
    return g_pair_mode.exchange(value);
  }
  if (key == SCF_TUNE_WINO1D4) {
    if (value < 0 || value > 2) return SCF_EINVAL;
    return g_wino1d4.exchange(value);
  }
  return SCF_EINVAL;
}

extern "C" int scf_conv_log_enable(int capacity) {
  std::lock_guard<std::mutex> lk(g_log_mu);
  g_log.clear();
  if (capacity > 0) g_log.reserve((size_t)capacity);
  else g_log.shrink_to_fit();
  g_log_cap.store(capacity > 0 ? capacity : 0);
  return SCF_OK;
}

extern "C" int scf_conv_log_read(scf_conv_log_entry* out, int max_entries) {
  std::lock_guard<std::mutex> lk(g_log_mu);
  const int n = (int)g_log.size();
  if (out)
    for (int i = 0; i < n && i < max_entries; ++i) out[i] = g_log[(size_t)i];
  return n;
}

static void conv_log_push(const scf_conv_desc* d, int which);
extern "C" int scf_conv2d(const scf_conv_desc* d, scf_stream_t stream) {
  int which = 0;
  const int rc = conv2d_launch(d, stream, &which);
  if (rc == SCF_OK) conv_log_push(d, which);
  return rc;
}

// ---------------------------------------------------------------------------------
// r6: two INDEPENDENT convolutions as one launch where both fall to the same small-grid kernel instantiation (the K-split
// LDS-DMA tile, the thin-input kernel): blocks [0, nA) run a, the rest b -- the branch-level concurrency of batch 1-4 without a
// second stream (conv_dma_pair_kernel).  Anything else: the two launches one after the other.  Same results either way.
// ---------------------------------------------------------------------------------
static void conv_log_push(const scf_conv_desc* d, int which) {
  if (g_log_cap.load(std::memory_order_relaxed) <= 0) return;
  std::lock_guard<std::mutex> lk(g_log_mu);
  if ((int)g_log.size() >= g_log_cap.load()) return;
  scf_conv_log_entry e;
  e.kernel = which; e.Cin = d->C0 + d->C1; e.Cout = d->Cout; e.KH = d->KH; e.KW = d->KW; e.stride = d->stride;
  e.Ho = (d->H + 2 * d->pad_h - d->KH) / d->stride + 1;
  e.Wo = (d->W + 2 * d->pad_w - d->KW) / d->stride + 1;
  e.N = d->N; e.mode = d->mode;
  g_log.push_back(e);
}

// the family that WOULD take d (same precedence as conv2d_launch) with its launch captured, or -1 when it is not a pairable one
static int conv2d_capture(const scf_conv_desc* d, scf_stream_t stream, ScfLaunchCap* cap) {
  ConvPlan pl;
  if (conv_plan(d, &pl) != SCF_OK) return -1;
  int32_t info[4];
  if (d->k_slices > 1) {
    if (!want_dma(d)) return -1;
    return (scf_conv_dma_dispatch(pl.k, d->N, true, info, nullptr, cap) == SCF_OK && cap->variant >= 0) ? SCF_KERNEL_DMA : -1;
  }
  if (scf_conv_thin_dispatch(pl.k, d->N, true, nullptr, cap) == SCF_OK) return SCF_KERNEL_THIN;
  if (d->wp_taps) {
    const int rp = scf_conv_taps_dispatch(pl.k, d->wp_taps, d->N, true, info, nullptr, cap);
    if (rp == SCF_OK) return cap->variant >= 1 ? SCF_KERNEL_TAPS : -1;
    if (rp != SCF_EUNSUPPORTED) return -1;
  }
  if (d->wp_wino) {
    int quarter = 0;
    if (scf_conv_wino_dispatch(pl.k, d->wp_wino, d->N, true, info, nullptr, &quarter, cap) == SCF_OK)
      return (quarter && cap->variant >= 10 && cap->variant < 20) ? SCF_KERNEL_WINO_Q : (!quarter && cap->variant >= 20) ? SCF_KERNEL_WINO : -1;
  }
  if (d->wp_wino1d4 && g_wino1d4.load(std::memory_order_relaxed) &&
      scf_conv_wino1d4_dispatch(pl.k, d->wp_wino1d4, d->N, g_wino1d4.load(std::memory_order_relaxed) == 2, true, info, nullptr) == SCF_OK)
    return -1;
  if (d->wp_wino1d && scf_conv_wino1d_dispatch(pl.k, d->wp_wino1d, d->N, true, info, nullptr) == SCF_OK) return -1;
  if (want_f16x3(d) && scf_conv_f16x3_dispatch(pl.k, d->N, true, info, nullptr) == SCF_OK) return -1;
  if (want_dma(d)) {
    float* ws = nullptr;
    int64_t ws_floats = 0;
    if (g_autoslice.load(std::memory_order_relaxed) && kws_lookup(scf_stream(stream), &ws, &ws_floats)) return -1;
    if (scf_conv_dma_dispatch(pl.k, d->N, true, info, nullptr, cap) == SCF_OK) return cap->variant >= 0 ? SCF_KERNEL_DMA : -1;
  }
  return -1;
}

extern "C" int scf_conv2d_pair(const scf_conv_desc* a, const scf_conv_desc* b, scf_stream_t stream) {
  if (!a || !b) return SCF_EINVAL;
  ScfLaunchCap ca, cb;
  const int fa = conv2d_capture(a, stream, &ca);
  if (fa >= 0) {
    const int fb = conv2d_capture(b, stream, &cb);
    // one launch only while both grids are resident together: past that the merged launch is a second round of blocks and
    // loses (batch 2, corr1 384 + flow1 128 blocks of 100+ KB: +11 us per pair, profiles/r6_b1_pairs.txt)
    // blocks that can be resident at once: thin-input and quarter-domain Winograd blocks fit two per CU, K-split blocks too while
    // their ring stays under half the LDS (the 32-channel-chunk packing of tiny grids takes up to 144 KB: one per CU)
    const size_t lmax = ca.ldsb > cb.ldsb ? ca.ldsb : cb.ldsb;
    long long slots = (long long)scf_cu_count() * (((fa == SCF_KERNEL_DMA || fb == SCF_KERNEL_DMA) && lmax > 80 * 1024) ? 1 : 2);
    if (g_pair_mode.load(std::memory_order_relaxed) == 1) slots = 0;      // scf_tune(SCF_TUNE_CONV_PAIR, 1): never one launch
    if (((fa == SCF_KERNEL_DMA && fb == SCF_KERNEL_TAPS) || (fa == SCF_KERNEL_TAPS && fb == SCF_KERNEL_DMA)) &&
        (long long)ca.nblk + cb.nblk <= slots && g_pair_mode.load(std::memory_order_relaxed) != 1) {
      // a K-split layer beside a thin-input layer (corr_net.0 | flow_net.0): one launch, the K-split grid first
      const bool a_dma = fa == SCF_KERNEL_DMA;
      const int rc = scf_conv_dma_taps_pair_launch(a_dma ? ca : cb, a_dma ? cb : ca, scf_stream(stream));
      if (rc == SCF_OK) {
        conv_log_push(a, fa);
        conv_log_push(b, fb);
        return SCF_OK;
      }
      if (rc != SCF_EUNSUPPORTED) return rc;
    }
    // ... or, for the MFMA-bound quarter-domain Winograd kernel, while one launch needs fewer ROUNDS of resident blocks than two
    // (batch 32: corr_net.1 768 + flow_net.1 256 blocks = 1.5 + 0.5 rounds apart, 2 full rounds together)
    // (thin-input blocks are light -- ~30 KB of LDS, ~100 registers -- four are resident per CU)
    const long long rslots = fa == SCF_KERNEL_TAPS ? 2 * slots : slots;
    const bool fewer_rounds = (fa == SCF_KERNEL_WINO_Q || fa == SCF_KERNEL_TAPS) && fb == fa && g_pair_mode.load(std::memory_order_relaxed) != 1 &&
                              (ca.nblk + cb.nblk + rslots - 1) / rslots < (ca.nblk + rslots - 1) / rslots + (cb.nblk + rslots - 1) / rslots;
    if (((fa == SCF_KERNEL_WINO_Q && fb == SCF_KERNEL_WINO) || (fa == SCF_KERNEL_WINO && fb == SCF_KERNEL_WINO_Q)) &&
        g_pair_mode.load(std::memory_order_relaxed) != 1 &&
        ((long long)ca.nblk + cb.nblk <= slots ||
         (ca.nblk + cb.nblk + slots - 1) / slots < (ca.nblk + slots - 1) / slots + (cb.nblk + slots - 1) / slots)) {
      // a quarter-domain layer beside a pair-kernel layer (delta_flow_encoder.1 | mask_encoder.1): the quarter-domain grid first
      const bool a_q = fa == SCF_KERNEL_WINO_Q;
      const int rc = scf_conv_wino_pair_launch(a_q ? ca : cb, a_q ? cb : ca, scf_stream(stream));
      if (rc == SCF_OK) {
        conv_log_push(a, fa);
        conv_log_push(b, fb);
        return SCF_OK;
      }
      if (rc != SCF_EUNSUPPORTED) return rc;
    }
    if (fb == fa && fa != SCF_KERNEL_WINO && ((long long)ca.nblk + cb.nblk <= slots || fewer_rounds)) {
      const int rc = fa == SCF_KERNEL_WINO_Q ? scf_conv_wino_pair_launch(ca, cb, scf_stream(stream))
                   : fa == SCF_KERNEL_THIN ? scf_conv_thin_pair_launch(ca, cb, scf_stream(stream)) : fa == SCF_KERNEL_TAPS ? scf_conv_taps_pair_launch(ca, cb, scf_stream(stream))
                                           : scf_conv_dma_pair_launch(ca, cb, scf_stream(stream));
      if (rc == SCF_OK) {
        conv_log_push(a, fa);
        conv_log_push(b, fb);
        return SCF_OK;
      }
      if (rc != SCF_EUNSUPPORTED) return rc;
    }
  }
  const int r1 = scf_conv2d(a, stream);
  return r1 != SCF_OK ? r1 : scf_conv2d(b, stream);
}

// ---------------------------------------------------------------------------------
// ConvGRU.forward (raft_decoder.py:235-253) as one entry point: per pass the fused z|r
// convolution (epilogue: z -> z buffer, r*h -> rh buffer) and the q convolution (epilogue:
// tanh + state update in place).
// ---------------------------------------------------------------------------------
extern "C" int scf_conv2d_query(const scf_conv_desc* d, int32_t* info);

// Short-chunk regime (the register-staged kernel with its smallest tile, WM = WN = 1): with KC = 8
// the MFMA phase of a chunk is shorter than the L2 round trip its prefetch has to hide, so the
// KC = 32 packing is used when the caller provides one and it fits (same rule as ops.conv2d).
static void gru_choose_packing(scf_conv_desc& d, const float* wp8, const float* wp32) {
  d.wp = wp8; d.KC = 8;
  if (!wp32 || (d.C1 > 0 && (d.C0 % 32) != 0)) return;
  int32_t info[4];
  if (scf_conv2d_query(&d, info) == SCF_OK && info[0] * info[1] == 1 && info[3] >= 0 && info[3] * 64 < 6000) {
    d.wp = wp32; d.KC = 32;
    if (scf_conv2d_query(&d, info) != SCF_OK) { d.wp = wp8; d.KC = 8; }
  }
}

// hx = [h (Ch) | skipped (Cskip) | x (Cx)]: the convolutions read h (or r*h) and x; ctx[i] (may be
// NULL) = pass i's pre-activation term (N, 3 Ch, H, W): channels [0, 2Ch) for z | r, [2Ch, 3Ch) for q
static int sepconv_gru_impl(float* hx, int64_t hx_nstride, int N, int Ch, int Cskip, int Cx, int H, int W,
                            const scf_gru_pass* passes, int npass, const float* const* ctx,
                            int64_t ctx_nstride, float* z, float* rh, scf_stream_t stream) {
  if (!hx || !passes || !z || !rh || N <= 0 || Ch <= 0 || Cx <= 0 || Cskip < 0 || H <= 0 || W <= 0 || npass <= 0)
    return SCF_EINVAL;
  if (Ch % 8 != 0) return SCF_EUNSUPPORTED;      // the h | x boundary must not split a channel chunk
  const int64_t hw = (int64_t)H * W;
  float* xin = hx + (int64_t)(Ch + Cskip) * hw;
  for (int i = 0; i < npass; ++i) {
    const scf_gru_pass& g = passes[i];
    if (!g.wp_zr || !g.wp_q || g.KH <= 0 || g.KW <= 0) return SCF_EINVAL;
    if (2 * g.pad_h != g.KH - 1 || 2 * g.pad_w != g.KW - 1) return SCF_EUNSUPPORTED;   // 'same' convolutions
    const float* cx = ctx ? ctx[i] : nullptr;
    scf_conv_desc d = {};
    d.N = N; d.H = H; d.W = W;
    d.KH = g.KH; d.KW = g.KW; d.stride = 1; d.pad_h = g.pad_h; d.pad_w = g.pad_w; d.KC = 8;
    d.out_div = 1.f;
    // z | r = sigmoid(conv([h | x]) [+ ctx]): z -> z, r*h -> rh
    if (Cskip == 0) {
      d.in0 = hx; d.C0 = Ch + Cx; d.in0_nstride = hx_nstride;
    } else {
      d.in0 = hx; d.C0 = Ch; d.in0_nstride = hx_nstride;
      d.in1 = xin; d.C1 = Cx; d.in1_nstride = hx_nstride;
    }
    d.Mld = (2 * Ch + 31) / 32 * 32; d.Cout = 2 * Ch; d.bias = g.bias_zr;
    d.wp_a4 = g.wp_zr_a4; d.a4_groups = g.a4_groups; d.a4_mld = d.Mld; d.wp_f16 = g.wp_zr_f16;
    d.wp_a4s = g.wp_zr_a4s; d.a4s_groups = g.a4s_groups;
    d.wp_a4t = g.wp_zr_a4t; d.a4t_groups = g.a4t_groups;
    d.wp_wino1d = g.wp_zr_wino1d; d.wp_wino1d4 = g.wp_zr_wino1d4;
    d.out = z; d.out_nstride = Ch * hw;
    d.mode = SCF_CONV_GRU_ZR; d.gru_h = hx; d.gru_h_nstride = hx_nstride;
    d.gru_aux = rh; d.gru_aux_nstride = Ch * hw;
    d.res = cx; d.res_nstride = cx ? ctx_nstride : 0;
    gru_choose_packing(d, g.wp_zr, g.wp_zr_k32);
    int rc = scf_conv2d(&d, stream);
    if (rc != SCF_OK) return rc;
    // q = tanh(conv([r*h | x]) [+ ctx]); h <- (1 - z) h + z q
    d.in0 = rh; d.C0 = Ch; d.in0_nstride = Ch * hw;
    d.in1 = xin; d.C1 = Cx; d.in1_nstride = hx_nstride;
    d.Mld = (Ch + 31) / 32 * 32; d.Cout = Ch; d.bias = g.bias_q;
    d.wp_a4 = g.wp_q_a4; d.a4_mld = d.Mld; d.wp_f16 = g.wp_q_f16;
    d.wp_a4s = g.wp_q_a4s;
    d.wp_a4t = g.wp_q_a4t;
    d.wp_wino1d = g.wp_q_wino1d; d.wp_wino1d4 = g.wp_q_wino1d4;
    d.out = hx; d.out_nstride = hx_nstride;
    d.mode = SCF_CONV_GRU_Q; d.gru_h = hx; d.gru_h_nstride = hx_nstride;
    d.gru_aux = nullptr; d.gru_aux_nstride = 0;
    d.gru_z = z; d.gru_z_nstride = Ch * hw;
    d.res = cx ? cx + (int64_t)2 * Ch * hw : nullptr;
    gru_choose_packing(d, g.wp_q, g.wp_q_k32);
    rc = scf_conv2d(&d, stream);
    if (rc != SCF_OK) return rc;
  }
  return SCF_OK;
}

extern "C" int scf_sepconv_gru(float* hx, int64_t hx_nstride, int N, int Ch, int Cx, int H, int W,
                               const scf_gru_pass* passes, int npass, float* z, float* rh,
                               scf_stream_t stream) {
  return sepconv_gru_impl(hx, hx_nstride, N, Ch, 0, Cx, H, W, passes, npass, nullptr, 0, z, rh, stream);
}

// The same update with the iteration-invariant part of x hoisted out of the refinement loop:
// hx = [h | c | x'] where c (Cc channels, e.g. RAFT's context features) does not change between
// iterations.  conv([h | c | x']) = conv_hx'([h | x']) + conv_c(c): the caller computes
// ctx[i] = conv_c(c) + bias once per pair (a plain scf_conv2d over c with the z | r | q weight
// columns of c stacked into 3 Ch output rows) and passes packings over the Ch + Cx remaining input
// channels (bias_zr / bias_q = NULL: folded into ctx).  Saves Cc / (Ch + Cc + Cx) of the GRU's
// multiply-adds in every iteration but the first.
extern "C" int scf_sepconv_gru_ctx(float* hx, int64_t hx_nstride, int N, int Ch, int Cc, int Cx, int H,
                                   int W, const scf_gru_pass* passes, int npass,
                                   const float* const* ctx, int64_t ctx_nstride, float* z, float* rh,
                                   scf_stream_t stream) {
  if (!ctx || Cc <= 0) return SCF_EINVAL;
  for (int i = 0; i < npass; ++i)
    if (!ctx[i]) return SCF_EINVAL;
  return sepconv_gru_impl(hx, hx_nstride, N, Ch, Cc, Cx, H, W, passes, npass, ctx, ctx_nstride, z, rh,
                          stream);
}

// Dry run of the tile selection for a descriptor: info[0..3] = WM, WN, grid blocks, MFMA
// instructions per wave per staged channel chunk.  Lets the host side choose between weight
// packings (KC = 8 vs 32) without launching.
extern "C" int scf_conv2d_query(const scf_conv_desc* d, int32_t* info) {
  if (!info) return SCF_EINVAL;
  ConvPlan pl;
  const int rc = conv_plan(d, &pl);
  if (rc != SCF_OK) return rc;
  if (d->k_slices > 1) {
    if (!want_dma(d)) return SCF_EUNSUPPORTED;
    const int rs = scf_conv_dma_dispatch(pl.k, d->N, true, info, nullptr);
    if (rs == SCF_OK) info[3] = -info[3];
    return rs;
  }
  if (scf_conv_thin_dispatch(pl.k, d->N, true, nullptr) == SCF_OK) {
    info[0] = info[1] = 0; info[2] = 0; info[3] = -1;      // vector-ALU thin-output kernel
    return SCF_OK;
  }
  if (d->wp_taps && scf_conv_taps_dispatch(pl.k, d->wp_taps, d->N, true, info, nullptr) == SCF_OK) {
    info[3] = -info[3];      // negative: the thin-input kernel will run, KC of d is irrelevant
    return SCF_OK;
  }
  if (d->wp_wino && scf_conv_wino_dispatch(pl.k, d->wp_wino, d->N, true, info, nullptr) == SCF_OK) {
    info[3] = -info[3];      // negative: the Winograd kernel will run (info = 16 positions, fragments per block, blocks, -LDS bytes)
    return SCF_OK;
  }
  if (d->wp_wino1d4 && g_wino1d4.load(std::memory_order_relaxed) &&
      scf_conv_wino1d4_dispatch(pl.k, d->wp_wino1d4, d->N, g_wino1d4.load(std::memory_order_relaxed) == 2, true, info, nullptr) == SCF_OK) {
    info[3] = -info[3];      // negative: the F(4, 5) kernel will run (info = 8 positions, 4 fragments per block, blocks, -LDS bytes)
    return SCF_OK;
  }
  if (d->wp_wino1d && scf_conv_wino1d_dispatch(pl.k, d->wp_wino1d, d->N, true, info, nullptr) == SCF_OK) {
    info[3] = -info[3];      // negative: the F(2, 5) kernel will run (info = 6 positions, 4 fragments per block, blocks, -LDS bytes)
    return SCF_OK;
  }
  if (want_f16x3(d) && scf_conv_f16x3_dispatch(pl.k, d->N, true, info, nullptr) == SCF_OK) {
    info[3] = -info[3];      // negative: the split-fp16 kernel will run
    return SCF_OK;
  }
  if (want_dma(d) && scf_conv_dma_dispatch(pl.k, d->N, true, info, nullptr) == SCF_OK) {
    info[3] = -info[3];      // negative: not the register-staged kernel, KC of d is irrelevant
    return SCF_OK;
  }
  info[0] = pl.WM;
  info[1] = pl.WN;
  info[2] = (int32_t)pl.nblk;
  info[3] = pl.k.T * (pl.k.KC / 2) * pl.WM * pl.WN / (pl.k.ksplit ? 4 : 1);
  return SCF_OK;
}
