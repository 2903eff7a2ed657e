// fp32-accurate convolution on the fp16 matrix cores of gfx950 ("split-fp16, 3 x MFMA").
//
// Every fp32 operand x is split exactly into two fp16 numbers
//        x = hi + lo' * 2^-11,      hi = fp16(x),  lo' = fp16((x - hi) * 2^11)
// (lo' is pre-scaled so it stays in fp16's normal range) and a product is evaluated as
//        a*b ~= a_hi*b_hi + (a_hi*b_lo' + a_lo'*b_hi) * 2^-11            (lo*lo dropped: 2^-22)
// with v_mfma_f32_32x32x16_f16: fp16 products are exact in the fp32 accumulator, so the result
// carries ~22 mantissa bits -- the same class as fp32 re-association noise -- at 3 MFMAs of 32
// cycles per 32x32x16 block instead of 8 fp32 MFMAs of 64 cycles (5.3x the matrix rate).
// End-to-end effect on the SCFlow path (CPU emulation, 8 iterations): flow EPE 8e-5 px vs the
// fp32 reference, |dR| 5e-7 (tolerance of the north star: 1e-3 px).  Two accumulators per
// output fragment keep the scaled cross terms separate; they are merged in the epilogue.
//
// GEMM view and tiling are those of conv_mfma.hip (D[cout, pixel], NCHW in / NCHW out, block =
// 4 waves side by side along the pixel axis, WM x WN fragments per wave) with k = 16 channels
// of one tap per MFMA:
//   A[cout][k8] : lane (cout = l&31, k8 = l>>5) holds 8 consecutive channels -> weights are
//                 pre-packed in global memory as 16-byte cells [tap][k8][cout][8] (hi plane, lo
//                 plane) and copied straight into LDS; ds_read_b128, lane <-> consecutive cell.
//   B[k8][pix]  : lane (pix = l&31, k8 = l>>5) holds 8 consecutive channels of one pixel ->
//                 the input window is staged as 16-byte cells [k8 group][py][px][8]; the
//                 fp32 -> (hi, lo') split and the channel transposition happen once per
//                 staged element, in registers, between the global load and the LDS write.
// Staging is software pipelined exactly as in conv_mfma.hip (gather table in registers, loads
// of chunk c+1 issued before the MFMA phase of chunk c).
#include <hip/hip_fp16.h>
#include <stdlib.h>

#include "scf_common.h"
#include <atomic>
#include "conv_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int F16_T_MAX = 9;         // taps (3x3, 1x5, 5x1); larger kernels stay on the fp32 path
constexpr int F16_WU_BATCH = 5;     // weight cells in flight per thread while staging
constexpr float F16_LO_SCALE = 2048.f;
__host__ __device__ constexpr int f16_cu_max(int nk) { return nk == 1 ? 3 : 4; }  // prefetched cells

__device__ __forceinline__ void split_f16(const float (&x)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const _Float16 h = (_Float16)x[j];
    hi[j] = h;
    lo[j] = (_Float16)((x[j] - (float)h) * F16_LO_SCALE);
  }
}

// NK = MFMA k-steps (16 channels each) per tap and staged chunk: KC = 16*NK channels.
template <int WM, int WN, int NK>
__global__ __launch_bounds__(256, 2) void conv_f16x3_kernel(ConvK p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int BM = WM * 32;
  constexpr int NFRAG = WN * 4;
  constexpr int KC = 16 * NK;
  constexpr int G = 2 * NK;          // 8-channel groups per chunk
  constexpr int CU_MAX = f16_cu_max(NK);

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

  // LDS: weights [T][NK][2 k8][2 planes][BM] cells | patch hi [G][PH][PW] cells | patch lo
  f16x8* wl = reinterpret_cast<f16x8*>(lds_raw);
  const int WE = T * NK * 4 * BM;    // weight cells per chunk for this block
  f16x8* ph_ = wl + WE;
  const int PCELLS = G * PHW;
  f16x8* plo = ph_ + PCELLS;

  const int fr = l32 >> p.fc_log2, fc = l32 & (FC - 1);
  int boff[WN];   // cell index of this lane's pixel for tap (0,0), k8 group = half
#pragma unroll
  for (int j = 0; j < WN; ++j) boff[j] = (((wave * WN + j) * FR + fr) * s) * PW + fc * s + half * PHW;

  f32x16 acc0[WM][WN], acc1[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }

  const int HWin = p.H * p.W;
  const float* in0n = p.in0 + (long long)n * p.in0_ns;
  const float* in1n = p.in1 ? p.in1 + (long long)n * p.in1_ns : nullptr;
  // packed fp16 weights: row = (chunk16*T + tap)*2 + k8, each row = [2 planes][Mld] cells
  const f16x8* wpk = reinterpret_cast<const f16x8*>(p.wp16);

  // gather table (chunk invariant): cell e = tid + 256u -> float offset of its first channel
  int toff[CU_MAX];
#pragma unroll
  for (int u = 0; u < CU_MAX; ++u) {
    const int e = tid + u * 256;
    int o = -1;
    if (e < PCELLS) {
      const int g = e / PHW, r = e - g * PHW;
      const int py = r / PW, px = r - py * PW;
      const int iy = iy0 + py, ix = ix0 + px;
      if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) o = g * 8 * HWin + iy * p.W + ix;
    }
    toff[u] = o;
  }

  float preg[CU_MAX][8];
  unsigned pmask = 0;   // validity bit per prefetched element

  for (int chunk = -1; chunk < p.nchunk; ++chunk) {
    if (chunk >= 0) {
      __syncthreads();
      // split + transpose the prefetched window, park it in LDS
#pragma unroll
      for (int u = 0; u < CU_MAX; ++u) {
        const int e = tid + u * 256;
        if (e < PCELLS) {
          f16x8 hi, lo;
          float xv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) xv[j] = ((pmask >> (u * 8 + j)) & 1u) ? preg[u][j] : 0.f;
          split_f16(xv, hi, lo);
          ph_[e] = hi;
          plo[e] = lo;
        }
      }
      // weights (L2 resident, shared by every block): global -> registers -> LDS here; the
      // co-resident block's MFMA phase covers this latency, and not prefetching them keeps the
      // kernel at 2 waves per SIMD.
      // LDS cell e = ((((t*NK + ks)*2 + k8)*2 + plane)*BM + c  <-  row ((chunk*NK+ks)*T + t)*2 + k8
      for (int e0 = tid; e0 < WE; e0 += 256 * F16_WU_BATCH) {
        f32x4 wv[F16_WU_BATCH];
#pragma unroll
        for (int u = 0; u < F16_WU_BATCH; ++u) {
          const int e = e0 + u * 256;
          wv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (e < WE) {
            const int c = e % BM, rp = e / BM;
            const int plane = rp & 1, k8 = (rp >> 1) & 1, tk = rp >> 2;   // tk = t*NK + ks
            const int t = tk / NK, ks = tk - t * NK;
            const long long row = (((long long)chunk * NK + ks) * T + t) * 2 + k8;
            if (m0 + c < p.Mld)
              wv[u] = *reinterpret_cast<const f32x4*>(wpk + (row * 2 + plane) * p.Mld + m0 + c);
          }
        }
#pragma unroll
        for (int u = 0; u < F16_WU_BATCH; ++u) {
          const int e = e0 + u * 256;
          if (e < WE) *reinterpret_cast<f32x4*>(wl + e) = wv[u];
        }
      }
      // The weight loads above sit in lane-predicated blocks; on the (never taken) all-lanes-off
      // path hipcc's waitcnt pass sees them as still pending and would drain vmcnt(0) in front
      // of the first LDS read of the MFMA phase -- together with the prefetch issued below.  An
      // explicit vmcnt(0) here (nothing else is in flight yet) clears its scoreboard.
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) expcnt(7) lgkmcnt(15)
      __syncthreads();
    }
    if (chunk + 1 < p.nchunk) {
      const int c0 = (chunk + 1) * KC;
      const float* base;
      int nvalid;
      if (c0 < p.C0) { base = in0n + (long long)c0 * HWin; nvalid = p.C0 - c0; }
      else { base = in1n + (long long)(c0 - p.C0) * HWin; nvalid = p.Cin - c0; }
      const unsigned limit = (unsigned)(nvalid < KC ? nvalid : KC) * (unsigned)HWin;
      pmask = 0;
#pragma unroll
      for (int u = 0; u < CU_MAX; ++u) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          // branch-free: padding / tail channels read element 0 of the chunk (always valid
          // memory) and are zeroed when the value is consumed
          const unsigned o = (unsigned)toff[u] + (unsigned)(j * HWin);
          const bool ok = toff[u] >= 0 && o < limit;
          preg[u][j] = base[ok ? o : 0u];
          pmask |= (ok ? 1u : 0u) << (u * 8 + j);
        }
      }
    }
    if (chunk >= 0) {
      // fully unrolled with a uniform guard instead of a counted loop (keeps hipcc from
      // flushing vmcnt in a loop preheader in front of the MFMA phase)
#pragma unroll
      for (int t = 0; t < F16_T_MAX; ++t) {
        if (t < T) {
          const int ky = t / p.KW, kx = t - ky * p.KW;
          const int po = ky * PW + kx;
#pragma unroll
          for (int ks = 0; ks < NK; ++ks) {
            const f16x8* wt = wl + (((t * NK + ks) * 2 + half) * 2) * BM + l32;
            f16x8 ah[WM], al[WM], bh[WN], bl[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) { ah[i] = wt[i * 32]; al[i] = wt[BM + i * 32]; }
#pragma unroll
            for (int j = 0; j < WN; ++j) {
              bh[j] = ph_[boff[j] + po + ks * 2 * PHW];
              bl[j] = plo[boff[j] + po + ks * 2 * PHW];
            }
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
              for (int j = 0; j < WN; ++j) {
                acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc0[i][j], 0, 0, 0);
                acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc1[i][j], 0, 0, 0);
                acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc1[i][j], 0, 0, 0);
              }
          }
        }
      }
    }
  }

  // ---- epilogue (same fused forms as conv_mfma.hip) ----
  const ConvEpi epi = scf_conv_epi(p, n);
  const bool use_div = p.out_div != 1.0f;
  constexpr float inv_scale = 1.0f / F16_LO_SCALE;
  int pix[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int oy = ty0 + (wave * WN + j) * FR + fr, ox = tx0 + fc;
    pix[j] = (oy < p.Ho && ox < p.Wo) ? oy * p.Wo + ox : -1;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[i][j][r] = acc0[i][j][r] + acc1[i][j][r] * inv_scale;
  }
  scf_conv_epilogue_tile<WM, WN>(p, epi, acc0, m0, half, pix, use_div);
}

template <int WM, int WN, int NK>
static int launch_f16(const ConvK& k, int nblk, size_t lds_bytes, hipStream_t st) {
  if (lds_bytes > 64 * 1024) {       // opt in to > 64 KiB of dynamic LDS: once per instantiation AND device
    static std::atomic<unsigned long long> raised{0};      // bit d: done on device d
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return SCF_ELAUNCH;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(raised.load(std::memory_order_relaxed) & bit)) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16x3_kernel<WM, WN, NK>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess)
        return SCF_ELAUNCH;
      raised.fetch_or(bit, std::memory_order_relaxed);
    }
  }
  scf_launch((conv_f16x3_kernel<WM, WN, NK>), dim3(nblk), dim3(256), lds_bytes, st, k);
  return scf_launch_status();
}

// tile selection + launch for the split-fp16 path; returns SCF_EUNSUPPORTED when the shape does
// not fit (caller falls back to the fp32 MFMA kernel).
int scf_conv_f16x3_dispatch(ConvK k, int N, bool dry_run, int* info, hipStream_t st) {
  if (!k.wp16 || k.Cin < 16 || k.T > F16_T_MAX) return SCF_EUNSUPPORTED;
  if (k.in1 && (k.C0 % 32) != 0) return SCF_EUNSUPPORTED;
  const int FC = 1 << k.fc_log2, FR = 32 / FC;
  const int frags_m = (k.Cout + 31) / 32;
  auto tiles = [&](int WN) {
    const int TR = WN * 4 * FR;
    return (long long)N * ((k.Ho + TR - 1) / TR) * ((k.Wo + FC - 1) / FC);
  };
  auto fits = [&](int WM, int WN, int NK, size_t* lds_out) {
    const int TR = WN * 4 * FR;
    const int PH = (TR - 1) * k.stride + k.KH, PW = (FC - 1) * k.stride + k.KW;
    const long long pcells = (long long)(2 * NK) * PH * PW;
    const long long wcells = (long long)k.T * NK * 4 * WM * 32;
    const size_t lds = (size_t)(wcells + 2 * pcells) * 16;
    if (lds_out) *lds_out = lds;
    // two blocks per CU must fit the 160 KiB LDS
    return (pcells + 255) / 256 <= f16_cu_max(NK) && lds <= 80 * 1024;
  };
  // candidates (WM, WN), largest tile first; take the first that fits and fills the chip
  // (>= 2 blocks per CU), otherwise the fitting candidate with the most blocks.  Two k-steps
  // per staged chunk (NK = 2: half the barriers, twice the MFMA work hiding each prefetch)
  // whenever the chunk fits.
  const int cand[3][2] = {{2, 1}, {1, 2}, {1, 1}};
  int WM = 0, WN = 0, NK = 1;
  long long best = -1;
  for (int c = 0; c < 3; ++c) {
    const int wm = cand[c][0], wn = cand[c][1];
    if (wm > frags_m) continue;
    const int nk = (k.Cin >= 32 && wn == 1 && fits(wm, wn, 2, nullptr)) ? 2 : 1;
    if (!fits(wm, wn, nk, nullptr)) continue;
    const long long nb = tiles(wn) * ((frags_m + wm - 1) / wm);
    if (nb >= 512) { WM = wm; WN = wn; NK = nk; best = nb; break; }
    if (nb > best) { WM = wm; WN = wn; NK = nk; best = nb; }
  }
  if (WM == 0) return SCF_EUNSUPPORTED;
  size_t lds_bytes = 0;
  fits(WM, WN, NK, &lds_bytes);
  k.KC = 16 * NK;
  k.nchunk = (k.Cin + k.KC - 1) / k.KC;
  k.mblocks = (frags_m + WM - 1) / WM;
  const int TR = WN * 4 * FR;
  k.PH = (TR - 1) * k.stride + k.KH;
  k.PW = (FC - 1) * k.stride + k.KW;
  k.tiles_y = (k.Ho + TR - 1) / TR;
  k.tiles_x = (k.Wo + FC - 1) / FC;
  const long long nblk = (long long)N * k.tiles_y * k.tiles_x * k.mblocks;
  if (nblk > 0x7fffffffLL) return SCF_EUNSUPPORTED;
  if (info) { info[0] = WM; info[1] = WN; info[2] = (int)nblk; info[3] = k.T * NK * 3 * WM * WN; }
  if (dry_run) return SCF_OK;
#define SCF_F16_CASE(M, Nn, K) \
  if (WM == M && WN == Nn && NK == K) return launch_f16<M, Nn, K>(k, (int)nblk, lds_bytes, st);
  SCF_F16_CASE(2, 1, 1) SCF_F16_CASE(2, 1, 2) SCF_F16_CASE(1, 2, 1)
  SCF_F16_CASE(1, 1, 1) SCF_F16_CASE(1, 1, 2)
#undef SCF_F16_CASE
  return SCF_EUNSUPPORTED;
}
