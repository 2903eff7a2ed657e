// conv_taps_body: the thin-input kernel's body (conv_taps.hip describes the kernel) as a function of its arguments, block index
// and grid size.  Lives in a header since r6: conv_taps.hip wraps it in conv_taps_kernel / conv_taps_pair_kernel, conv_dma.hip in
// conv_dma_taps_pair_kernel (a K-split layer and a thin-input layer -- corr_net.0 | flow_net.0 of the motion encoder -- in one launch).
#pragma once
#include "scf_common.h"
#include "conv_kernels.h"
#include "scf_dma.h"     // LDS-DMA through raw buffer descriptors: out-of-range lanes write zeros

typedef float ct_f32x16 __attribute__((ext_vector_type(16)));

// (the kernel's body as a function of its arguments, block index and grid size: conv_taps_pair_kernel below runs two layers' grids
// in one launch, r6)
template <int WM>
__device__ __forceinline__ void conv_taps_body(const ConvK& p, const float* __restrict__ wt, const int Kp, const int PWp, const int bid,
                                               const int nblk) {
  extern __shared__ __attribute__((aligned(16))) float ct_lds[];
  constexpr int BM = WM * 32;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, half = lane >> 5;

  const int lb = scf_xcd_remap(bid, nblk);
  const int mblk = __builtin_amdgcn_readfirstlane(lb % p.mblocks);
  const int tile = __builtin_amdgcn_readfirstlane(lb / p.mblocks);
  const int m0 = mblk * BM;
  const int FC = 1 << p.fc_log2, FR = 32 >> p.fc_log2, TR = 4 * FR;
  const int txi = __builtin_amdgcn_readfirstlane(tile % p.tiles_x);
  const int t2 = tile / p.tiles_x;
  const int tyi = __builtin_amdgcn_readfirstlane(t2 % p.tiles_y);
  const int n = __builtin_amdgcn_readfirstlane(t2 / p.tiles_y);
  const int ty0 = tyi * TR, tx0 = txi * FC;
  const int st = p.stride;
  const int iy0 = ty0 * st - p.pad_h, ix0 = tx0 * st - p.pad_w;
  const int PH = p.PH, PHW = PH * PWp, T = p.T;
  // LDS: weights [Kp][BM] | patch [Cin][PH][PWp] | tap table [Kp]; the first two are filled by LDS-DMA in
  // whole wave-instructions (64 cells), so each area is padded to a multiple of 64 cells: lanes past an
  // area's end carry an out-of-range offset and write zeros into the padding
  const int q4 = BM / 4, n4 = Kp * q4;                     // weight cells (float4)
  const int w_cells = (n4 + 63) & ~63;
  const int PE = p.Cin * PHW;                              // patch cells (floats)
  const int x_cells = (PE + 63) & ~63;
  float* Ws = ct_lds;
  float* Xs = Ws + w_cells * 4;
  int* koff = (int*)(Xs + x_cells);

  // ---- everything is requested up front (asynchronous memory -> LDS copies), then ONE wait ----
  {
    const scf_rsrc4 wrs = scf_make_rsrc(wt + m0, (unsigned)(((long long)(Kp - 1) * p.Mld + BM) * 4));
    const unsigned wl = scf_lds_addr(Ws);
    for (int c0 = wave * 64; c0 < w_cells; c0 += 256) {   // wave-uniform trip count
      const int e = c0 + lane;
      const int row = e / q4, c4 = e - row * q4;
      const unsigned voff = e < n4 ? (unsigned)((row * p.Mld + 4 * c4) * 4) : SCF_BUF_OOB;
      scf_bdma_b128(wrs, voff, wl + (unsigned)c0 * 16u);
    }
    const int HW = p.H * p.W;
    const scf_rsrc4 xrs = scf_make_rsrc(p.in0 + (long long)n * p.in0_ns, (unsigned)((long long)p.Cin * HW * 4));
    const unsigned xl = scf_lds_addr(Xs);
    const float rPHW = 1.0f / (float)PHW, rPW = 1.0f / (float)PWp;
    for (int c0 = wave * 64; c0 < x_cells; c0 += 256) {
      const int e = c0 + lane;
      int c = (int)((float)e * rPHW);
      int r = e - c * PHW;
      if (r < 0) { --c; r += PHW; } else if (r >= PHW) { ++c; r -= PHW; }
      int py = (int)((float)r * rPW);
      int px = r - py * PWp;
      if (px < 0) { --py; px += PWp; } else if (px >= PWp) { ++py; px -= PWp; }
      const int iy = iy0 + py, ix = ix0 + px;
      const bool ok = e < PE && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      scf_bdma_b32(xrs, ok ? (unsigned)((c * HW + iy * p.W + ix) * 4) : SCF_BUF_OOB, xl + (unsigned)c0 * 4u);
    }
  }
  // ---- tap table: k = c * T + t -> patch offset; padding rows (k >= Cin * T: zero weights) read cell 0 ----
  for (int k = tid; k < Kp; k += 256) {
    int o = 0;
    if (k < p.Cin * T) {
      const int c = k / T, t = k - c * T;
      const int ky = t / p.KW, kx = t - ky * p.KW;
      o = c * PHW + ky * PWp + kx;
    }
    koff[k] = o;
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): this wave's copies have landed
  __syncthreads();

  ct_f32x16 acc[WM][1];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;

  const int fr = l32 >> p.fc_log2, fc = l32 & (FC - 1);
  const float* xb = Xs + ((wave * FR + fr) * st) * PWp + fc * st;      // this lane's pixel, tap (0, 0)
  const float* wb = Ws + half * BM + l32;                               // row k = 2 ks + half
  const int* kb = koff + half;
  // Kp is a multiple of 8: trips of four k-steps, software-pipelined by hand -- while trip t is on the
  // matrix pipe, the operands of trip t+1 and the table entries of trip t+2 are already being read (the
  // table -> operand-address dependency is the only chain; hipcc left to itself issues each operand
  // read right in front of its MFMA and waits for it).  The last trips re-read the final one instead of
  // branching around the prefetch.
  const int ntrip = Kp >> 3;
  int oN[4];
  float bC[4], aC[4][WM];
#pragma unroll
  for (int u = 0; u < 4; ++u) oN[u] = kb[2 * u];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    bC[u] = xb[oN[u]];
#pragma unroll
    for (int i = 0; i < WM; ++i) aC[u][i] = wb[2 * u * BM + 32 * i];
  }
  {
    const int t1 = ntrip > 1 ? 1 : 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) oN[u] = kb[8 * t1 + 2 * u];
  }
  for (int t = 0; t < ntrip; ++t) {
    const int tn = t + 1 < ntrip ? t + 1 : ntrip - 1, tnn = t + 2 < ntrip ? t + 2 : ntrip - 1;
    float bN[4], aN[4][WM];
    int oNN[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      bN[u] = xb[oN[u]];
#pragma unroll
      for (int i = 0; i < WM; ++i) aN[u][i] = wb[(8 * tn + 2 * u) * BM + 32 * i];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) oNN[u] = kb[8 * tnn + 2 * u];
    __builtin_amdgcn_sched_barrier(0);       // keep the reads AHEAD of this trip's MFMAs
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < WM; ++i)
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(aC[u][i], bC[u], acc[i][0], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      bC[u] = bN[u];
      oN[u] = oNN[u];
#pragma unroll
      for (int i = 0; i < WM; ++i) aC[u][i] = aN[u][i];
    }
  }

  const ConvEpi epi = scf_conv_epi(p, n);
  int pix[1];
  {
    const int oy = ty0 + wave * FR + fr, ox = tx0 + fc;
    pix[0] = (oy < p.Ho && ox < p.Wo) ? oy * p.Wo + ox : -1;
  }
  scf_conv_epilogue_tile<WM, 1>(p, epi, acc, m0, half, pix, p.out_div != 1.0f);
}

