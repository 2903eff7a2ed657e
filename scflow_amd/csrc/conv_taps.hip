// Thin-INPUT convolution (Cin <= 4): the 7x7 stems of the RAFT encoders (3 -> 64, stride 2,
// raft_encoder.py:210-217), the 7x7 first layers of the flow / delta-flow encoders (2 -> 128,
// raft_decoder.py:141-148, scflow_decoder.py:102-103) and the mask encoder's first 3x3 (1 -> 64,
// scflow_decoder.py:104-105).
//
// The channel-chunked kernels (conv_mfma.hip / conv_dma.hip) contract over channels chunk by chunk;
// with 1-3 channels that is a single short chunk -- no pipeline at all, every block a chain of
// dependent round trips (2 -> 128 7x7: 19 us per launch for 6 us of matrix work) -- and the
// 2-channel k-step of the fp32 MFMA leaves a third of it multiplying zeros at Cin = 3.  Here the
// contraction runs over TAPS x channels as one dense K = Cin * KH * KW dimension:
//     D[cout, pixel] = sum_k W[k][cout] * X[k][pixel],   k = c * T + t   (k-steps of 2, K padded to a multiple of 8)
//   block   = BM = 32 WM output channels x 128 pixels (4 waves x one 32-pixel fragment = FR rows x
//             FC columns of the output, like the other kernels), 2-3 blocks per CU;
//   staging = ONCE per block, all of it requested up front by LDS-DMA (buffer descriptors: out-of-image
//             taps are zero-filled by the range check): the whole [Kp][BM] weight slab, the input patch of
//             all channels, and a table koff[k] = offset of tap k inside the patch; then a
//             single MFMA loop over Kp / 2 steps with every operand in LDS (lane half h takes row
//             k = 2 ks + h: exactly the A[l&31][l>>5] / B[l>>5][l&31] operand layout);
//   epilogue = the shared fused epilogue (conv_kernels.h).
// Blocks overlap each other's staging; nothing is chunk-pipelined because nothing needs to be.
#include "scf_common.h"
#include "conv_kernels.h"
#include "scf_dma.h"     // LDS-DMA through raw buffer descriptors: out-of-range lanes write zeros

#include "conv_taps_body.h"

template <int WM>
__global__ __launch_bounds__(256, 2) void conv_taps_kernel(ConvK p, const float* __restrict__ wt, int Kp, int PWp) {
  conv_taps_body<WM>(p, wt, Kp, PWp, (int)blockIdx.x, (int)gridDim.x);
}

// r6: two independent thin-input layers (the delta-flow and mask encoders' first layers) in one launch; see conv_dma_pair_kernel
template <int WMA, int WMB>
__global__ __launch_bounds__(256, 2) void conv_taps_pair_kernel(ConvK pa, const float* __restrict__ wta, int Kpa, int PWpa, ConvK pb,
                                                                const float* __restrict__ wtb, int Kpb, int PWpb, int nba) {
  if ((int)blockIdx.x < nba) conv_taps_body<WMA>(pa, wta, Kpa, PWpa, (int)blockIdx.x, nba);
  else conv_taps_body<WMB>(pb, wtb, Kpb, PWpb, (int)blockIdx.x - nba, (int)gridDim.x - nba);
}

#define CT_MAXU 12      // patch cells per lane the gather table keeps in registers (256 * 12 floats per patch)

// r5: launches with more tiles than the chip holds blocks (the 7x7 / stride-2 stems at batch >= 8: 16 rounds of blocks
// that were 2 us of staging, 4-8 us of MFMAs and their stores, one after the other) run PERSISTENT over the tiles of one
// channel block -- the weight slab and the tap table are staged once per block, the patch of the block's NEXT tile
// streams into the second patch buffer while the current tile is on the matrix cores.  Grids that fit the chip in one
// go keep conv_taps_kernel above (one tile per block: nothing to amortise, and its shorter prologue wins).
template <int WM>
__global__ __launch_bounds__(256, 2) void conv_taps_persist_kernel(ConvK p, const float* __restrict__ wt, int Kp, int PWp, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float ct_lds[];
  constexpr int BM = WM * 32;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, half = lane >> 5;

  const int lb = scf_xcd_remap(blockIdx.x, gridDim.x);
  const int mblk = __builtin_amdgcn_readfirstlane(lb % p.mblocks);
  const int slot = __builtin_amdgcn_readfirstlane(lb / p.mblocks);
  const int nslots = __builtin_amdgcn_readfirstlane((int)gridDim.x / p.mblocks);      // the host launches a multiple of mblocks
  const int m0 = mblk * BM;
  const int FC = 1 << p.fc_log2, FR = 32 >> p.fc_log2, TR = 4 * FR;
  const int st = p.stride;
  const int PH = p.PH, PHW = PH * PWp, T = p.T;
  // LDS: weights [Kp][BM] | patch [2][Cin][PH][PWp] | tap table [Kp]; the first two are filled by LDS-DMA in
  // whole wave-instructions (64 cells), so each area is padded to a multiple of 64 cells: lanes past an
  // area's end carry an out-of-range offset and write zeros into the padding
  const int q4 = BM / 4, n4 = Kp * q4;                     // weight cells (float4)
  const int w_cells = (n4 + 63) & ~63;
  const int PE = p.Cin * PHW;                              // patch cells (floats)
  const int x_cells = (PE + 63) & ~63;
  float* Ws = ct_lds;
  float* Xs = Ws + w_cells * 4;
  int* koff = (int*)(Xs + 2 * x_cells);
  const int HW = p.H * p.W;

  // ---- gather table of the patch, tile-invariant part: cell e = lane's u-th -> (channel, row, column) ----
  int rel[CT_MAXU], pyx[CT_MAXU];
  {
    const float rPHW = 1.0f / (float)PHW, rPW = 1.0f / (float)PWp;
#pragma unroll
    for (int u = 0; u < CT_MAXU; ++u) {
      rel[u] = -1; pyx[u] = 0;
      if (wave * 64 + u * 256 >= x_cells) continue;        // wave-uniform: no cell of this slot in the patch
      const int e = wave * 64 + u * 256 + lane;
      int c = (int)((float)e * rPHW);
      int r = e - c * PHW;
      if (r < 0) { --c; r += PHW; } else if (r >= PHW) { ++c; r -= PHW; }
      int py = (int)((float)r * rPW);
      int px = r - py * PWp;
      if (px < 0) { --py; px += PWp; } else if (px >= PWp) { ++py; px -= PWp; }
      rel[u] = e < PE ? c * HW + py * p.W + px : -1;
      pyx[u] = (py << 16) | px;
    }
  }
  auto tile_coords = [&](int tile, int& n, int& ty0, int& tx0) {
    const int txi = __builtin_amdgcn_readfirstlane(tile % p.tiles_x);
    const int t2 = tile / p.tiles_x;
    const int tyi = __builtin_amdgcn_readfirstlane(t2 % p.tiles_y);
    n = __builtin_amdgcn_readfirstlane(t2 / p.tiles_y);
    ty0 = tyi * TR; tx0 = txi * FC;
  };
  const unsigned xl = scf_lds_addr(Xs);
  auto stage_x = [&](int tile, int b) {
    int n, ty0, tx0;
    tile_coords(tile, n, ty0, tx0);
    const int iy0 = ty0 * st - p.pad_h, ix0 = tx0 * st - p.pad_w;
    const scf_rsrc4 xrs = scf_make_rsrc(p.in0 + (long long)n * p.in0_ns, (unsigned)((long long)p.Cin * HW * 4));
    const int tbase = iy0 * p.W + ix0;
    const unsigned dst = xl + (unsigned)(b * x_cells + wave * 64) * 4u;
#pragma unroll
    for (int u = 0; u < CT_MAXU; ++u) {
      if (wave * 64 + u * 256 < x_cells) {                 // wave-uniform
        const int iy = iy0 + (pyx[u] >> 16), ix = ix0 + (pyx[u] & 0xffff);
        const bool ok = rel[u] >= 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        scf_bdma_b32(xrs, ok ? (unsigned)((rel[u] + tbase) * 4) : SCF_BUF_OOB, dst + (unsigned)u * 1024u);
      }
    }
  };

  // ---- once per block: the weight slab, the first tile's patch, the tap table; then ONE wait ----
  {
    const scf_rsrc4 wrs = scf_make_rsrc(wt + m0, (unsigned)(((long long)(Kp - 1) * p.Mld + BM) * 4));
    const unsigned wl = scf_lds_addr(Ws);
    for (int c0 = wave * 64; c0 < w_cells; c0 += 256) {   // wave-uniform trip count
      const int e = c0 + lane;
      const int row = e / q4, c4 = e - row * q4;
      const unsigned voff = e < n4 ? (unsigned)((row * p.Mld + 4 * c4) * 4) : SCF_BUF_OOB;
      scf_bdma_b128(wrs, voff, wl + (unsigned)c0 * 16u);
    }
  }
  if (slot < ntiles) stage_x(slot, 0);
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

  const int fr = l32 >> p.fc_log2, fc = l32 & (FC - 1);
  const float* wb = Ws + half * BM + l32;                               // row k = 2 ks + half
  const int* kb = koff + half;
  const int ntrip = Kp >> 3;
  int buf = 0;
#pragma nounroll
  for (int tile = slot; tile < ntiles; tile += nslots) {
    if (tile + nslots < ntiles) stage_x(tile + nslots, buf ^ 1);       // the next tile's patch: lands under this tile's MFMAs

    ct_f32x16 acc[WM][1];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;

    const float* xb = Xs + buf * x_cells + ((wave * FR + fr) * st) * PWp + fc * st;      // this lane's pixel, tap (0, 0)
    // Kp is a multiple of 8: trips of four k-steps, software-pipelined by hand -- while trip t is on the
    // matrix pipe, the operands of trip t+1 and the table entries of trip t+2 are already being read (the
    // table -> operand-address dependency is the only chain; hipcc left to itself issues each operand
    // read right in front of its MFMA and waits for it).  The last trips re-read the final one instead of
    // branching around the prefetch.
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
#pragma nounroll
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

    // the next patch has landed before this tile's stores go out (loads and stores share the in-order vmcnt);
    // the barrier below is a bare s_barrier: it orders LDS use between the waves and does not wait for the stores
    __builtin_amdgcn_s_waitcnt(0x0F70);
    int n, ty0, tx0;
    tile_coords(tile, n, ty0, tx0);
    const ConvEpi epi = scf_conv_epi(p, n);
    int pix[1];
    {
      const int oy = ty0 + wave * FR + fr, ox = tx0 + fc;
      pix[0] = (oy < p.Ho && ox < p.Wo) ? oy * p.Wo + ox : -1;
    }
    // (m0 / half pass through an empty asm per tile: left loop-invariant, hipcc hoists every epilogue kind's address
    // arithmetic and constant loads out of the tile loop and spills 245 registers to keep them)
    int m0t = m0, halft = half;
    asm volatile("" : "+s"(m0t), "+v"(halft));
    scf_conv_epilogue_tile<WM, 1>(p, epi, acc, m0t, halft, pix, p.out_div != 1.0f);
    if (tile + nslots < ntiles) {
      __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): this wave's LDS reads of the tile are done
      __builtin_amdgcn_s_barrier();            // every wave is done with `buf`, every wave's copies into the other one have landed
    }
    buf ^= 1;
  }
}

// Tile selection + launch; SCF_EUNSUPPORTED -> the caller goes on to the channel-chunked kernels.
int scf_conv_taps_dispatch(ConvK k, const float* wt, int N, bool dry_run, int* info, hipStream_t st, ScfLaunchCap* cap) {
  if (!wt || k.Cin > 4 || k.in1 || k.w_ns != 0 || k.out_tile || (k.Mld & 3) || ((uintptr_t)wt & 15)) return SCF_EUNSUPPORTED;
  const int FC = 1 << k.fc_log2, FR = 32 / FC, TR = 4 * FR;
  const int frags_m = (k.Cout + 31) / 32;
  const int Kp = (k.Cin * k.T + 7) & ~7;         // rows of the packing: a whole number of four-k-step trips
  const int PH = (TR - 1) * k.stride + k.KH, PWin = (FC - 1) * k.stride + k.KW;
  const int PWp = PWin | 1;                        // odd row pitch: stride-2 column reads spread over the banks
  const long long tiles = (long long)N * ((k.Ho + TR - 1) / TR) * ((k.Wo + FC - 1) / FC);
  // 64 output channels per block when that still gives every CU two blocks, else 32
  int WM = (frags_m % 2 == 0 && tiles * (frags_m / 2) >= 2LL * scf_cu_count()) ? 2 : 1;
  const size_t xc = ((size_t)k.Cin * PH * PWp + 63) & ~(size_t)63;
  auto lds_for = [&](int wm, int patch_buffers) {          // weight cells are float4, patch cells floats
    const size_t wc = ((size_t)Kp * wm * 8 + 63) & ~(size_t)63;
    return (wc * 4 + patch_buffers * xc + Kp) * sizeof(float);
  };
  size_t ldsb1 = 0;
  for (;; WM = 1) {
    ldsb1 = lds_for(WM, 1);
    if (ldsb1 <= 64 * 1024 || WM == 1) break;
  }
  if (ldsb1 > 64 * 1024) return SCF_EUNSUPPORTED;
  k.PH = PH; k.PW = PWp; k.PWin = PWin;
  k.tiles_y = (k.Ho + TR - 1) / TR;
  k.tiles_x = (k.Wo + FC - 1) / FC;
  k.mblocks = frags_m / WM;
  const long long nblk = tiles * k.mblocks;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return SCF_EUNSUPPORTED;
  if (info) { info[0] = WM; info[1] = 1; info[2] = (int)nblk; info[3] = Kp / 2 * WM; }
  if (dry_run && !cap) return SCF_OK;
  // more tiles than the chip holds blocks (by LDS with two patch buffers, at most three per CU): the persistent kernel,
  // a block walks tiles slot, slot + nslots, ... of its channel block
  const size_t ldsb = lds_for(WM, 2);
  long long grid = nblk;
  if (xc <= 256 * CT_MAXU && ldsb <= 64 * 1024 && tiles <= 0x7fffffffLL) {
    long long per_cu = (long long)(160 * 1024 / ldsb);
    per_cu = per_cu > 3 ? 3 : per_cu;
    const long long cap = per_cu * scf_cu_count() / k.mblocks * k.mblocks;
    if (cap >= k.mblocks && nblk > cap) grid = cap;
  }
  if (cap) {                               // r6: hand the launch back instead of issuing it (scf_conv2d_pair)
    cap->k = k; cap->nblk = (int)nblk; cap->ldsb = ldsb1; cap->wt = wt; cap->Kp = Kp; cap->PWp = PWp;
    cap->variant = grid == nblk ? WM : -1;      // the one-tile-per-block kernel only
    return SCF_OK;
  }
  if (grid == nblk) {                      // one tile per block
    if (WM == 2) scf_launch((conv_taps_kernel<2>), dim3((unsigned)nblk), dim3(256), ldsb1, st, k, wt, Kp, PWp);
    else scf_launch((conv_taps_kernel<1>), dim3((unsigned)nblk), dim3(256), ldsb1, st, k, wt, Kp, PWp);
    return scf_launch_status();
  }
  if (WM == 2) scf_launch((conv_taps_persist_kernel<2>), dim3((unsigned)grid), dim3(256), ldsb, st, k, wt, Kp, PWp, (int)tiles);
  else scf_launch((conv_taps_persist_kernel<1>), dim3((unsigned)grid), dim3(256), ldsb, st, k, wt, Kp, PWp, (int)tiles);
  return scf_launch_status();
}

int scf_conv_taps_pair_launch(const ScfLaunchCap& a, const ScfLaunchCap& b, hipStream_t st) {
  // variant = output channels per block / 32 (1 or 2) of the one-tile-per-block kernel; the two layers may differ (the delta-flow
  // encoder's 2 -> 128 takes 64 channels per block on full grids, the mask encoder's 1 -> 64 takes 32)
  if (a.variant < 1 || a.variant > 2 || b.variant < 1 || b.variant > 2 || a.nblk <= 0 || b.nblk <= 0) return SCF_EUNSUPPORTED;
  const size_t lds = a.ldsb > b.ldsb ? a.ldsb : b.ldsb;
  if (lds > 64 * 1024) return SCF_EUNSUPPORTED;
  const dim3 grid((unsigned)(a.nblk + b.nblk)), blk(256);
#define SCF_GO(A_, B_) scf_launch((conv_taps_pair_kernel<A_, B_>), grid, blk, lds, st, a.k, a.wt, a.Kp, a.PWp, b.k, b.wt, b.Kp, b.PWp, a.nblk)
  if (a.variant == 1 && b.variant == 1) SCF_GO(1, 1);
  else if (a.variant == 1) SCF_GO(1, 2);
  else if (b.variant == 1) SCF_GO(2, 1);
  else SCF_GO(2, 2);
#undef SCF_GO
  return scf_launch_status();
}
