// 1 x 5 / 5 x 1 stride-1 convolutions (the SepConvGRU gates, raft_decoder.py:198-253) in the one-dimensional
// Winograd form F(4, 5), fp32 throughout: FOUR outputs along the filter axis from a window of eight inputs with 8
// multiplies per channel pair instead of 20 (points 0, +-1, +-2, +-1/2, infinity):
//     y = A^T [ (G g) . (B^T d) ],   A^T 4 x 8, G 8 x 5, B^T 8 x 8
// -- 1.5x fewer matrix-core flops than F(2, 5) (conv_wino1d.hip: 6 per 2 outputs), 2.5x fewer than the direct kernels.
// Read conv_wino.hip / conv_wino1d.hip first: copy streams through buffer descriptors, three-deep rings, operands of
// the next chunk read / computed under the MFMAs of the current one.  What differs:
//   * a wave holds ONE 32 x 32 fragment (32 output channels x 32 tiles of FOUR outputs = 128 pixels) for all 8 transform
//     positions = 128 accumulator registers; the output transform is lane-local; after it a lane holds 16 channel rows
//     of four pixels = four accumulator fragments of the direct kernels, finished by the shared fused epilogue
//     (conv_kernels.h: bias, BN, residual, activations) or, for the two GRU gates, by w4_gru_epilogue below;
//   * a block = 2 channel fragments x 2 tile groups (64 channels x 256 pixels); 4 channels per chunk: one ds_read_b64 per
//     position feeds both k-steps, 16 MFMAs per barrier (the 128 accumulators leave no room for deeper operand buffers);
//   * the input transform is 26 fmas per window (2 windows per lane and chunk):
//         r0 = (d0 - d6) + 21/4 (d4 - d2)                       r7 = (d7 - d1) + 21/4 (d3 - d5)
//         r1 | r2 = (d2 + d6 - 17/4 d4) +- (d1 + d5 - 17/4 d3)
//         r3 | r4 = (d6 + 1/4 d2 - 5/4 d4) +- (1/2 d1 - 5/2 d3 + 2 d5)
//         r5 | r6 = (d6 + 4 d2 - 5 d4) +- (2 d1 - 5/2 d3 + 1/2 d5)
//     and the output transform  y0 = M0 + ... + M6,  y1 = (M1 - M2) + 2 (M3 - M4) + 1/2 (M5 - M6),
//         y2 = (M1 + M2) + 4 (M3 + M4) + 1/4 (M5 + M6),  y3 = (M1 - M2) + 8 (M3 - M4) + 1/8 (M5 - M6) + M7;
//   * a GRU launch's pre-activation term (the hoisted context part, one value per output) goes INTO the accumulators:
//     positions 0, 1, 2, 7 take res0 - res2, (res1 + res2) / 2, (res2 - res1) / 2, res3 - res1 -- A^T maps exactly these
//     back onto (res0, res1, res2, res3) -- so it costs no registers over the loop.  It arrives during the first
//     chunks, one output column per chunk, as a fourth copy stream (memory -> 4 KB of LDS per wave -> 16 reads): read
//     at kernel start (r4m) it was a memory round trip of every block of the launch at the same time with the matrix
//     cores idle, read in the epilogue it is the same at the end.
// Error vs fp64 in units of eps sum|w||x| (tests/test_gpu_ops.py, MI355X): 12.6 ... 16 on N(0, 1) operands (F(2, 5):
// 8.9 ... 11), 3.7 ... 10.7 on DC-offset / one-signed ones, 29 ... 37 with weights spread over three decades (F(2, 5): 22 ... 30).
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "scf_common.h"
#include "conv_kernels.h"
#include "scf_dma.h"

typedef float w4_f32x16 __attribute__((ext_vector_type(16)));
typedef float w4_f32x2 __attribute__((ext_vector_type(2)));

#define W4_KC 4             // channels per chunk
#define W4_UF 1024          // floats of one fragment's U chunk: [8 positions][2 k-halves][32 channels][2 k-steps]
// patch copy instructions per wave per chunk (256 cells per block-instruction) = template parameter NPI: 16-byte cells 2
// (vertical 16-column blocks: 3), dword cells 5

struct Wino4K {
  const float* wu;          // [nchunk][F][8][2][32][2]
  int F;                    // channel fragments in the packing
  int txl;                  // log2(tile columns of a wave's 32-tile group)
  int PH, PWp, PPL;         // patch rows, row pitch, plane stride (floats)
  int nchunk;
  int sx, sy;               // block strips per image
  int mblocks;
};

__device__ __forceinline__ int w4_div(int e, int d, float rd) {       // floor(e / d), 0 <= e < 2^20, 0 < d < 2^12
  int q = (int)((float)e * rd);
  const int r = e - q * d;
  q += (r >= d) ? 1 : 0;
  q -= (r < 0) ? 1 : 0;
  return q;
}

// The two GRU gate epilogues on a lane's 16 channel rows x 4 pixels, same arithmetic as the shared ones
// (scf_epi_general_frag: + bias, sigmoid -> z | r h;  tanh -> (1 - z) h + z q), different memory schedule: the shared
// code handles one pixel's fragment at a time = four dependent round trips for h (and z) at the end of every block of
// the launch at the same time.  Here the operands of 16 rows x 4 pixels (z | r: h) or 8 rows x 4 pixels (q: h and z) are
// requested together -- 64 registers, free by then; VEC (horizontal
// passes, rows of 16-byte aligned pixel quadruples): one 16-byte access per row instead of four 4-byte ones 16 bytes
// apart.  Needs whole channel fragments, Ch % 32 == 0 (a wave is all z rows or all r rows) and a 16-byte aligned bias.
template <int KIND, bool VEC>
__device__ __forceinline__ void w4_gru_epilogue(const ConvK& p, const ConvEpi& e, const w4_f32x16 (&o)[1][4], int cb,
                                                const int (&pix)[4]) {
  const int hc = p.Cout >> 1;
  const bool upper = KIND == SCF_EPI_GRU_ZR && cb >= hc;              // wave-uniform
  const float* hsrc = KIND == SCF_EPI_GRU_ZR ? e.gru_h - hc * e.HWo : e.gru_h;      // row co -> hsrc + co * HWo
  float* dst = upper ? e.gru_aux - hc * e.HWo : e.out;
  const scf_f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  scf_f32x4 bv[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bv[g] = p.bias ? *reinterpret_cast<const scf_f32x4*>(p.bias + cb + 8 * g) : zero4;
  // operand loads are unconditional (a pixel outside the image reads pixel 0 of its row instead: one branch and one
  // wait per element otherwise), stores are masked once per pixel column
  int pl[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) pl[j] = pix[j] >= 0 ? pix[j] : 0;
  auto load4 = [&](const float* row) __attribute__((always_inline)) {
    scf_f32x4 v;
    if (VEC) {
      v = *reinterpret_cast<const scf_f32x4*>(row + pl[0]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = row[pl[j]];
    }
    return v;
  };
  constexpr int RB = KIND == SCF_EPI_GRU_Q ? 8 : 16;      // rows per batch of operand requests: 64 registers of them
#pragma unroll
  for (int hb = 0; hb < 16 / RB; ++hb) {
    scf_f32x4 hv[RB], zv[RB];
#pragma unroll
    for (int q = 0; q < RB; ++q) {
      const int r = hb * RB + q;
      const int off = (cb + 8 * (r >> 2) + (r & 3)) * e.HWo;
      hv[q] = (KIND == SCF_EPI_GRU_Q || upper) ? load4(hsrc + off) : zero4;
      zv[q] = KIND == SCF_EPI_GRU_Q ? load4(e.gru_z + off) : zero4;
    }
    __builtin_amdgcn_sched_barrier(0);
    scf_f32x4 w[RB];
#pragma unroll
    for (int q = 0; q < RB; ++q) {
      const int r = hb * RB + q;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v = o[0][j][r] + bv[r >> 2][r & 3];
        if (KIND == SCF_EPI_GRU_ZR) {
          const float sg = scf_fast_sigmoid(v);
          w[q][j] = upper ? sg * hv[q][j] : sg;
        } else {
          w[q][j] = (1.f - zv[q][j]) * hv[q][j] + zv[q][j] * scf_fast_tanh(v);
        }
      }
    }
    if (VEC) {
      if (pix[0] >= 0) {
#pragma unroll
        for (int q = 0; q < RB; ++q) {
          const int r = hb * RB + q;
          scf_store4<(SCF_ST_SC1 & 16) != 0>(dst + (cb + 8 * (r >> 2) + (r & 3)) * e.HWo + pix[0], w[q][0], w[q][1], w[q][2], w[q][3]);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (pix[j] >= 0) {
#pragma unroll
          for (int q = 0; q < RB; ++q) {
            const int r = hb * RB + q;
            scf_store1<(SCF_ST_SC1 & 16) != 0>(dst + (cb + 8 * (r >> 2) + (r & 3)) * e.HWo + pix[j], w[q][j]);
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <bool VERT, bool PX4, int NPI>
__global__ __launch_bounds__(256, 2)
void conv_wino1d4_kernel(ConvK p, Wino4K q) {
  static_assert(!VERT || PX4, "the vertical kernel copies 16-byte cells only");
  extern __shared__ __attribute__((aligned(16))) float w4_lds[];
  constexpr int CW = 2, TW = 2;
  constexpr int USLOT = CW * W4_UF;                             // floats per ring slot
  constexpr int NUI = 2;                                        // U copy instructions (16 B per lane) per wave per chunk
  constexpr int PSLOT = NPI * (PX4 ? 1024 : 256);
  constexpr int GRP = NUI + NPI;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cw = wave >> 1, tw = wave & 1;
  const int half = lane >> 5, l32 = lane & 31;

  int lb = scf_xcd_remap(blockIdx.x, gridDim.x);
  const int mb = __builtin_amdgcn_readfirstlane(lb % q.mblocks);
  lb /= q.mblocks;
  const int xs = __builtin_amdgcn_readfirstlane(lb % q.sx);
  lb /= q.sx;
  const int ys = __builtin_amdgcn_readfirstlane(lb % q.sy);
  const int n = __builtin_amdgcn_readfirstlane(lb / q.sy);
  const int TXW = 1 << q.txl, TYW = 32 >> q.txl;
  // first output pixel of the block; a tile = 4 pixels along the filter axis
  const int y0 = ys * (VERT ? 4 * TW * TYW : TW * TYW), x0 = xs * (VERT ? TXW : 4 * TXW);
  const int f0 = mb * CW;
  const int HW = p.H * p.W;

  float* Us = w4_lds;
  float* Ps = Us + 3 * USLOT;
  float* Rs = Ps + 3 * PSLOT + wave * 1024;                     // GRU launches: [16 rows][64 lanes] of the pre-activation term
  const unsigned u_lds = scf_lds_addr(Us), p_lds = scf_lds_addr(Ps), r_lds = scf_lds_addr(Rs);

  // ---- chunk-invariant copy offsets ----
  // patch = 4 channel planes of PH rows x PWp floats: horizontal: the block's rows, columns from x0 - 2 (x0 - 4 with
  // 16-byte cells: windows then start at column 4 tx + 2); vertical: rows from y0 - 2, the block's columns (pitch 32)
  unsigned pvo[NPI];
  {
    const int NC = PX4 ? q.PWp >> 2 : q.PWp, PPC = q.PH * NC;
    const float rPPC = 1.0f / (float)PPC, rNC = 1.0f / (float)NC;
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
      const int e = i * 256 + tid;
      const int c = w4_div(e, PPC, rPPC), r = e - c * PPC;
      const int py = w4_div(r, NC, rNC), px = r - py * NC;
      const int iy = VERT ? y0 - 2 + py : y0 + py;
      const int ix = VERT ? x0 + 4 * px : (PX4 ? x0 - 4 + 4 * px : x0 - 2 + px);
      const bool ok = c < W4_KC && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      pvo[i] = ok ? (unsigned)((c * HW + iy * p.W + ix) * 4) : SCF_BUF_OOB;
    }
  }
  unsigned uvo[NUI], uld[NUI];
#pragma unroll
  for (int i = 0; i < NUI; ++i) {               // a fragment's chunk is 4 KB contiguous: 4 instructions of 1 KB
    const int j = wave + 4 * i;
    const int f = j >> 2, part = j & 3;
    uvo[i] = (unsigned)((f0 + f) * (W4_UF * 4) + part * 1024 + lane * 16);
    uld[i] = (unsigned)(f * (W4_UF * 4) + part * 1024);
  }
  const unsigned u_chunk_bytes = (unsigned)(q.F * W4_UF * 4);
  const unsigned u_total = (unsigned)q.nchunk * u_chunk_bytes;
  scf_rsrc4 urs = scf_make_rsrc(q.wu, u_total);
  int u_left = (int)u_total;
  auto issue_u = [&](int slot) {                   // the next U chunk -> ring slot (past the end: zeros)
    const unsigned dst = u_lds + (unsigned)(slot * USLOT * 4);
#pragma unroll
    for (int i = 0; i < NUI; ++i) scf_bdma_b128(urs, uvo[i], dst + uld[i]);
    const unsigned lo = (unsigned)urs[0] + u_chunk_bytes;
    urs[1] += lo < u_chunk_bytes ? 1 : 0;
    urs[0] = (int)lo;
    u_left -= (int)u_chunk_bytes;
    urs[2] = u_left > 0 ? u_left : 0;
  };
  const unsigned p_chunk_bytes = (unsigned)(W4_KC * HW * 4);
  int p_left = p.C0;                                 // channels of the current input segment still to copy
  bool p_second = p.in1 == nullptr;
  scf_rsrc4 prs = scf_make_rsrc(p.in0 + (long long)n * p.in0_ns, (unsigned)((p_left < W4_KC ? p_left : W4_KC) * HW * 4));
  auto issue_p = [&](int slot) {                   // the next patch chunk -> ring slot
    const unsigned dst = p_lds + (unsigned)((slot * PSLOT + wave * (PX4 ? 256 : 64)) * 4);
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
      if (PX4) scf_bdma_b128(prs, pvo[i], dst + (unsigned)(i * 4096));
      else scf_bdma_b32(prs, pvo[i], dst + (unsigned)(i * 1024));
    }
    p_left -= W4_KC;
    if (p_left <= 0 && !p_second) {                  // on to the second input segment (C0 % 4 == 0 there)
      p_second = true;
      p_left = p.Cin - p.C0;
      prs = scf_make_rsrc(p.in1 + (long long)n * p.in1_ns, 0u);
    } else {
      const unsigned lo = (unsigned)prs[0] + p_chunk_bytes;
      prs[1] += lo < p_chunk_bytes ? 1 : 0;
      prs[0] = (int)lo;
    }
    const int cl = p_left < W4_KC ? p_left : W4_KC;
    prs[2] = cl > 0 ? cl * HW * 4 : 0;
  };

  // ---- input transform in registers: lane (tile l32, k-half) turns the 8-windows of channels half and 2 + half into
  //      its B operands ----
  const int ty = tw * TYW + (l32 >> q.txl), tx = l32 & (TXW - 1);
  unsigned prow[2];                                  // absolute LDS byte address of the window of channel 2 s + half, slot 0
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int off = VERT ? 4 * ty * q.PWp + tx : ty * q.PWp + 4 * tx + (PX4 ? 2 : 0);
    prow[s] = p_lds + (unsigned)(((2 * s + half) * q.PPL + off) * 4);
  }
  float dw[2][8];
  auto win_load = [&](unsigned slot_bytes, int s) {
    const __attribute__((address_space(3))) float* r =
        (const __attribute__((address_space(3))) float*)(uintptr_t)(prow[s] + slot_bytes);
#pragma unroll
    for (int i = 0; i < 8; ++i) dw[s][i] = VERT ? r[i * 32] : r[i];       // vertical: pitch 32 floats
  };
  auto win_transform = [&](w4_f32x2 (&bo)[8], int s) {                    // -> bo[position][s]
    const float (&d)[8] = dw[s];
    const float t1 = __builtin_fmaf(-4.25f, d[4], d[2] + d[6]), t2 = __builtin_fmaf(-4.25f, d[3], d[1] + d[5]);
    const float t3 = __builtin_fmaf(-1.25f, d[4], __builtin_fmaf(0.25f, d[2], d[6]));
    const float t4 = __builtin_fmaf(2.f, d[5], __builtin_fmaf(-2.5f, d[3], 0.5f * d[1]));
    const float t5 = __builtin_fmaf(-5.f, d[4], __builtin_fmaf(4.f, d[2], d[6]));
    const float t6 = __builtin_fmaf(0.5f, d[5], __builtin_fmaf(-2.5f, d[3], 2.f * d[1]));
    bo[0][s] = __builtin_fmaf(5.25f, d[4] - d[2], d[0] - d[6]);
    bo[1][s] = t1 + t2;
    bo[2][s] = t1 - t2;
    bo[3][s] = t3 + t4;
    bo[4][s] = t3 - t4;
    bo[5][s] = t5 + t6;
    bo[6][s] = t5 - t6;
    bo[7][s] = __builtin_fmaf(5.25f, d[3] - d[5], d[7] - d[1]);
  };

  // ---- output pixels of this lane (four along the filter axis) ----
  const ConvEpi e = scf_conv_epi(p, n);
  int pix[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int oy = VERT ? y0 + 4 * ty + j : y0 + ty, ox = VERT ? x0 + tx : x0 + 4 * tx + j;
    pix[j] = (oy < p.Ho && ox < p.Wo) ? oy * p.Wo + ox : -1;
  }

  w4_f32x16 acc[8];
#pragma unroll
  for (int x = 0; x < 8; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
  // GRU launches: the pre-activation term enters through the accumulators (see the header).  Chunk k = 0 ... 3 requests
  // output column k of the wave's fragment (16 copies of one row x 64 lanes, issued IN FRONT of the chunk's U / patch
  // copies, so the counted wait at its end covers them), chunk k + 1 adds it in.  Rows past Cout: the descriptor range.
  // Straight-line code in the first five chunks of EVERY launch (a branch around accumulator updates costs register
  // copies): without a term the descriptor's range is empty, the copies deliver zeros and the adds add them.
  const bool pre_res = e.res && (p.mode == SCF_CONV_GRU_ZR || p.mode == SCF_CONV_GRU_Q) && p.out_div == 1.0f;
  const scf_rsrc4 rrs = scf_make_rsrc(pre_res ? e.res : p.out, pre_res ? (unsigned)(p.Cout * e.HWo * 4) : 0u);
  const unsigned r_row0 = (unsigned)((((f0 + cw) * 32 + 4 * half) * e.HWo) * 4);
  float rt[16];
  auto res_issue = [&](int j) __attribute__((always_inline)) {
    const unsigned vo = pix[j] >= 0 ? r_row0 + (unsigned)(pix[j] * 4) : SCF_BUF_OOB;
    const unsigned dst = r_lds;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      scf_bdma_b32(rrs, vo + (unsigned)((8 * (r >> 2) + (r & 3)) * e.HWo * 4), dst + (unsigned)(r * 256));
  };
  auto res_read = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 16; ++r) rt[r] = Rs[r * 64 + lane];
  };
  auto res_add = [&](int j) __attribute__((always_inline)) {            // column j of the term -> the positions A^T maps back onto it
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (j == 0) acc[0][r] += rt[r];
      if (j == 1) { acc[1][r] = __builtin_fmaf(0.5f, rt[r], acc[1][r]); acc[2][r] = __builtin_fmaf(-0.5f, rt[r], acc[2][r]); acc[7][r] -= rt[r]; }
      if (j == 2) { acc[0][r] -= rt[r]; acc[1][r] = __builtin_fmaf(0.5f, rt[r], acc[1][r]); acc[2][r] = __builtin_fmaf(0.5f, rt[r], acc[2][r]); }
      if (j == 3) acc[7][r] += rt[r];
    }
  };
  // ---- prologue: three chunks requested, two awaited ----
  issue_p(0); issue_u(0);
  issue_u(1); issue_p(1);
  issue_u(2); issue_p(2);
  scf_wait_vmcnt_imm<GRP>();
  __syncthreads();
  const float* ua = Us + cw * W4_UF + lane * 2;                        // + position * 128
  w4_f32x2 a0[8], b0[8], a1[8], b1[8];
#pragma unroll
  for (int x = 0; x < 8; ++x) a0[x] = *reinterpret_cast<const w4_f32x2*>(ua + x * 128);
  win_load(0u, 0); win_transform(b0, 0);
  win_load(0u, 1); win_transform(b0, 1);
  __syncthreads();                     // slot 0 of both rings is free again

  int s1 = 1;                          // ring slot of the next chunk
  auto chunk = [&](auto kc, const w4_f32x2 (&a)[8], const w4_f32x2 (&b)[8], w4_f32x2 (&an)[8], w4_f32x2 (&bn)[8]) {
    constexpr int K = decltype(kc)::value;        // 0 ... 4: the chunk's number (the pre-activation term's stream), 5: any later one
    const float* uc = ua + s1 * USLOT;
    const unsigned pcb = (unsigned)(s1 * PSLOT * 4);
    int s3 = s1 + 2;                   // ring slot of the chunk three ahead
    s3 = s3 >= 3 ? s3 - 3 : s3;
#define W4_M(X, S)                                                                              \
    acc[X] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[X][S], b[X][S], acc[X], 0, 0, 0);           \
    __builtin_amdgcn_sched_barrier(0);
    if (K >= 1 && K <= 4) {              // read back the column the last chunk requested, then request the next one
      res_read();
      __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): the reads are done before the next copies can land there
    }
    if (K <= 3) res_issue(K);
    __builtin_amdgcn_sched_barrier(0);
    W4_M(0, 0) win_load(pcb, 0); __builtin_amdgcn_sched_barrier(0);
    W4_M(1, 0) win_load(pcb, 1); __builtin_amdgcn_sched_barrier(0);
    W4_M(2, 0) issue_u(s3); __builtin_amdgcn_sched_barrier(0);
    W4_M(3, 0) issue_p(s3); __builtin_amdgcn_sched_barrier(0);
    W4_M(4, 0)
#pragma unroll
    for (int x = 0; x < 8; ++x) an[x] = *reinterpret_cast<const w4_f32x2*>(uc + x * 128);
    __builtin_amdgcn_sched_barrier(0);
    W4_M(5, 0) W4_M(6, 0) win_transform(bn, 0); __builtin_amdgcn_sched_barrier(0);
    W4_M(7, 0) W4_M(0, 1) W4_M(1, 1) win_transform(bn, 1); __builtin_amdgcn_sched_barrier(0);
    W4_M(2, 1) W4_M(3, 1) W4_M(4, 1) W4_M(5, 1) W4_M(6, 1) W4_M(7, 1)
#undef W4_M
    if (K >= 1 && K <= 4) res_add(K - 1);
    __builtin_amdgcn_sched_barrier(0);
    scf_wait_vmcnt_imm<GRP>();
    __syncthreads();
    s1 = s1 == 2 ? 0 : s1 + 1;
  };
#define W4_K(k) std::integral_constant<int, k>()
  chunk(W4_K(0), a0, b0, a1, b1);      // (the dispatch takes layers of at least five chunks)
  chunk(W4_K(1), a1, b1, a0, b0);
  chunk(W4_K(2), a0, b0, a1, b1);
  chunk(W4_K(3), a1, b1, a0, b0);
  chunk(W4_K(4), a0, b0, a1, b1);
  int c = 5;
  for (; c + 1 < q.nchunk; c += 2) {
    chunk(W4_K(5), a1, b1, a0, b0);
    chunk(W4_K(5), a0, b0, a1, b1);
  }
  if (c < q.nchunk) chunk(W4_K(5), a1, b1, a0, b0);
#undef W4_K
  scf_wait_vmcnt_imm<0>();             // the zero-filled groups past the end

  // ---- output transform, then the shared fused epilogue on the four pixels' 16-row fragments ----
  w4_f32x16 o[1][4];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float s12 = acc[1][r] + acc[2][r], d12 = acc[1][r] - acc[2][r];
    const float s34 = acc[3][r] + acc[4][r], d34 = acc[3][r] - acc[4][r];
    const float s56 = acc[5][r] + acc[6][r], d56 = acc[5][r] - acc[6][r];
    o[0][0][r] = ((acc[0][r] + s12) + s34) + s56;
    o[0][1][r] = __builtin_fmaf(0.5f, d56, __builtin_fmaf(2.f, d34, d12));
    o[0][2][r] = __builtin_fmaf(0.25f, s56, __builtin_fmaf(4.f, s34, s12));
    o[0][3][r] = __builtin_fmaf(0.125f, d56, __builtin_fmaf(8.f, d34, d12)) + acc[7][r];
  }
  ConvEpi e2 = e;
  if (pre_res) e2.res = nullptr;       // consumed through the accumulators
  const int kind = scf_conv_epi_kind(p);
  if ((kind == SCF_EPI_GRU_ZR || kind == SCF_EPI_GRU_Q) && !e2.res && p.out_div == 1.0f && (p.Cout & 63) == 0 &&
      ((uintptr_t)p.bias & 15) == 0) {
    const int cb = (f0 + cw) * 32 + 4 * half;
    // pixel quadruples: a row's four outputs are one aligned 16-byte cell (x0 and Wo are multiples of 4: all in or all out)
    const bool vec = !VERT && (p.Wo & 3) == 0 &&
                     ((((uintptr_t)e.out | (uintptr_t)e.gru_h | (uintptr_t)e.gru_aux | (uintptr_t)e.gru_z) & 15) == 0);
    if (kind == SCF_EPI_GRU_ZR) {
      if (!VERT && vec) w4_gru_epilogue<SCF_EPI_GRU_ZR, true>(p, e2, o, cb, pix);
      else w4_gru_epilogue<SCF_EPI_GRU_ZR, false>(p, e2, o, cb, pix);
    } else {
      if (!VERT && vec) w4_gru_epilogue<SCF_EPI_GRU_Q, true>(p, e2, o, cb, pix);
      else w4_gru_epilogue<SCF_EPI_GRU_Q, false>(p, e2, o, cb, pix);
    }
    return;
  }
  scf_conv_epilogue_tile<1, 4>(p, e2, o, (f0 + cw) * 32, half, pix, p.out_div != 1.0f);
}

// ===================================================================================================
// Half-domain form (r5): one computed B operand feeds TWO channel fragments.
// Above, a wave holds all 8 positions of ONE fragment, so the two waves of a tile group (one per fragment) each run the
// whole 26-operation input transform on the same windows -- and vector-ALU work costs matrix-pipe time.  Here
// a wave holds HALF the transform domain (positions 0 1 2 7, or 3 4 5 6) for BOTH fragments of the block = the same 128
// accumulators and 16 MFMAs per chunk, but 12 / 14 transform operations per window instead of 26.  A^T is linear in the
// positions: each wave applies it to its half, the two shares of an output meet through LDS once per block (the ring
// space, 16 KB per wave), and every wave finishes one fragment x four outputs in the layout of the kernel above -- the
// same epilogues.  B operands, U operands and every accumulator are bit-identical to the kernel above; the outputs
// differ by the association of the last adds.  The pre-activation term (positions 0 1 2 7 = one wave per tile group,
// both fragments) streams in over chunks 0 ... 7.
// ===================================================================================================
template <bool VERT, bool PX4, int NPI>
__global__ __launch_bounds__(256, 2)
void conv_wino1d4h_kernel(ConvK p, Wino4K q) {
  static_assert(!VERT || PX4, "the vertical kernel copies 16-byte cells only");
  extern __shared__ __attribute__((aligned(16))) float w4_lds[];
  constexpr int CW = 2, TW = 2;
  constexpr int USLOT = CW * W4_UF;                             // floats per ring slot
  constexpr int NUI = 2;                                        // U copy instructions (16 B per lane) per wave per chunk
  constexpr int PSLOT = NPI * (PX4 ? 1024 : 256);
  constexpr int GRP = NUI + NPI;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ph = wave >> 1, tw = wave & 1;           // half of the transform domain (positions 0 1 2 7 | 3 4 5 6), tile group
  const int half = lane >> 5, l32 = lane & 31;

  int lb = scf_xcd_remap(blockIdx.x, gridDim.x);
  const int mb = __builtin_amdgcn_readfirstlane(lb % q.mblocks);
  lb /= q.mblocks;
  const int xs = __builtin_amdgcn_readfirstlane(lb % q.sx);
  lb /= q.sx;
  const int ys = __builtin_amdgcn_readfirstlane(lb % q.sy);
  const int n = __builtin_amdgcn_readfirstlane(lb / q.sy);
  const int TXW = 1 << q.txl, TYW = 32 >> q.txl;
  // first output pixel of the block; a tile = 4 pixels along the filter axis
  const int y0 = ys * (VERT ? 4 * TW * TYW : TW * TYW), x0 = xs * (VERT ? TXW : 4 * TXW);
  const int f0 = mb * CW;
  const int HW = p.H * p.W;

  float* Us = w4_lds;
  float* Ps = Us + 3 * USLOT;
  float* Rs = Ps + 3 * PSLOT + wave * 1024;                     // GRU launches: [16 rows][64 lanes] of the pre-activation term
  const unsigned u_lds = scf_lds_addr(Us), p_lds = scf_lds_addr(Ps), r_lds = scf_lds_addr(Rs);

  // ---- chunk-invariant copy offsets ----
  // patch = 4 channel planes of PH rows x PWp floats: horizontal: the block's rows, columns from x0 - 2 (x0 - 4 with
  // 16-byte cells: windows then start at column 4 tx + 2); vertical: rows from y0 - 2, the block's columns (pitch 32)
  unsigned pvo[NPI];
  {
    const int NC = PX4 ? q.PWp >> 2 : q.PWp, PPC = q.PH * NC;
    const float rPPC = 1.0f / (float)PPC, rNC = 1.0f / (float)NC;
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
      const int e = i * 256 + tid;
      const int c = w4_div(e, PPC, rPPC), r = e - c * PPC;
      const int py = w4_div(r, NC, rNC), px = r - py * NC;
      const int iy = VERT ? y0 - 2 + py : y0 + py;
      const int ix = VERT ? x0 + 4 * px : (PX4 ? x0 - 4 + 4 * px : x0 - 2 + px);
      const bool ok = c < W4_KC && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      pvo[i] = ok ? (unsigned)((c * HW + iy * p.W + ix) * 4) : SCF_BUF_OOB;
    }
  }
  unsigned uvo[NUI], uld[NUI];
#pragma unroll
  for (int i = 0; i < NUI; ++i) {               // a fragment's chunk is 4 KB contiguous: 4 instructions of 1 KB
    const int j = wave + 4 * i;
    const int f = j >> 2, part = j & 3;
    uvo[i] = (unsigned)((f0 + f) * (W4_UF * 4) + part * 1024 + lane * 16);
    uld[i] = (unsigned)(f * (W4_UF * 4) + part * 1024);
  }
  const unsigned u_chunk_bytes = (unsigned)(q.F * W4_UF * 4);
  const unsigned u_total = (unsigned)q.nchunk * u_chunk_bytes;
  scf_rsrc4 urs = scf_make_rsrc(q.wu, u_total);
  int u_left = (int)u_total;
  auto issue_u = [&](int slot) {                   // the next U chunk -> ring slot (past the end: zeros)
    const unsigned dst = u_lds + (unsigned)(slot * USLOT * 4);
#pragma unroll
    for (int i = 0; i < NUI; ++i) scf_bdma_b128(urs, uvo[i], dst + uld[i]);
    const unsigned lo = (unsigned)urs[0] + u_chunk_bytes;
    urs[1] += lo < u_chunk_bytes ? 1 : 0;
    urs[0] = (int)lo;
    u_left -= (int)u_chunk_bytes;
    urs[2] = u_left > 0 ? u_left : 0;
  };
  const unsigned p_chunk_bytes = (unsigned)(W4_KC * HW * 4);
  int p_left = p.C0;                                 // channels of the current input segment still to copy
  bool p_second = p.in1 == nullptr;
  scf_rsrc4 prs = scf_make_rsrc(p.in0 + (long long)n * p.in0_ns, (unsigned)((p_left < W4_KC ? p_left : W4_KC) * HW * 4));
  auto issue_p = [&](int slot) {                   // the next patch chunk -> ring slot
    const unsigned dst = p_lds + (unsigned)((slot * PSLOT + wave * (PX4 ? 256 : 64)) * 4);
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
      if (PX4) scf_bdma_b128(prs, pvo[i], dst + (unsigned)(i * 4096));
      else scf_bdma_b32(prs, pvo[i], dst + (unsigned)(i * 1024));
    }
    p_left -= W4_KC;
    if (p_left <= 0 && !p_second) {                  // on to the second input segment (C0 % 4 == 0 there)
      p_second = true;
      p_left = p.Cin - p.C0;
      prs = scf_make_rsrc(p.in1 + (long long)n * p.in1_ns, 0u);
    } else {
      const unsigned lo = (unsigned)prs[0] + p_chunk_bytes;
      prs[1] += lo < p_chunk_bytes ? 1 : 0;
      prs[0] = (int)lo;
    }
    const int cl = p_left < W4_KC ? p_left : W4_KC;
    prs[2] = cl > 0 ? cl * HW * 4 : 0;
  };

  // ---- input transform in registers: lane (tile l32, k-half) turns the 8-windows of channels half and 2 + half into
  //      its B operands ----
  const int ty = tw * TYW + (l32 >> q.txl), tx = l32 & (TXW - 1);
  unsigned prow[2];                                  // absolute LDS byte address of the window of channel 2 s + half, slot 0
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int off = VERT ? 4 * ty * q.PWp + tx : ty * q.PWp + 4 * tx + (PX4 ? 2 : 0);
    prow[s] = p_lds + (unsigned)(((2 * s + half) * q.PPL + off) * 4);
  }
  float dw[2][8];
  auto win_load = [&](unsigned slot_bytes, int s) {
    const __attribute__((address_space(3))) float* r =
        (const __attribute__((address_space(3))) float*)(uintptr_t)(prow[s] + slot_bytes);
#pragma unroll
    for (int i = 0; i < 8; ++i) dw[s][i] = VERT ? r[i * 32] : r[i];       // vertical: pitch 32 floats
  };
  auto win_transform = [&](w4_f32x2 (&bo)[4], int s) {                    // -> bo[position of this wave's half][s]
    const float (&d)[8] = dw[s];
    if (ph == 0) {                                   // wave-uniform: 12 of the 26 operations
      const float t1 = __builtin_fmaf(-4.25f, d[4], d[2] + d[6]), t2 = __builtin_fmaf(-4.25f, d[3], d[1] + d[5]);
      bo[0][s] = __builtin_fmaf(5.25f, d[4] - d[2], d[0] - d[6]);
      bo[1][s] = t1 + t2;
      bo[2][s] = t1 - t2;
      bo[3][s] = __builtin_fmaf(5.25f, d[3] - d[5], d[7] - d[1]);
    } else {                                         // 14
      const float t3 = __builtin_fmaf(-1.25f, d[4], __builtin_fmaf(0.25f, d[2], d[6]));
      const float t4 = __builtin_fmaf(2.f, d[5], __builtin_fmaf(-2.5f, d[3], 0.5f * d[1]));
      const float t5 = __builtin_fmaf(-5.f, d[4], __builtin_fmaf(4.f, d[2], d[6]));
      const float t6 = __builtin_fmaf(0.5f, d[5], __builtin_fmaf(-2.5f, d[3], 2.f * d[1]));
      bo[0][s] = t3 + t4;
      bo[1][s] = t3 - t4;
      bo[2][s] = t5 + t6;
      bo[3][s] = t5 - t6;
    }
  };

  // ---- output pixels of this lane (four along the filter axis) ----
  const ConvEpi e = scf_conv_epi(p, n);
  int pix[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int oy = VERT ? y0 + 4 * ty + j : y0 + ty, ox = VERT ? x0 + tx : x0 + 4 * tx + j;
    pix[j] = (oy < p.Ho && ox < p.Wo) ? oy * p.Wo + ox : -1;
  }

  w4_f32x16 acc[8];
#pragma unroll
  for (int x = 0; x < 8; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
  // GRU launches: the pre-activation term enters through the accumulators (see the header).  Chunk k = 0 ... 3 requests
  // output column k of the wave's fragment (16 copies of one row x 64 lanes, issued IN FRONT of the chunk's U / patch
  // copies, so the counted wait at its end covers them), chunk k + 1 adds it in.  Rows past Cout: the descriptor range.
  // Straight-line code in the first five chunks of EVERY launch (a branch around accumulator updates costs register
  // copies): without a term the descriptor's range is empty, the copies deliver zeros and the adds add them.
  // slot group g (accumulators 4 g ... 4 g + 3) holds channel fragment g ^ ph: group 0 is the fragment this wave finishes
  // after the exchange.  The pre-activation term lives on positions 0, 1, 2, 7 = the ph = 0 waves; chunks 0 ... 3 stream its
  // four columns for group 0, chunks 4 ... 7 for group 1 (the other waves run the same code on an empty descriptor range).
  const bool pre_res = e.res && (p.mode == SCF_CONV_GRU_ZR || p.mode == SCF_CONV_GRU_Q) && p.out_div == 1.0f;
  const bool my_res = pre_res && ph == 0;
  const scf_rsrc4 rrs = scf_make_rsrc(my_res ? e.res : p.out, my_res ? (unsigned)(p.Cout * e.HWo * 4) : 0u);
  float rt[16];
  auto res_issue = [&](int j, int g) __attribute__((always_inline)) {
    const unsigned r_row0 = (unsigned)((((f0 + (g ^ ph)) * 32 + 4 * half) * e.HWo) * 4);
    const unsigned vo = pix[j] >= 0 ? r_row0 + (unsigned)(pix[j] * 4) : SCF_BUF_OOB;
    const unsigned dst = r_lds;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      scf_bdma_b32(rrs, vo + (unsigned)((8 * (r >> 2) + (r & 3)) * e.HWo * 4), dst + (unsigned)(r * 256));
  };
  auto res_read = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 16; ++r) rt[r] = Rs[r * 64 + lane];
  };
  // column j of the term -> the positions A^T maps back onto it: slots 0 1 2 3 of the group = positions 0 1 2 7
  auto res_add = [&](int j, auto gc) __attribute__((always_inline)) {
    constexpr int B = 4 * decltype(gc)::value;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (j == 0) acc[B][r] += rt[r];
      if (j == 1) { acc[B + 1][r] = __builtin_fmaf(0.5f, rt[r], acc[B + 1][r]); acc[B + 2][r] = __builtin_fmaf(-0.5f, rt[r], acc[B + 2][r]); acc[B + 3][r] -= rt[r]; }
      if (j == 2) { acc[B][r] -= rt[r]; acc[B + 1][r] = __builtin_fmaf(0.5f, rt[r], acc[B + 1][r]); acc[B + 2][r] = __builtin_fmaf(0.5f, rt[r], acc[B + 2][r]); }
      if (j == 3) acc[B + 3][r] += rt[r];
    }
  };
  // ---- prologue: three chunks requested, two awaited ----
  issue_p(0); issue_u(0);
  issue_u(1); issue_p(1);
  issue_u(2); issue_p(2);
  scf_wait_vmcnt_imm<GRP>();
  __syncthreads();
  // U operands: group g = fragment g ^ ph; slots 0 1 2 = positions pa, pa + 1, pa + 2, slot 3 = position pb
  const int pa = ph ? 3 : 0, pb = ph ? 6 : 7;
  const float* uaA[2] = {Us + ph * W4_UF + pa * 128 + lane * 2, Us + (ph ^ 1) * W4_UF + pa * 128 + lane * 2};
  const int ub = (pb - pa) * 128;
  w4_f32x2 a0[8], b0[4], a1[8], b1[4];
#pragma unroll
  for (int x = 0; x < 8; ++x) a0[x] = *reinterpret_cast<const w4_f32x2*>(uaA[x >> 2] + ((x & 3) < 3 ? (x & 3) * 128 : ub));
  win_load(0u, 0); win_transform(b0, 0);
  win_load(0u, 1); win_transform(b0, 1);
  __syncthreads();                     // slot 0 of both rings is free again

  int s1 = 1;                          // ring slot of the next chunk
  auto chunk = [&](auto kc, const w4_f32x2 (&a)[8], const w4_f32x2 (&b)[4], w4_f32x2 (&an)[8], w4_f32x2 (&bn)[4]) {
    constexpr int K = decltype(kc)::value;        // 0 ... 8: the chunk's number (the pre-activation term's stream), 9: any later one
    const int uoff = s1 * USLOT;
    const unsigned pcb = (unsigned)(s1 * PSLOT * 4);
    int s3 = s1 + 2;                   // ring slot of the chunk three ahead
    s3 = s3 >= 3 ? s3 - 3 : s3;
#define W4_M(X, S)                                                                              \
    acc[X] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[X][S], b[(X) & 3][S], acc[X], 0, 0, 0);     \
    __builtin_amdgcn_sched_barrier(0);
    if (K >= 1 && K <= 8) {              // read back the column the last chunk requested, then request the next one
      res_read();
      __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): the reads are done before the next copies can land there
    }
    if (K <= 7) res_issue(K & 3, K >> 2);
    __builtin_amdgcn_sched_barrier(0);
    W4_M(0, 0) win_load(pcb, 0); __builtin_amdgcn_sched_barrier(0);
    W4_M(1, 0) win_load(pcb, 1); __builtin_amdgcn_sched_barrier(0);
    W4_M(2, 0) issue_u(s3); __builtin_amdgcn_sched_barrier(0);
    W4_M(3, 0) issue_p(s3); __builtin_amdgcn_sched_barrier(0);
    W4_M(4, 0)
#pragma unroll
    for (int x = 0; x < 8; ++x) an[x] = *reinterpret_cast<const w4_f32x2*>(uaA[x >> 2] + uoff + ((x & 3) < 3 ? (x & 3) * 128 : ub));
    __builtin_amdgcn_sched_barrier(0);
    W4_M(5, 0) W4_M(6, 0) win_transform(bn, 0); __builtin_amdgcn_sched_barrier(0);
    W4_M(7, 0) W4_M(0, 1) W4_M(1, 1) win_transform(bn, 1); __builtin_amdgcn_sched_barrier(0);
    W4_M(2, 1) W4_M(3, 1) W4_M(4, 1) W4_M(5, 1) W4_M(6, 1) W4_M(7, 1)
#undef W4_M
    if (K >= 1 && K <= 8) res_add((K - 1) & 3, std::integral_constant<int, ((K - 1) >> 2) & 1>());
    __builtin_amdgcn_sched_barrier(0);
    scf_wait_vmcnt_imm<GRP>();
    __syncthreads();
    s1 = s1 == 2 ? 0 : s1 + 1;
  };
#define W4_K(k) std::integral_constant<int, k>()
  chunk(W4_K(0), a0, b0, a1, b1);      // (the dispatch takes layers of at least nine chunks)
  chunk(W4_K(1), a1, b1, a0, b0);
  chunk(W4_K(2), a0, b0, a1, b1);
  chunk(W4_K(3), a1, b1, a0, b0);
  chunk(W4_K(4), a0, b0, a1, b1);
  chunk(W4_K(5), a1, b1, a0, b0);
  chunk(W4_K(6), a0, b0, a1, b1);
  chunk(W4_K(7), a1, b1, a0, b0);
  chunk(W4_K(8), a0, b0, a1, b1);
  int c = 9;
  for (; c + 1 < q.nchunk; c += 2) {
    chunk(W4_K(9), a1, b1, a0, b0);
    chunk(W4_K(9), a0, b0, a1, b1);
  }
  if (c < q.nchunk) chunk(W4_K(9), a1, b1, a0, b0);
#undef W4_K
  scf_wait_vmcnt_imm<0>();             // the zero-filled groups past the end

  __syncthreads();                     // every wave is out of the loop: the rings are free for the exchange

  // ---- output transform: each wave forms ITS half's share of the four outputs for both slot groups (A^T is linear in the
  //      positions), hands group 1's share to the partner wave (the other half of the same tile group: wave ^ 2) through
  //      LDS and adds the partner's share of group 0: y = (M0 + (M1 + M2) | (M1 - M2) | (M1 + M2) | (M1 - M2) + M7)
  //                                                       + ((M3 + M4) + (M5 + M6) | 2 (M3 - M4) + (M5 - M6) / 2 | ...) ----
  w4_f32x16 yp[2][4];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float m0 = acc[4 * g][r], m1 = acc[4 * g + 1][r], m2 = acc[4 * g + 2][r], m3 = acc[4 * g + 3][r];
      if (ph == 0) {                     // m = M0, M1, M2, M7
        const float s12 = m1 + m2, d12 = m1 - m2;
        yp[g][0][r] = m0 + s12; yp[g][1][r] = d12; yp[g][2][r] = s12; yp[g][3][r] = d12 + m3;
      } else {                           // m = M3, M4, M5, M6
        const float s34 = m0 + m1, d34 = m0 - m1, s56 = m2 + m3, d56 = m2 - m3;
        yp[g][0][r] = s34 + s56;
        yp[g][1][r] = __builtin_fmaf(0.5f, d56, 2.f * d34);
        yp[g][2][r] = __builtin_fmaf(0.25f, s56, 4.f * s34);
        yp[g][3][r] = __builtin_fmaf(0.125f, d56, 8.f * d34);
      }
    }
  {
    float* box = w4_lds + wave * 4096 + lane * 4;         // [16 cells][64 lanes][4 floats] per wave
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)
        *reinterpret_cast<scf_f32x4*>(box + (j * 4 + r4) * 256) =
            scf_f32x4{yp[1][j][4 * r4], yp[1][j][4 * r4 + 1], yp[1][j][4 * r4 + 2], yp[1][j][4 * r4 + 3]};
  }
  __syncthreads();
  w4_f32x16 o[1][4];
  {
    const float* box = w4_lds + (wave ^ 2) * 4096 + lane * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const scf_f32x4 v = *reinterpret_cast<const scf_f32x4*>(box + (j * 4 + r4) * 256);
#pragma unroll
        for (int u = 0; u < 4; ++u) o[0][j][4 * r4 + u] = yp[0][j][4 * r4 + u] + v[u];
      }
  }
  const int cw = ph;                   // the fragment this wave finishes
  ConvEpi e2 = e;
  if (pre_res) e2.res = nullptr;       // consumed through the accumulators
  const int kind = scf_conv_epi_kind(p);
  if ((kind == SCF_EPI_GRU_ZR || kind == SCF_EPI_GRU_Q) && !e2.res && p.out_div == 1.0f && (p.Cout & 63) == 0 &&
      ((uintptr_t)p.bias & 15) == 0) {
    const int cb = (f0 + cw) * 32 + 4 * half;
    // pixel quadruples: a row's four outputs are one aligned 16-byte cell (x0 and Wo are multiples of 4: all in or all out)
    const bool vec = !VERT && (p.Wo & 3) == 0 &&
                     ((((uintptr_t)e.out | (uintptr_t)e.gru_h | (uintptr_t)e.gru_aux | (uintptr_t)e.gru_z) & 15) == 0);
    if (kind == SCF_EPI_GRU_ZR) {
      if (!VERT && vec) w4_gru_epilogue<SCF_EPI_GRU_ZR, true>(p, e2, o, cb, pix);
      else w4_gru_epilogue<SCF_EPI_GRU_ZR, false>(p, e2, o, cb, pix);
    } else {
      if (!VERT && vec) w4_gru_epilogue<SCF_EPI_GRU_Q, true>(p, e2, o, cb, pix);
      else w4_gru_epilogue<SCF_EPI_GRU_Q, false>(p, e2, o, cb, pix);
    }
    return;
  }
  scf_conv_epilogue_tile<1, 4>(p, e2, o, (f0 + cw) * 32, half, pix, p.out_div != 1.0f);
}

// ---------------------------------------------------------------------------------------------------
// Host side: packing and launch
// ---------------------------------------------------------------------------------------------------
extern "C" int64_t scf_pack_conv_weight_wino1d4_size(int32_t cout, int32_t cin) {
  if (cout <= 0 || cin <= 0) return 0;
  const int64_t F = (cout + 31) / 32, nchunk = (cin + W4_KC - 1) / W4_KC;
  return nchunk * F * W4_UF;
}

// w: (Cout, Cin, 5) = a 1 x 5 or 5 x 1 kernel's taps in filter order
extern "C" int scf_pack_conv_weight_wino1d4(const float* w, int32_t cout, int32_t cin, float* out) {
  if (!w || !out || cout <= 0 || cin <= 0) return SCF_EINVAL;
  const int F = (cout + 31) / 32;
  memset(out, 0, sizeof(float) * (size_t)scf_pack_conv_weight_wino1d4_size(cout, cin));
  // G (8 x 5) for the points 0, 1, -1, 2, -2, 1/2, -1/2, infinity: row i = (1, a_i, ..., a_i^4) / prod_{k != i} (a_i - a_k),
  // row 0 with the sign that goes with B^T row 0 = (d0 - d6) + 21/4 (d4 - d2)
  static const double pts[7] = {0.0, 1.0, -1.0, 2.0, -2.0, 0.5, -0.5};
  double G[8][5];
  for (int i = 0; i < 7; ++i) {
    double nrm = 1.0;
    for (int k = 0; k < 7; ++k)
      if (k != i) nrm *= pts[i] - pts[k];
    if (i == 0) nrm = -nrm;
    double pw = 1.0;
    for (int k = 0; k < 5; ++k) { G[i][k] = pw / nrm; pw *= pts[i]; }
  }
  for (int k = 0; k < 5; ++k) G[7][k] = k == 4 ? 1.0 : 0.0;
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci) {
      const float* g = w + ((size_t)co * cin + ci) * 5;
      const int chunk = ci / W4_KC, cl = ci % W4_KC, s = cl >> 1, kh = cl & 1;
      const int frag = co / 32, m = co % 32;
      for (int i = 0; i < 8; ++i) {
        double u = 0.0;
        for (int k = 0; k < 5; ++k) u += G[i][k] * g[k];
        out[(((size_t)chunk * F + frag) * 8 + i) * 128 + kh * 64 + m * 2 + s] = (float)u;
      }
    }
  return SCF_OK;
}

// scf_tune(SCF_TUNE_WINO1D4_HALF, v): 1 = the half-domain kernel wherever it applies (>= 9 chunks), 0 (default) = the
// full-domain kernel.  Measured (MI355X, tools/lab/wino1d4_half_ab.py, profiles/r5_wino1d4_half_ab.txt): plain layers
// x1.05 (384->256 1x5) ... x0.95 (q 256->128 1x5), the GRU cell 0.299 -> 0.305 ms, the batch-32 step 14.04 -> 14.10 ms:
// halving the transform work buys ~5 % of the loop on the longest layers, the exchange and the longer term stream take it
// back -- the vector-ALU share of this kernel's loop is not what holds it at 0.55.  Kept behind the knob.
static std::atomic<int> g_w4_half{0};
int scf_wino1d4_half_set(int v) { return (v < 0 || v > 1) ? SCF_EINVAL : g_w4_half.exchange(v); }

// Tile selection + launch; SCF_EUNSUPPORTED -> the caller goes on to the F(2, 5) / direct kernels.
// info: {8 transform positions, 4 fragments per block, blocks, LDS bytes}.  any_grid: also on grids the F(2, 5) kernel fills
// better (tests of ragged shapes).
int scf_conv_wino1d4_dispatch(ConvK k, const float* wu, int N, bool any_grid, bool dry_run, int* info, hipStream_t st) {
  const bool vert = k.KH == 5 && k.KW == 1 && k.pad_h == 2 && k.pad_w == 0;
  const bool horz = k.KH == 1 && k.KW == 5 && k.pad_h == 0 && k.pad_w == 2;
  if (!wu || !(vert || horz) || k.stride != 1 || k.w_ns != 0 || k.out_tile) return SCF_EUNSUPPORTED;
  if (k.in1 && (k.C0 % W4_KC) != 0) return SCF_EUNSUPPORTED;
  if (((uintptr_t)wu & 15) || (long long)W4_KC * k.H * k.W * 4 >= 0x7fffffffLL) return SCF_EUNSUPPORTED;
  const int F = (k.Cout + 31) / 32;
  if (F % 2) return SCF_EUNSUPPORTED;                   // blocks take pairs of channel fragments
  const bool px4 = (k.W % 4) == 0 && (((uintptr_t)k.in0 | (uintptr_t)k.in1) & 15) == 0 && (k.in0_ns % 4) == 0 && (k.in1_ns % 4) == 0;
  if (vert && !px4) return SCF_EUNSUPPORTED;
  const int TW = 2;
  int txl;
  Wino4K q;
  q.wu = wu; q.F = F;
  if (vert) {
    txl = k.Wo > 16 ? 5 : 4;                            // 32 columns x 1 tile row, or 16 x 2
    const int TXW = 1 << txl, TYW = 32 >> txl;
    q.PH = 4 * TW * TYW + 4; q.PWp = 32;
    q.sx = (k.Wo + TXW - 1) / TXW;
    q.sy = (k.Ho + 4 * TW * TYW - 1) / (4 * TW * TYW);
  } else {
    const int tcols = (k.Wo + 3) / 4;
    txl = 3;                                            // 8 tiles = 32 pixels per group row unless narrower wastes less
    if (tcols < 8) { txl = 1; while ((1 << txl) < tcols) ++txl; }
    else {
      int best = (tcols + 15) / 16 * 16;                // 16 tiles per row (64 pixels) as the widest candidate
      txl = 4;
      for (int l = 3; l >= 2; --l) {
        const int wpad = (tcols + (1 << l) - 1) >> l << l;
        if (wpad * 10 <= best * 9) { best = wpad; txl = l; }
      }
    }
    const int TXW = 1 << txl, TYW = 32 >> txl;
    q.PH = TW * TYW; q.PWp = px4 ? 4 * TXW + 8 : 4 * TXW + 4;
    q.sx = (k.Wo + 4 * TXW - 1) / (4 * TXW);
    q.sy = (k.Ho + TW * TYW - 1) / (TW * TYW);
  }
  q.txl = txl;
  q.PPL = q.PH * q.PWp;
  const int cells = W4_KC * q.PPL / (px4 ? 4 : 1);
  const int npi = px4 ? (cells <= 512 ? 2 : 3) : 5;
  if (cells > npi * 256 || (!vert && px4 && npi != 2)) return SCF_EUNSUPPORTED;
  q.nchunk = (k.Cin + W4_KC - 1) / W4_KC;
  if (q.nchunk < 5) return SCF_EUNSUPPORTED;           // the kernel's first five chunks are peeled
  q.mblocks = F / 2;
  const long long nblk = (long long)N * q.sx * q.sy * q.mblocks;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return SCF_EUNSUPPORTED;
  // a block is twice the pixels of an F(2, 5) block: up to CUs / 2 blocks the F(2, 5) kernel, one block per CU too, is the
  // faster one (1152 vs 1536 MFMAs per wave at Cin 384); above that its blocks share SIMDs and this kernel wins even with
  // one wave per SIMD (MI355X, GRU q launch at batch 32 = 256 blocks: 73 ... 85 us vs 95 ... 105 us)
  if (!any_grid && nblk * 2 <= (long long)scf_cu_count()) return SCF_EUNSUPPORTED;
  size_t ldsb = (size_t)(3 * 2 * W4_UF + 3 * npi * (px4 ? 1024 : 256) + 4 * 1024) * sizeof(float);    // rings + the term's 4 KB per wave
  const bool halfdom = g_w4_half.load(std::memory_order_relaxed) != 0 && q.nchunk >= 9;       // its first nine chunks are peeled
  if (halfdom && ldsb < 64 * 1024) ldsb = 64 * 1024;                                          // the output exchange: 16 KB per wave
  if (info) { info[0] = 8; info[1] = 4; info[2] = (int)nblk; info[3] = (int)ldsb; }      // positions, fragments per block
  if (dry_run) return SCF_OK;
  const int cfg = vert ? (npi == 2 ? 0 : 1) : (px4 ? 2 : 3);
  if (halfdom) {
    if (ldsb > 64 * 1024) {
      static std::atomic<unsigned long long> raised_h[4];
      const void* fn = cfg == 0 ? (const void*)conv_wino1d4h_kernel<true, true, 2> : cfg == 1 ? (const void*)conv_wino1d4h_kernel<true, true, 3>
                     : cfg == 2 ? (const void*)conv_wino1d4h_kernel<false, true, 2> : (const void*)conv_wino1d4h_kernel<false, false, 5>;
      const int rc = scf_raise_dynamic_lds(raised_h[cfg], fn, 80 * 1024);
      if (rc != SCF_OK) return rc;
    }
    if (vert && npi == 2) scf_launch((conv_wino1d4h_kernel<true, true, 2>), dim3((unsigned)nblk), dim3(256), ldsb, st, k, q);
    else if (vert) scf_launch((conv_wino1d4h_kernel<true, true, 3>), dim3((unsigned)nblk), dim3(256), ldsb, st, k, q);
    else if (px4) scf_launch((conv_wino1d4h_kernel<false, true, 2>), dim3((unsigned)nblk), dim3(256), ldsb, st, k, q);
    else scf_launch((conv_wino1d4h_kernel<false, false, 5>), dim3((unsigned)nblk), dim3(256), ldsb, st, k, q);
    return scf_launch_status();
  }
  if (ldsb > 64 * 1024) {
    static std::atomic<unsigned long long> raised[4];
    const void* fn = cfg == 0 ? (const void*)conv_wino1d4_kernel<true, true, 2> : cfg == 1 ? (const void*)conv_wino1d4_kernel<true, true, 3>
                   : cfg == 2 ? (const void*)conv_wino1d4_kernel<false, true, 2> : (const void*)conv_wino1d4_kernel<false, false, 5>;
    const int rc = scf_raise_dynamic_lds(raised[cfg], fn, 80 * 1024);
    if (rc != SCF_OK) return rc;
  }
  if (vert && npi == 2) scf_launch((conv_wino1d4_kernel<true, true, 2>), dim3((unsigned)nblk), dim3(256), ldsb, st, k, q);
  else if (vert) scf_launch((conv_wino1d4_kernel<true, true, 3>), dim3((unsigned)nblk), dim3(256), ldsb, st, k, q);
  else if (px4) scf_launch((conv_wino1d4_kernel<false, true, 2>), dim3((unsigned)nblk), dim3(256), ldsb, st, k, q);
  else scf_launch((conv_wino1d4_kernel<false, false, 5>), dim3((unsigned)nblk), dim3(256), ldsb, st, k, q);
  return scf_launch_status();
}
