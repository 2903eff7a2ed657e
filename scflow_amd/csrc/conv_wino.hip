// 3x3 / stride-1 / pad-1 convolution in the Winograd F(2x2, 3x3) form, fp32 throughout:
//     Y = A^T [ (G g G^T) . (B^T d B) ] A        (Lavin & Gray; the reference reaches the same
// algorithm through cuDNN for these layers: ConvModule / nn.Conv2d 3x3 in resnet.py:75-86,
// raft_decoder.py:141-160, 430-440, scflow_decoder.py:100-105.)
// 16 multiplies per 2x2 outputs and channel pair instead of 36: 2.25x fewer matrix-core flops than
// the direct kernels of conv_dma.hip, on the layers that are matrix-core bound there.
//
//   GEMM view   for each of the 16 transform positions xi:  M[xi][cout][tile] = sum_cin U[xi][cout][cin] V[xi][cin][tile]
//               v_mfma_f32_32x32x2_f32, M = 32 output channels, N = 32 tiles (2 x 2 outputs each), k = 2 channels.
//   wave pair   one 32 x 32 fragment: each of the two waves accumulates 8 of the 16 xi (two rows of the 4 x 4
//               transform domain) = 128 accumulator registers, so TWO waves fit a SIMD and two blocks a CU:
//               one block's staging, barriers and epilogue run under the other's MFMAs (a first version with
//               all 16 xi in one wave = one wave per SIMD measured MFMA time + everything else, no overlap).
//               The output transform A^T M A is linear in the rows: each wave transforms its two rows, the
//               pair exchanges half of the 4-value partial results through LDS and each finishes 8 of the
//               fragment's 16 channel rows.
//   block       4 waves = 2 fragments: one channel fragment of two vertically stacked tile groups (an 8 x 32
//               output region on a wide map); the two wave pairs share the U chunk.
//   U           = G g G^T, transformed and packed on the host (scf_pack_conv_weight_wino) in the exact
//               LDS image [chunk][fragment][xi][k-half][cout][2] (rows of the transform domain in the order
//               0, 1, 3, 2, see below): one ds_read_b64 per xi feeds both k-steps of a 4-channel chunk;
//               staged by LDS-DMA into a 3-deep ring.
//   V           = B^T d B, computed IN REGISTERS: the raw input patch of a later chunk is staged by LDS-DMA
//               (descriptor range check = zero padding); lane (tile, k-half) reads the three input rows its
//               wave's two transform rows need and computes its own B operands (16 packed adds per chunk),
//               between the MFMAs of the current chunk.  No V buffer, one barrier per chunk.
//   pipeline    chunk c: MFMAs on operands already in registers; the operands of chunk c+1 are read (U) /
//               computed (patch) under them; the copies of chunk c+3 are issued, those of c+2 awaited.
//   epilogue    output transform, pair exchange, then the affine epilogue (bias, BN scale/shift, residual,
//               ReLU) and float2 stores: 16 lanes cover one full 128-byte line of an output row.
//   grid        small grids (< CUs / 2 blocks) stay on the direct kernels (dispatch below).
//
// Arithmetic: fp32 adds / fmas only; the transforms re-associate the sum, so results differ from the
// direct kernel by a few ulp of the accumulated magnitude (measured in tests/test_gpu_ops.py).
#include <stdlib.h>
#include <string.h>
#include "scf_common.h"
#include "conv_kernels.h"
#include "scf_dma.h"

typedef float wn_f32x16 __attribute__((ext_vector_type(16)));
typedef float wn_f32x4 __attribute__((ext_vector_type(4)));
typedef float wn_f32x2 __attribute__((ext_vector_type(2)));

// patch DMA instructions per wave per chunk, fixed per block shape (lanes past the patch write zeros into the
// slot's padding): 16-byte cells (256 cells per block-instruction) when the rows are 16-byte aligned, else dwords
#define WN_NPI(TW, PX4) ((PX4) ? ((TW) == 2 ? 3 : 2) : ((TW) == 2 ? 8 : 5))
#define WN_KC 4             // channels per chunk
#ifndef SCF_WINO_DEFAULT_VARIANT
#define SCF_WINO_DEFAULT_VARIANT 2      // 1 = pair kernel, 2 = quarter-domain kernel (4 waves); 3 = its 8-wave form, lab builds only
#endif

// tools/lab/wino_phases.py builds this file with compile-time phase ablations (tools/lab/wino_lab_hooks.h,
// -DSCF_WINO_LAB -DSCF_WINO_LAB_MASK=m); the product build sees constants
#ifdef SCF_WINO_LAB
#include "wino_lab_hooks.h"      // lab builds only: -I tools/lab
#else
#define WN_LAB(bit) 0
#define WN_LAB_FIELDS
#endif

struct WinoK {
  const float* wu;          // [nchunk][F][16][2][32][2]
  int F;                    // channel fragments in the packing
  int txl;                  // log2(tiles per row of a wave's 32-tile group)
  int PH, PW, PWp, PPL;     // patch rows, columns, row pitch, plane stride (floats)
  int nchunk;
  int sx, sy;               // strips per image
  int mblocks;
  WN_LAB_FIELDS
};

// floor(e / d) for 0 <= e < 2^20, 0 < d < 2^12 without the integer-division expansion
__device__ __forceinline__ int wn_div(int e, int d, float rd) {
  int q = (int)((float)e * rd);
  const int r = e - q * d;
  q += (r >= d) ? 1 : 0;
  q -= (r < 0) ? 1 : 0;
  return q;
}

// (body as a function of its arguments, block index and grid size; conv_wino_mixed_pair_kernel below, r6)
template <int CW, int TW, bool PX4>
__device__ __forceinline__ void conv_wino_body(const ConvK& p, const WinoK& q, const int bid, const int nblk) {
  static_assert(CW * TW == 2, "two fragments (four waves) per block");
  extern __shared__ __attribute__((aligned(16))) float wn_lds[];
  constexpr int USLOT = CW * 2048;                              // floats per ring slot
  constexpr int NUI = 2 * CW;                                   // U DMA instructions (16 B per lane) per wave per chunk
  constexpr int NPI = WN_NPI(TW, PX4);                          // patch DMA instructions per wave per chunk
  constexpr int PSLOT = NPI * (PX4 ? 1024 : 256);
  constexpr int GRP = NUI + NPI;                                // DMA instructions per wave per group
  static_assert(3 * USLOT + 3 * PSLOT >= 4 * 2048, "the pair exchange reuses the rings");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fs = wave >> 1, xh = wave & 1;                      // fragment of the block, half of the transform domain
  const int cw = CW == 2 ? fs : 0, tw = TW == 2 ? fs : 0;
  const int half = lane >> 5, l32 = lane & 31;

  int lb = scf_xcd_remap(bid, nblk);
  const int mb = __builtin_amdgcn_readfirstlane(lb % q.mblocks);
  lb /= q.mblocks;
  const int xs = __builtin_amdgcn_readfirstlane(lb % q.sx);
  lb /= q.sx;
  const int ys = __builtin_amdgcn_readfirstlane(lb % q.sy);
  const int n = __builtin_amdgcn_readfirstlane(lb / q.sy);
  const int TXW = 1 << q.txl, TYW = 32 >> q.txl;
  const int y0 = ys * (2 * TW * TYW), x0 = xs * (2 * TXW);      // first output pixel of the block
  const int f0 = mb * CW;
  const int HW = p.H * p.W;

  float* Us = wn_lds;
  float* Ps = Us + 3 * USLOT;
  const unsigned u_lds = scf_lds_addr(Us), p_lds = scf_lds_addr(Ps);

  // ---- chunk-invariant DMA offsets --------------------------------------------------------------
  unsigned pvo[NPI];                            // patch: byte offset inside the chunk's 4 channel planes
  {
    // PX4: cells of 4 floats, rows start 4 columns left of the block (16-byte aligned: W % 4 == 0), the
    // windows then start at column 3; else single floats, rows start 1 column left
    const int NC = PX4 ? q.PWp >> 2 : q.PWp, PPC = q.PH * NC;        // cells per row / per plane
    const float rPPC = 1.0f / (float)PPC, rNC = 1.0f / (float)NC;
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
      const int e = i * 256 + tid;
      const int c = wn_div(e, PPC, rPPC), r = e - c * PPC;
      const int py = wn_div(r, NC, rNC), px = r - py * NC;
      const int iy = y0 - 1 + py, ix = PX4 ? x0 - 4 + 4 * px : x0 - 1 + px;
      const bool ok = c < WN_KC && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && (PX4 || px < q.PW);
      pvo[i] = ok ? (unsigned)((c * HW + iy * p.W + ix) * 4) : SCF_BUF_OOB;
    }
  }
  unsigned uvo[NUI], uld[NUI];                  // U: byte offset inside the chunk's slab / inside the ring slot
#pragma unroll
  for (int i = 0; i < NUI; ++i) {               // a fragment's 16 xi are 8 KB contiguous: 8 instructions of 1 KB
    const int j = wave + 4 * i;
    const int f = j >> 3, part = j & 7;
    uvo[i] = (unsigned)((f0 + f) * 8192 + part * 1024 + lane * 16);
    uld[i] = (unsigned)(f * 8192 + part * 1024);
  }
  const unsigned u_chunk_bytes = (unsigned)(q.F * 8192);
  const unsigned u_total = (unsigned)q.nchunk * u_chunk_bytes;

  // Both copy streams walk their chunks in order, so the buffer descriptors are kept in SGPRs and advanced
  // by a constant per chunk (a handful of scalar adds; rebuilding them from the chunk index is ~50 scalar
  // instructions and two branches per chunk, in front of the wave's next MFMA).  Chunks past the end get
  // num_records = 0: every lane is out of range and its cell is zeroed.
  scf_rsrc4 urs = scf_make_rsrc(q.wu, u_total);
  int u_left = (int)u_total;
  auto issue_u = [&](int slot) {                   // the next U chunk -> ring slot
    const unsigned dst = u_lds + (unsigned)(slot * USLOT * 4);
#pragma unroll
    for (int i = 0; i < NUI; ++i) scf_bdma_b128(urs, uvo[i], dst + uld[i]);
    const unsigned lo = (unsigned)urs[0] + u_chunk_bytes;
    urs[1] += lo < u_chunk_bytes ? 1 : 0;          // carry into base[47:32] (stride bits stay 0: the carry never gets there)
    urs[0] = (int)lo;
    u_left -= (int)u_chunk_bytes;
    urs[2] = u_left > 0 ? u_left : 0;
  };
  const unsigned p_chunk_bytes = (unsigned)(WN_KC * HW * 4);
  const float* p_seg = p.in0 + (long long)n * p.in0_ns;      // current input segment (in0, then in1)
  int p_left = p.C0;                                 // its channels still to copy
  bool p_second = p.in1 == nullptr;                  // no (further) segment behind this one
  scf_rsrc4 prs = scf_make_rsrc(p_seg, (unsigned)((p_left < WN_KC ? p_left : WN_KC) * HW * 4));
  auto issue_p = [&](int slot) {                   // the next patch chunk -> ring slot
    const unsigned dst = p_lds + (unsigned)((slot * PSLOT + wave * (PX4 ? 256 : 64)) * 4);
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
      if (PX4) scf_bdma_b128(prs, pvo[i], dst + (unsigned)(i * 4096));
      else scf_bdma_b32(prs, pvo[i], dst + (unsigned)(i * 1024));
    }
    p_left -= WN_KC;
    if (p_left <= 0 && !p_second) {                  // rare: on to the second segment (C0 % 4 == 0 there)
      p_second = true;
      p_seg = p.in1 + (long long)n * p.in1_ns;
      p_left = p.Cin - p.C0;
      prs = scf_make_rsrc(p_seg, 0u);
    } else {
      const unsigned lo = (unsigned)prs[0] + p_chunk_bytes;
      prs[1] += lo < p_chunk_bytes ? 1 : 0;
      prs[0] = (int)lo;
    }
    const int cl = p_left < WN_KC ? p_left : WN_KC;
    prs[2] = cl > 0 ? cl * HW * 4 : 0;
  };

  // ---- input transform in registers: lane (tile l32, k-half) turns the 4 x 4 windows of channels
  //      half and 2 + half into the B operands of its wave's two rows of the transform domain.
  //      Rows (0, 1) of B^T d need input rows 0-2, rows (2, 3) need 1-3: three rows from row xh on ----
  const int ty = tw * TYW + (l32 >> q.txl), tx = l32 & (TXW - 1);
  // Vector-ALU instructions of a wave run on the same SIMD as the MFMAs of both co-resident waves and do
  // NOT hide behind them: every one costs matrix-pipe time.  So the loop keeps them to the transform's own
  // adds: (a) the six row addresses of the two windows are absolute LDS addresses computed once (one add of
  // the slot offset per row and chunk); (b) both halves of the transform domain run the SAME code: rows
  // (0, 1) take input rows (d0, d1, d2) as (e0, e1, e2) and compute e0 - e2, e1 + e2; rows (3, 2) take
  // (d3, d2, d1) and compute e0 - e2 = -(row 3), e1 - e2 = row 2 -- one fma with sigma = +-1, the sign of row
  // 3 is undone in the output transform and the packing stores U's rows in the order 0, 1, 3, 2; (c) the
  // operand double buffer is a 2x unrolled loop, not 32 moves.
  const float sigma = xh == 0 ? 1.0f : -1.0f;
  unsigned prow[2][3];                              // absolute LDS byte address of row e_i, slot 0
  {
    const int poff = half * q.PPL + 2 * ty * q.PWp + 2 * tx + (PX4 ? 3 : 0);      // channel `half`, input row d0
#pragma unroll
    for (int sI = 0; sI < 2; ++sI)
#pragma unroll
      for (int i = 0; i < 3; ++i)
        prow[sI][i] = p_lds + (unsigned)((poff + 2 * sI * q.PPL + (xh == 0 ? i : 3 - i) * q.PWp) * 4);
  }
  // Packed fp32 adds on the register pairs the LDS reads return (columns 0-1, 2-3 of a window row): 16
  // instructions per chunk instead of 32.  Written as asm because the op_sel / neg forms of the column
  // stage are not something hipcc derives (it pairs operands up with moves instead).
  wn_f32x2 dws[2][3][2];
  auto win_load = [&](unsigned slot_bytes, int sI) {                  // channel 2 sI + half
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const __attribute__((address_space(3))) float* r =
          (const __attribute__((address_space(3))) float*)(uintptr_t)(prow[sI][i] + slot_bytes);
      dws[sI][i][0] = wn_f32x2{r[0], r[1]};
      dws[sI][i][1] = wn_f32x2{r[2], r[3]};
    }
  };
  const wn_f32x2 sigma2 = {sigma, sigma};
  auto win_transform = [&](wn_f32x2 (&bo)[8], int sI) {                // -> bo[4 il + j][sI]
    const wn_f32x2 (&e)[3][2] = dws[sI];
    wn_f32x2 w[2][2];                  // [il][column pair]
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(w[0][h]) : "v"(e[0][h]), "v"(e[2][h]));   // e0 - e2
      asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(w[1][h]) : "v"(sigma2), "v"(e[2][h]), "v"(e[1][h]));             // e1 + sigma e2
    }
#pragma unroll
    for (int il = 0; il < 2; ++il) {
      wn_f32x2 v01, v23;
      // (w0 - w2, w1 + w2): both halves take the low half of (w2, w3), negated for the low result
      asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(v01) : "v"(w[il][0]), "v"(w[il][1]));
      // (w2 - w1, w1 - w3): low = w2 - w1, high = -w3 + w1
      asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(v23) : "v"(w[il][1]), "v"(w[il][0]));
      bo[4 * il + 0][sI] = v01[0];
      bo[4 * il + 1][sI] = v01[1];
      bo[4 * il + 2][sI] = v23[0];
      bo[4 * il + 3][sI] = v23[1];
    }
  };

  wn_f32x16 acc[8];
#pragma unroll
  for (int x = 0; x < 8; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;

  // ---- prologue: patch[0], U[0]; (U[1], patch[1]); (U[2], patch[2]) ---------------------------------
  issue_p(0); issue_u(0);
  issue_u(1); issue_p(1);
  issue_u(2); issue_p(2);
  scf_wait_vmcnt_imm<GRP>();           // all but the last group have landed
  __syncthreads();
  const float* ua = Us + cw * 2048 + xh * 1024 + lane * 2;          // + xi_local * 128
  wn_f32x2 a0[8], b0[8], a1[8], b1[8];
#pragma unroll
  for (int x = 0; x < 8; ++x) a0[x] = *reinterpret_cast<const wn_f32x2*>(ua + x * 128);
  win_load(0u, 0); win_transform(b0, 0);
  win_load(0u, 1); win_transform(b0, 1);
  __syncthreads();                     // slot 0 of both rings is free again

  // chunk c: the operands of chunk c are in registers (a, b); those of chunk c + 1 are read (U) / computed
  // (patch) into (an, bn) under its MFMAs; the copies of chunk c + 3 are issued, those of chunk c + 2 must have
  // landed at its end.  The MFMAs go first; everything else is placed between them in the order its results
  // are needed.
  int s1 = 1;                          // ring slot of U[c + 1], patch[c + 1]
  auto chunk = [&](int c, const wn_f32x2 (&a)[8], const wn_f32x2 (&b)[8], wn_f32x2 (&an)[8], wn_f32x2 (&bn)[8]) {
    const float* uc = ua + s1 * USLOT;               // past the last chunk: zeros, results unused
    const unsigned pcb = (unsigned)(s1 * PSLOT * 4);
#define WN_M(X, S)                                                                                    \
    if (!WN_LAB(0)) acc[X] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[X][S], b[X][S], acc[X], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);
    WN_M(0, 0)
    if (!WN_LAB(1)) win_load(pcb, 0);
    __builtin_amdgcn_sched_barrier(0);
    WN_M(1, 0)
    if (!WN_LAB(1)) win_load(pcb, 1);
    __builtin_amdgcn_sched_barrier(0);
    WN_M(2, 0)
    int s3 = s1 + 2;                                 // (c + 3) % 3
    s3 = s3 >= 3 ? s3 - 3 : s3;
    if (!WN_LAB(2)) issue_u(s3);
    __builtin_amdgcn_sched_barrier(0);
    WN_M(3, 0)
    if (!WN_LAB(2)) issue_p(s3);
    __builtin_amdgcn_sched_barrier(0);
    WN_M(4, 0)
    if (!WN_LAB(5)) {
#pragma unroll
      for (int x = 0; x < 8; ++x) an[x] = *reinterpret_cast<const wn_f32x2*>(uc + x * 128);
    }
    __builtin_amdgcn_sched_barrier(0);
    WN_M(5, 0)
    if (!WN_LAB(1)) win_transform(bn, 0);
    __builtin_amdgcn_sched_barrier(0);
    WN_M(6, 0) WN_M(7, 0)
    WN_M(0, 1) WN_M(1, 1)
    if (!WN_LAB(1)) win_transform(bn, 1);
    __builtin_amdgcn_sched_barrier(0);
    WN_M(2, 1) WN_M(3, 1) WN_M(4, 1) WN_M(5, 1) WN_M(6, 1) WN_M(7, 1)
#undef WN_M
    scf_wait_vmcnt_imm<GRP>();
    if (!WN_LAB(4)) __syncthreads();
    s1 = s1 == 2 ? 0 : s1 + 1;
  };
  int c = 0;
  for (; c + 1 < q.nchunk; c += 2) {
    chunk(c, a0, b0, a1, b1);
    chunk(c + 1, a1, b1, a0, b0);
  }
  if (c < q.nchunk) chunk(c, a0, b0, a1, b1);
  scf_wait_vmcnt_imm<0>();             // the zero-filled groups past the end
  __syncthreads();                     // every wave's copies have landed: the rings are free for the exchange

  // ---- output transform of this wave's two rows of the transform domain -----------------------------
  // rows (0, 1): A^T contributes (M0 + M1, M1); rows (3, 2), row 3 negated (above): (M2, -M2 - M3) =
  // (acc[4 + j], acc[j] - acc[4 + j]); then the column transform
  float* xw = wn_lds + wave * 2048;             // this wave's outbox: [8 channel rows][64 lanes][4]
  const float* xr = wn_lds + (wave ^ 1) * 2048; // the partner's
  wn_f32x4 own[8];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float t0[4], t1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (xh == 0) { t0[j] = acc[j][r] + acc[4 + j][r]; t1[j] = acc[4 + j][r]; }
      else { t0[j] = acc[4 + j][r]; t1[j] = acc[j][r] - acc[4 + j][r]; }
    }
    wn_f32x4 y;
    y[0] = (t0[0] + t0[1]) + t0[2];
    y[1] = (t0[1] - t0[2]) - t0[3];
    y[2] = (t1[0] + t1[1]) + t1[2];
    y[3] = (t1[1] - t1[2]) - t1[3];
    // this wave finishes channel rows r in [8 xh, 8 xh + 8); the others go to the partner
    if ((r >> 3) == xh) own[r & 7] = y;
    else *reinterpret_cast<wn_f32x4*>(xw + ((r & 7) * 64 + lane) * 4) = y;
  }
  __syncthreads();

  const ConvEpi e = scf_conv_epi(p, n);
  const int oy = y0 + 2 * ty, ox = x0 + 2 * tx;
  if (ox >= p.Wo || oy >= p.Ho || WN_LAB(3)) return;
  const bool row1 = oy + 1 < p.Ho;
  const bool relu = p.act == SCF_ACT_RELU;
  const int cb = (f0 + cw) * 32 + 4 * half + 16 * xh;
  // every load of the epilogue (partner's partial sums, bias / BN constants, residual) is issued before the
  // first store: loads and stores share the in-order vmcnt counter, so a load behind a store waits for the
  // store's acknowledgement -- one memory round trip per channel row otherwise
  wn_f32x4 o[8];
  float bv[8], sc[8], sh[8];
  wn_f32x2 rr[8][2];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int co = cb + 8 * (r >> 2) + (r & 3);
    const int cc = co < p.Cout ? co : 0;
    o[r] = *reinterpret_cast<const wn_f32x4*>(xr + (r * 64 + lane) * 4);
    bv[r] = p.bias ? p.bias[cc] : 0.f;
    sc[r] = p.scale ? p.scale[cc] : 1.f;
    sh[r] = p.scale ? p.shift[cc] : 0.f;
    const int off = cc * e.HWo + oy * p.Wo + ox;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      rr[r][i] = (e.res && (i == 0 || row1)) ? *reinterpret_cast<const wn_f32x2*>(e.res + off + i * p.Wo) : wn_f32x2{0.f, 0.f};
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int co = cb + 8 * (r >> 2) + (r & 3);
    if (co >= p.Cout) continue;
    // rows (0, 1) + rows (2, 3), the same order in both waves
    const wn_f32x4 y = xh == 0 ? own[r] + o[r] : o[r] + own[r];
    const int off = co * e.HWo + oy * p.Wo + ox;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i == 1 && !row1) break;
      wn_f32x2 v = {y[2 * i] + bv[r], y[2 * i + 1] + bv[r]};
      if (p.scale) { v[0] = v[0] * sc[r] + sh[r]; v[1] = v[1] * sc[r] + sh[r]; }
      v[0] += rr[r][i][0]; v[1] += rr[r][i][1];
      if (relu) { v[0] = v[0] > 0.f ? v[0] : 0.f; v[1] = v[1] > 0.f ? v[1] : 0.f; }
      scf_store2<(SCF_ST_SC1 & 8) != 0>(e.out + off + i * p.Wo, v[0], v[1]);
    }
  }
}

template <int CW, int TW, bool PX4>
__global__ __launch_bounds__(256, 2)
void conv_wino_kernel(ConvK p, WinoK q) {
  conv_wino_body<CW, TW, PX4>(p, q, (int)blockIdx.x, (int)gridDim.x);
}

// ===================================================================================================
// Quarter-domain kernel (r4): one computed B operand feeds TWO output-channel fragments.
//
// In the pair kernel above every MFMA has its own B operand: a wave holds 8 transform positions of ONE channel
// fragment, so the input transform (16 packed adds per chunk) is paid once per fragment -- a 128 -> 512 layer
// transforms every patch 16 times -- and vector-ALU instructions cost matrix-pipe time here (see above).  In this
// kernel a wave holds ONE ROW of the transform domain (4 positions) for TWO channel fragments = the same 128
// accumulator registers: 16 MFMAs per chunk again, but only 8 packed instructions of transform (one fma per
// column pair for the row stage, two adds for the column stage, per channel) and 8 instead of 12 window reads.
//   block   4 TW waves: wave = (tile group tw, storage row of the transform domain); 2 channel fragments x TW
//           tile groups.  TW = 1: 256 threads, two blocks per CU (as above).  TW = 2: 512 threads, one block per
//           CU, the U chunk (16 KB) serves both tile groups: least L2 -> LDS traffic per MFMA.
//   rows    storage rows 0, 1, 2, 3 = transform rows 0, 1, 3, 2 (the packing's order): B^T d row = e_a + sigma e_b
//           with (a, b, sigma) = (0, 2, -), (1, 2, +), (1, 3, -), (2, 1, -): one packed fma per column pair.
//   output  A^T M A: the column transform is lane-local (2 values per accumulator row), the row sums need all four
//           waves: each wave finishes 8 of the 32 (fragment, row) combinations of its tile group and passes the
//           other 24 pairs through LDS (12 b128 writes, 12 reads); sums in the fixed order (t0 + t1) + t2,
//           (t1 - t2) - t3.
// Needs an even number of channel fragments (the dispatch keeps other layers on the pair kernel).
// ===================================================================================================
#define WQ_NPI(TW, PX4) ((PX4) ? 1 : ((TW) == 2 ? 3 : 4))

// NR = slots of the U / patch rings (r5).  3: chunk c + 3 is requested while chunk c is on the matrix cores and has to
// have landed one chunk later.  4 (80 KB of LDS: still two blocks per CU): one more chunk of latency tolerance -- a
// block whose CU partner is in its prologue / epilogue then keeps its chunk rate instead of waiting on its copies.
// BURST (r5, NR = 4): a chunk is ONE run of everything that is not an MFMA -- the transform of the next chunk's
// windows (read one chunk earlier), the window reads of the chunk after that, the U reads, the copies -- followed by its
// 16 MFMAs back to back, instead of seven gaps between MFMAs.  tools/lab/coissue.hip (profiles/r5_coissue_burst_and_
// priorities.txt): vector-ALU / LDS / copy instructions never overlap the MFMAs of their SIMD (4.2 / 3.3 / 10 clk each on
// top of 64 per MFMA, with any wave priorities), and every switch between the two kinds costs ~20 clk more.
// (the kernel's body as a function of its arguments, block index and grid size: conv_wino_q_pair_kernel below runs two layers' grids
// in one launch, r6)
template <int TW, bool PX4, int NR = 3, bool BURST = false>
__device__ __forceinline__ void conv_wino_q_body(const ConvK& p, const WinoK& q, const int bid, const int nblk) {
  static_assert(!BURST || NR == 4, "the burst form reads two chunks ahead");
  extern __shared__ __attribute__((aligned(16))) float wn_lds[];
  constexpr int NW = 4 * TW, NT = 64 * NW;                      // waves, threads
  constexpr int CW = 2;
  constexpr int USLOT = CW * 2048;                              // floats per ring slot
  constexpr int NUI = 16 / NW;                                  // U copy instructions (1 KB each) per wave per chunk
  constexpr int NPI = WQ_NPI(TW, PX4);
  constexpr int PSLOT = NPI * NT * (PX4 ? 4 : 1);
  constexpr int GRP = NUI + NPI;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tw = TW == 2 ? wave >> 2 : 0, row = wave & 3;
  const int half = lane >> 5, l32 = lane & 31;

  int lb = scf_xcd_remap(bid, nblk);
  const int mb = __builtin_amdgcn_readfirstlane(lb % q.mblocks);
  lb /= q.mblocks;
  const int xs = __builtin_amdgcn_readfirstlane(lb % q.sx);
  lb /= q.sx;
  const int ys = __builtin_amdgcn_readfirstlane(lb % q.sy);
  const int n = __builtin_amdgcn_readfirstlane(lb / q.sy);
  const int TXW = 1 << q.txl, TYW = 32 >> q.txl;
  const int y0 = ys * (2 * TW * TYW), x0 = xs * (2 * TXW);
  const int f0 = mb * CW;
  const int HW = p.H * p.W;

  float* Us = wn_lds;
  float* Ps = Us + NR * USLOT;
  const unsigned u_lds = scf_lds_addr(Us), p_lds = scf_lds_addr(Ps);

  // ---- chunk-invariant copy offsets (as in the pair kernel) ----
  unsigned pvo[NPI];
  {
    const int NC = PX4 ? q.PWp >> 2 : q.PWp, PPC = q.PH * NC;
    const float rPPC = 1.0f / (float)PPC, rNC = 1.0f / (float)NC;
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
      const int e = i * NT + tid;
      const int c = wn_div(e, PPC, rPPC), r = e - c * PPC;
      const int py = wn_div(r, NC, rNC), px = r - py * NC;
      const int iy = y0 - 1 + py, ix = PX4 ? x0 - 4 + 4 * px : x0 - 1 + px;
      const bool ok = c < WN_KC && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && (PX4 || px < q.PW);
      pvo[i] = ok ? (unsigned)((c * HW + iy * p.W + ix) * 4) : SCF_BUF_OOB;
    }
  }
  unsigned uvo[NUI], uld[NUI];
#pragma unroll
  for (int i = 0; i < NUI; ++i) {               // the block's two fragments: 2 x 8 KB per chunk, 16 pieces of 1 KB
    const int j = wave + NW * i;
    const int f = j >> 3, part = j & 7;
    uvo[i] = (unsigned)((f0 + f) * 8192 + part * 1024 + lane * 16);
    uld[i] = (unsigned)(f * 8192 + part * 1024);
  }
  const unsigned u_chunk_bytes = (unsigned)(q.F * 8192);
  const unsigned u_total = (unsigned)q.nchunk * u_chunk_bytes;
  scf_rsrc4 urs = scf_make_rsrc(q.wu, u_total);
  int u_left = (int)u_total;
  auto issue_u = [&](int slot) {
    const unsigned dst = u_lds + (unsigned)(slot * USLOT * 4);
#pragma unroll
    for (int i = 0; i < NUI; ++i) scf_bdma_b128(urs, uvo[i], dst + uld[i]);
    const unsigned lo = (unsigned)urs[0] + u_chunk_bytes;
    urs[1] += lo < u_chunk_bytes ? 1 : 0;
    urs[0] = (int)lo;
    u_left -= (int)u_chunk_bytes;
    urs[2] = u_left > 0 ? u_left : 0;
  };
  const unsigned p_chunk_bytes = (unsigned)(WN_KC * HW * 4);
  const float* p_seg = p.in0 + (long long)n * p.in0_ns;
  int p_left = p.C0;
  bool p_second = p.in1 == nullptr;
  scf_rsrc4 prs = scf_make_rsrc(p_seg, (unsigned)((p_left < WN_KC ? p_left : WN_KC) * HW * 4));
  auto issue_p = [&](int slot) {
    const unsigned dst = p_lds + (unsigned)((slot * PSLOT + wave * (PX4 ? 256 : 64)) * 4);
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
      if (PX4) scf_bdma_b128(prs, pvo[i], dst + (unsigned)(i * NT * 16));
      else scf_bdma_b32(prs, pvo[i], dst + (unsigned)(i * NT * 4));
    }
    p_left -= WN_KC;
    if (p_left <= 0 && !p_second) {
      p_second = true;
      p_seg = p.in1 + (long long)n * p.in1_ns;
      p_left = p.Cin - p.C0;
      prs = scf_make_rsrc(p_seg, 0u);
    } else {
      const unsigned lo = (unsigned)prs[0] + p_chunk_bytes;
      prs[1] += lo < p_chunk_bytes ? 1 : 0;
      prs[0] = (int)lo;
    }
    const int cl = p_left < WN_KC ? p_left : WN_KC;
    prs[2] = cl > 0 ? cl * HW * 4 : 0;
  };

  // ---- input transform: lane (tile l32, k-half) computes ITS ROW of B^T d B for channels half and 2 + half ----
  const int ty = tw * TYW + (l32 >> q.txl), tx = l32 & (TXW - 1);
  const int ra = row == 0 ? 0 : row == 3 ? 2 : 1, rb = row == 2 ? 3 : row == 3 ? 1 : 2;
  const float sigma = row == 1 ? 1.0f : -1.0f;
  const wn_f32x2 sigma2 = {sigma, sigma};
  unsigned prow[2][2];                              // absolute LDS byte address of input rows a, b; slot 0
  {
    const int poff = half * q.PPL + 2 * ty * q.PWp + 2 * tx + (PX4 ? 3 : 0);
#pragma unroll
    for (int sI = 0; sI < 2; ++sI) {
      prow[sI][0] = p_lds + (unsigned)((poff + 2 * sI * q.PPL + ra * q.PWp) * 4);
      prow[sI][1] = p_lds + (unsigned)((poff + 2 * sI * q.PPL + rb * q.PWp) * 4);
    }
  }
  wn_f32x2 dws[2][2][2];                            // [channel][row a / b][column pair]
  auto win_load = [&](unsigned slot_bytes, int sI) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const __attribute__((address_space(3))) float* r =
          (const __attribute__((address_space(3))) float*)(uintptr_t)(prow[sI][i] + slot_bytes);
      dws[sI][i][0] = wn_f32x2{r[0], r[1]};
      dws[sI][i][1] = wn_f32x2{r[2], r[3]};
    }
  };
  auto win_transform = [&](wn_f32x2 (&bo)[2][2], int sI) {       // -> bo[sI] = {(j0, j1), (j2, j3)}
    wn_f32x2 v01, v23;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(v01) : "v"(sigma2), "v"(dws[sI][1][0]), "v"(dws[sI][0][0]));     // e_a + sigma e_b
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(v23) : "v"(sigma2), "v"(dws[sI][1][1]), "v"(dws[sI][0][1]));
    // (v0 - v2, v1 + v2) and (v2 - v1, v1 - v3): the column stage of the pair kernel
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(bo[sI][0]) : "v"(v01), "v"(v23));
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(bo[sI][1]) : "v"(v23), "v"(v01));
  };

  wn_f32x16 acc[8];                                 // [fragment][column j]
#pragma unroll
  for (int x = 0; x < 8; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;

  issue_p(0); issue_u(0);
  issue_u(1); issue_p(1);
  issue_u(2); issue_p(2);
  if (NR == 4) { issue_u(3); issue_p(3); }
  if (BURST) scf_wait_vmcnt_imm<GRP>();               // chunks 0, 1 and 2 have landed
  else scf_wait_vmcnt_imm<(NR - 2) * GRP>();          // chunks 0 and 1 have landed
  __syncthreads();
  const float* ua = Us + row * 512 + lane * 2;      // + fragment * 2048 + j * 128
  wn_f32x2 a0[8], a1[8], b0[2][2], b1[2][2];
#pragma unroll
  for (int x = 0; x < 8; ++x) a0[x] = *reinterpret_cast<const wn_f32x2*>(ua + (x >> 2) * 2048 + (x & 3) * 128);
  win_load(0u, 0); win_transform(b0, 0);
  win_load(0u, 1); win_transform(b0, 1);
  if (BURST) {                                       // the raw windows of chunk 1 wait in registers for the first burst
    win_load((unsigned)(PSLOT * 4), 0);
    win_load((unsigned)(PSLOT * 4), 1);
  }
  __syncthreads();

  int s1 = 1;
  auto chunk = [&](const wn_f32x2 (&a)[8], const wn_f32x2 (&b)[2][2], wn_f32x2 (&an)[8], wn_f32x2 (&bn)[2][2]) {
    const float* uc = ua + s1 * USLOT;
    const unsigned pcb = (unsigned)(s1 * PSLOT * 4);
#define WQ_M(X, S)                                                                                             \
    if (!WN_LAB(0)) acc[X] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[X][S], b[S][((X) & 3) >> 1][(X) & 1], acc[X], 0, 0, 0);    \
    __builtin_amdgcn_sched_barrier(0);
    int s3 = s1 + NR - 1;                            // the slot of the chunk in the registers: free
    s3 = s3 >= NR ? s3 - NR : s3;
    if (BURST) {
      int s2 = s1 + 1;                               // the slot of the chunk after the next: its windows are read now
      s2 = s2 >= NR ? s2 - NR : s2;
      win_transform(bn, 0);
      win_transform(bn, 1);
      __builtin_amdgcn_sched_barrier(0);
      win_load((unsigned)(s2 * PSLOT * 4), 0);
      win_load((unsigned)(s2 * PSLOT * 4), 1);
#pragma unroll
      for (int x = 0; x < 8; ++x) an[x] = *reinterpret_cast<const wn_f32x2*>(uc + (x >> 2) * 2048 + (x & 3) * 128);
      __builtin_amdgcn_sched_barrier(0);
      issue_u(s3);
      issue_p(s3);
      __builtin_amdgcn_sched_barrier(0);
      WQ_M(0, 0) WQ_M(1, 0) WQ_M(2, 0) WQ_M(3, 0) WQ_M(4, 0) WQ_M(5, 0) WQ_M(6, 0) WQ_M(7, 0)
      WQ_M(0, 1) WQ_M(1, 1) WQ_M(2, 1) WQ_M(3, 1) WQ_M(4, 1) WQ_M(5, 1) WQ_M(6, 1) WQ_M(7, 1)
      scf_wait_vmcnt_imm<GRP>();                     // all but the newest chunk have landed
      __syncthreads();
      s1 = s1 == NR - 1 ? 0 : s1 + 1;
      return;
    }
    // (measured, r4: the same work as groups of 2-2-4-8 MFMAs with the non-MFMA instructions in three gaps -- what
    // tools/lab/coissue.hip suggests -- runs within +-2 % of this placement on every layer shape; not kept)
    WQ_M(0, 0)
    if (!WN_LAB(1)) win_load(pcb, 0);
    __builtin_amdgcn_sched_barrier(0);
    WQ_M(1, 0)
    if (!WN_LAB(1)) win_load(pcb, 1);
    __builtin_amdgcn_sched_barrier(0);
    WQ_M(2, 0)
    if (!WN_LAB(2)) issue_u(s3);
    __builtin_amdgcn_sched_barrier(0);
    WQ_M(3, 0)
    if (!WN_LAB(2)) issue_p(s3);
    __builtin_amdgcn_sched_barrier(0);
    WQ_M(4, 0)
    if (!WN_LAB(5)) {
#pragma unroll
      for (int x = 0; x < 8; ++x) an[x] = *reinterpret_cast<const wn_f32x2*>(uc + (x >> 2) * 2048 + (x & 3) * 128);
    }
    __builtin_amdgcn_sched_barrier(0);
    WQ_M(5, 0)
    if (!WN_LAB(1)) win_transform(bn, 0);
    __builtin_amdgcn_sched_barrier(0);
    WQ_M(6, 0) WQ_M(7, 0)
    WQ_M(0, 1) WQ_M(1, 1)
    if (!WN_LAB(1)) win_transform(bn, 1);
    __builtin_amdgcn_sched_barrier(0);
    WQ_M(2, 1) WQ_M(3, 1) WQ_M(4, 1) WQ_M(5, 1) WQ_M(6, 1) WQ_M(7, 1)
#undef WQ_M
    scf_wait_vmcnt_imm<(NR - 2) * GRP>();            // all but the newest NR - 2 chunks have landed: the next one to be read has
    if (!WN_LAB(4)) __syncthreads();
    s1 = s1 == NR - 1 ? 0 : s1 + 1;
  };
  int c = 0;
  for (; c + 1 < q.nchunk; c += 2) {
    chunk(a0, b0, a1, b1);
    chunk(a1, b1, a0, b0);
  }
  if (c < q.nchunk) chunk(a0, b0, a1, b1);
  scf_wait_vmcnt_imm<0>();
  __syncthreads();                     // the rings are free for the exchange

  // ---- output transform.  Column stage (lane-local): t = ((m0 + m1) + m2, (m1 - m2) - m3) per (fragment, row r).
  //      Combination q = 16 f + r is finished by wave q >> 3 of the tile group; the others' pairs go through LDS:
  //      outbox[source wave][slot of the owner: 3][pair of combinations: 4][lane][4 floats] ----
  float* box = wn_lds + tw * (4 * 3072);
  // the per-channel constants of the epilogue are requested HERE: their round trip runs under the output transform
  // and the exchange instead of after them
  const int cb = (f0 + (row >> 1)) * 32 + 16 * (row & 1) + 4 * half;
  float bv[8], sc[8], sh[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int co = cb + 8 * (k >> 2) + (k & 3);
    const int cc = co < p.Cout ? co : 0;
    bv[k] = p.bias ? p.bias[cc] : 0.f;
    sc[k] = p.scale ? p.scale[cc] : 1.f;
    sh[k] = p.scale ? p.shift[cc] : 0.f;
  }
  __builtin_amdgcn_sched_barrier(0);
  wn_f32x4 own[4];
#pragma unroll
  for (int qq = 0; qq < 32; qq += 2) {
    wn_f32x4 t;
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int f = (qq + h2) >> 4, r = (qq + h2) & 15;
      const float m0 = acc[4 * f][r], m1 = acc[4 * f + 1][r], m2 = acc[4 * f + 2][r], m3 = acc[4 * f + 3][r];
      t[2 * h2] = (m0 + m1) + m2;
      t[2 * h2 + 1] = (m1 - m2) - m3;
    }
    const int d = qq >> 3, kp = (qq & 7) >> 1;          // owner wave, pair index
    if (d == row) own[kp] = t;
    else *reinterpret_cast<wn_f32x4*>(box + (((row * 3 + (d < row ? d : d - 1)) * 4 + kp) * 64 + lane) * 4) = t;
  }
  __syncthreads();

  const ConvEpi e = scf_conv_epi(p, n);
  const int oy = y0 + 2 * ty, ox = x0 + 2 * tx;
  if (ox >= p.Wo || oy >= p.Ho || WN_LAB(3)) return;
  const bool row1 = oy + 1 < p.Ho;
  const bool relu = p.act == SCF_ACT_RELU;
  // every load of the epilogue is issued before the first store (loads and stores share the in-order vmcnt)
  wn_f32x4 in[4][4];                                    // [storage row of the source][pair]
#pragma unroll
  for (int sw = 0; sw < 4; ++sw)
#pragma unroll
    for (int kp = 0; kp < 4; ++kp) {
      if (sw == row) in[sw][kp] = own[kp];
      else in[sw][kp] = *reinterpret_cast<const wn_f32x4*>(box + (((sw * 3 + (row < sw ? row : row - 1)) * 4 + kp) * 64 + lane) * 4);
    }
  wn_f32x2 rr[8][2];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int co = cb + 8 * (k >> 2) + (k & 3);
    const int cc = co < p.Cout ? co : 0;
    const int off = cc * e.HWo + oy * p.Wo + ox;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      rr[k][i] = (e.res && (i == 0 || row1)) ? *reinterpret_cast<const wn_f32x2*>(e.res + off + i * p.Wo) : wn_f32x2{0.f, 0.f};
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int co = cb + 8 * (k >> 2) + (k & 3);
    if (co >= p.Cout) continue;
    const int kp = k >> 1, h2 = 2 * (k & 1);
    // storage rows 0, 1, 2, 3 hold transform rows 0, 1, 3, 2:  Y0 = (t0 + t1) + t2,  Y1 = (t1 - t2) - t3
    wn_f32x2 y[2];
#pragma unroll
    for (int cI = 0; cI < 2; ++cI) {
      const float t0 = in[0][kp][h2 + cI], t1 = in[1][kp][h2 + cI], t3 = in[2][kp][h2 + cI], t2 = in[3][kp][h2 + cI];
      y[0][cI] = (t0 + t1) + t2;
      y[1][cI] = (t1 - t2) - t3;
    }
    const int off = co * e.HWo + oy * p.Wo + ox;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i == 1 && !row1) break;
      wn_f32x2 v = {y[i][0] + bv[k], y[i][1] + bv[k]};
      if (p.scale) { v[0] = v[0] * sc[k] + sh[k]; v[1] = v[1] * sc[k] + sh[k]; }
      v[0] += rr[k][i][0]; v[1] += rr[k][i][1];
      if (relu) { v[0] = v[0] > 0.f ? v[0] : 0.f; v[1] = v[1] > 0.f ? v[1] : 0.f; }
      scf_store2<(SCF_ST_SC1 & 8) != 0>(e.out + off + i * p.Wo, v[0], v[1]);
    }
  }
}

template <int TW, bool PX4, int NR = 3, bool BURST = false>
__global__ __launch_bounds__(256 * TW, TW == 1 ? 2 : 1)
void conv_wino_q_kernel(ConvK p, WinoK q) {
  conv_wino_q_body<TW, PX4, NR, BURST>(p, q, (int)blockIdx.x, (int)gridDim.x);
}

// r6: two independent layers' grids in one launch (blocks [0, nba) = layer a, the rest layer b): at batch 1-4 the context
// encoder's 64 -> 64 layers (128 blocks per image) ride in the feature encoder's launches (256 blocks for a pair of images)
// instead of running after them or on a second stream (see conv_dma_pair_kernel)
template <bool PX4>
__global__ __launch_bounds__(256, 2)
void conv_wino_q_pair_kernel(ConvK pa, WinoK qa, ConvK pb, WinoK qb, int nba) {
  if ((int)blockIdx.x < nba) conv_wino_q_body<1, PX4>(pa, qa, (int)blockIdx.x, nba);
  else conv_wino_q_body<1, PX4>(pb, qb, (int)blockIdx.x - nba, (int)gridDim.x - nba);
}

// r6: a quarter-domain layer (even fragment count) beside a PAIR-kernel layer (one fragment): the delta-flow encoder's 128 -> 64 and the
// mask encoder's 64 -> 32 at batch 32 are 256 + 128 blocks = half a round of resident blocks each
template <bool PXA, bool PXB>
__global__ __launch_bounds__(256, 2)
void conv_wino_mixed_pair_kernel(ConvK pa, WinoK qa, ConvK pb, WinoK qb, int nba) {
  if ((int)blockIdx.x < nba) conv_wino_q_body<1, PXA>(pa, qa, (int)blockIdx.x, nba);
  else conv_wino_body<1, 2, PXB>(pb, qb, (int)blockIdx.x - nba, (int)gridDim.x - nba);
}

// ---------------------------------------------------------------------------------------------------
// Host side: packing and launch
// ---------------------------------------------------------------------------------------------------
extern "C" int64_t scf_pack_conv_weight_wino_size(int32_t cout, int32_t cin) {
  if (cout <= 0 || cin <= 0) return 0;
  const int64_t F = (cout + 31) / 32, nchunk = (cin + WN_KC - 1) / WN_KC;
  return nchunk * 16 * F * 128;
}

extern "C" int scf_pack_conv_weight_wino(const float* w, int32_t cout, int32_t cin, float* out) {
  if (!w || !out || cout <= 0 || cin <= 0) return SCF_EINVAL;
  const int F = (cout + 31) / 32;
  memset(out, 0, sizeof(float) * (size_t)scf_pack_conv_weight_wino_size(cout, cin));
  static const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci) {
      const float* g = w + ((size_t)co * cin + ci) * 9;
      double t[4][3];
      for (int i = 0; i < 4; ++i)
        for (int b = 0; b < 3; ++b) t[i][b] = G[i][0] * g[b] + G[i][1] * g[3 + b] + G[i][2] * g[6 + b];
      const int chunk = ci / WN_KC, cl = ci % WN_KC, s = cl >> 1, kh = cl & 1;
      const int frag = co / 32, m = co % 32;
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
          const double u = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
          const int pos = 4 * (i < 2 ? i : 5 - i) + j;        // rows of the transform domain stored in the order 0, 1, 3, 2
          out[(((size_t)chunk * F + frag) * 16 + pos) * 128 + kh * 64 + m * 2 + s] = (float)u;
        }
    }
  return SCF_OK;
}

// measurement knob (scflow_hip_prof.h: scf_tune(SCF_TUNE_WINO_VARIANT, v)): 0 = the dispatch's own choice,
// 1 = pair kernel, 2 = quarter-domain kernel with 4 waves, 3 = quarter-domain kernel with 8 waves
static std::atomic<int> g_wino_variant{0};
int scf_wino_variant_set(int v) {
#ifndef SCF_WINO_LAB
  if (v < 0 || v > 2) return SCF_EINVAL;      // 3 / 4 / 5 exist in -DSCF_WINO_LAB builds only: refuse instead of silently running 2
#else
  if (v < 0 || v > 5) return SCF_EINVAL;
#endif
  return g_wino_variant.exchange(v);
}

// Tile selection + launch; SCF_EUNSUPPORTED -> the caller goes on to the direct kernels.
// info (optional): {16 transform positions, fragments per block, blocks, LDS bytes}.
int scf_conv_wino_dispatch(ConvK k, const float* wu, int N, bool dry_run, int* info, hipStream_t st, int* which, ScfLaunchCap* cap) {
  if (which) *which = 0;      // 1: the quarter-domain kernel took the launch
  if (!wu || k.KH != 3 || k.KW != 3 || k.stride != 1 || k.pad_h != 1 || k.pad_w != 1) return SCF_EUNSUPPORTED;
  if (k.w_ns != 0 || k.out_tile || k.mode != SCF_CONV_PLAIN || k.out_div != 1.0f || k.act_split > 0) return SCF_EUNSUPPORTED;
  if (k.act != SCF_ACT_NONE && k.act != SCF_ACT_RELU) return SCF_EUNSUPPORTED;
  if ((k.Wo & 1) || (k.out_ns & 1) || ((uintptr_t)k.out & 7) || (k.res && ((k.res_ns & 1) || ((uintptr_t)k.res & 7)))) return SCF_EUNSUPPORTED;
  if (k.in1 && (k.C0 % WN_KC) != 0) return SCF_EUNSUPPORTED;
  if (((uintptr_t)wu & 15) || (long long)WN_KC * k.H * k.W * 4 >= 0x7fffffffLL) return SCF_EUNSUPPORTED;
  const int F = (k.Cout + 31) / 32;
  // tiles per row of a wave's group: 16 unless a narrower group wastes clearly fewer columns
  const int tcols = (k.Wo + 1) / 2;
  int txl = 4;
  {
    int best = (tcols + 15) / 16 * 16;
    if (tcols < 16) { txl = 0; while ((1 << txl) < tcols) ++txl; if (txl < 2) txl = 2; best = 0; }
    for (int l = 3; l >= 2 && best; --l) {
      const int wpad = (tcols + (1 << l) - 1) >> l << l;
      if (wpad * 10 <= best * 9) { best = wpad; txl = l; }
    }
  }
  const int TXW = 1 << txl, TYW = 32 >> txl;
  // 16-byte patch cells when every row of every plane is 16-byte aligned
  const bool px4 = (k.W % 4) == 0 && (((uintptr_t)k.in0 | (uintptr_t)k.in1) & 15) == 0 && (k.in0_ns % 4) == 0 && (k.in1_ns % 4) == 0;
  const int cfg = px4 ? 1 : 0;
  WinoK q;
  q.wu = wu; q.F = F; q.txl = txl;
  q.nchunk = (k.Cin + WN_KC - 1) / WN_KC;
  q.sx = (k.Wo + 2 * TXW - 1) / (2 * TXW);
  // geometry of a variant: CW channel fragments x TW tile groups per block, pslot floats per patch ring slot
  auto shape = [&](int CW, int TW, int pslot) -> long long {
    q.PH = 2 * TW * TYW + 2; q.PW = px4 ? 2 * TXW + 8 : 2 * TXW + 2;
    // row pitch = TXW mod 32 floats: the TYW tile rows a half-wave reads together then start 2 TXW banks apart
    // (fewer bank conflicts in the window reads), when the wider patch still fits the slot
    q.PWp = q.PW + ((TXW - q.PW) & 31);
    if (WN_KC * q.PH * q.PWp > pslot) q.PWp = q.PW;
    q.PPL = q.PH * q.PWp;
    if (WN_KC * q.PPL > pslot || (F % CW) != 0) return -1;
    q.sy = (k.Ho + 2 * TW * TYW - 1) / (2 * TW * TYW);
    q.mblocks = F / CW;
    const long long nblk = (long long)N * q.sx * q.sy * q.mblocks;
    return (nblk <= 0 || nblk > 0x7fffffffLL) ? -1 : nblk;
  };
  // Small grids stay on the direct kernels: a block here is a serial chain of Cin / 4 chunks (27 us at 128
  // input channels, 48 us at 256, whatever the grid) and the direct path has a K-split tile for them.  Measured
  // on all layer shapes of the refiner (tools/lab/wino_sweep.py): 0.4-0.87x at <= 96 blocks, 1.25-1.35x at 128,
  // 1.5-2x from 192 blocks on.  (In units of 4-wave blocks: an 8-wave block counts twice.)
  const int min_blocks4 = scf_cu_count() / 2;
  int variant = g_wino_variant.load(std::memory_order_relaxed);
  if (variant == 0) variant = SCF_WINO_DEFAULT_VARIANT;
#ifndef SCF_WINO_LAB
  // lab builds only (each measured, none faster -- docs/lab_notebook.md): 3 = the 8-wave form, 4 = four ring slots
  // (r5: 14.04 -> 14.07 ms per step), 5 = four ring slots + the burst form of a chunk (r5: 14.00 -> 14.82 ms)
  if (variant >= 3) variant = 2;
#endif
  if (variant == 2 || variant == 3 || variant == 4 || variant == 5) {
    const int TW = variant == 3 ? 2 : 1;
    const int NR = variant >= 4 ? 4 : 3;                 // ring slots (4: 80 KB of LDS, two blocks per CU still fit)
    const bool burst = variant == 5; (void)burst;
    const int npi = WQ_NPI(TW, px4), pslot = npi * 256 * TW * (px4 ? 4 : 1);
    const long long nblk = shape(2, TW, pslot);
    if (nblk > 0 && nblk * TW >= min_blocks4) {
      size_t ldsf = (size_t)NR * (2 * 2048 + pslot);
      if (ldsf < (size_t)TW * 4 * 3072) ldsf = (size_t)TW * 4 * 3072;      // the output exchange reuses the rings
      const size_t ldsb = ldsf * sizeof(float);
      if (info) { info[0] = 16; info[1] = 2 * TW; info[2] = (int)nblk; info[3] = (int)ldsb; }
      if (cap) {              // r6: hand the launch back (scf_conv2d_pair): the product's 4-wave, 3-slot form only
        static_assert(sizeof(WinoK) <= sizeof(cap->aux), "ScfLaunchCap::aux holds a WinoK");
        cap->k = k; cap->nblk = (int)nblk; cap->ldsb = ldsb;
        memcpy(cap->aux, &q, sizeof(WinoK));
        cap->variant = (TW == 1 && NR == 3) ? 10 + cfg : -1;
        if (which) *which = 1;
        return SCF_OK;
      }
      if (dry_run) return SCF_OK;
      static std::atomic<unsigned long long> raised_q[4];
#ifdef SCF_WINO_LAB
      if (TW == 2) {
        const void* fn8 = cfg ? (const void*)conv_wino_q_kernel<2, true> : (const void*)conv_wino_q_kernel<2, false>;
        const int rc8 = scf_raise_dynamic_lds(raised_q[2 + cfg], fn8, 96 * 1024);
        if (rc8 != SCF_OK) return rc8;
        if (cfg) scf_launch((conv_wino_q_kernel<2, true>), dim3((unsigned)nblk), dim3(512), ldsb, st, k, q);
        else scf_launch((conv_wino_q_kernel<2, false>), dim3((unsigned)nblk), dim3(512), ldsb, st, k, q);
        return scf_launch_status();
      }
#endif
      if (which) *which = 1;
#ifdef SCF_WINO_LAB
      if (NR == 4 && burst) {
        static std::atomic<unsigned long long> raised_qb[2];
        const void* fnb = cfg ? (const void*)conv_wino_q_kernel<1, true, 4, true> : (const void*)conv_wino_q_kernel<1, false, 4, true>;
        const int rcb = scf_raise_dynamic_lds(raised_qb[cfg], fnb, 80 * 1024);
        if (rcb != SCF_OK) return rcb;
        if (cfg) scf_launch((conv_wino_q_kernel<1, true, 4, true>), dim3((unsigned)nblk), dim3(256), ldsb, st, k, q);
        else scf_launch((conv_wino_q_kernel<1, false, 4, true>), dim3((unsigned)nblk), dim3(256), ldsb, st, k, q);
        return scf_launch_status();
      }
      if (NR == 4) {
        static std::atomic<unsigned long long> raised_q4[2];
        const void* fn4 = cfg ? (const void*)conv_wino_q_kernel<1, true, 4> : (const void*)conv_wino_q_kernel<1, false, 4>;
        const int rc4 = scf_raise_dynamic_lds(raised_q4[cfg], fn4, 80 * 1024);
        if (rc4 != SCF_OK) return rc4;
        if (cfg) scf_launch((conv_wino_q_kernel<1, true, 4>), dim3((unsigned)nblk), dim3(256), ldsb, st, k, q);
        else scf_launch((conv_wino_q_kernel<1, false, 4>), dim3((unsigned)nblk), dim3(256), ldsb, st, k, q);
        return scf_launch_status();
      }
#endif
      const void* fn = cfg ? (const void*)conv_wino_q_kernel<1, true> : (const void*)conv_wino_q_kernel<1, false>;
      const int rc = scf_raise_dynamic_lds(raised_q[cfg], fn, 64 * 1024);
      if (rc != SCF_OK) return rc;
      if (cfg) scf_launch((conv_wino_q_kernel<1, true>), dim3((unsigned)nblk), dim3(256), ldsb, st, k, q);
      else scf_launch((conv_wino_q_kernel<1, false>), dim3((unsigned)nblk), dim3(256), ldsb, st, k, q);
      return scf_launch_status();
    }
    // odd fragment counts, patches that do not fit the quarter kernel's slot: the pair kernel
  }
  // pair kernel: a block = one channel fragment of two vertically stacked tile groups (8 x 32 outputs on a wide
  // map): the two wave pairs share the U chunk
  const int CW = 1, TW = 2;
  const int npi = WN_NPI(TW, px4);
  const long long nblk = shape(CW, TW, npi * (px4 ? 1024 : 256));
  if (nblk < 0) return SCF_EUNSUPPORTED;
  if (nblk < min_blocks4) return SCF_EUNSUPPORTED;
  const size_t ldsb = (size_t)(3 * CW * 2048 + 3 * npi * (px4 ? 1024 : 256)) * sizeof(float);
  if (ldsb > 80 * 1024) return SCF_EUNSUPPORTED;
  if (info) { info[0] = 16; info[1] = CW * TW; info[2] = (int)nblk; info[3] = (int)ldsb; }     // positions, fragments per block
  if (cap) {                  // r6: the pair kernel's launch, for a launch shared with a quarter-domain layer
    cap->k = k; cap->nblk = (int)nblk; cap->ldsb = ldsb;
    memcpy(cap->aux, &q, sizeof(WinoK));
    cap->variant = 20 + cfg;
    return SCF_OK;
  }
  if (dry_run) return SCF_OK;
  {                                  // more than 64 KB of dynamic LDS needs the attribute, once per kernel and device
    static std::atomic<unsigned long long> raised[2];
    const int rc = scf_raise_dynamic_lds(raised[cfg], cfg ? (const void*)conv_wino_kernel<1, 2, true>
                                                          : (const void*)conv_wino_kernel<1, 2, false>, 80 * 1024);
    if (rc != SCF_OK) return rc;
  }
  if (cfg) scf_launch((conv_wino_kernel<1, 2, true>), dim3((unsigned)nblk), dim3(256), ldsb, st, k, q);
  else scf_launch((conv_wino_kernel<1, 2, false>), dim3((unsigned)nblk), dim3(256), ldsb, st, k, q);
  return scf_launch_status();
}

int scf_conv_wino_pair_launch(const ScfLaunchCap& a, const ScfLaunchCap& b, hipStream_t st) {
  if (a.variant >= 10 && a.variant < 20 && b.variant >= 20 && a.nblk > 0 && b.nblk > 0) {      // quarter-domain | pair kernel
    const size_t ldsm = a.ldsb > b.ldsb ? a.ldsb : b.ldsb;
    WinoK qa, qb;
    memcpy(&qa, a.aux, sizeof(WinoK));
    memcpy(&qb, b.aux, sizeof(WinoK));
    const int ca = a.variant - 10, cb2 = b.variant - 20, id = ca * 2 + cb2;
    static std::atomic<unsigned long long> raised_m[4];
    const void* fn = id == 0 ? (const void*)conv_wino_mixed_pair_kernel<false, false> : id == 1 ? (const void*)conv_wino_mixed_pair_kernel<false, true>
                   : id == 2 ? (const void*)conv_wino_mixed_pair_kernel<true, false> : (const void*)conv_wino_mixed_pair_kernel<true, true>;
    const int rc = scf_raise_dynamic_lds(raised_m[id], fn, 80 * 1024);
    if (rc != SCF_OK) return rc;
    const dim3 grid((unsigned)(a.nblk + b.nblk));
    if (id == 0) scf_launch((conv_wino_mixed_pair_kernel<false, false>), grid, dim3(256), ldsm, st, a.k, qa, b.k, qb, a.nblk);
    else if (id == 1) scf_launch((conv_wino_mixed_pair_kernel<false, true>), grid, dim3(256), ldsm, st, a.k, qa, b.k, qb, a.nblk);
    else if (id == 2) scf_launch((conv_wino_mixed_pair_kernel<true, false>), grid, dim3(256), ldsm, st, a.k, qa, b.k, qb, a.nblk);
    else scf_launch((conv_wino_mixed_pair_kernel<true, true>), grid, dim3(256), ldsm, st, a.k, qa, b.k, qb, a.nblk);
    return scf_launch_status();
  }
  if (a.variant < 10 || a.variant >= 20 || a.variant != b.variant || a.nblk <= 0 || b.nblk <= 0) return SCF_EUNSUPPORTED;
  const size_t lds = a.ldsb > b.ldsb ? a.ldsb : b.ldsb;
  WinoK qa, qb;
  memcpy(&qa, a.aux, sizeof(WinoK));
  memcpy(&qb, b.aux, sizeof(WinoK));
  const int cfg = a.variant - 10;
  static std::atomic<unsigned long long> raised[2];
  const void* fn = cfg ? (const void*)conv_wino_q_pair_kernel<true> : (const void*)conv_wino_q_pair_kernel<false>;
  const int rc = scf_raise_dynamic_lds(raised[cfg], fn, 64 * 1024);
  if (rc != SCF_OK) return rc;
  const unsigned grid = (unsigned)(a.nblk + b.nblk);
  if (cfg) scf_launch((conv_wino_q_pair_kernel<true>), dim3(grid), dim3(256), lds, st, a.k, qa, b.k, qb, a.nblk);
  else scf_launch((conv_wino_q_pair_kernel<false>), dim3(grid), dim3(256), lds, st, a.k, qa, b.k, qb, a.nblk);
  return scf_launch_status();
}
