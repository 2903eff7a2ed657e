// 1 x 5 / 5 x 1 stride-1 convolutions (the SepConvGRU gates, raft_decoder.py:198-253) in the one-dimensional
// Winograd form F(2, 5), fp32 throughout:  two outputs along the filter axis from a window of six inputs with
// 6 multiplies per channel pair instead of 10 (points 0, +-1, +-2, infinity):
//     y = A^T [ (G g) . (B^T d) ],   A^T 2 x 6, G 6 x 5, B^T 6 x 6.
// Same structure as conv_wino.hip (read that file first: copy streams, rings, operand double buffer, why the
// vector-ALU count per MFMA is what matters), with these differences:
//   * a wave holds ONE 32 x 32 fragment (32 output channels x 32 tiles of 2 outputs) for all 6 transform
//     positions = 96 accumulator registers: the output transform is lane-local, no pair exchange;
//   * a block = 2 channel fragments x 2 tile groups (the copy of a U chunk serves two waves, the copy of a patch
//     chunk two); 8 channels per chunk: one ds_read_b128 per position feeds four k-steps, 24 MFMAs per barrier;
//   * the input transform is 6 packed fmas per window (4 windows per lane and chunk = 1 vector instruction per MFMA;
//     r4: was 12 scalar fmas);
//   * after the output transform a lane holds 16 channel rows of two pixels -- exactly two accumulator
//     fragments of the direct kernels, so the shared fused epilogue (conv_kernels.h) finishes them: bias, BN,
//     residual, activations, both GRU gates with the hoisted context term.
// Error vs fp64 about 5e-6 on unit-scale outputs (the direct kernel: 1e-6; B^T scales the inputs by up to 5 and
// G by down to 1/24 before they cancel again); the GRU state after two iterations differs from the direct
// kernels' by 4.5e-6 (tests/test_gpu_ops.py).
#include <stdlib.h>
#include <string.h>
#include "scf_common.h"
#include "conv_kernels.h"
#include "scf_dma.h"

typedef float w1_f32x16 __attribute__((ext_vector_type(16)));
typedef float w1_f32x4 __attribute__((ext_vector_type(4)));
typedef float w1_f32x2 __attribute__((ext_vector_type(2)));

#define W1_KC 8             // channels per chunk
#define W1_UF 1536          // floats of one fragment's U chunk: [6 positions][2 k-halves][32 channels][4 k-steps]
// patch copy instructions per wave per chunk (256 cells per block-instruction; lanes past the patch write zeros
// into the slot's padding): horizontal with 16-byte cells 2 (a conflict-free row pitch of 96 floats was measured:
// no difference), with dword cells 6; vertical (16-byte cells only) 3
#define W1_NPI(VERT, PX4) ((VERT) ? 3 : ((PX4) ? 2 : 6))

struct Wino1K {
  const float* wu;          // [nchunk][F][6][2][32][4]
  int F;                    // channel fragments in the packing
  int txl;                  // log2(tile columns of a wave's 32-tile group)
  int PH, PWp, PPL;         // patch rows, row pitch, plane stride (floats)
  int nchunk;
  int sx, sy;               // block strips per image
  int mblocks;
};

__device__ __forceinline__ int w1_div(int e, int d, float rd) {       // floor(e / d), 0 <= e < 2^20, 0 < d < 2^12
  int q = (int)((float)e * rd);
  const int r = e - q * d;
  q += (r >= d) ? 1 : 0;
  q -= (r < 0) ? 1 : 0;
  return q;
}

template <bool VERT, bool PX4>
__global__ __launch_bounds__(256, 2)
void conv_wino1d_kernel(ConvK p, Wino1K q) {
  static_assert(!VERT || PX4, "the vertical kernel copies 16-byte cells only");
  extern __shared__ __attribute__((aligned(16))) float w1_lds[];
  constexpr int CW = 2, TW = 2;
  constexpr int USLOT = CW * W1_UF;                             // floats per ring slot
  constexpr int NUI = 3;                                        // U copy instructions (16 B per lane) per wave per chunk
  constexpr int NPI = W1_NPI(VERT, PX4);
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
  // first output pixel of the block; a tile = 2 pixels along the filter axis
  const int y0 = ys * (VERT ? 2 * TW * TYW : TW * TYW), x0 = xs * (VERT ? TXW : 2 * TXW);
  const int f0 = mb * CW;
  const int HW = p.H * p.W;

  float* Us = w1_lds;
  float* Ps = Us + 3 * USLOT;
  const unsigned u_lds = scf_lds_addr(Us), p_lds = scf_lds_addr(Ps);

  // ---- chunk-invariant copy offsets ------------------------------------------------------------------
  // patch = 8 channel planes of PH rows x PWp floats: horizontal: the block's rows, columns from x0 - 2
  // (x0 - 4 with 16-byte cells, windows then start at column 2 tx + 2); vertical: rows from y0 - 2, the block's
  // columns (pitch 32 floats: the six window rows are immediates apart)
  unsigned pvo[NPI];
  {
    const int NC = PX4 ? q.PWp >> 2 : q.PWp, PPC = q.PH * NC;
    const float rPPC = 1.0f / (float)PPC, rNC = 1.0f / (float)NC;
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
      const int e = i * 256 + tid;
      const int c = w1_div(e, PPC, rPPC), r = e - c * PPC;
      const int py = w1_div(r, NC, rNC), px = r - py * NC;
      const int iy = VERT ? y0 - 2 + py : y0 + py;
      const int ix = VERT ? x0 + 4 * px : (PX4 ? x0 - 4 + 4 * px : x0 - 2 + px);
      const bool ok = c < W1_KC && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      pvo[i] = ok ? (unsigned)((c * HW + iy * p.W + ix) * 4) : SCF_BUF_OOB;
    }
  }
  unsigned uvo[NUI], uld[NUI];
#pragma unroll
  for (int i = 0; i < NUI; ++i) {               // a fragment's chunk is 6 KB contiguous: 6 instructions of 1 KB
    const int j = wave + 4 * i;
    const int f = j / 6, part = j - 6 * f;
    uvo[i] = (unsigned)((f0 + f) * (W1_UF * 4) + part * 1024 + lane * 16);
    uld[i] = (unsigned)(f * (W1_UF * 4) + part * 1024);
  }
  const unsigned u_chunk_bytes = (unsigned)(q.F * W1_UF * 4);
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
  const unsigned p_chunk_bytes = (unsigned)(W1_KC * HW * 4);
  int p_left = p.C0;                                 // channels of the current input segment still to copy
  bool p_second = p.in1 == nullptr;
  scf_rsrc4 prs = scf_make_rsrc(p.in0 + (long long)n * p.in0_ns, (unsigned)((p_left < W1_KC ? p_left : W1_KC) * HW * 4));
  auto issue_p = [&](int slot) {                   // the next patch chunk -> ring slot
    const unsigned dst = p_lds + (unsigned)((slot * PSLOT + wave * (PX4 ? 256 : 64)) * 4);
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
      if (PX4) scf_bdma_b128(prs, pvo[i], dst + (unsigned)(i * 4096));
      else scf_bdma_b32(prs, pvo[i], dst + (unsigned)(i * 1024));
    }
    p_left -= W1_KC;
    if (p_left <= 0 && !p_second) {                  // on to the second input segment (C0 % 8 == 0 there)
      p_second = true;
      p_left = p.Cin - p.C0;
      prs = scf_make_rsrc(p.in1 + (long long)n * p.in1_ns, 0u);
    } else {
      const unsigned lo = (unsigned)prs[0] + p_chunk_bytes;
      prs[1] += lo < p_chunk_bytes ? 1 : 0;
      prs[0] = (int)lo;
    }
    const int cl = p_left < W1_KC ? p_left : W1_KC;
    prs[2] = cl > 0 ? cl * HW * 4 : 0;
  };

  // ---- input transform in registers: lane (tile l32, k-half) turns the windows of channels half, 2 + half,
  //      4 + half, 6 + half into its B operands: r0 = 4 d0 - 5 d2 + d4, r1 / r2 = (d4 - 4 d2) +- (d3 - 4 d1),
  //      r3 / r4 = (d4 - d2) +- 2 (d3 - d1), r5 = 4 d1 - 5 d3 + d5 ------------------------------------------
  const int ty = tw * TYW + (l32 >> q.txl), tx = l32 & (TXW - 1);
  unsigned prow[4];                                  // absolute LDS byte address of the window of channel 2 s + half, slot 0
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int off = VERT ? 2 * ty * q.PWp + tx : ty * q.PWp + 2 * tx + (PX4 ? 2 : 0);
    prow[s] = p_lds + (unsigned)(((2 * s + half) * q.PPL + off) * 4);
  }
  // The transform runs on PACKED fp32 pairs (r4): vector-ALU instructions of either co-resident wave cost
  // matrix-pipe time on their SIMD (conv_wino.hip), and the scalar form was 12 fmas per window = 2 per MFMA.
  // With the window held as the pairs P01 = (d0, d1), P23 = (d2, d3), P45 = (d4, d5) -- exactly what the three
  // ds_read_b64 of a horizontal window return -- six v_pk_fma_f32 give all six positions, each fma the one the
  // scalar form evaluated (same operands, same order: identical bits):
  //     (r0, r5) = 4 P01 + (-5 P23 + P45)
  //     T = (t1, t3) = (-4, -1) (d2, d2) + (d4, d4)        S = (t2, sd) = (-4, -1) (d1, d1) + (d3, d3)
  //     (r1, r3) = (1, 2) S + T                             (r2, r4) = (-1, -2) S + T
  // The op_sel forms (both result halves from the same half of an operand) are written in asm; an MFMA takes its
  // B operand from either half of a result pair directly, so nothing is moved.
  w1_f32x2 dwp[4][3];
  auto win_load = [&](unsigned slot_bytes, int s) {
    const __attribute__((address_space(3))) float* r =
        (const __attribute__((address_space(3))) float*)(uintptr_t)(prow[s] + slot_bytes);
#pragma unroll
    for (int i = 0; i < 3; ++i)       // vertical: pitch 32 floats
      dwp[s][i] = VERT ? w1_f32x2{r[(2 * i) * 32], r[(2 * i + 1) * 32]} : w1_f32x2{r[2 * i], r[2 * i + 1]};
  };
  const w1_f32x2 k_4 = {4.f, 4.f}, k_m5 = {-5.f, -5.f}, k_m4m1 = {-4.f, -1.f}, k_12 = {1.f, 2.f}, k_m12 = {-1.f, -2.f};
  auto win_transform = [&](w1_f32x2 (&bo)[4][3], int s) {      // -> bo[s] = {(r0, r5), (r1, r3), (r2, r4)}
    const w1_f32x2 (&d)[3] = dwp[s];
    w1_f32x2 in, T, S;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(in) : "v"(k_m5), "v"(d[1]), "v"(d[2]));
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(bo[s][0]) : "v"(k_4), "v"(d[0]), "v"(in));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(T) : "v"(k_m4m1), "v"(d[1]), "v"(d[2]));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,1,1]" : "=v"(S) : "v"(k_m4m1), "v"(d[0]), "v"(d[1]));
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(bo[s][1]) : "v"(k_12), "v"(S), "v"(T));
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(bo[s][2]) : "v"(k_m12), "v"(S), "v"(T));
  };
// B operand of position X, k-step S
#define W1_B(b, X, S) ((X) == 0 ? b[S][0][0] : (X) == 5 ? b[S][0][1] : (X) == 1 ? b[S][1][0] : (X) == 3 ? b[S][1][1] : (X) == 2 ? b[S][2][0] : b[S][2][1])

  w1_f32x16 acc[6];
#pragma unroll
  for (int x = 0; x < 6; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;

  // GRU launches: the pre-activation term (`res`: the hoisted context part, one value per output) is requested
  // NOW, in front of the first copies (vector loads return in order, so the counted waits below still cover
  // exactly the copies), and added after the output transform -- read in the epilogue it is one more memory
  // round trip at the end of every block, when all blocks of a round are there at the same time.
  const ConvEpi e = scf_conv_epi(p, n);
  int pix[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int oy = VERT ? y0 + 2 * ty + j : y0 + ty, ox = VERT ? x0 + tx : x0 + 2 * tx + j;
    pix[j] = (oy < p.Ho && ox < p.Wo) ? oy * p.Wo + ox : -1;
  }
  const bool pre_res = e.res && (p.mode == SCF_CONV_GRU_ZR || p.mode == SCF_CONV_GRU_Q) && p.out_div == 1.0f;
  float resv[2][16];
  if (pre_res) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = (f0 + cw) * 32 + 8 * (r >> 2) + (r & 3) + 4 * half;
        resv[j][r] = (pix[j] >= 0 && co < p.Cout) ? e.res[co * e.HWo + pix[j]] : 0.f;
      }
  }

  // ---- prologue: three chunks requested, two awaited ----------------------------------------------------
  issue_p(0); issue_u(0);
  issue_u(1); issue_p(1);
  issue_u(2); issue_p(2);
  scf_wait_vmcnt_imm<GRP>();
  __syncthreads();
  const float* ua = Us + cw * W1_UF + lane * 4;                        // + position * 256
  w1_f32x4 a0[6], a1[6];
  w1_f32x2 b0[4][3], b1[4][3];
#pragma unroll
  for (int x = 0; x < 6; ++x) a0[x] = *reinterpret_cast<const w1_f32x4*>(ua + x * 256);
#pragma unroll
  for (int s = 0; s < 4; ++s) { win_load(0u, s); win_transform(b0, s); }
  __syncthreads();                     // slot 0 of both rings is free again

  // chunk: operands of the current chunk in registers (a, b); those of the next one are read (U) / computed
  // (patch) under its MFMAs; the copies of the chunk three ahead are issued, those two ahead must have landed
  // at its end.  The MFMAs go first, everything else sits between them in the order its results are needed.
  int s1 = 1;                          // ring slot of the next chunk
  auto chunk = [&](const w1_f32x4 (&a)[6], const w1_f32x2 (&b)[4][3], w1_f32x4 (&an)[6], w1_f32x2 (&bn)[4][3]) {
    const float* uc = ua + s1 * USLOT;
    const unsigned pcb = (unsigned)(s1 * PSLOT * 4);
    int s3 = s1 + 2;                   // ring slot of the chunk three ahead
    s3 = s3 >= 3 ? s3 - 3 : s3;
#define W1_M(X, S)                                                                              \
    acc[X] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[X][S], W1_B(b, X, S), acc[X], 0, 0, 0);     \
    __builtin_amdgcn_sched_barrier(0);
    W1_M(0, 0) win_load(pcb, 0); __builtin_amdgcn_sched_barrier(0);
    W1_M(1, 0) win_load(pcb, 1); __builtin_amdgcn_sched_barrier(0);
    W1_M(2, 0) win_load(pcb, 2); __builtin_amdgcn_sched_barrier(0);
    W1_M(3, 0) win_load(pcb, 3); __builtin_amdgcn_sched_barrier(0);
    W1_M(4, 0) issue_u(s3); __builtin_amdgcn_sched_barrier(0);
    W1_M(5, 0) issue_p(s3); __builtin_amdgcn_sched_barrier(0);
    W1_M(0, 1)
#pragma unroll
    for (int x = 0; x < 6; ++x) an[x] = *reinterpret_cast<const w1_f32x4*>(uc + x * 256);
    __builtin_amdgcn_sched_barrier(0);
    W1_M(1, 1) W1_M(2, 1) win_transform(bn, 0); __builtin_amdgcn_sched_barrier(0);
    W1_M(3, 1) W1_M(4, 1) W1_M(5, 1) win_transform(bn, 1); __builtin_amdgcn_sched_barrier(0);
    W1_M(0, 2) W1_M(1, 2) W1_M(2, 2) win_transform(bn, 2); __builtin_amdgcn_sched_barrier(0);
    W1_M(3, 2) W1_M(4, 2) W1_M(5, 2) win_transform(bn, 3); __builtin_amdgcn_sched_barrier(0);
    W1_M(0, 3) W1_M(1, 3) W1_M(2, 3) W1_M(3, 3) W1_M(4, 3) W1_M(5, 3)
#undef W1_M
    scf_wait_vmcnt_imm<GRP>();
    __syncthreads();
    s1 = s1 == 2 ? 0 : s1 + 1;
  };
  int c = 0;
  for (; c + 1 < q.nchunk; c += 2) {
    chunk(a0, b0, a1, b1);
    chunk(a1, b1, a0, b0);
  }
  if (c < q.nchunk) chunk(a0, b0, a1, b1);
  scf_wait_vmcnt_imm<0>();             // the zero-filled groups past the end

  // ---- output transform: y0 = M0 + M1 + M2 + M3 + M4, y1 = M1 - M2 + 2 (M3 - M4) + M5, then the shared fused
  //      epilogue on the two pixels' 16-row fragments ----------------------------------------------------------
  w1_f32x16 o[1][2];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    o[0][0][r] = ((acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r])) + acc[4][r];
    o[0][1][r] = __builtin_fmaf(2.f, acc[3][r] - acc[4][r], acc[1][r] - acc[2][r]) + acc[5][r];
  }
  ConvEpi e2 = e;
  if (pre_res) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[0][j][r] += resv[j][r];
    e2.res = nullptr;                  // consumed
  }
  scf_conv_epilogue_tile<1, 2>(p, e2, o, (f0 + cw) * 32, half, pix, p.out_div != 1.0f);
}

// ---------------------------------------------------------------------------------------------------
// Host side: packing and launch
// ---------------------------------------------------------------------------------------------------
extern "C" int64_t scf_pack_conv_weight_wino1d_size(int32_t cout, int32_t cin) {
  if (cout <= 0 || cin <= 0) return 0;
  const int64_t F = (cout + 31) / 32, nchunk = (cin + W1_KC - 1) / W1_KC;
  return nchunk * F * W1_UF;
}

// w: (Cout, Cin, 5) = a 1 x 5 or 5 x 1 kernel's taps in filter order
extern "C" int scf_pack_conv_weight_wino1d(const float* w, int32_t cout, int32_t cin, float* out) {
  if (!w || !out || cout <= 0 || cin <= 0) return SCF_EINVAL;
  const int F = (cout + 31) / 32;
  memset(out, 0, sizeof(float) * (size_t)scf_pack_conv_weight_wino1d_size(cout, cin));
  // G (6 x 5) for the points 0, 1, -1, 2, -2, infinity: row i = (1, a_i, ..., a_i^4) / prod_{k != i} (a_i - a_k)
  static const double pts[5] = {0.0, 1.0, -1.0, 2.0, -2.0};
  double G[6][5];
  for (int i = 0; i < 5; ++i) {
    double nrm = 1.0;
    for (int k = 0; k < 5; ++k)
      if (k != i) nrm *= pts[i] - pts[k];
    double pw = 1.0;
    for (int k = 0; k < 5; ++k) { G[i][k] = pw / nrm; pw *= pts[i]; }
  }
  for (int k = 0; k < 5; ++k) G[5][k] = k == 4 ? 1.0 : 0.0;
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci) {
      const float* g = w + ((size_t)co * cin + ci) * 5;
      const int chunk = ci / W1_KC, cl = ci % W1_KC, s = cl >> 1, kh = cl & 1;
      const int frag = co / 32, m = co % 32;
      for (int i = 0; i < 6; ++i) {
        double u = 0.0;
        for (int k = 0; k < 5; ++k) u += G[i][k] * g[k];
        out[(((size_t)chunk * F + frag) * 6 + i) * 256 + kh * 128 + m * 4 + s] = (float)u;
      }
    }
  return SCF_OK;
}

// Tile selection + launch; SCF_EUNSUPPORTED -> the caller goes on to the direct kernels.  info: {6 transform positions, 4 fragments per block, blocks, LDS bytes}.
int scf_conv_wino1d_dispatch(ConvK k, const float* wu, int N, bool dry_run, int* info, hipStream_t st) {
  const bool vert = k.KH == 5 && k.KW == 1 && k.pad_h == 2 && k.pad_w == 0;
  const bool horz = k.KH == 1 && k.KW == 5 && k.pad_h == 0 && k.pad_w == 2;
  if (!wu || !(vert || horz) || k.stride != 1 || k.w_ns != 0 || k.out_tile) return SCF_EUNSUPPORTED;
  if (k.in1 && (k.C0 % W1_KC) != 0) return SCF_EUNSUPPORTED;
  if (((uintptr_t)wu & 15) || (long long)W1_KC * k.H * k.W * 4 >= 0x7fffffffLL) return SCF_EUNSUPPORTED;
  const int F = (k.Cout + 31) / 32;
  if (F % 2) return SCF_EUNSUPPORTED;                   // blocks take pairs of channel fragments
  const bool px4 = (k.W % 4) == 0 && (((uintptr_t)k.in0 | (uintptr_t)k.in1) & 15) == 0 && (k.in0_ns % 4) == 0 && (k.in1_ns % 4) == 0;
  if (vert && !px4) return SCF_EUNSUPPORTED;
  const int TW = 2;
  int txl;
  Wino1K q;
  q.wu = wu; q.F = F;
  if (vert) {
    txl = k.Wo > 16 ? 5 : 4;                            // 32 columns x 1 tile row, or 16 x 2
    const int TXW = 1 << txl, TYW = 32 >> txl;
    q.PH = 2 * TW * TYW + 4; q.PWp = 32;
    q.sx = (k.Wo + TXW - 1) / TXW;
    q.sy = (k.Ho + 2 * TW * TYW - 1) / (2 * TW * TYW);
  } else {
    const int tcols = (k.Wo + 1) / 2;
    txl = 4;
    if (tcols < 16) { txl = 2; while ((1 << txl) < tcols) ++txl; }
    else {
      int best = (tcols + 15) / 16 * 16;
      for (int l = 3; l >= 2; --l) {
        const int wpad = (tcols + (1 << l) - 1) >> l << l;
        if (wpad * 10 <= best * 9) { best = wpad; txl = l; }
      }
    }
    const int TXW = 1 << txl, TYW = 32 >> txl;
    q.PH = TW * TYW; q.PWp = px4 ? 2 * TXW + 8 : 2 * TXW + 4;
    q.sx = (k.Wo + 2 * TXW - 1) / (2 * TXW);
    q.sy = (k.Ho + TW * TYW - 1) / (TW * TYW);
  }
  q.txl = txl;
  q.PPL = q.PH * q.PWp;
  const int npi = W1_NPI(vert, px4);
  if (W1_KC * q.PPL > npi * (px4 ? 1024 : 256)) return SCF_EUNSUPPORTED;
  q.nchunk = (k.Cin + W1_KC - 1) / W1_KC;
  q.mblocks = F / 2;
  const long long nblk = (long long)N * q.sx * q.sy * q.mblocks;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return SCF_EUNSUPPORTED;
  if (nblk < scf_cu_count() / 2) return SCF_EUNSUPPORTED;             // small grids: the direct kernels' K-split tile
  const size_t ldsb = (size_t)(3 * 2 * W1_UF + 3 * npi * (px4 ? 1024 : 256)) * sizeof(float);
  if (info) { info[0] = 6; info[1] = 4; info[2] = (int)nblk; info[3] = (int)ldsb; }      // positions, fragments per block
  if (dry_run) return SCF_OK;
  const int cfg = vert ? 2 : (px4 ? 1 : 0);
  const void* fn = cfg == 2 ? (const void*)conv_wino1d_kernel<true, true> : cfg == 1 ? (const void*)conv_wino1d_kernel<false, true>
                                                                                    : (const void*)conv_wino1d_kernel<false, false>;
  if (ldsb > 64 * 1024) {
    static std::atomic<unsigned long long> raised[3];
    const int rc = scf_raise_dynamic_lds(raised[cfg], fn, 80 * 1024);
    if (rc != SCF_OK) return rc;
  }
  if (cfg == 2) scf_launch((conv_wino1d_kernel<true, true>), dim3((unsigned)nblk), dim3(256), ldsb, st, k, q);
  else if (cfg == 1) scf_launch((conv_wino1d_kernel<false, true>), dim3((unsigned)nblk), dim3(256), ldsb, st, k, q);
  else scf_launch((conv_wino1d_kernel<false, false>), dim3((unsigned)nblk), dim3(256), ldsb, st, k, q);
  return scf_launch_status();
}
