// Correlation volume (CorrelationPyramid.forward, models/decoder/raft_decoder.py:35-58) as a
// dedicated fp32 MFMA GEMM for gfx950:
//     level0[n][i][j] = (1/sqrt(C)) * sum_c feat1[n][c][i] * feat2[n][c][j],   i, j in [0, h*w)
//     level1[n][i]    = 2x2 average pool of level0[n][i] over the target map (fused, optional)
//
// Both operands are NCHW feature maps = K-major matrices [C][hw]: exactly what the fp32 matrix
// instruction wants (v_mfma_f32_32x32x2_f32: lane l holds A[l&31][l>>5] and B[l>>5][l&31], so a
// k-row of 32 consecutive queries / targets is one conflict-free ds_read_b32).  Nothing is
// transposed, packed or converted anywhere.
//
//   block tile  : 128 queries (i) x 128 targets (j); 4 waves as 2 x 2, each 64 x 64 = 2 x 2
//                 accumulator fragments (64 VGPRs).
//   staging     : LDS-DMA (global_load_lds_dwordx4: memory -> LDS, no staging registers), three
//                 chunk buffers of 16 channels x (128 + 128) floats = 48 KB: chunk c+2 streams in
//                 while chunk c is on the matrix cores; one barrier per chunk.
//   target order: with the tiled level-0 layout a 32-target fragment IS one 8x4-float tile of the
//                 query's map.  The permutation costs nothing: it lives in the per-lane source
//                 offsets of the B-operand DMA (computed once per block); LDS holds the fragment
//                 in tile order, the MFMA D fragment comes out in tile order, and every output row
//                 of a fragment is ONE full 128-byte line (row-major layout: 32 consecutive targets).
//   epilogue    : x 1/sqrt(C) (exact for power-of-two sqrt(C), correctly rounded division
//                 otherwise), level-0 store, and -- tiled layout -- the first 2x2 average pool from
//                 the fragment itself: lane (x, y) of a tile fetches its three window partners with
//                 DPP (quad_perm / row_ror:8), sums them in AvgPool2d's order ((a+b)+c)+d, x 0.25.
//
// The kernel that ran this contraction in round 1 (the register-staged convolution kernel with
// per-sample "weights") reached 0.48 of the fp32 MFMA peak.
#include "scf_common.h"

typedef float cg_f32x16 __attribute__((ext_vector_type(16)));

struct CorrGemmParams {
  const float* f1; const float* f2;
  float* lvl0; float* lvl1;          // lvl1 == nullptr: no fused pool
  int C, h, w, hw;
  int nchunk;                         // C / 16
  int mblocks, nblocks;               // 128-query / 128-target blocks per sample
  int tiled;                          // level 0 (and the target order) in 8x4 tiles
  int ntiles_x;                       // w / 8 (tiled)
  float scale; int exact_scale;       // exact_scale: multiply by scale (= 1/sqrt(C), power of two); else divide by `divisor`
  float divisor;
  int l1_pw4, l1_msz;                 // level-1 layout: 4 x padded width when tiled (0 = row-major), floats per query map
};

#define CG_KC 16
#define CG_BM 128
#define CG_BN 128
#define CG_STAGE_FLOATS (CG_KC * (CG_BM + CG_BN))      // 4096 floats = 16 KB
#define CG_NSTAGE 3

__device__ __forceinline__ unsigned cg_lds_addr(const void* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ const void* cg_sgpr_ptr(const void* p) {
  const unsigned long long u = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return (const void*)(((unsigned long long)hi << 32) | lo);
}
// 16 bytes per lane: lane l's data lands at lds + l*16; source = scalar base + 32-bit lane byte
// offset; a lane whose offset is 0xFFFFFFFF stays off (its LDS slot keeps the zero written at
// block start).  The compiler does not count these loads: the caller waits on vmcnt itself.
__device__ __forceinline__ void cg_dma_b128(const void* sbase, unsigned voff, unsigned lds) {
  asm volatile("v_cmp_ne_u32_e32 vcc, -1, %1\n\ts_mov_b64 exec, vcc\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %0\n\ts_mov_b64 exec, -1"
               : : "s"(sbase), "v"(voff), "s"(lds) : "memory", "vcc");
}

template <bool TILED, bool POOL>
__global__ __launch_bounds__(256, 3) void corr_gemm_kernel(CorrGemmParams p) {
  extern __shared__ __attribute__((aligned(16))) float cg_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, half = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware: consecutive logical blocks (same sample, same query block -> same A slab) share an L2
  const int lb = scf_xcd_remap(blockIdx.x, gridDim.x);
  const int nb = __builtin_amdgcn_readfirstlane(lb % p.nblocks);
  const int t2 = lb / p.nblocks;
  const int mb = __builtin_amdgcn_readfirstlane(t2 % p.mblocks);
  const int n = __builtin_amdgcn_readfirstlane(t2 / p.mblocks);
  const int i0 = mb * CG_BM;                     // first query of the block
  const int hw = p.hw;

  // ---- zero all stages once: lanes outside the matrices (ragged last blocks) are never written ----
  {
    typedef float __attribute__((ext_vector_type(4))) f4;
    for (int i = tid; i < CG_NSTAGE * CG_STAGE_FLOATS / 4; i += 256) ((f4*)cg_lds)[i] = f4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- per-lane DMA source offsets (bytes, relative to the chunk's first channel plane) ----
  // One wave instruction stages 2 channel rows (kk = lane >> 5) x 128 floats (32 lanes x 16 B).
  // A: queries i0 + 4*(lane & 31) .. +3 in natural order.
  // B: targets of the block in FRAGMENT order: fragment F = nb*4 + (l5 >> 3), element (l5 & 7)*4..+3
  //    of the fragment; tiled: fragment = tile (ty, tx), element e = y*8 + x -> pixel (4ty + y, 8tx + x).
  const int kk = lane >> 5, l5 = lane & 31;
  unsigned offA, offB;
  {
    const int i = i0 + l5 * 4;
    offA = i < hw ? (unsigned)((kk * hw + i) * 4) : 0xFFFFFFFFu;
    const int F = nb * 4 + (l5 >> 3), e = (l5 & 7) * 4;
    int pix = -1;
    if (TILED) {
      const int ntile = (p.h >> 2) * p.ntiles_x;
      if (F < ntile) {
        const int ty = F / p.ntiles_x, tx = F - ty * p.ntiles_x;
        pix = (ty * 4 + (e >> 3)) * p.w + tx * 8 + (e & 7);
      }
    } else {
      const int j = F * 32 + e;
      if (j < hw) pix = j;
    }
    offB = pix >= 0 ? (unsigned)((kk * hw + pix) * 4) : 0xFFFFFFFFu;
  }
  const float* a_n = p.f1 + (long long)n * p.C * hw;
  const float* b_n = p.f2 + (long long)n * p.C * hw;
  const unsigned lds0 = cg_lds_addr(cg_lds);
  const long long rowpair = (long long)2 * hw;   // floats per 2 channel rows

  // wave w stages channel rows 4w .. 4w+3 of every chunk: 2 instructions per operand
  auto stage = [&](int chunk, int buf) {
    const float* ab = (const float*)cg_sgpr_ptr(a_n + ((long long)chunk * CG_KC + wave * 4) * hw);
    const float* bb = (const float*)cg_sgpr_ptr(b_n + ((long long)chunk * CG_KC + wave * 4) * hw);
    const unsigned la = lds0 + (unsigned)(buf * CG_STAGE_FLOATS + wave * 4 * CG_BM) * 4u;
    const unsigned lbb = lds0 + (unsigned)(buf * CG_STAGE_FLOATS + CG_KC * CG_BM + wave * 4 * CG_BN) * 4u;
    cg_dma_b128(ab, offA, la);
    cg_dma_b128(ab + rowpair, offA, la + 2 * CG_BM * 4);
    cg_dma_b128(bb, offB, lbb);
    cg_dma_b128(bb + rowpair, offB, lbb + 2 * CG_BN * 4);
  };

  cg_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  __syncthreads();                       // zero fill complete before any DMA data can land
  stage(0, 0);
  if (p.nchunk > 1) stage(1, 1);

  for (int c = 0; c < p.nchunk; ++c) {
    // this wave's DMA of chunk c has landed (chunk c+1's 4 instructions may still be in flight)
    if (c + 1 < p.nchunk) __builtin_amdgcn_s_waitcnt(0x0F74);      // vmcnt(4)
    else __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0)
    __syncthreads();                     // everyone's has; everyone is done with chunk c-1's buffer
    if (c + 2 < p.nchunk) stage(c + 2, (c + 2) % CG_NSTAGE);
    const float* As = cg_lds + (c % CG_NSTAGE) * CG_STAGE_FLOATS + half * CG_BM + wm * 64 + l32;
    const float* Bs = cg_lds + (c % CG_NSTAGE) * CG_STAGE_FLOATS + CG_KC * CG_BM + half * CG_BN + wn * 64 + l32;
    __builtin_amdgcn_s_setprio(0);
    float a[2][2], b[2][2];
    a[0][0] = As[0]; a[0][1] = As[32]; b[0][0] = Bs[0]; b[0][1] = Bs[32];
#pragma unroll
    for (int ks = 0; ks < CG_KC / 2; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (ks + 1 < CG_KC / 2) {          // operands of the next k-step, one step ahead of the MFMAs
        a[nxt][0] = As[(ks + 1) * 2 * CG_BM];
        a[nxt][1] = As[(ks + 1) * 2 * CG_BM + 32];
        b[nxt][0] = Bs[(ks + 1) * 2 * CG_BN];
        b[nxt][1] = Bs[(ks + 1) * 2 * CG_BN + 32];
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the reads AHEAD of this step's MFMAs (hipcc sinks them)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(3);
  }

  // ---- epilogue: D layout col = lane & 31 (target element of the fragment), row = query
  //      i0 + wm*64 + 32 fi + (r & 3) + 8 (r >> 2) + 4 half ----
  const int hw1 = p.l1_msz, w1 = p.w >> 1;
  float* o0 = p.lvl0 + (long long)n * hw * hw;
  float* o1 = POOL ? p.lvl1 + (long long)n * hw * hw1 : nullptr;
#pragma unroll
  for (int fj = 0; fj < 2; ++fj) {
    const int F = nb * 4 + wn * 2 + fj;            // target fragment
    bool fok;
    int pos0, pos1 = 0;                            // level-0 / level-1 position of this lane's target
    bool pool_lane = false;
    if (TILED) {
      const int ntile = (p.h >> 2) * p.ntiles_x;
      fok = F < ntile;
      pos0 = F * 32 + l32;
      if (POOL) {
        const int ty = F / p.ntiles_x, tx = F - ty * p.ntiles_x;
        const int x = l32 & 7, y = l32 >> 3;
        pool_lane = fok && !(x & 1) && !(y & 1);
        const int y1 = ty * 2 + (y >> 1), x1 = tx * 4 + (x >> 1);      // level-1 pixel of this window
        pos1 = p.l1_pw4 ? (y1 >> 2) * p.l1_pw4 + (x1 >> 3) * 32 + (y1 & 3) * 8 + (x1 & 7) : y1 * w1 + x1;
      }
    } else {
      pos0 = F * 32 + l32;
      fok = pos0 < hw;
    }
#pragma unroll
    for (int fi = 0; fi < 2; ++fi) {
      const int ib = i0 + wm * 64 + fi * 32 + 4 * half;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = ib + (r & 3) + 8 * (r >> 2);
        float v = acc[fi][fj][r];
        v = p.exact_scale ? v * p.scale : v / p.divisor;
        if (fok && i < hw) o0[(long long)i * hw + pos0] = v;
        if (POOL) {
          // window partners: x+1 (quad_perm [1,0,3,2]), y+1 (row_ror:8 swaps lanes l <-> l^8)
          const int vi = __builtin_bit_cast(int, v);
          const float vx = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, vi, 0xB1, 0xF, 0xF, false));
          const float vy = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, vi, 0x128, 0xF, 0xF, false));
          const float vxy = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, vx), 0x128, 0xF, 0xF, false));
          const float s = (((v + vx) + vy) + vxy) * 0.25f;        // AvgPool2d's order
          if (pool_lane && i < hw) o1[(long long)i * hw1 + pos1] = s;
        }
      }
    }
  }
}

// levels[0] (and levels[1] when pooled) of the pyramid.  SCF_EUNSUPPORTED: the caller falls back
// to the generic path (register-staged kernel + separate pool).
static int corr_gemm_dispatch(const float* feat1, const float* feat2, float* lvl0, float* lvl1, int l1_tiled,
                              int N, int C, int h, int w, int tiled, hipStream_t st) {
  const long long hw = (long long)h * w;
  if (C % CG_KC != 0 || (hw & 3) != 0 || hw > 0x3fffffffLL) return SCF_EUNSUPPORTED;
  if ((long long)C * hw * 4 > 0x7fffffffLL) return SCF_EUNSUPPORTED;          // 32-bit lane byte offsets
  if (tiled && ((w & 7) || (h & 3))) return SCF_EUNSUPPORTED;
  if (lvl1 && (!tiled || h < 2 || w < 2)) return SCF_EUNSUPPORTED;
  if ((((uintptr_t)feat1 | (uintptr_t)feat2) & 15) != 0) return SCF_EUNSUPPORTED;
  CorrGemmParams p;
  p.f1 = feat1; p.f2 = feat2; p.lvl0 = lvl0; p.lvl1 = lvl1;
  p.C = C; p.h = h; p.w = w; p.hw = (int)hw;
  p.nchunk = C / CG_KC;
  p.mblocks = (int)((hw + CG_BM - 1) / CG_BM);
  p.tiled = tiled ? 1 : 0;
  p.ntiles_x = w >> 3;
  const long long nfrag = tiled ? (long long)(h >> 2) * (w >> 3) : (hw + 31) / 32;
  p.nblocks = (int)((nfrag + 3) / 4);
  const float sq = sqrtf((float)C);
  int ex = 0;
  p.exact_scale = (frexpf(sq, &ex) == 0.5f) ? 1 : 0;       // sqrt(C) is a power of two
  p.scale = 1.0f / sq;
  p.divisor = sq;
  p.l1_pw4 = (lvl1 && l1_tiled) ? (((w >> 1) + 7) / 8 * 8) * 4 : 0;
  p.l1_msz = lvl1 ? (int)scf_corr_level_floats(h, w, 1, l1_tiled) : 0;
  const long long nblk = (long long)N * p.mblocks * p.nblocks;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return SCF_EUNSUPPORTED;
  const size_t lds = (size_t)CG_NSTAGE * CG_STAGE_FLOATS * sizeof(float);      // 48 KB
  if (tiled) {
    if (lvl1) scf_launch((corr_gemm_kernel<true, true>), dim3((unsigned)nblk), dim3(256), lds, st, p);
    else scf_launch((corr_gemm_kernel<true, false>), dim3((unsigned)nblk), dim3(256), lds, st, p);
  } else {
    scf_launch((corr_gemm_kernel<false, false>), dim3((unsigned)nblk), dim3(256), lds, st, p);
  }
  return scf_launch_status();
}

// ---------------------------------------------------------------------------------
// Correlation volume + pyramid: CorrelationPyramid.forward, raft_decoder.py:35-58.
// level 0 = the GEMM above (level 1 from the same fragments when level 0 is tiled); shapes the GEMM
// does not take (C % 16, hw % 4, unaligned pointers) run as a 1x1 convolution with per-sample
// "weights" feat1[n] ([C][hw] is already the packed [K][M] layout) divided by sqrt(C); the remaining
// levels are cascaded 2x2 average pools in the layout the caller names per level.
// ---------------------------------------------------------------------------------
int scf_avgpool2x2_layout(const float* x, float* out, int64_t planes, int Hin, int Win, int in_tiled,
                          int out_tiled, hipStream_t st);

extern "C" int scf_corr_build_ex(const float* feat1, const float* feat2, float* const* levels, int N,
                                 int C, int h, int w, int L, unsigned tiled_levels, scf_stream_t stream) {
  if (!feat1 || !feat2 || !levels || N <= 0 || C <= 0 || h <= 0 || w <= 0 || L <= 0) return SCF_EINVAL;
  if (L > SCF_MAX_LEVELS) return SCF_EUNSUPPORTED;
  if ((tiled_levels & 1u) && ((w & 7) || (h & 3))) return SCF_EUNSUPPORTED;
  // everything is validated BEFORE the first launch: no level may be empty
  for (int l = 0; l < L; ++l)
    if (!levels[l] || (h >> l) < 1 || (w >> l) < 1) return SCF_EINVAL;
  hipStream_t st = scf_stream(stream);
  const int hw = h * w;
  const int t0 = tiled_levels & 1u, t1 = (tiled_levels >> 1) & 1u;
  float* l1 = (t0 && L >= 2) ? levels[1] : nullptr;           // first pool fused into the GEMM epilogue
  int rc = corr_gemm_dispatch(feat1, feat2, levels[0], l1, t1, N, C, h, w, t0, st);
  int done = (rc == SCF_OK && l1) ? 2 : 1;
  if (rc == SCF_EUNSUPPORTED) {
    scf_conv_desc d = {};
    d.in0 = feat2; d.C0 = C; d.in0_nstride = (int64_t)C * hw;
    d.N = N; d.H = h; d.W = w;
    d.wp = feat1; d.w_nstride = (int64_t)C * hw; d.Mld = hw; d.Cout = hw;
    d.KH = d.KW = 1; d.stride = 1; d.pad_h = d.pad_w = 0;
    d.KC = (C % 32 == 0) ? 32 : (C % 8 == 0) ? 8 : 2;
    d.out = levels[0]; d.out_nstride = (int64_t)hw * hw;
    d.out_div = sqrtf((float)C);
    d.act = SCF_ACT_NONE; d.mode = SCF_CONV_PLAIN;
    d.out_tile8x4 = t0;
    rc = scf_conv2d(&d, stream);
    done = 1;
  }
  if (rc != SCF_OK) return rc;
  for (int l = done; l < L; ++l) {
    rc = scf_avgpool2x2_layout(levels[l - 1], levels[l], (int64_t)N * hw, h >> (l - 1), w >> (l - 1),
                               (tiled_levels >> (l - 1)) & 1u, (tiled_levels >> l) & 1u, st);
    if (rc != SCF_OK) return rc;
  }
  return SCF_OK;
}

extern "C" int scf_corr_build(const float* feat1, const float* feat2, float* const* levels, int N,
                              int C, int h, int w, int L, scf_stream_t stream) {
  return scf_corr_build_ex(feat1, feat2, levels, N, C, h, w, L, 0u, stream);
}
