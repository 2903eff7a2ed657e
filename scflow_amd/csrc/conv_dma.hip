// Stride-1 / stride-2 fp32 convolution, second generation of the implicit-GEMM kernel in conv_mfma.hip
// (same GEMM view, same v_mfma_f32_32x32x2_f32 arithmetic and k order, same fused epilogue):
//
//   * operands are staged by LDS-DMA (global_load_lds): memory -> LDS without passing through
//     VGPRs, into a DOUBLE-BUFFERED chunk area -- chunk c+1 streams in while chunk c is on the
//     matrix cores, one workgroup barrier per chunk, no staging registers;
//   * the k dimension inside a chunk is interleaved four-deep: channel c = 8g + 2s + h sits at
//     float s of cell (g, h), for the weights   [tap][g][h][cout][4]
//                              and for the patch [g][h][PH][PW][4],
//     so ONE ds_read_b128 per fragment feeds four consecutive MFMA k-steps (lane half h takes
//     channel 2s+h at step s, exactly the A[l&31][l>>5] / B[l>>5][l&31] operand layout);
//   * those reads are software-pipelined one (tap, group) step ahead of the MFMAs.
// The MFMA phase alone runs at 138-147 TFLOP/s from LDS (tools/probes/mfma_loop_probe.hip) vs
// 122-131 for the b32 / read-then-use loop.
//
// Zero padding: the patch areas are zero-filled once per block and out-of-image positions are
// never written afterwards (the gather table is chunk-invariant).
#include <stdlib.h>
#include "scf_common.h"
#include <atomic>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#include "conv_kernels.h"
#include "scf_dma.h"
#include "conv_taps_body.h"

#define SCF_DMA_PU 20   // patch gathers per thread per chunk (256 * 20 floats)
#define SCF_DMA_WU 7    // weight float4 per thread per chunk
#define SCF_DMA_PU_KSP 24   // K-split tile (small grids: 32-channel chunks, one wave per SIMD: registers are free)
#define SCF_DMA_WU_KSP 9
#define SCF_PX4_MODE 3      // aligned-x4 patch staging: bit 0 full-grid tiles, bit 1 small-grid tiles
#define SCF_DMA_PU_X4 8     // PX4: float4 patch cells per thread per chunk (256 * 8 * 4 floats)
#define SCF_DMA_LDS_MAX (80 * 1024)   // two blocks per CU (160 KB)

// LDS-DMA through raw buffer descriptors: scf_dma.h (scf_make_rsrc, scf_bdma_b128 / _b32; a lane whose offset
// is >= num_records writes ZEROS to its LDS cell, so zero padding, out-of-image positions and the channels
// past the end of a short last chunk need no EXEC mask, no pre-zeroed LDS and no special path)
typedef scf_rsrc4 scf_rsrc_t;
#define SCF_DMA_OOB SCF_BUF_OOB
__device__ __forceinline__ unsigned lds_addr(const void* p) { return scf_lds_addr(p); }
__device__ __forceinline__ scf_rsrc_t make_rsrc(const void* base, unsigned bytes) { return scf_make_rsrc(base, bytes); }
__device__ __forceinline__ void bdma_b128(scf_rsrc_t rsrc, unsigned voff, unsigned lds_base) { scf_bdma_b128(rsrc, voff, lds_base); }
__device__ __forceinline__ void bdma_b32(scf_rsrc_t rsrc, unsigned voff, unsigned lds_base) { scf_bdma_b32(rsrc, voff, lds_base); }
// the first n (0 < n < 64) lanes only: the last, partial slot of an LDS area
__device__ __forceinline__ void bdma_b128_n(scf_rsrc_t rsrc, unsigned voff, unsigned lds_base, int n) {
  asm volatile("s_bfm_b64 exec, %3, 0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "buffer_load_dwordx4 %1, %0, 0 offen lds\n\ts_mov_b64 exec, -1"
               : : "s"(rsrc), "v"(voff), "s"(lds_base), "s"(n) : "memory");
}
__device__ __forceinline__ void bdma_b32_n(scf_rsrc_t rsrc, unsigned voff, unsigned lds_base, int n) {
  asm volatile("s_bfm_b64 exec, %3, 0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "buffer_load_dword %1, %0, 0 offen lds\n\ts_mov_b64 exec, -1"
               : : "s"(rsrc), "v"(voff), "s"(lds_base), "s"(n) : "memory");
}
// one slot of an area of `count` cells: this wave's 64 lanes hold cells [first, first + 64)
template <bool X4>
__device__ __forceinline__ void bdma_slot(scf_rsrc_t rsrc, unsigned voff, unsigned lds_base, int rem) {
  if (rem >= 64) {
    if (X4) bdma_b128(rsrc, voff, lds_base); else bdma_b32(rsrc, voff, lds_base);
  } else if (rem > 0) {
    if (X4) bdma_b128_n(rsrc, voff, lds_base, rem); else bdma_b32_n(rsrc, voff, lds_base, rem);
  }
}

// a pointer the compiler keeps in SGPRs (block-uniform by construction)
__device__ __forceinline__ const void* sgpr_ptr(const void* p) {
  const unsigned long long u = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return (const void*)(((unsigned long long)hi << 32) | lo);
}

// Same, with the lane mask derived from the offset itself (0xFFFFFFFF = this lane stays off): one
// v_cmp instead of a lane mask kept in (spilled) SGPRs -- VALU issue slots are scarce while the
// co-resident wave streams MFMAs.
__device__ __forceinline__ void dma_b128_v(const void* sbase, unsigned voff, unsigned lds_base) {
  asm volatile("v_cmp_ne_u32_e32 vcc, -1, %1\n\ts_mov_b64 exec, vcc\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %0\n\ts_mov_b64 exec, -1"
               : : "s"(sbase), "v"(voff), "s"(lds_base) : "memory", "vcc");
}
__device__ __forceinline__ void dma_b32_v(const void* sbase, unsigned voff, unsigned lds_base) {
  asm volatile("v_cmp_ne_u32_e32 vcc, -1, %1\n\ts_mov_b64 exec, vcc\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "global_load_lds_dword %1, %0\n\ts_mov_b64 exec, -1"
               : : "s"(sbase), "v"(voff), "s"(lds_base) : "memory", "vcc");
}

// floor(e / d) for 0 <= e < 2^22, 0 < d < 2^12 without the integer-division expansion
__device__ __forceinline__ int fast_div(int e, int d, float rd) {
  int q = (int)((float)e * rd);
  const int r = e - q * d;
  q += (r >= d) ? 1 : 0;
  q -= (r < 0) ? 1 : 0;
  return q;
}

// tools/lab/conv_trace*.py build this file with a per-chunk timeline, tools/lab/conv_phases.py with compile-time
// phase ablations (tools/lab/conv_lab_hooks.h, -DSCF_CONV_LAB [-DSCF_CONV_LAB_MASK=m]); the product build sees
// empty hooks
#ifdef SCF_CONV_LAB
#include "conv_lab_hooks.h"      // lab builds only: -I tools/lab
#else
#define CTRACE(slot) do { } while (0)
#define CLAB(bit) 0
#endif

// s_waitcnt vmcnt(n) for a wave-uniform run-time n: scf_wait_vmcnt_le (scf_dma.h)
__device__ __forceinline__ void wait_vmcnt_le(int n) { scf_wait_vmcnt_le(n); }

// NST   : chunk buffers in the LDS ring.  2 = chunk c+1 streams in while chunk c is on the matrix
//         cores (large grids: co-resident blocks hide each other's latency).  Small grids (batch 1:
//         one block per CU, nothing else to hide a ~2 us memory round trip per chunk) use a deeper
//         ring: NST-1 chunks in flight per block.
// KSP   : K-split tile for small grids: 32 channels x ONE 32-pixel fragment per block (4x the
//         blocks of the smallest pixel-split tile); the four waves take every fourth (tap, group)
//         step of each chunk and combine their partial sums through LDS in a fixed order.
// PX4   : (W % 4 == 0) the patch is staged as PLAIN full-resolution channel planes [KC][PH][PWa] with
//         dwordx4 DMA: a patch row starts at the 16-byte-aligned column ixa = ix0 - px_off, so every
//         lane moves one aligned group of 4 columns that lies wholly inside or wholly outside the
//         image (4x fewer patch DMA instructions, contiguous instead of 4-plane interleaved gathers:
//         ~125 vs 4 x 117 cycles of texture-path time per KiB).  B operands are then read with four
//         ds_read_b32 per (tap, group) step instead of one ds_read_b128 (stride 2: every second
//         column, a 2-way bank conflict the MFMA-bound loop does not notice).
// NG    : (K-split tile only, r4) wave GROUPS per block: group g stages and contracts chunks g, g + NG, ... in its own
//         double buffer, so a block's serial chain of "one memory round trip per chunk" is 1 / NG as long -- the
//         split of K across more waves of the SAME block (the partial sums already meet in LDS at the end), for
//         grids with at most one block per CU (batch 1: a launch was 8 ... 16 such round trips whatever it computed).
// The kernel's body as a function of (its arguments, its block index, its grid size): conv_dma_kernel runs it with the launch's
// own blockIdx / gridDim, conv_dma_pair_kernel (r6, below) runs TWO independent layers' grids in one launch.
template <int WM, int WN, int NST = 2, bool KSP = false, bool PX4 = false, int NG = 1>
__device__ __forceinline__ void conv_dma_body(ConvK p, const int bid, const int nblk) {
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  static_assert(!KSP || (WM == 1 && WN == 1), "K-split tile is one 32x32 fragment");
  static_assert(NG == 1 || (KSP && NST == 2), "wave groups: K-split tile, double buffer");
  constexpr int BM = WM * 32;
  constexpr int NFRAG = KSP ? 1 : WN * 4;
  constexpr int PU = PX4 ? SCF_DMA_PU_X4 : KSP ? SCF_DMA_PU_KSP : SCF_DMA_PU;

  __builtin_amdgcn_s_setprio(3);       // setup / staging / epilogue instructions go first
  CTRACE(0);
  const int tid = threadIdx.x & 255, lane = tid & 63;            // thread / wave index inside the wave group
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6) & 3);
  const int grp = NG == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
  const int l32 = lane & 31, half = lane >> 5;

  // block-uniform tile coordinates, pinned to SGPRs (the integer divisions are expanded on the
  // vector ALU and would otherwise leave n / m0 / the tile origin -- and every pointer derived
  // from them -- in VGPRs)
  int lb = scf_xcd_remap(bid, nblk);
  // K split across blocks (r5): slice ksl contracts chunks [cb, cb + nch) and stores raw partial sums into its own
  // output tensor; the consumer adds the slices in order.  kslices <= 1: one slice = everything.
  int cb = 0, nch = p.nchunk;
  if (p.kslices > 1) {
    const int ksl = __builtin_amdgcn_readfirstlane(lb / p.slice_blocks);
    lb -= ksl * p.slice_blocks;
    const int cps = (p.nchunk + p.kslices - 1) / p.kslices;
    cb = ksl * cps;
    nch = min(cps, p.nchunk - cb);
    nch = nch < 0 ? 0 : nch;
    p.out += (long long)ksl * p.slice_ns;
  }
  const int mblk = __builtin_amdgcn_readfirstlane(lb % p.mblocks);
  const int tile = __builtin_amdgcn_readfirstlane(lb / p.mblocks);
  const int m0 = mblk * BM;

  const int FC = 1 << p.fc_log2, FR = 32 >> p.fc_log2, TR = NFRAG * FR;
  const int txi = __builtin_amdgcn_readfirstlane(tile % p.tiles_x);
  const int t2 = tile / p.tiles_x;
  const int tyi = __builtin_amdgcn_readfirstlane(t2 % p.tiles_y);
  const int n = __builtin_amdgcn_readfirstlane(t2 / p.tiles_y);
  const int ty0 = tyi * TR, tx0 = txi * FC;
  const int st = p.stride;             // 1 or 2
  const int iy0 = ty0 * st - p.pad_h, ix0 = tx0 * st - p.pad_w;
  // PW = LDS row pitch of the patch.  Stride 2: a row is stored even columns first, then odd
  // columns (PWh each), so that the 32 lanes of a fragment read consecutive float4 cells for
  // every tap (no bank conflicts): input column 2*fc + kx lives at (kx&1)*PWh + fc + (kx>>1).
  const int PW = p.PW, PHW = p.PH * p.PW, PWh = PW >> 1;
  const int T = p.T, G = p.G4, KC = 8 * G;
  const int NIT = T * G;               // (tap, group) steps per chunk, 4 k-steps each
  const int WF4 = NIT * 2 * BM;        // weight float4 per chunk
  const int PE = KC * PHW;             // patch floats per chunk
  const int bufsz = WF4 * 4 + PE;      // floats per buffer (multiple of 4)
  float* const lds = lds_all + grp * (NST * bufsz);      // this wave group's ring

  // weights: float4 e = tid + 256u of the chunk's [NIT*2 rows][BM] slab out of [rows][Mld4]
  constexpr int WU = KSP ? SCF_DMA_WU_KSP : SCF_DMA_WU;
  unsigned woff[WU];
#pragma unroll
  for (int u = 0; u < WU; ++u) {
    const int e = tid + u * 256;
    const int row = e / BM, m = e - row * BM;
    const bool ok = e < WF4 && m0 + m < p.Mld4;
    woff[u] = ok ? (unsigned)((row * p.Mld4 + m) * 16) : SCF_DMA_OOB;
  }
  const long long wrow = (long long)p.Mld4 * 4;      // floats per (chunk, tap, g, h) weight row
  const long long wslab = (long long)NIT * 2 * wrow; // floats per chunk of packed weights
  const unsigned wbytes = (unsigned)(((long long)(NIT * 2 - 1) * p.Mld4 + min(BM, p.Mld4 - m0)) * 16);
  const int wrem0 = WF4 - wave * 64;                 // weight cells from this wave's first lane on, slot 0
  // a chunk's weight slab: WU pieces per wave.  Chunk 0's go out right here, before the patch table
  // below is computed (~1.5 us of integer arithmetic the first memory round trip can hide behind).
  auto stage_w = [&](int chunk, int b) {
    const scf_rsrc_t wrs = make_rsrc(p.wp4 + (long long)chunk * wslab + (long long)m0 * 4, wbytes);
    int wrem = wrem0;
    asm volatile("" : "+s"(wrem));
    const unsigned wl0 = lds_addr(lds + b * bufsz) + wave * 1024;
#pragma unroll
    for (int u = 0; u < WU; ++u)
      bdma_slot<true>(wrs, woff[u], wl0 + u * 4096, wrem - u * 256);
  };
  CTRACE(1);
  if (grp < nch) stage_w(cb + grp, 0);
  __builtin_amdgcn_sched_barrier(0);

  // ---- gather table: LDS patch float e = tid + 256u <-> (group g, half h, py, px, s) ----
  // Chunk-invariant: byte offset from the chunk's first channel plane; out-of-image positions get
  // SCF_DMA_OOB (the descriptor's range check writes zeros there).
  const int HWin = p.H * p.W;
  const float rPHW = 1.0f / (float)PHW, rPW = 1.0f / (float)PW;
  unsigned toff[PU];
#pragma unroll
  for (int u = 0; u < PU; ++u) {
    const int e = tid + u * 256;
    bool ok = false;
    unsigned o = 0;
    if (PX4) {                         // e = float4 cell: (channel c, row py, aligned column group p4)
      if (u * 1024 < PE) {
        const int PW4 = PW >> 2, PHW4 = PHW >> 2;
        const int c = fast_div(e, PHW4, 1.0f / (float)PHW4), r = e - c * PHW4;
        const int py = fast_div(r, PW4, 1.0f / (float)PW4), p4 = r - py * PW4;
        const int iy = iy0 + py, ix = ix0 - p.px_off + 4 * p4;
        ok = e * 4 < PE && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        o = (unsigned)(c * HWin + iy * p.W + ix) * 4u;
      }
    } else if (u * 256 < PE) {
      const int s = e & 3, q = e >> 2;
      const int gh = fast_div(q, PHW, rPHW), r = q - gh * PHW;
      const int py = fast_div(r, PW, rPW), pxs = r - py * PW;
      const int px = st == 1 ? pxs : (pxs >= PWh ? 2 * (pxs - PWh) + 1 : 2 * pxs);
      const int c = 8 * (gh >> 1) + 2 * s + (gh & 1);
      const int iy = (iy0 + py) * p.in_step, ix = (ix0 + px) * p.in_step;      // in_step 2: the dilated gather of a 1x1 / s2 layer
      ok = e < PE && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && px < p.PWin;
      o = (unsigned)(c * HWin + iy * p.W + ix) * 4u;
    }
    toff[u] = ok ? o : SCF_DMA_OOB;
  }

  const int fr = l32 >> p.fc_log2, fc = l32 & (FC - 1);
  int boff[WN];                        // float4 index of this lane's pixel, fragment j, tap (0,0)
#pragma unroll
  for (int j = 0; j < WN; ++j)
    boff[j] = ((KSP ? 0 : (wave * WN + j)) * FR + fr) * st * PW + half * PHW + (PX4 ? fc * st + p.px_off : fc);
  // (PX4: FLOAT index into the plain planes; lane half h reads channel 2s + h: + h * PHW)

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  f32x16 acc2;                         // K-split tile: second accumulator (odd steps of this wave)
#pragma unroll
  for (int r = 0; r < 16; ++r) acc2[r] = 0.f;

  // GRU launches: the pre-activation term (`res`: the hoisted context part) goes into the
  // accumulators NOW, while the first chunks are still on their way, instead of being read in
  // the epilogue -- a launch of one round of blocks has all its epilogues at the same moment, and
  // every byte they read or write there is unoverlapped HBM time at the end of the kernel.
  if (p.res && (p.mode == SCF_CONV_GRU_ZR || p.mode == SCF_CONV_GRU_Q) && p.out_div == 1.0f && !p.out_tile) {
    if (!KSP || (wave == 0 && grp == 0)) {
      const float* rn = p.res + (long long)n * p.res_ns;
      const int HWo = p.Ho * p.Wo;
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int oy = ty0 + (KSP ? 0 : (wave * WN + j)) * FR + fr, ox = tx0 + fc;
        if (oy < p.Ho && ox < p.Wo) {
          const int pj = oy * p.Wo + ox;
#pragma unroll
          for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int co = m0 + 32 * i + 8 * (r >> 2) + (r & 3) + 4 * half;
              if (co < p.Cout) acc[i][j][r] = rn[co * HWo + pj];
            }
        }
      }
    }
    p.res = nullptr;                   // consumed: the epilogue adds nothing
  }

  const float* in0n = p.in0 + (long long)n * p.in0_ns;
  const float* in1n = p.in1 ? p.in1 + (long long)n * p.in1_ns : nullptr;

  // cells of the patch / weight areas (a cell = one lane's DMA unit) and this wave's share of them
  const int pcells = PX4 ? PE >> 2 : PE;
  const int prem0 = pcells - wave * 64;              // cells from this wave's first lane on, slot 0

  auto stage_p = [&](int chunk, int b) {
    float* pb = lds + b * bufsz + WF4 * 4;
    const int c0 = chunk * KC;
    const float* base;
    int nvalid;
    if (c0 < p.C0) { base = in0n + (long long)c0 * HWin; nvalid = p.C0 - c0; }
    else { base = in1n + (long long)(c0 - p.C0) * HWin; nvalid = p.Cin - c0; }
    // channels past the end of a segment's last chunk fall outside the descriptor: zeros
    const scf_rsrc_t prs = make_rsrc(base, (unsigned)min(nvalid, KC) * (unsigned)HWin * 4u);
    // slot counts re-materialised per call (scalar compares per site): left to itself hipcc hoists
    // the loop-invariant guards out of the chunk loop as 64-bit masks, spills them, and reloads
    // each with two v_readlane -- VALU slots the co-resident wave's MFMA stream leaves scarce
    int prem = prem0;
    asm volatile("" : "+s"(prem));
    const unsigned pl0 = lds_addr(pb) + wave * (PX4 ? 1024 : 256);
#pragma unroll
    for (int u = 0; u < PU; ++u)
      bdma_slot<PX4>(prs, toff[u], pl0 + u * (PX4 ? 4096 : 1024), prem - u * 256);
  };
  auto stage = [&](int chunk, int b) {   // weights first (chunk 0: already out), then the patch
    stage_w(chunk, b);
    stage_p(chunk, b);
  };

  // DMA instructions this wave issues per chunk (the same for every chunk): vmcnt bookkeeping
  const int cnt = __builtin_amdgcn_readfirstlane(max(0, (prem0 + 255) >> 8) + max(0, (wrem0 + 255) >> 8));

  if (grp < nch) stage_p(cb + grp, 0);   // (its weights went out before the table)
#pragma unroll
  for (int c = 1; c < NST - 1; ++c)
    if (c < nch) stage(cb + c, c);
  CTRACE(2);

  int buf = 0;                         // ring slot of the current chunk
  // wave group g walks chunks g, g + NG, ...; every group runs the same number of iterations (the barriers are the
  // block's), an iteration past the group's last chunk does nothing between them
  const int niter = (nch + NG - 1) / NG;
  for (int iter = 0; iter < niter; ++iter) {
    const int chunk = iter * NG + grp;
    const bool live = NG == 1 || chunk < nch;
    __builtin_amdgcn_s_setprio(3);
    // this wave's DMA of THIS chunk has landed; up to NST-2 later chunks stay in flight
    if (NST == 2) {
      __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0)
    } else {
      const int later = min(NST - 2, nch - 1 - chunk);
      wait_vmcnt_le(later * cnt);
    }
    CTRACE(4 + chunk * 4);
    __syncthreads();                                   // everyone's has; previous MFMA phase done
    CTRACE(5 + chunk * 4);
    if (!CLAB(0) && chunk + NG * (NST - 1) < nch) stage(cb + chunk + NG * (NST - 1), buf == 0 ? NST - 1 : buf - 1);
    CTRACE(6 + chunk * 4);
    __builtin_amdgcn_s_setprio(0);                     // the MFMA stream yields to the other waves
    if (!live) continue;

    const f32x4* wl = reinterpret_cast<const f32x4*>(lds + buf * bufsz) + half * BM + l32;
    const f32x4* pl = reinterpret_cast<const f32x4*>(lds + buf * bufsz + WF4 * 4);
    if (++buf == NST) buf = 0;
    // dense 1x1 (T == 1: a step = one group of 8 channels): a short last chunk of a segment runs only the groups that
    // hold channels (324 -> 256: 10 chunks of 32 + one of 4 = 41 steps instead of 44); the skipped steps would
    // multiply the zeros the range check staged
    int NITc = NIT;
    if (T == 1) {
      const int c0 = (cb + chunk) * KC;
      const int nv = c0 < p.C0 ? p.C0 - c0 : p.Cin - c0;
      NITc = __builtin_amdgcn_readfirstlane(min(G, (nv + 7) >> 3));
    }
    if (KSP) {
      // wave w takes the (tap, group) steps it == w (mod 4): one ds_read_b128 pair feeds 4 MFMAs
      // G is 1, 2 or 4: four steps ahead is the same group g, 4 / G taps further.  The operands of
      // the next step are read while this step's MFMAs run, and consecutive steps alternate between
      // two accumulators (one wave per SIMD here: a single dependent MFMA chain would leave the
      // matrix pipe waiting for its own result).
      const int gshift = G == 4 ? 2 : G == 2 ? 1 : 0, g = wave & (G - 1), tstep = 4 >> gshift;
      int ky = 0, kx = wave >> gshift;
      while (kx >= p.KW) { kx -= p.KW; ++ky; }
      const f32x4* pg = pl + g * 2 * PHW + boff[0];
      const float* pgf = reinterpret_cast<const float*>(pl) + g * 8 * PHW + boff[0];   // PX4
      auto opnd = [&](int it, f32x4& aa, f32x4& bb) {
        aa = wl[it * 2 * BM];
        if (PX4) {
          const float* q = pgf + ky * PW + kx;
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) bb[s4] = q[2 * s4 * PHW];
        } else {
          bb = pg[ky * PW + (st == 1 ? kx : (kx & 1) * PWh + (kx >> 1))];
        }
        kx += tstep;
        while (kx >= p.KW) { kx -= p.KW; ++ky; }
      };
      f32x4 a0, b0, a1, b1;
      int it = wave;
      const int NIT = NITc;                  // (shadows the full count: a short last chunk of a 1x1 layer)
      if (it < NIT) opnd(it, a0, b0);
      while (it < NIT) {
        if (it + 4 < NIT) opnd(it + 4, a1, b1);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s4], b0[s4], acc[0][0], 0, 0, 0);
        it += 4;
        if (it >= NIT) break;
        if (it + 4 < NIT) opnd(it + 4, a0, b0);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
          acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s4], b1[s4], acc2, 0, 0, 0);
        it += 4;
      }
      CTRACE(7 + chunk * 4);
      continue;
    }
    f32x4 a[2][WM], b[2][WN];
    int lg = 0, lky = 0, lkx = 0;                     // (tap, group) of the next operand load
    auto load = [&](f32x4 (&aa)[WM], f32x4 (&bb)[WN], int it) {
      const f32x4* wt = wl + it * 2 * BM;
#pragma unroll
      for (int i = 0; i < WM; ++i) aa[i] = wt[i * 32];
      if (PX4) {
        const float* ptf = reinterpret_cast<const float*>(pl) + lg * 8 * PHW + lky * PW + lkx;
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) bb[j][s4] = ptf[boff[j] + 2 * s4 * PHW];
      } else {
        const f32x4* pt = pl + lg * 2 * PHW + lky * PW + (st == 1 ? lkx : (lkx & 1) * PWh + (lkx >> 1));
#pragma unroll
        for (int j = 0; j < WN; ++j) bb[j] = pt[boff[j]];
      }
      if (++lg == G) {
        lg = 0;
        if (++lkx == p.KW) { lkx = 0; ++lky; }
      }
    };
    auto mma = [&](const f32x4 (&aa)[WM], const f32x4 (&bb)[WN]) {
      if (CLAB(1)) { asm volatile("" : : "v"(aa[0]), "v"(bb[0])); return; }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[i][s], bb[j][s], acc[i][j], 0, 0, 0);
    };
    // Steady state without conditions around the loads: with a conditional load the compiler's
    // wait-count insertion falls back to lgkmcnt(0) at the join, i.e. it waits for the operands it
    // has just requested for the NEXT step before starting this step's MFMAs.
    load(a[0], b[0], 0);
    int it = 0;
    for (; it + 2 < NITc; it += 2) {
      load(a[1], b[1], it + 1);
      mma(a[0], b[0]);
      load(a[0], b[0], it + 2);
      mma(a[1], b[1]);
    }
    if (it + 1 < NITc) {
      load(a[1], b[1], it + 1);
      mma(a[0], b[0]);
      mma(a[1], b[1]);
    } else {
      mma(a[0], b[0]);
    }
    CTRACE(7 + chunk * 4);
  }

  if (KSP) {
    // ---- cross-wave reduction (fixed order) + epilogue: wave w finalises accumulator rows 4w..4w+3 ----
    __builtin_amdgcn_s_setprio(3);
    __syncthreads();                          // every wave is done reading the ring
    float* red = lds_all;                     // [4 NG waves][16 regs][64 lanes] = 16 KB per wave group
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((grp * 4 + wave) * 16 + r) * 64 + lane] = acc[0][0][r] + acc2[r];
    __syncthreads();
    if (grp != 0) return;                     // group 0 adds the partial tiles in wave order and finishes them
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = 4 * wave + q;
      float sum = red[r * 64 + lane];
#pragma unroll
      for (int w = 1; w < 4 * NG; ++w) sum += red[(w * 16 + r) * 64 + lane];
      v[q] = sum;
    }
    const int oy = ty0 + fr, ox = tx0 + fc;
    if (oy < p.Ho && ox < p.Wo) {
      const ConvEpi epi = scf_conv_epi(p, n);
      const int pixk = p.out_tile ? (((oy >> 2) * (p.Wo >> 3) + (ox >> 3)) * 32 + (oy & 3) * 8 + (ox & 7))
                                  : oy * p.Wo + ox;
      // accumulator register r <-> channel row (r & 3) + 8 (r >> 2) + 4 half: registers 4w..4w+3 are
      // the four consecutive rows 8w + 4 half .. + 3
      scf_conv_epilogue_group(p, epi, v, m0 + 8 * wave + 4 * half, pixk, p.out_div != 1.0f);
    }
    CTRACE(3);
    return;
  }

  // ---- epilogue: C/D layout col = lane&31 (pixel), row = (reg&3) + 8*(reg>>2) + 4*half ----
  __builtin_amdgcn_s_setprio(3);
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
  if (!CLAB(2)) scf_conv_epilogue_tile<WM, WN>(p, epi, acc, m0, half, pix, use_div);
  CTRACE(3);
}

template <int WM, int WN, int NST = 2, bool KSP = false, bool PX4 = false, int NG = 1>
__global__ __launch_bounds__(256 * NG, (KSP || NST > 2) ? 1 : 2) void conv_dma_kernel(ConvK p) {
  conv_dma_body<WM, WN, NST, KSP, PX4, NG>(p, (int)blockIdx.x, (int)gridDim.x);
}

// r6: TWO independent small-grid layers in ONE launch (blocks [0, nba) run layer a, the rest layer b; K-split tile only).
// At batch 1-4 a layer fills a fraction of the chip and two independent layers used to run side by side on two streams;
// a hipGraph replay on this runtime pays ~1.2 us per node as soon as the graph holds a parallel branch
// (tools/lab/graph_fork_penalty.py), so the concurrency moves INTO the launch: no cross-queue dependency, one dispatch,
// the same arithmetic per layer (a block cannot tell which launch form it runs in: bit-identical outputs).
template <bool PX4, int NG>
__global__ __launch_bounds__(256 * NG, 1) void conv_dma_pair_kernel(ConvK pa, ConvK pb, int nba) {
  if ((int)blockIdx.x < nba) conv_dma_body<1, 1, 2, true, PX4, NG>(pa, (int)blockIdx.x, nba);
  else conv_dma_body<1, 1, 2, true, PX4, NG>(pb, (int)blockIdx.x - nba, (int)gridDim.x - nba);
}

#define SCF_DMA_LDS_DEEP (144 * 1024)  // tiny grids (one block per CU): 32-channel chunks of 3x3 layers

template <int WM, int WN, int NST = 2, bool KSP = false, bool PX4 = false, int NG = 1>
static int launch_dma(const ConvK& k, int nblk, size_t lds_bytes, hipStream_t st) {
  if (lds_bytes > 64 * 1024) {       // opt in to > 64 KiB of dynamic LDS: once per instantiation AND device
    static std::atomic<unsigned long long> raised{0};      // bit d: done on device d
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return SCF_ELAUNCH;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(raised.load(std::memory_order_relaxed) & bit)) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_dma_kernel<WM, WN, NST, KSP, PX4, NG>),
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (NST == 2 && !KSP) ? SCF_DMA_LDS_MAX : SCF_DMA_LDS_DEEP) != hipSuccess)
        return SCF_ELAUNCH;
      raised.fetch_or(bit, std::memory_order_relaxed);
    }
  }
  scf_launch((conv_dma_kernel<WM, WN, NST, KSP, PX4, NG>), dim3(nblk), dim3(256 * NG), lds_bytes, st, k);
  return scf_launch_status();
}

// Tile selection + launch.  k comes from conv_plan() (geometry fields are overwritten here).
// SCF_EUNSUPPORTED -> the caller falls back to the register-staged kernel (thin inputs, 7x7,
// shapes that exceed the DMA kernel's staging budget).
//   large grids  (>= 2 blocks per CU): pixel-split tiles, double buffer (co-resident blocks overlap)
//   small grids  : the K-split tile (32 channels x 32 pixels, 4x the blocks), double buffer too
// measurement knob (scf_tune(SCF_TUNE_DMA_FORCE_KSPLIT, 1)): every launch takes the K-split tile (32 channels x 32
// pixels per block: each block stages the full weight slab of its 32 output channels for 32 pixels)
static std::atomic<int> g_force_ksp{0};
int scf_dma_force_ksplit_set(int v) { return g_force_ksp.exchange(v); }
// scf_tune(SCF_TUNE_DMA_KSPLIT_GROUPS, 1): K-split blocks keep ONE wave group (the r3 kernel) on every grid
static std::atomic<int> g_ksp_groups{0};
int scf_dma_ksplit_groups_set(int v) { return g_ksp_groups.exchange(v); }

int scf_conv_dma_dispatch(ConvK k, int N, bool dry_run, int* info, hipStream_t st, ScfLaunchCap* cap) {
  if ((!k.wp4 && !k.wp4s) || (k.stride != 1 && k.stride != 2) || k.w_ns != 0) return SCF_EUNSUPPORTED;
  // K split across blocks: S slices are S times the blocks (grid-size decisions below see N * S samples) and 1 / S of
  // every block's chunk chain
  const int S = k.kslices > 1 ? k.kslices : 1;
  const long long NE = (long long)N * S;
  // A 1x1 / stride-2 / pad-0 layer (the ResNet shortcuts, resnet.py:721-730) reads every second row and column of
  // its input and nothing else: it runs as a DENSE 1x1 over that sub-grid -- the gather table of the dword staging
  // path doubles its row / column steps (in_step), everything behind the staging sees a stride-1 layer on an
  // Ho x Wo map.  (Staged as a stride-2 patch it would move four times the floats it uses; on the register-staged
  // kernel these layers ran at 23-37 TF/s.)
  const bool dilated = k.T == 1 && k.stride == 2 && k.pad_h == 0 && k.pad_w == 0;
  if (dilated) { k.stride = 1; k.in_step = 2; }
  // byte offsets inside a staged chunk (<= 32 channel planes) are 32-bit and must stay below SCF_DMA_OOB
  if ((long long)k.H * k.W * 32 * 4 >= 0x7fffffffLL) return SCF_EUNSUPPORTED;
  const int FC = 1 << k.fc_log2, FR = 32 / FC;
  const int frags_m = (k.Cout + 31) / 32;
  int G = k.wp4 ? k.G4 : 0;
  int KC = 8 * G;
  // aligned dwordx4 patch staging (PX4): stride 1, rows and planes that keep 16-byte alignment
  const bool px4_ok = !dilated && FC * k.stride >= 4 && (k.W & 3) == 0 && (k.in0_ns & 3) == 0 &&
                      ((uintptr_t)k.in0 & 15) == 0 &&
                      (!k.in1 || ((k.in1_ns & 3) == 0 && ((uintptr_t)k.in1 & 15) == 0));
  const bool px4_large = px4_ok && (SCF_PX4_MODE & 1), px4_small = px4_ok && (SCF_PX4_MODE & 2);
  const int px_off = (4 - (k.pad_w & 3)) & 3;
  auto pitch = [&](int PWin, bool px4) {
    return px4 ? (px_off + PWin + 3) & ~3 : k.stride == 1 ? PWin : ((PWin + 1) / 2) * 2;
  };
  const bool pix_ok = k.wp4 && (G == 1 || G == 2 || G == 4) && !(k.in1 && (k.C0 % KC) != 0);
  // candidates in order of preference; WM must divide the channel fragments (no idle MFMA
  // rows) unless nothing else fits.  Among the tiles that give >= 2 blocks per CU the choice is
  // by an estimate of (slot utilisation over the launch's rounds) x (what the tile itself
  // reaches): blocks fill 2 x CUs slots per round, and a last round that leaves CUs with one or
  // no block costs a good part of a full round (a block alone on its CU runs ~1.7x as fast) --
  // 640 blocks of the (2,2) tile on a 60 x 80 map are 1.6 round-times for 1.25 rounds of work,
  // 1280 blocks of the (2,1) tile 2.6 for 2.5.  Without such a tile: the one with the most blocks.
  const int cand[4][2] = {{2, 2}, {3, 1}, {2, 1}, {1, 1}};
  const float tile_eff[4] = {1.00f, 0.97f, 0.90f, 0.75f};     // measured rate relative to the (2,2) tile
  const long long slots = 2LL * scf_cu_count();
  int best = -1;
  long long best_blk = 0;
  size_t best_lds = 0;
  float best_score = 0.f;
  for (int pass = 0; pix_ok && pass < 2 && best < 0; ++pass) {
    for (int c = 0; c < 4; ++c) {
      const int WM = cand[c][0], WN = cand[c][1];
      if (WM > frags_m) continue;
      if (pass == 0 && frags_m % WM != 0) continue;
      const int TR = WN * 4 * FR;
      const int PH = (TR - 1) * k.stride + k.KH, PWin = (FC - 1) * k.stride + k.KW;
      const int PW = pitch(PWin, px4_large);
      const long long PE = (long long)KC * PH * PW;
      const long long WF4 = (long long)k.T * G * 2 * WM * 32;
      const size_t ldsb = (size_t)(WF4 * 4 + PE) * 2 * sizeof(float);
      if (PE > (px4_large ? 1024 * SCF_DMA_PU_X4 : 256 * SCF_DMA_PU) || WF4 > 256 * SCF_DMA_WU ||
          ldsb > SCF_DMA_LDS_MAX) continue;
      const long long tiles_y = (k.Ho + TR - 1) / TR;
      const long long blk = NE * tiles_y * ((k.Wo + FC - 1) / FC) * ((frags_m + WM - 1) / WM);
      if (blk >= slots) {
        const long long full = blk / slots, rem = blk % slots;
        const float rounds = (float)full + (rem == 0 ? 0.f : rem * 2 <= slots ? 0.6f : 1.0f);
        // useful fraction of the tiles' rows / channel fragments (ragged last tile row, WM not dividing)
        const float useful = (float)k.Ho / (float)(tiles_y * TR) *
                             (float)frags_m / (float)(((frags_m + WM - 1) / WM) * WM);
        const float score = (float)blk / (rounds * (float)slots) * useful * tile_eff[c];
        if (best < 0 || best_blk < slots || score > best_score * 1.03f) {
          best = c; best_blk = blk; best_lds = ldsb; best_score = score;
        }
      } else if (best < 0 || (best_blk < slots && blk > best_blk)) {
        best = c; best_blk = blk; best_lds = ldsb;
      }
    }
  }
  const bool large = best >= 0 && best_blk >= slots;
  // dense 1x1 on a full grid: stride 1 runs here since the aligned-x4 staging (90 vs 78 TF/s for the
  // register-staged KC = 32 kernel); stride 2 would stage four times the columns it uses
  if (k.T == 1 && large && !dilated && (k.stride != 1 || !px4_large)) return SCF_EUNSUPPORTED;
  if (!pix_ok) {
    // only the small-grid packing is present (dense 1x1): it is for small grids only -- fewer than
    // 256 blocks even with the smallest pixel-split tile (32 channels x 128 pixels)
    const long long blk11 = NE * ((k.Ho + 4 * FR - 1) / (4 * FR)) * ((k.Wo + FC - 1) / FC) * frags_m;
    if (blk11 >= 256) return SCF_EUNSUPPORTED;
  }
  if (KC) k.nchunk = (k.Cin + KC - 1) / KC;
  int WM = 1, WN = 1, ngroups = 1;
  bool ksp = false, px4 = false;
  long long nblk = 0;
  size_t ldsb = 0;
  // Grids that do not fill the chip twice over take the K-split tile (32 channels x ONE 32-pixel
  // fragment per block: 4x the blocks of the smallest pixel-split tile) with the plain double buffer:
  // several small blocks per CU hide each other's memory round trips better than one block with a
  // deep ring does (measured, graph replay of get_pose: batch 1 3.84 -> 3.45 ms, batch 4 6.55 -> 5.40,
  // batch 8 8.95 -> 8.14; the 4- and 6-deep rings of round 2 lost at every batch size).
  bool use_ksp = !large || g_force_ksp.load(std::memory_order_relaxed) != 0;
  const ConvK k_in = k;
  const long long ksp_blk = NE * ((k.Ho + FR - 1) / FR) * ((k.Wo + FC - 1) / FC) * frags_m;
  // TINY grids (no more blocks than CUs: every block alone on its CU): a launch is a chain of one
  // memory round trip per staged chunk (~1.8 us each, whatever the arithmetic: 17 us for a 128 -> 128
  // 3x3 onto a 4 x 4 map), so 3x3 layers take 32-channel chunks there when the caller provides that
  // packing -- half the chunks, half the chain (batch 1: 256 -> 192 29.8 -> 21.5 us, pose-head convs
  // 17 -> 13.7 us).  With more blocks than CUs the 100 KB stages would cost the co-residency that
  // small grids live on (128 -> 512 at batch 1: 19.4 -> 23.5 us): those keep 16-channel chunks.
  // r4: on such a grid a block may also run TWO wave groups that walk alternate chunks (conv_dma_kernel NG = 2) if
  // four stages fit its LDS -- so the choice is by the length of the serial chain, chunks / groups, over the packings
  // the caller provides (ties: the bigger chunks, fewer barriers).
  const bool one_per_cu = ksp_blk <= scf_cu_count();
  const bool tiny = one_per_cu && k.wp4t && k.G4t == 4 && !(k.in1 && (k.C0 % 32) != 0);
  const int PHk = (FR - 1) * k.stride + k.KH, PWink = (FC - 1) * k.stride + k.KW;
  const int PWk = pitch(PWink, px4_small);
  auto stage_bytes = [&](int g) { return (size_t)((long long)k.T * g * 2 * 32 * 4 + (long long)8 * g * PHk * PWk) * sizeof(float); };
  auto fits_ksp = [&](int g, int ng) {
    const long long PE = (long long)8 * g * PHk * PWk, WF4 = (long long)k.T * g * 2 * 32;
    const size_t l = stage_bytes(g) * 2 * ng;
    return PE <= (px4_small ? 1024 * SCF_DMA_PU_X4 : 256 * SCF_DMA_PU_KSP) && WF4 <= 256 * SCF_DMA_WU_KSP &&
           l <= ((one_per_cu && (ng > 1 || tiny)) ? SCF_DMA_LDS_DEEP : SCF_DMA_LDS_MAX);
  };
  if (use_ksp) {
    const bool groups_ok = one_per_cu && g_ksp_groups.load(std::memory_order_relaxed) != 1;
    const float* cand_w[3] = {tiny ? k.wp4t : nullptr, k.wp4s, pix_ok ? k.wp4 : nullptr};
    const int cand_g[3] = {k.G4t, k.G4s, k.G4};
    int best_c = -1, best_chain = 1 << 30, best_ng = 1;
    for (int c = 0; c < 3; ++c) {
      const int g = cand_g[c];
      if (!cand_w[c] || !(g == 1 || g == 2 || g == 4) || (k.in1 && (k.C0 % (8 * g)) != 0)) continue;
      const int nch = ((k.Cin + 8 * g - 1) / (8 * g) + S - 1) / S;      // chunks of one slice
      const int ng = (groups_ok && nch >= 4 && fits_ksp(g, 2)) ? 2 : 1;
      if (!fits_ksp(g, ng)) continue;
      const int chain = (nch + ng - 1) / ng;
      // without wave groups the order of preference is the r3 one (tiny-grid packing, small-grid packing, the
      // full-grid one): a later candidate must be strictly shorter
      if (best_c < 0 || chain < best_chain) { best_c = c; best_chain = chain; best_ng = ng; }
    }
    if (best_c < 0) {
      use_ksp = false;
    } else {
      k.wp4 = cand_w[best_c]; k.G4 = cand_g[best_c];
      ngroups = best_ng;
    }
  }
  if (use_ksp) {
    G = k.G4; KC = 8 * G;
    k.nchunk = (k.Cin + KC - 1) / KC;
    const int PH = PHk;
    px4 = px4_small;
    const long long PE = (long long)KC * PH * PWk, WF4 = (long long)k.T * G * 2 * 32;
    const size_t stage_b = stage_bytes(G);
    ldsb = stage_b * 2 * ngroups;
    if (ldsb < (size_t)ngroups * 16 * 1024) ldsb = (size_t)ngroups * 16 * 1024;      // cross-wave reduction area
    if (PE > (px4 ? 1024 * SCF_DMA_PU_X4 : 256 * SCF_DMA_PU_KSP) || WF4 > 256 * SCF_DMA_WU_KSP ||
        ldsb > ((tiny || ngroups > 1) ? SCF_DMA_LDS_DEEP : SCF_DMA_LDS_MAX)) {
      use_ksp = false;
    } else {
      ksp = true;
      nblk = ksp_blk;
      k.PH = PH;
      k.tiles_y = (k.Ho + FR - 1) / FR;
    }
  }
  if (!use_ksp) {                                      // pixel-split tile, double buffer
    k = k_in;
    if (best < 0 || best_blk < 256) return SCF_EUNSUPPORTED;
    G = k.G4; KC = 8 * G;
    if (KC) k.nchunk = (k.Cin + KC - 1) / KC;
    WM = cand[best][0]; WN = cand[best][1];
    nblk = best_blk;
    ldsb = best_lds;
    const int TR = WN * 4 * FR;
    k.PH = (TR - 1) * k.stride + k.KH;
    k.tiles_y = (k.Ho + TR - 1) / TR;
    px4 = px4_large;
  }
  if (nblk <= 0 || nblk > 0x7fffffffLL) return SCF_EUNSUPPORTED;
  k.PWin = (FC - 1) * k.stride + k.KW;                       // input columns a tile needs
  k.PW = pitch(k.PWin, px4);                                  // LDS row pitch
  k.px_off = px4 ? px_off : 0;

  k.tiles_x = (k.Wo + FC - 1) / FC;
  k.mblocks = (frags_m + WM - 1) / WM;
  k.slice_blocks = (int)(nblk / S);
  // every slice owns at least one chunk: slice i starts at chunk i * ceil(nchunk / S), so the LAST one is empty as soon
  // as (S - 1) * ceil(nchunk / S) >= nchunk (5 chunks in 4 slices: 2 2 1 0) -- such a launch would run blocks that only
  // store zeros for the consumer to add; the caller asks again with fewer slices (ops.conv_kslices_for)
  if (S > 1 && (long long)(S - 1) * ((k.nchunk + S - 1) / S) >= k.nchunk) return SCF_EUNSUPPORTED;
  if (info) { info[0] = WM; info[1] = ksp ? ngroups : WN; info[2] = (int)nblk; info[3] = k.T * G * 4 * WM * WN / (ksp ? 4 : 1); }
  if (cap) {                // r6: hand the launch back instead of issuing it (scf_conv2d_pair)
    cap->k = k; cap->nblk = (int)nblk; cap->ldsb = ldsb;
    cap->variant = ksp ? (ngroups == 2 ? 2 : 0) + (px4 ? 1 : 0) : -1;      // K-split tile: {NG 1 | 2} x {dword | x4 patch staging}
    return SCF_OK;
  }
  if (dry_run) return SCF_OK;
#define SCF_GO(...) return px4 ? launch_dma<__VA_ARGS__, true>(k, (int)nblk, ldsb, st)            \
                               : launch_dma<__VA_ARGS__, false>(k, (int)nblk, ldsb, st)
  if (ksp && ngroups == 2) {
    return px4 ? launch_dma<1, 1, 2, true, true, 2>(k, (int)nblk, ldsb, st) : launch_dma<1, 1, 2, true, false, 2>(k, (int)nblk, ldsb, st);
  }
  if (ksp) SCF_GO(1, 1, 2, true);
#define SCF_CASE(M, Nn) if (WM == M && WN == Nn) SCF_GO(M, Nn, 2, false);
  SCF_CASE(2, 2) SCF_CASE(3, 1) SCF_CASE(2, 1) SCF_CASE(1, 1)
#undef SCF_CASE
#undef SCF_GO
  return SCF_EUNSUPPORTED;
}


// r6: a K-split layer and a THIN-INPUT layer (conv_taps_body, WM = 1) in one launch: the motion encoder's corr_net.0 (1x1 324 -> 256) and
// flow_net.0 (7x7 2 -> 128) read different tensors and feed different branches.  The thin-input blocks use the first four waves.
template <bool PX4, int NG>
__global__ __launch_bounds__(256 * NG, 1) void conv_dma_taps_pair_kernel(ConvK pa, ConvK pb, const float* __restrict__ wtb, int Kpb, int PWpb,
                                                                         int nba) {
  if ((int)blockIdx.x < nba) {
    conv_dma_body<1, 1, 2, true, PX4, NG>(pa, (int)blockIdx.x, nba);
  } else {
    if (NG > 1 && threadIdx.x >= 256) return;      // (whole waves: the body's barriers count the waves that are still alive)
    conv_taps_body<1>(pb, wtb, Kpb, PWpb, (int)blockIdx.x - nba, (int)gridDim.x - nba);
  }
}

template <bool PX4, int NG>
static int launch_dma_taps_pair(const ScfLaunchCap& a, const ScfLaunchCap& b, size_t lds_bytes, hipStream_t st) {
  if (lds_bytes > 64 * 1024) {
    static std::atomic<unsigned long long> raised{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return SCF_ELAUNCH;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(raised.load(std::memory_order_relaxed) & bit)) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_dma_taps_pair_kernel<PX4, NG>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, SCF_DMA_LDS_DEEP) != hipSuccess)
        return SCF_ELAUNCH;
      raised.fetch_or(bit, std::memory_order_relaxed);
    }
  }
  scf_launch((conv_dma_taps_pair_kernel<PX4, NG>), dim3((unsigned)(a.nblk + b.nblk)), dim3(256 * NG), lds_bytes, st, a.k, b.k, b.wt, b.Kp,
             b.PWp, a.nblk);
  return scf_launch_status();
}

// a = a captured K-split launch, b = a captured thin-input launch (one tile per block, 32 output channels per block)
int scf_conv_dma_taps_pair_launch(const ScfLaunchCap& a, const ScfLaunchCap& b, hipStream_t st) {
  if (a.variant < 0 || a.variant > 3 || b.variant != 1 || a.nblk <= 0 || b.nblk <= 0) return SCF_EUNSUPPORTED;
  const size_t lds = a.ldsb > b.ldsb ? a.ldsb : b.ldsb;
  if (lds > SCF_DMA_LDS_DEEP) return SCF_EUNSUPPORTED;
  switch (a.variant) {
    case 0: return launch_dma_taps_pair<false, 1>(a, b, lds, st);
    case 1: return launch_dma_taps_pair<true, 1>(a, b, lds, st);
    case 2: return launch_dma_taps_pair<false, 2>(a, b, lds, st);
    case 3: return launch_dma_taps_pair<true, 2>(a, b, lds, st);
  }
  return SCF_EUNSUPPORTED;
}

// r6: two captured K-split launches of the same instantiation as one launch (see conv_dma_pair_kernel)
template <bool PX4, int NG>
static int launch_dma_pair(const ScfLaunchCap& a, const ScfLaunchCap& b, size_t lds_bytes, hipStream_t st) {
  if (lds_bytes > 64 * 1024) {
    static std::atomic<unsigned long long> raised{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return SCF_ELAUNCH;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(raised.load(std::memory_order_relaxed) & bit)) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_dma_pair_kernel<PX4, NG>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, SCF_DMA_LDS_DEEP) != hipSuccess)
        return SCF_ELAUNCH;
      raised.fetch_or(bit, std::memory_order_relaxed);
    }
  }
  scf_launch((conv_dma_pair_kernel<PX4, NG>), dim3((unsigned)(a.nblk + b.nblk)), dim3(256 * NG), lds_bytes, st, a.k, b.k, a.nblk);
  return scf_launch_status();
}

int scf_conv_dma_pair_launch(const ScfLaunchCap& a, const ScfLaunchCap& b, hipStream_t st) {
  if (a.variant < 0 || a.variant != b.variant || a.nblk <= 0 || b.nblk <= 0) return SCF_EUNSUPPORTED;
  const size_t lds = a.ldsb > b.ldsb ? a.ldsb : b.ldsb;
  if (lds > SCF_DMA_LDS_DEEP) return SCF_EUNSUPPORTED;
  switch (a.variant) {
    case 0: return launch_dma_pair<false, 1>(a, b, lds, st);
    case 1: return launch_dma_pair<true, 1>(a, b, lds, st);
    case 2: return launch_dma_pair<false, 2>(a, b, lds, st);
    case 3: return launch_dma_pair<true, 2>(a, b, lds, st);
  }
  return SCF_EUNSUPPORTED;
}


// ---------------------------------------------------------------------------------------------------------------------
// K slices WITHOUT a cooperating consumer (r5): a small-grid launch is a serial chain of one memory round trip per staged
// chunk whatever it computes (batch 1: 256 -> 192 3x3 = 8 chunks = 21 us for 1.2 GFLOP).  With a workspace registered for the
// launch stream (scf_conv_workspace) scf_conv2d splits such a launch into S slices writing partial tensors to the workspace
// and one combine launch that adds them IN SLICE ORDER and runs the layer's own fused epilogue (the same code the
// convolution kernels use): same result as one launch up to the re-association of S partial sums, deterministic.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_kcombine_kernel(ConvK p, const float* __restrict__ parts, int S, long long slice_ns,
                                                            int N) {
  const int HWo = p.Ho * p.Wo, CG = (p.Cout + 3) >> 2;
  const long long total = (long long)N * CG * HWo;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int pix = (int)(idx % HWo);
    const long long t = idx / HWo;
    const int cg = (int)(t % CG), n = (int)(t / CG);
    const int cb = cg * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const float* q0 = parts + ((long long)n * p.Cout + cb) * HWo + pix;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (cb + q < p.Cout) {
        float a = q0[(long long)q * HWo];
        for (int sl = 1; sl < S; ++sl) a += q0[(long long)sl * slice_ns + (long long)q * HWo];
        v[q] = a;
      }
    const ConvEpi e = scf_conv_epi(p, n);
    scf_conv_epilogue_group(p, e, v, cb, pix, p.out_div != 1.0f);
  }
}

int scf_conv_kcombine_launch(const ConvK& k, const float* parts, int S, long long slice_ns, int N, hipStream_t st) {
  const long long total = (long long)N * ((k.Cout + 3) / 4) * k.Ho * k.Wo;
  long long nb = scf_cdiv(total, 256);
  const long long cap = 8LL * scf_cu_count();
  nb = nb > cap ? cap : nb;
  scf_launch(conv_kcombine_kernel, dim3((unsigned)nb), dim3(256), 0, st, k, parts, S, slice_ns, N);
  return scf_launch_status();
}

// slice count for a launch nobody asked to slice: only grids on which every K-split block is alone on its CU, only
// chains of at least four chunks; the combine launch costs ~4.5 us, a chunk ~2 us
int scf_conv_dma_autoslice(const ConvK& k, int N) {
  if ((!k.wp4 && !k.wp4s) || (k.stride != 1 && k.stride != 2) || k.w_ns != 0 || k.out_tile || k.kslices > 1) return 1;
  if (k.T == 1 && k.stride == 2) return 1;                        // dilated 1x1 shortcuts: short chains
  const int FC = 1 << k.fc_log2, FR = 32 / FC;
  const long long ksp_blk = (long long)N * ((k.Ho + FR - 1) / FR) * ((k.Wo + FC - 1) / FC) * ((k.Cout + 31) / 32);
  if (ksp_blk > scf_cu_count()) return 1;
  int chain = 1 << 30;
  const int gs[3] = {k.wp4t ? k.G4t : 0, k.wp4s ? k.G4s : 0, k.wp4 ? k.G4 : 0};
  for (int i = 0; i < 3; ++i)
    if (gs[i] == 1 || gs[i] == 2 || gs[i] == 4) {
      const int nch = (k.Cin + 8 * gs[i] - 1) / (8 * gs[i]);
      chain = nch < chain ? nch : chain;
    }
  if (chain == (1 << 30) || chain < 4) return 1;
  int S = chain / 2 < 4 ? chain / 2 : 4;
  while (S > 1 && ksp_blk * S > 4LL * scf_cu_count()) --S;
  return S;
}
