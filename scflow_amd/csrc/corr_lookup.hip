// Multi-scale correlation lookup (RAFT/SCFlow "CorrLookup") for gfx950 -- v7.
//
// Reference semantics: models/utils/corr_lookup.py:102-136 (+ bilinear_sample :31-67).
//
// Roofline: HBM-bound gather.  Per query and level the kernel touches the (2r+2)^2
// footprint of that query's private correlation map once (400 B at r=4) and writes
// (2r+1)^2 outputs (324 B); no byte is shared between queries, so the algorithmic
// traffic is 4*(400+324)+8 = 2904 B/query.
//
// What bounded v5 (round 1) was not memory but its own instruction stream: with gathers AND
// stores removed it still ran 11.6 us at batch 32 (tools/lab/lookup_lab.hip) -- a branchy loop
// with ~45 issue slots and 13 address VALU per (query, DMA), three v_readlane per query.  Two
// lessons from the lab shape v7:
//   * ONE DMA instruction must cover whole cache lines' worth of one query's window (lane =
//     footprint element of ONE query).  A row-of-six-queries mapping (v6) needs a third of the
//     instructions but asks the L1/L2 for every 128-byte line four times (once per window row)
//     and lost 2-3 us on the request path.
//   * the end-of-kernel write-back of the XCD L2s (up to 32 MB of dirty output lines) is part of
//     the kernel's duration: write-through (sc1) stores drain during the kernel instead (-2 us).
//
//   block = 4 waves x the same 32 consecutive queries ("group"); a wave owns one pyramid level
//   (levels w, w+4, ... when L > 4); blocks are persistent over groups.
//   tables  : per level the query lanes (lane = query, the two half-waves split rows / columns)
//             write a 20-entry u16 table per query to LDS: the offset (in floats) of each window
//             row and of each window column inside the query's map, 0x8000 when outside it.
//             This is where the layout lives (row-major, or 8x4-float tiles of 128 B: any level
//             whose rows are about a cache line long, see scf_corr_preferred_layout).
//   gather  : lane = footprint element e = (row, col) of ONE query, two DMA instructions per
//             query.  Per DMA: two ds_read_u16 with IMMEDIATE offsets (the query loop is
//             unrolled), v_add_lshl (row + col -> byte offset), v_cmp (any 0x8000 -> lane off),
//             the rest scalar.  Straight-line inline asm: no branches, no v_readlane.
//             Zero padding = zero-filled staging + switched-off lanes: only in-map bytes move.
//   small   : a level whose whole map fits in the footprint is staged whole (one DMA per query
//             and 64 map elements); out-of-map rows are redirected to a shared zero row at
//             read-back, out-of-map columns are clamped and lose their weights.
//   emit    : lane = (query, half); per-lane row addresses + immediate column offsets for both
//             layouts; bilinear weights are per (query, level) constants; each half-wave
//             writes 32 consecutive queries of one channel = one full 128-B line, write-through.
//   generic : radius > 4, maps of more than 32767 floats or more LDS than a block may have take
//             corr_lookup_generic_kernel (one thread per output element; same arithmetic).
#include "scf_common.h"
#include <type_traits>

#ifndef SCF_LOOKUP_STORE_MODE
#define SCF_LOOKUP_STORE_MODE 2      // sc1 = write-through, see lk_store
#endif
// cache policy of the gather instructions (A/B builds of tools/lab/lookup_lab.hip): footprint taps / whole maps
#ifndef SCF_LOOKUP_TAP_POL
#define SCF_LOOKUP_TAP_POL " nt"
#endif
#ifndef SCF_LOOKUP_MAP_POL
#define SCF_LOOKUP_MAP_POL " nt"
#endif

// tools/lab/lookup_lab.hip compiles this file with per-wave timeline stamps and ablation switches;
// their code lives in tools/lab/lookup_lab_hooks.h.  The product build sees empty hooks.
#ifdef SCF_LOOKUP_LAB
#include "lookup_lab_hooks.h"    // lab builds only: -I tools/lab
#else
#define LK_LAB_PARAMS
#define LK_TRACE(slot) do { } while (0)
#define LK_TRACE_END(lvl) do { } while (0)
#define LK_TRACE_U(g, lvl, slot) do { } while (0)
#define LK_TRACE_END_U(g, lvl) do { } while (0)
#define LK_LAB_PIPE_MODE(v) (v)
#define LK_SKIP_DMA false
#define LK_SKIP_STORE false
#define LK_LAB_LAUNCH(p, nblk) do { } while (0)
#define LK_LAB_SETUP(p, nblk) do { } while (0)
#define LK_LAB_STAGGER do { } while (0)
#define LK_EARLY_MAPS true
#endif

#define LK_MAXU 3        // units per wave of the pipelined kernel
struct LookupParams {
  const float* lvl[SCF_MAX_LEVELS];
  int lh[SCF_MAX_LEVELS];
  int lw[SCF_MAX_LEVELS];
  int pw4[SCF_MAX_LEVELS];   // level stored in 8x4-float tiles: 4 * padded width (= floats per row of tiles); 0 = row-major
  int msz[SCF_MAX_LEVELS];   // floats per query map (padded size when tiled)
  const float* flow;
  float* out;
  int N, h, w, L;
  int woff[4];            // LDS offset (floats) of each wave slot's staging region
  int ngroups;            // ceil(total_q / 32)
  long long total_q;
  // pipelined kernel (corr_lookup_pipe_kernel): a block owns `gpb` consecutive groups; its gpb * L
  // (group, level) units are dealt to the four waves by cost, each wave runs its units cheapest first
  int gpb, nsg;                     // groups per block, ceil(ngroups / gpb)
  unsigned char unit[4][4];         // wave slot, position: (group within the block << 4) | level; 0xff = none
  int uoff[4][LK_MAXU];             // LDS offset (floats) of that unit's staging region
  LK_LAB_PARAMS
};

// explicit LDS (address space 3) pointers: 32 bits each, ds_read / ds_write without relying on
// address-space inference (the per-lane row address arrays would otherwise be 64-bit flat pointers)
typedef __attribute__((address_space(3))) float* lds_fp_t;
typedef __attribute__((address_space(3))) const float* lds_cfp_t;
typedef __attribute__((address_space(3))) unsigned short* lds_u16p_t;
typedef __attribute__((address_space(3))) const unsigned short* lds_cu16p_t;

__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

// a pointer the compiler keeps in SGPRs (wave-uniform by construction)
__device__ __forceinline__ const void* lk_sgpr_ptr(const void* p) {
  const unsigned long long u = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return (const void*)(((unsigned long long)hi << 32) | lo);
}

// Footprint gather through a raw buffer descriptor (base = ONE query's map, num_records = its size
// in bytes, at most 128 KB): `buffer_load_dword voff, rsrc, 0 offen lds` moves one dword per lane
// from base + voff to LDS byte M0 + 4 * lane, and a lane whose offset is past num_records gets ZERO
// written to its cell (checked on gfx950: tools/lab/buf_lds_test.hip).  Taps outside the map carry
// an offset >= 0x20000 (a 0x8000 table entry), so zero padding needs no EXEC mask per instruction
// and no pre-zeroed staging; stepping to the next query is two scalar adds on the descriptor.  The whole gather phase runs under ONE divergent branch
// (the footprint's FS elements are split into equal shares of LPS <= 64 lanes; lanes >= LPS sit it out).  The compiler does not
// count these loads: the caller waits on vmcnt itself.
typedef int lk_rsrc_t __attribute__((ext_vector_type(4)));
#define LK_OOB 0x8000u        // table marker (floats): (row + column) << 2 >= 0x20000 > any num_records
__device__ __forceinline__ lk_rsrc_t lk_make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  lk_rsrc_t r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xFFFFu));   // stride 0: raw buffer
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);                             // num_records (bytes)
  r[3] = 0x00020000;
  return r;
}
__device__ __forceinline__ void lk_dma_tap(lk_rsrc_t rsrc, unsigned voff, unsigned lds) {
  asm volatile("s_mov_b32 m0, %2\n\t"
               "s_nop 0\n\t"
               "buffer_load_dword %1, %0, 0 offen" SCF_LOOKUP_TAP_POL " lds"
               : : "s"(rsrc), "v"(voff), "s"(lds) : "memory");
}
// contiguous variant with an explicit lane mask (whole small maps)
__device__ __forceinline__ void lk_dma_mask(const void* sbase, unsigned voff, unsigned lds,
                                            unsigned long long mask) {
  asm volatile("s_mov_b64 exec, %3\n\t"
               "s_mov_b32 m0, %2\n\t"
               "s_nop 0\n\t"
               "global_load_lds_dword %1, %0" SCF_LOOKUP_MAP_POL "\n\t"
               "s_mov_b64 exec, -1"
               : : "s"(sbase), "v"(voff), "s"(lds), "s"(mask) : "memory");
}

// window read-back + bilinear blend + store for one (wave, level).  rowp[r] = LDS address of
// window row r at this lane's first column (out-of-map rows of a small level point at the zero
// row); column `it` is an immediate offset (non-small) or a clamped per-lane offset (small).
// output store with an explicit cache policy (SM): 0 plain (write-back: the lines stay dirty in
// this XCD's L2 until the end-of-kernel write-back), 1 nt, 2 sc1 (write-through), 3 sc0 sc1,
// 4 sc1 nt.  addr = running scalar channel base + 32-bit lane offset (scoped atomic stores would
// take 64-bit vector addresses and spill).  No "memory" clobber: nothing reads the output back.
template <int SM>
__device__ __forceinline__ void lk_store(char* sbase, unsigned voff, float v) {
  if constexpr (SM == 0) *(float*)(sbase + voff) = v;
  else if constexpr (SM == 1) asm volatile("global_store_dword %0, %1, %2 nt" : : "v"(voff), "v"(v), "s"(sbase));
  else if constexpr (SM == 2) asm volatile("global_store_dword %0, %1, %2 sc1" : : "v"(voff), "v"(v), "s"(sbase));
  else if constexpr (SM == 3) asm volatile("global_store_dword %0, %1, %2 sc0 sc1" : : "v"(voff), "v"(v), "s"(sbase));
  else asm volatile("global_store_dword %0, %1, %2 sc1 nt" : : "v"(voff), "v"(v), "s"(sbase));
}

// Window centre of one (query, level) and its bilinear weights -- shared by the LDS-DMA kernel and
// the generic kernel so that both are the same arithmetic.  Reference: corr_lookup.py:127 (centre /
// 2^l), :64-67 (normalise with max(size-1, 1), grid_sample de-normalises with size-1).
//   * along a size-1 axis every tap lands exactly on index 0 (flat: centre := R, weights 1 | 0);
//   * a centre far outside any map is clamped to +-30000: all taps read zero padding;
//   * NaN / inf flow, or a centre whose normalisation overflows fp32 (|c| * 2 = inf): the reference's
//     grid_sample returns NaN for every tap of that query (torch CPU) -- reproduced by NaN weights
//     on an all-padding window (0 * NaN = NaN).
struct LkCentre {
  float x0f, y0f;       // floor of the (clamped) centre
  int x0, y0;           // map coordinates of window column / row 0
  float nw, ne, sw, se;
};
template <int R_>
__device__ __forceinline__ LkCentre lk_centre(float qx, float qy, float inv, bool flat_x, bool flat_y, int r_rt = 0) {
  const int R = R_ > 0 ? R_ : r_rt;
  const float cxu = qx * inv, cyu = qy * inv;               // exact power-of-two scaling
  const bool bad = !(fabsf(cxu * 2.f) < __builtin_inff()) || !(fabsf(cyu * 2.f) < __builtin_inff());
  float cx = flat_x ? (float)R : cxu;
  float cy = flat_y ? (float)R : cyu;
  cx = fminf(fmaxf(cx, -30000.f), 30000.f);                 // far outside any map -> all taps 0
  cy = fminf(fmaxf(cy, -30000.f), 30000.f);
  if (bad) { cx = -30000.f; cy = -30000.f; }
  LkCentre c;
  c.x0f = floorf(cx); c.y0f = floorf(cy);
  c.x0 = (int)c.x0f - R; c.y0 = (int)c.y0f - R;
  const float tx = cx - c.x0f, ty = cy - c.y0f;
  const float wx0 = (c.x0f + 1.f) - cx, wy0 = (c.y0f + 1.f) - cy;   // grid_sample: (x_se - x)
  const float nanv = __builtin_nanf("");
  c.nw = bad ? nanv : wx0 * wy0; c.ne = bad ? nanv : tx * wy0;
  c.sw = bad ? nanv : wx0 * ty;  c.se = bad ? nanv : tx * ty;
  return c;
}
// the blend, as one explicit fma chain (same bits in both kernels)
__device__ __forceinline__ float lk_blend(float a, float b, float c, float d, float nw, float ne, float sw, float se) {
  return __builtin_fmaf(d, se, __builtin_fmaf(c, sw, __builtin_fmaf(b, ne, a * nw)));
}

template <int R, bool SMALL, int SM>
__device__ __forceinline__ void lookup_emit(const lds_cfp_t (&rowp)[2 * R + 2], int lw, int xfirst,
                                            bool flat_x, float nw, float ne, float sw, float se,
                                            int part, char* obase, unsigned lane_off, size_t cs,
                                            bool qvalid) {
  constexpr int FW = 2 * R + 2, D = 2 * R + 1, NP = 2;
  // The two half-waves split the D x-offsets: half g takes columns [i0(g), i0(g+1)), i0(g) =
  // ceil(g*D / 2).  The loop itself is wave-uniform (NI = the larger share; the shorter one is
  // masked) so that store bases stay in SGPRs and only a 32-bit per-lane offset goes into the
  // address VGPR.
  constexpr int NI = (D + NP - 1) / NP;
  const int i0 = (part * D + NP - 1) / NP;
  const int cnt = ((part + 1) * D + NP - 1) / NP - i0;
  const unsigned lane_off2 = lane_off + (unsigned)((size_t)i0 * D * cs);
  // small: column validity (xfirst = map column of window column i0)
  auto column = [&](int it, float (&col)[FW], float& cv) {
    if constexpr (SMALL) {
      const int xx = flat_x ? 0 : xfirst + it;
      const bool cok = (unsigned)xx < (unsigned)lw;
      const int xs = cok ? xx : 0;
      cv = cok ? 1.f : 0.f;
#pragma unroll
      for (int r = 0; r < FW; ++r) col[r] = rowp[r][xs];
    } else {
      cv = 1.f;
#pragma unroll
      for (int r = 0; r < FW; ++r) col[r] = rowp[r][it];      // immediate offsets
    }
  };
  float col[2][FW];
  float cva, cvb;
  column(0, col[0], cva);
#pragma unroll
  for (int it = 0; it < NI; ++it) {
    float (&cA)[FW] = col[it & 1];
    float (&cB)[FW] = col[(it & 1) ^ 1];
    column(it + 1, cB, cvb);
    float w0 = nw, w1 = ne, w2 = sw, w3 = se;
    if constexpr (SMALL) {       // out-of-map columns: clamped reads, zero weights
      w0 = nw * cva; w2 = sw * cva; w1 = ne * cvb; w3 = se * cvb;
    }
    if (qvalid && it < cnt) {
      char* oc = obase + (size_t)(it * D) * cs;      // wave-uniform channel base (SGPRs)
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const float v = lk_blend(cA[j], cB[j], cA[j + 1], cB[j + 1], w0, w1, w2, w3);
        // a RUNNING scalar base (opaque to the optimiser: it would otherwise precompute all 81
        // channel bases and spill them): one s_add_u32 / s_addc_u32 per store
        asm volatile("" : "+s"(oc));
        lk_store<SM>(oc, lane_off2, v);
        oc += cs;
      }
    }
    cva = cvb;
  }
}

// GPB = groups per block: 2 = the same waves in half as many workgroups (the four waves of a group never
// synchronise with the other group's; A/B knob lookup_pipe = 4)
template <int R, int SM, int GPB = 1>
__global__ __launch_bounds__(256 * GPB, GPB == 3 ? 1 : 4 / GPB) void corr_lookup_kernel(LookupParams p) {
  constexpr int FW = 2 * R + 2;       // footprint width
  constexpr int FS = FW * FW;         // footprint size
  constexpr int FSP = FS | 1;         // odd LDS stride per query (conflict-free lane = query reads)
  constexpr int D = 2 * R + 1;        // window width
  constexpr int QB = 32;              // queries per group
  constexpr int NSET = (FS + 63) / 64;  // DMA instructions per query footprint
  constexpr int LPS = FS / NSET;      // lanes per DMA instruction (equal shares: 50 + 50 at r = 4)
  static_assert(LPS * NSET == FS && LPS <= 64, "the footprint splits into equal lane shares");
  constexpr int TQ = 2 * FW;          // table entries (u16) per query: FW row + FW column offsets
  extern __shared__ __attribute__((aligned(16))) float lds_fp[];

  const bool skip_dma = LK_SKIP_DMA, skip_store = LK_SKIP_STORE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
  const int wave = wave_all & 3, gsel = GPB == 1 ? 0 : wave_all >> 2;
  const int l32 = lane & 31, half = lane >> 5;                // query within the group, half-wave
  const int hw = p.h * p.w;
  const int ktot = p.L * D * D;
  const size_t cs = (size_t)hw * 4;   // channel stride in bytes
  const lds_fp_t myfp = (lds_fp_t)lds_fp + gsel * p.nsg + p.woff[wave];      // (GPB > 1: p.nsg = LDS floats per group)

  // gather-lane roles: this lane fetches footprint elements e = lane + 64 s of EVERY query
  const lds_u16p_t tbl = (lds_u16p_t)(myfp + QB * FSP);   // [QB][TQ]
  lds_cu16p_t trow[NSET];
  lds_cu16p_t tcol[NSET];
#pragma unroll
  for (int s = 0; s < NSET; ++s) {
    const int e = (lane < LPS ? lane : 0) + LPS * s;      // lanes >= LPS are switched off in the gather phase
    trow[s] = tbl + e / FW;
    tcol[s] = tbl + FW + (e - (e / FW) * FW);
  }

  // query indices fit 31 bits (the launcher refuses more): 32-bit divisions only
  const unsigned total_q = (unsigned)p.total_q;
  auto flow_of = [&](int gg, float& fx_, float& fy_) {
    const unsigned gq_ = (unsigned)gg * QB + l32;
    fx_ = 0.f; fy_ = 0.f;
    if (gq_ < total_q) {
      const unsigned n_ = gq_ / (unsigned)hw;
      const unsigned q_ = gq_ - n_ * (unsigned)hw;
      const float* fl = p.flow + (size_t)n_ * 2 * hw + q_;
      fx_ = fl[0];
      fy_ = fl[hw];
    }
  };
  const int gstep = (int)gridDim.x * GPB;
  int g = (int)blockIdx.x * GPB + gsel;
  float fx = 0.f, fy = 0.f;
  LK_LAB_STAGGER;
  if (g < p.ngroups) flow_of(g, fx, fy);       // issued before any setup

  for (; g < p.ngroups; g += gstep) {
    LK_TRACE(0);
    const unsigned gq0 = (unsigned)g * QB;
    const int n0 = (int)(gq0 / (unsigned)hw);     // sample of the group's first query (wave-uniform)
    const unsigned gq = gq0 + l32;
    const bool qvalid = gq < total_q;
    // maps of >= 32 pixels (every real one): the group spans at most two samples, no per-lane division
    unsigned q0 = gq - (unsigned)n0 * (unsigned)hw;
    int n = n0;
    if (hw >= QB) {
      if (q0 >= (unsigned)hw) { q0 -= (unsigned)hw; ++n; }
    } else {
      const unsigned dn = q0 / (unsigned)hw;
      n += (int)dn;
      q0 -= dn * (unsigned)hw;
    }
    const int q = qvalid ? (int)q0 : 0;
    const int y = (int)((unsigned)q / (unsigned)p.w), x = q - y * p.w;
    const float xf = (float)x, yf = (float)y;
    // byte offset of this lane's query from the group-uniform base out[n0, k, 0, 0]
    const unsigned lane_off = ((unsigned)(n - n0) * (unsigned)(ktot * hw) + (unsigned)q) * 4u;
    const int nq = (int)((total_q - gq0) < (unsigned)QB ? (total_q - gq0) : (unsigned)QB);   // wave-uniform
    const bool more = g + gstep < p.ngroups;
    float fxn = 0.f, fyn = 0.f;

    for (int lvl = wave; lvl < p.L; lvl += 4) {
      const int lh = p.lh[lvl], lw = p.lw[lvl];
      const int msz = p.msz[lvl];
      const int pw4 = p.pw4[lvl];
      const bool tiled = pw4 != 0;
      // Reference quirk at degenerate sizes: coordinates are normalised with max(size-1, 1) and
      // grid_sample(align_corners=True) de-normalises with (size-1), so along a size-1 axis
      // EVERY tap lands exactly on index 0 (corr_lookup.py:64-67).
      const bool flat_x = lw == 1, flat_y = lh == 1;
      const bool small = !tiled && (lh <= FW && lw <= FW);
      if (small && lane < lw) (myfp + QB * (msz | 1))[lane] = 0.f;       // the shared zero row
      const float inv = 1.0f / (float)(1 << lvl);
      const char* lbase = (const char*)lk_sgpr_ptr(p.lvl[lvl] + (size_t)gq0 * msz);
      unsigned st0 = (unsigned)(uintptr_t)myfp;
      asm volatile("" : "+s"(st0));                     // row / query LDS addresses: s_add from here
      // A level that is staged whole does not need the flow to be fetched: its DMAs leave before the flow
      // load is waited for (~0.8 us earlier; the memory system has work while the other waves build tables)
      if (small && LK_EARLY_MAPS) {
        const int S = msz | 1;
        const unsigned long long m0mask = msz >= 64 ? ~0ull : ((1ull << msz) - 1ull);
        const unsigned long long m1mask = msz > 64 ? ((1ull << (msz - 64)) - 1ull) : 0ull;
        const unsigned vlane4 = (unsigned)lane * 4u;
        LK_TRACE(2);
        if (!skip_dma) {
          for (int qq = 0; qq < nq; ++qq) {
            const char* mb = lbase + (size_t)qq * msz * 4;
            lk_dma_mask(mb, vlane4, st0 + (unsigned)(qq * S) * 4u, m0mask);
            if (msz > 64) lk_dma_mask(mb + 256, vlane4, st0 + (unsigned)(qq * S + 64) * 4u, m1mask);
          }
        }
        asm volatile("" : "+v"(fx), "+v"(fy));          // the first use of the flow stays behind the DMAs
      }
      const LkCentre c = lk_centre<R>(xf + fx, yf + fy, inv, flat_x, flat_y);   // first use of the flow
      const int x0 = c.x0, y0 = c.y0;
      const int i0 = (half * D + 1) / 2;                // first x-offset of this half-wave
      lds_cfp_t rowp[FW];

      if (small) {
        // ---- whole map per query, stride S (odd), + ONE shared zero row behind them ----
        const int S = msz | 1;
        const lds_fp_t zrow = myfp + QB * S;
        const unsigned long long m0mask = msz >= 64 ? ~0ull : ((1ull << msz) - 1ull);
        const unsigned long long m1mask = msz > 64 ? ((1ull << (msz - 64)) - 1ull) : 0ull;
        const unsigned vlane4 = (unsigned)lane * 4u;
        if (!LK_EARLY_MAPS) {
          LK_TRACE(2);
          if (!skip_dma) {
            for (int qq = 0; qq < nq; ++qq) {
              const char* mb = lbase + (size_t)qq * msz * 4;
              lk_dma_mask(mb, vlane4, st0 + (unsigned)(qq * S) * 4u, m0mask);
              if (msz > 64) lk_dma_mask(mb + 256, vlane4, st0 + (unsigned)(qq * S + 64) * 4u, m1mask);
            }
          }
        }
        const lds_cfp_t f = myfp + l32 * S;
#pragma unroll
        for (int r = 0; r < FW; ++r) {
          const int yy = flat_y ? 0 : y0 + r;
          rowp[r] = (unsigned)yy < (unsigned)lh ? f + yy * lw : (lds_cfp_t)zrow;
        }
      } else {
        // ---- zero-padded footprints, stride FSP; NSET DMA instructions per query ----
        // offset tables: half-wave 0 writes this query's FW row offsets, half-wave 1 its FW column
        // offsets (in floats, inside the query's map; LK_OOB = outside).  The map layout lives here
        // and nowhere else: row-major, or 8x4-float tiles of 128 B (rows / columns past the map's
        // real size -- tile padding -- count as outside like everything else past it).
        {
          const int c0 = half ? x0 : y0, lim = half ? lw : lh;
          const bool flat = half ? flat_x : flat_y;
          const int sh = half ? 3 : 2, msk = half ? 7 : 3;
          const int mula = tiled ? (half ? 32 : pw4) : 0, mulb = tiled ? (half ? 1 : 8) : (half ? 1 : lw);
          const lds_u16p_t tq = tbl + l32 * TQ + half * FW;
#pragma unroll
          for (int j = 0; j < FW; j += 2) {
            unsigned pr = 0;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
              const int v = flat ? 0 : c0 + j + jj;
              const bool ok = qvalid && (unsigned)v < (unsigned)lim;
              // 24-bit multiplies (full rate; v_mul_lo_u32 is quarter rate): |v| <= 30010, factors < 2^15
              const int val = tiled ? __mul24(v >> sh, mula) + __mul24(v & msk, mulb) : __mul24(v, mulb);
              pr |= (ok ? (unsigned)val : LK_OOB) << (16 * jj);
            }
            *(__attribute__((address_space(3))) unsigned*)(tq + j) = pr;
          }
        }
        const unsigned mbytes = (unsigned)msz * 4u;
        __builtin_amdgcn_s_waitcnt(0xC07F);             // lgkmcnt(0): the tables are in LDS
        __builtin_amdgcn_wave_barrier();
        LK_TRACE(2);
        // ONE exec setup for the whole phase: an ordinary divergent branch (the compiler owns EXEC,
        // nothing it schedules into the region runs with the wrong lanes)
        if (!skip_dma && (LPS == 64 || lane < LPS)) {
          constexpr int QBATCH = 4;                      // table reads of a batch issue together
#pragma unroll
          for (int qb = 0; qb < QB; qb += QBATCH) {
            unsigned tr_[QBATCH][NSET], tc_[QBATCH][NSET];
#pragma unroll
            for (int qi = 0; qi < QBATCH; ++qi)
#pragma unroll
              for (int s = 0; s < NSET; ++s) {
                tr_[qi][s] = trow[s][(qb + qi) * TQ];     // immediate offsets
                tc_[qi][s] = tcol[s][(qb + qi) * TQ];
              }
#pragma unroll
            for (int qi = 0; qi < QBATCH; ++qi) {
              const int qq = qb + qi;
              const lk_rsrc_t rsrc = lk_make_rsrc(lbase + (size_t)qq * mbytes, mbytes);
#pragma unroll
              for (int s = 0; s < NSET; ++s)
                lk_dma_tap(rsrc, (tr_[qi][s] + tc_[qi][s]) << 2, st0 + (unsigned)(qq * FSP + LPS * s) * 4u);
            }
          }
        }
        const lds_cfp_t f = myfp + l32 * FSP + i0;
#pragma unroll
        for (int r = 0; r < FW; ++r) rowp[r] = f + r * FW;
      }
      LK_TRACE(3);
      // flow of this block's next group: issued behind the gathers, consumed after the stores
      if (more && lvl + 4 >= p.L) flow_of(g + gstep, fxn, fyn);

      // ---- lane = (query, half); halves split the x-offsets ----
      char* obase = (char*)p.out + ((size_t)n0 * ktot + (size_t)lvl * D * D) * cs;
      __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): the DMA data is in LDS
      __builtin_amdgcn_wave_barrier();
      LK_TRACE(4);
      if (small)
        lookup_emit<R, true, SM>(rowp, lw, x0 + i0, flat_x, c.nw, c.ne, c.sw, c.se, half, obase, lane_off, cs, qvalid && !skip_store);
      else
        lookup_emit<R, false, SM>(rowp, lw, 0, flat_x, c.nw, c.ne, c.sw, c.se, half, obase, lane_off, cs, qvalid && !skip_store);
      __builtin_amdgcn_wave_barrier();
      LK_TRACE_END(lvl);
    }
    fx = fxn;
    fy = fyn;
  }
}

// ---------------------------------------------------------------------------------
// v9 (round 5): the same gather / emit code, de-phased.  v8's 1024 blocks all run tables -> gather ->
// emit in step: the chip reads for the first half of the kernel and writes for the second.  Here a
// block owns G consecutive groups; its G * L (group, level) units are dealt to the four waves so that
// every wave carries the same number of DMA instructions (a big level of one group + a small level
// of another), and a wave runs its units as a software pipeline:
//   flows of all G groups (plain loads, waited for before the first DMA)
//   issue  : every unit, cheapest first (a level that is staged whole, then the footprint level)
//   emit   : in the same order, each behind a COUNTED wait -- s_waitcnt vmcnt(n), n = the DMA
//            instructions of the units behind it (loads return in order; stores in flight only add
//            to the counter, so the wait is conservative, never early).  The first unit's stores
//            leave while the later units' gathers are still in flight: every CU reads and writes
//            at the same time for most of the kernel.
// Half as many waves as v8 for the same LDS bytes in flight: the launch ramp halves as well.
// ---------------------------------------------------------------------------------
// counted wait on the vector-memory counter.  Every DMA and store of this kernel is inline asm the compiler does
// not count; the only loads it knows of (the flows) are consumed before the first DMA is issued.
#define LK_WAITVM(N) asm volatile("s_waitcnt vmcnt(" #N ")" : : : "memory")

template <int R, int SM, int G, int NU>
__global__ __launch_bounds__(256, 2) void corr_lookup_pipe_kernel(LookupParams p) {
  constexpr int FW = 2 * R + 2, FS = FW * FW, FSP = FS | 1, D = 2 * R + 1, QB = 32;
  constexpr int NSET = (FS + 63) / 64, LPS = FS / NSET, TQ = 2 * FW;
  static_assert(LPS * NSET == FS && LPS <= 64, "the footprint splits into equal lane shares");
  static_assert(G == 2 || G == 3, "two or three groups per block");
  extern __shared__ __attribute__((aligned(16))) float lds_fp[];

  const bool skip_dma = LK_SKIP_DMA, skip_store = LK_SKIP_STORE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, half = lane >> 5;
  const int hw = p.h * p.w;
  const int ktot = p.L * D * D;
  const size_t cs = (size_t)hw * 4;
  const unsigned total_q = (unsigned)p.total_q;

  // geometry of group g as this lane sees it (maps of >= 32 pixels: a group spans at most two samples)
  struct Geom { unsigned gq0; int n0, nq, x, y; unsigned lane_off; bool qvalid; };
  auto geom = [&](int g) {
    Geom m;
    m.gq0 = (unsigned)g * QB;
    m.n0 = (int)(m.gq0 / (unsigned)hw);
    const unsigned gq = m.gq0 + l32;
    m.qvalid = gq < total_q;
    unsigned q0 = gq - (unsigned)m.n0 * (unsigned)hw;
    int n = m.n0;
    if (hw >= QB) {
      if (q0 >= (unsigned)hw) { q0 -= (unsigned)hw; ++n; }
    } else {
      const unsigned dn = q0 / (unsigned)hw;
      n += (int)dn;
      q0 -= dn * (unsigned)hw;
    }
    const int q = m.qvalid ? (int)q0 : 0;
    m.y = (int)((unsigned)q / (unsigned)p.w);
    m.x = q - m.y * p.w;
    m.lane_off = ((unsigned)(n - m.n0) * (unsigned)(ktot * hw) + (unsigned)q) * 4u;
    m.nq = (int)((total_q - m.gq0) < (unsigned)QB ? (total_q - m.gq0) : (unsigned)QB);
    return m;
  };

  for (int sg = blockIdx.x; sg < p.nsg; sg += gridDim.x) {
    // ---- flows of the block's groups: plain loads, consumed (centres of the first footprint unit, or the
    //      opaque use below) BEFORE the first DMA leaves -- the compiler's own wait for them must not sit
    //      behind DMAs it does not know of ----
    float fx[G], fy[G];
#pragma unroll
    for (int gs = 0; gs < G; ++gs) {
      const int g = sg * G + gs;
      const unsigned gq = (unsigned)g * QB + l32;
      fx[gs] = 0.f; fy[gs] = 0.f;
      if (g < p.ngroups && gq < total_q) {
        const unsigned n_ = gq / (unsigned)hw;
        const float* fl = p.flow + (size_t)n_ * 2 * hw + (gq - n_ * (unsigned)hw);
        fx[gs] = fl[0];
        fy[gs] = fl[hw];
      }
    }
#pragma unroll
    for (int gs = 0; gs < G; ++gs) asm volatile("" : "+v"(fx[gs]), "+v"(fy[gs]));   // used: the loads are waited for here
    auto wait_vm = [&](int n) {      // at most n (wave-uniform) vector-memory operations stay in flight
      if (n >= 63) LK_WAITVM(63);
      else if (n >= 32) LK_WAITVM(32);
      else LK_WAITVM(0);
    };
    auto flow_of = [&](int gs, float& ax, float& ay) {
      ax = fx[0]; ay = fy[0];
#pragma unroll
      for (int k = 1; k < G; ++k)
        if (gs == k) { ax = fx[k]; ay = fy[k]; }
    };

    int cnt[NU];
    int issued = 0;
    // ================= issue: every unit of this wave, cheapest first =================
#pragma unroll
    for (int s = 0; s < NU; ++s) {
      cnt[s] = 0;
      const int uc = p.unit[wave][s];
      const int gs = uc >> 4, lvl = uc & 15;
      const int g = sg * G + gs;
      if (uc == 0xff || g >= p.ngroups) continue;
      LK_TRACE_U(g, lvl, 0);
      const Geom m = geom(g);
      const int lh = p.lh[lvl], lw = p.lw[lvl], msz = p.msz[lvl], pw4 = p.pw4[lvl];
      const bool tiled = pw4 != 0;
      const bool flat_x = lw == 1, flat_y = lh == 1;
      const bool small = !tiled && (lh <= FW && lw <= FW);
      const lds_fp_t myfp = (lds_fp_t)lds_fp + p.uoff[wave][s];
      const char* lbase = (const char*)lk_sgpr_ptr(p.lvl[lvl] + (size_t)m.gq0 * msz);
      unsigned st0 = (unsigned)(uintptr_t)myfp;
      asm volatile("" : "+s"(st0));
      if (small) {
        const int S = msz | 1;
        if (lane < lw) (myfp + QB * S)[lane] = 0.f;       // the shared zero row
        const unsigned long long m0mask = msz >= 64 ? ~0ull : ((1ull << msz) - 1ull);
        const unsigned long long m1mask = msz > 64 ? ((1ull << (msz - 64)) - 1ull) : 0ull;
        const unsigned vlane4 = (unsigned)lane * 4u;
        LK_TRACE_U(g, lvl, 2);
        if (!skip_dma) {
          for (int qq = 0; qq < m.nq; ++qq) {
            const char* mb = lbase + (size_t)qq * msz * 4;
            lk_dma_mask(mb, vlane4, st0 + (unsigned)(qq * S) * 4u, m0mask);
            if (msz > 64) lk_dma_mask(mb + 256, vlane4, st0 + (unsigned)(qq * S + 64) * 4u, m1mask);
          }
          cnt[s] = m.nq * (msz > 64 ? 2 : 1);
        }
      } else {
        float qx, qy;
        flow_of(gs, qx, qy);
        const float inv = 1.0f / (float)(1 << lvl);
        const LkCentre c = lk_centre<R>((float)m.x + qx, (float)m.y + qy, inv, flat_x, flat_y);
        const lds_u16p_t tbl = (lds_u16p_t)(myfp + QB * FSP);
        {
          const int c0 = half ? c.x0 : c.y0, lim = half ? lw : lh;
          const bool flat = half ? flat_x : flat_y;
          const int sh = half ? 3 : 2, msk = half ? 7 : 3;
          const int mula = tiled ? (half ? 32 : pw4) : 0, mulb = tiled ? (half ? 1 : 8) : (half ? 1 : lw);
          const lds_u16p_t tq = tbl + l32 * TQ + half * FW;
#pragma unroll
          for (int j = 0; j < FW; j += 2) {
            unsigned pr = 0;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
              const int v = flat ? 0 : c0 + j + jj;
              const bool ok = m.qvalid && (unsigned)v < (unsigned)lim;
              const int val = tiled ? __mul24(v >> sh, mula) + __mul24(v & msk, mulb) : __mul24(v, mulb);
              pr |= (ok ? (unsigned)val : LK_OOB) << (16 * jj);
            }
            *(__attribute__((address_space(3))) unsigned*)(tq + j) = pr;
          }
        }
        const unsigned mbytes = (unsigned)msz * 4u;
        __builtin_amdgcn_s_waitcnt(0xC07F);               // lgkmcnt(0): the tables are in LDS
        __builtin_amdgcn_wave_barrier();
        LK_TRACE_U(g, lvl, 2);
        if (!skip_dma && (LPS == 64 || lane < LPS)) {
          lds_cu16p_t trow[NSET];
          lds_cu16p_t tcol[NSET];
#pragma unroll
          for (int e_ = 0; e_ < NSET; ++e_) {
            const int e = lane + LPS * e_;
            trow[e_] = tbl + e / FW;
            tcol[e_] = tbl + FW + (e - (e / FW) * FW);
          }
          constexpr int QBATCH = 4;
#pragma unroll
          for (int qb = 0; qb < QB; qb += QBATCH) {
            unsigned tr_[QBATCH][NSET], tc_[QBATCH][NSET];
#pragma unroll
            for (int qi = 0; qi < QBATCH; ++qi)
#pragma unroll
              for (int e_ = 0; e_ < NSET; ++e_) {
                tr_[qi][e_] = trow[e_][(qb + qi) * TQ];
                tc_[qi][e_] = tcol[e_][(qb + qi) * TQ];
              }
#pragma unroll
            for (int qi = 0; qi < QBATCH; ++qi) {
              const int qq = qb + qi;
              const lk_rsrc_t rsrc = lk_make_rsrc(lbase + (size_t)qq * mbytes, mbytes);
#pragma unroll
              for (int e_ = 0; e_ < NSET; ++e_)
                lk_dma_tap(rsrc, (tr_[qi][e_] + tc_[qi][e_]) << 2, st0 + (unsigned)(qq * FSP + LPS * e_) * 4u);
            }
          }
        }
        if (!skip_dma) cnt[s] = QB * NSET;
      }
      issued += cnt[s];
      LK_TRACE_U(g, lvl, 3);
    }

    // ================= emit: same order, each behind a counted wait =================
#pragma unroll
    for (int s = 0; s < NU; ++s) {
      const int uc = p.unit[wave][s];
      const int gs = uc >> 4, lvl = uc & 15;
      const int g = sg * G + gs;
      if (uc == 0xff || g >= p.ngroups) continue;
      issued -= cnt[s];                                   // what is left behind this unit
      wait_vm(issued);
      __builtin_amdgcn_wave_barrier();
      LK_TRACE_U(g, lvl, 4);
      const Geom m = geom(g);
      const int lh = p.lh[lvl], lw = p.lw[lvl], msz = p.msz[lvl], pw4 = p.pw4[lvl];
      const bool tiled = pw4 != 0;
      const bool flat_x = lw == 1, flat_y = lh == 1;
      const bool small = !tiled && (lh <= FW && lw <= FW);
      const lds_fp_t myfp = (lds_fp_t)lds_fp + p.uoff[wave][s];
      float qx, qy;
      flow_of(gs, qx, qy);
      const float inv = 1.0f / (float)(1 << lvl);
      const LkCentre c = lk_centre<R>((float)m.x + qx, (float)m.y + qy, inv, flat_x, flat_y);
      const int i0 = (half * D + 1) / 2;
      char* obase = (char*)p.out + ((size_t)m.n0 * ktot + (size_t)lvl * D * D) * cs;
      lds_cfp_t rowp[FW];
      if (small) {
        const int S = msz | 1;
        const lds_fp_t zrow = myfp + QB * S;
        const lds_cfp_t f = myfp + l32 * S;
#pragma unroll
        for (int r = 0; r < FW; ++r) {
          const int yy = flat_y ? 0 : c.y0 + r;
          rowp[r] = (unsigned)yy < (unsigned)lh ? f + yy * lw : (lds_cfp_t)zrow;
        }
        lookup_emit<R, true, SM>(rowp, lw, c.x0 + i0, flat_x, c.nw, c.ne, c.sw, c.se, half, obase, m.lane_off, cs, m.qvalid && !skip_store);
      } else {
        const lds_cfp_t f = myfp + l32 * FSP + i0;
#pragma unroll
        for (int r = 0; r < FW; ++r) rowp[r] = f + r * FW;
        lookup_emit<R, false, SM>(rowp, lw, 0, flat_x, c.nw, c.ne, c.sw, c.se, half, obase, m.lane_off, cs, m.qvalid && !skip_store);
      }
      __builtin_amdgcn_wave_barrier();
      LK_TRACE_END_U(g, lvl);
    }
  }
}

// ---------------------------------------------------------------------------------
// Generic lookup: any radius, level count, map size and layout -- one thread per output element,
// four cached loads per tap.  It exists so that the operator seam (CorrLookup(radius, ...) on any
// pyramid, corr_lookup.py:91-102) never answers "unsupported"; every configuration the reference
// ships (r = 4, L = 4, maps up to 60 x 80) takes the LDS-DMA kernel above.  Same centre / weight /
// blend arithmetic (lk_centre, lk_blend), so both kernels produce the same bits.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void corr_lookup_generic_kernel(LookupParams p, int R) {
  const int D = 2 * R + 1, DD = D * D;
  const int hw = p.h * p.w;
  const long long total = p.total_q * p.L * DD;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int q = (int)(idx % hw);
    const long long t = idx / hw;
    const int k = (int)(t % ((long long)p.L * DD));
    const long long n = t / ((long long)p.L * DD);
    const int lvl = k / DD, kk = k - lvl * DD;
    const int i = kk / D, j = kk - i * D;           // x-offset index (slow), y-offset index (fast)
    const int y = q / p.w, x = q - y * p.w;
    const int lh = p.lh[lvl], lw = p.lw[lvl], pw4 = p.pw4[lvl];
    const bool flat_x = lw == 1, flat_y = lh == 1;
    const float fx = p.flow[(n * 2) * hw + q], fy = p.flow[(n * 2 + 1) * hw + q];
    const LkCentre c = lk_centre<0>((float)x + fx, (float)y + fy, 1.0f / (float)(1 << lvl), flat_x, flat_y, R);
    const float* map = p.lvl[lvl] + (n * hw + q) * (long long)p.msz[lvl];
    auto at = [&](int yy, int xx) -> float {
      if (flat_y) yy = 0;
      if (flat_x) xx = 0;
      if ((unsigned)yy >= (unsigned)lh || (unsigned)xx >= (unsigned)lw) return 0.f;
      const int off = pw4 ? (yy >> 2) * pw4 + (xx >> 3) * 32 + (yy & 3) * 8 + (xx & 7) : yy * lw + xx;
      return map[off];
    };
    const int xx = c.x0 + i, yy = c.y0 + j;
    p.out[idx] = lk_blend(at(yy, xx), at(yy, xx + 1), at(yy + 1, xx), at(yy + 1, xx + 1), c.nw, c.ne, c.sw, c.se);
  }
}

// floats per query of pyramid level `level` of an h x w map in the given layout
extern "C" int64_t scf_corr_level_floats(int h, int w, int level, int tiled) {
  if (h <= 0 || w <= 0 || level < 0 || level >= SCF_MAX_LEVELS) return SCF_EINVAL;
  const int64_t lh = h >> level, lw = w >> level;
  if (lh <= 0 || lw <= 0) return SCF_EINVAL;
  return tiled ? ((lh + 3) / 4 * 4) * ((lw + 7) / 8 * 8) : lh * lw;
}

// The layout the lookup likes best (bit l = level l in 8x4-float tiles).  A (2r+2)^2 window costs
// ~ (1 + (2r+1)/8) (1 + (2r+1)/4) 128-byte lines of a tiled map against one or two lines PER ROW of a
// row-major one: tiling pays once a map row is about a line long (>= 24 floats).  Maps that fit
// the window are staged whole and stay row-major; level 0 is written by the correlation GEMM, whose
// fragments are whole tiles (no padding there: w % 8 == 0, h % 4 == 0).
extern "C" unsigned scf_corr_preferred_layout(int h, int w, int r, int L) {
  unsigned mask = 0;
  if (h <= 0 || w <= 0 || r < 1 || L <= 0) return 0;
  const int FW = 2 * r + 2;
  for (int l = 0; l < L && l < SCF_MAX_LEVELS; ++l) {
    const int lh = h >> l, lw = w >> l;
    if (lh <= 0 || lw <= 0) break;
    const bool small = lh <= FW && lw <= FW;
    if (small || lw < 24 || lh < 4) continue;
    if (l == 0 && ((w & 7) || (h & 3))) continue;
    mask |= 1u << l;
  }
  return mask;
}

// scf_tune(SCF_TUNE_LOOKUP_PIPE, v): 0 = the dispatch's own choice, 1 = one group per block, 2 / 3 = the pipelined
// kernel with that many groups per block wherever it fits, 4 / 5 = two / four groups per block (the same waves in
// fewer workgroups)
static std::atomic<int> g_lookup_pipe{0};
int scf_lookup_pipe_set(int v) {
  if (v < 0 || v > 6) return SCF_EINVAL;
  return g_lookup_pipe.exchange(v);
}

// scf_tune(SCF_TUNE_LOOKUP_STORE, v): 0 = the build's policy, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt, 5 plain (r = 4, one group per block)
static std::atomic<int> g_lookup_store{0};
int scf_lookup_store_set(int v) {
  if (v < 0 || v > 5) return SCF_EINVAL;
  return g_lookup_store.exchange(v);
}

static int lookup_launch(const float* const* levels, const float* flow, float* out, int N, int h, int w,
                         int r, int L, unsigned tiled_levels, scf_stream_t stream) {
  if (!levels || !flow || !out || N <= 0 || h <= 0 || w <= 0 || L <= 0 || r < 1) return SCF_EINVAL;
  if (L > SCF_MAX_LEVELS) return SCF_EUNSUPPORTED;
  if ((tiled_levels & 1u) && ((w & 7) || (h & 3))) return SCF_EUNSUPPORTED;   // level 0 is never padded
  LookupParams p;
  bool fast = r <= 4;                                    // the LDS-DMA kernel is instantiated for r = 1..4
  for (int l = 0; l < L; ++l) {
    const int lh = h >> l, lw = w >> l;
    if (!levels[l] || lh <= 0 || lw <= 0) return SCF_EINVAL;
    const bool tiled = (tiled_levels >> l) & 1u;
    p.lvl[l] = levels[l];
    p.lh[l] = lh;
    p.lw[l] = lw;
    const long long msz = scf_corr_level_floats(h, w, l, tiled);
    if (msz > 0x7fffffffLL) return SCF_EUNSUPPORTED;
    p.msz[l] = (int)msz;
    p.pw4[l] = tiled ? ((lw + 7) / 8 * 8) * 4 : 0;
    if (msz > 32767) fast = false;                       // u16 offset tables; one map = one descriptor of < 128 KB
  }
  for (int l = L; l < SCF_MAX_LEVELS; ++l) { p.lvl[l] = nullptr; p.lh[l] = p.lw[l] = p.pw4[l] = p.msz[l] = 0; }
  p.flow = flow;
  p.out = out;
  p.N = N; p.h = h; p.w = w; p.L = L;
  p.total_q = (long long)N * h * w;
  constexpr int qb = 32;   // 32 queries per group: full 128-byte store lines
  const long long ngroups = scf_cdiv(p.total_q, qb);
  if (ngroups > 0x7fffffffLL) return SCF_EUNSUPPORTED;
  p.ngroups = (int)ngroups;
  // per-wave LDS region: wave w stages levels w, w+4, ...; a row-major level whose whole map fits in
  // the (2r+2)^2 footprint is staged whole (stride map|1) + one shared zero row, otherwise as
  // zero-padded footprints (stride FS|1) + the u16 offset tables (2*(2r+2) entries per query)
  const int FW = 2 * r + 2, FSP = (FW * FW) | 1;
  int off = 0;
  for (int wv = 0; wv < 4; ++wv) {
    int need = 0;
    for (int l = wv; l < L; l += 4) {
      const bool small = !p.pw4[l] && p.lh[l] <= FW && p.lw[l] <= FW;
      const int fl = small ? qb * (p.msz[l] | 1) + p.lw[l] : qb * FSP + qb * FW;   // 2*FW u16 = FW floats
      need = need > fl ? need : fl;
    }
    p.woff[wv] = off;
    off += (need + 3) & ~3;                            // 16-byte aligned regions (b128 zero fill)
  }
  const size_t lds = (size_t)off * sizeof(float);
  if (lds > 64 * 1024) fast = false;
  // 32-bit lane offsets from a group's output base: a group of 32 queries spans at most two samples
  if ((long long)2 * L * (2 * r + 1) * (2 * r + 1) * h * w * 4 > 0xffffffffLL) fast = false;
  if (!fast) {
    const long long total = p.total_q * L * (2 * r + 1) * (2 * r + 1);
    long long nb = scf_cdiv(total, 256);
    if (nb > 262144) nb = 262144;
    scf_launch(corr_lookup_generic_kernel, dim3((unsigned)nb), dim3(256), 0, scf_stream(stream), p, r);
    return scf_launch_status();
  }
  int per_cu = (int)((160 * 1024) / (lds + 512));
  per_cu = per_cu > 4 ? 4 : per_cu < 1 ? 1 : per_cu;    // launch bounds: 4 blocks (16 waves) per CU
  long long nblk = (long long)scf_cu_count() * per_cu;
  if (nblk > ngroups) nblk = ngroups;

  // ---- pipelined kernel: G groups per block, their G * L units dealt to the four waves by cost ----
  // (r = 4 and three or four levels -- every configuration the reference ships; anything else keeps v8)
  const int pipe_mode = LK_LAB_PIPE_MODE(g_lookup_pipe.load(std::memory_order_relaxed));
  // the dispatch's own choice (0) is the one-group kernel: the pipelined one is faster on a cache-resident pyramid
  // only and 0.7-1.0 us slower inside the step at batch 8 / 16 / 32 (profiles/r5_lookup_inpipe_ab.txt)
  const int G = (pipe_mode == 2 || pipe_mode == 3) ? pipe_mode : 0;
  if (G && r == 4 && (L == 3 || L == 4) && ngroups >= 2 * G) {
    struct Unit { int gs, lvl, cost, fl; };
    Unit u[3 * 4];
    int nu = 0;
    for (int l = 0; l < L; ++l)                            // levels are in descending cost order already
      for (int gs = 0; gs < G; ++gs) {
        const bool small = !p.pw4[l] && p.lh[l] <= FW && p.lw[l] <= FW;
        u[nu++] = {gs, l, small ? qb * (p.msz[l] > 64 ? 2 : 1) : qb * ((FW * FW + 63) / 64),
                   small ? qb * (p.msz[l] | 1) + p.lw[l] : qb * FSP + qb * FW};
      }
    for (int i = 1; i < nu; ++i)                            // stable insertion sort, most expensive first
      for (int j = i; j > 0 && u[j].cost > u[j - 1].cost; --j) { const Unit t = u[j]; u[j] = u[j - 1]; u[j - 1] = t; }
    int perwave[4] = {0, 0, 0, 0};
    Unit mine[4][LK_MAXU];
    bool fits = true;
    for (int i = 0; i < nu && fits; ++i) {                  // snake: 0 1 2 3 3 2 1 0 0 1 ...
      const int c = i & 3, wv = ((i >> 2) & 1) ? 3 - c : c;
      if (perwave[wv] >= LK_MAXU) { fits = false; break; }
      mine[wv][perwave[wv]++] = u[i];
    }
    int poff = 0, numax = 0;
    if (fits) {
      for (int wv = 0; wv < 4; ++wv) {
        numax = numax > perwave[wv] ? numax : perwave[wv];
        for (int k = 0; k < 4; ++k) p.unit[wv][k] = 0xff;
        for (int k = 0; k < perwave[wv]; ++k) {             // run order: cheapest first
          const Unit& t = mine[wv][perwave[wv] - 1 - k];
          p.unit[wv][k] = (unsigned char)((t.gs << 4) | t.lvl);
          p.uoff[wv][k] = poff;
          poff += (t.fl + 3) & ~3;
        }
      }
    }
    const size_t plds = (size_t)poff * sizeof(float);
    const int nu_k = G == 2 ? 2 : 3;                        // instantiated (G, NU) pairs: (2, 2), (3, 3)
    int pper = fits ? (int)((160 * 1024) / (plds + 512)) : 0;
    pper = pper > 2 ? 2 : pper;
    // worth it only with at least two blocks (8 waves) per CU issuing gathers
    if (fits && numax <= nu_k && pper >= 1) {
      p.gpb = G;
      p.nsg = (int)scf_cdiv(ngroups, G);
      long long pblk = (long long)scf_cu_count() * pper;
      if (pblk > p.nsg) pblk = p.nsg;
      int rc = SCF_OK;
      if (G == 2) {
        static std::atomic<unsigned long long> done{0};
        if (plds > 64 * 1024) rc = scf_raise_dynamic_lds(done, (const void*)corr_lookup_pipe_kernel<4, SCF_LOOKUP_STORE_MODE, 2, 2>, (int)plds);
        if (rc != SCF_OK) return rc;
        LK_LAB_SETUP(p, pblk);
        scf_launch((corr_lookup_pipe_kernel<4, SCF_LOOKUP_STORE_MODE, 2, 2>), dim3((unsigned)pblk), dim3(256), plds, scf_stream(stream), p);
      } else {
        static std::atomic<unsigned long long> done{0};
        if (plds > 64 * 1024) rc = scf_raise_dynamic_lds(done, (const void*)corr_lookup_pipe_kernel<4, SCF_LOOKUP_STORE_MODE, 3, 3>, (int)plds);
        if (rc != SCF_OK) return rc;
        LK_LAB_SETUP(p, pblk);
        scf_launch((corr_lookup_pipe_kernel<4, SCF_LOOKUP_STORE_MODE, 3, 3>), dim3((unsigned)pblk), dim3(256), plds, scf_stream(stream), p);
      }
      return scf_launch_status();
    }
  }
  p.gpb = 1; p.nsg = p.ngroups;
  // several groups per block: the same waves in fewer workgroups (the head of this kernel is the workgroup dispatch:
  // 1024 blocks enter over 2.0 us, 512 over 1.3).  Mode 4 / 5 = two / four groups per block of 512 / 1024 threads.
  // The dispatch's own choice (mode 0): four groups per block when that still fills every CU (>= 4 x CUs groups of
  // four-per-CU blocks: batch 32 at 256 x 256), measured in the step at batch 32 / 16 / 8 (profiles/r5_lookup_inpipe_ab.txt):
  // 20.2 -> 19.4 us at batch 32; at batch 16 / 8 one group per block stays (12.7 vs 16.0, 9.9 vs 15.0 us).
  int gpb = pipe_mode == 4 ? 2 : pipe_mode == 5 ? 4 : pipe_mode == 6 ? 3 : 1;
  // own choice: four groups per block where four one-group blocks fit a CU and the grid still fills every CU; maps
  // with three blocks per CU (configs[4]: 1200 groups on 768 slots) lose with three-group blocks (30.3 -> 32.6 us:
  // the 1.56 rounds are then made of three times coarser pieces)
  if (pipe_mode == 0 && per_cu == 4 && ngroups >= 4LL * scf_cu_count()) gpb = 4;
  if (gpb > 1 && r == 4 && (size_t)gpb * lds + 512 <= 160 * 1024 && ngroups >= gpb) {
    const size_t ldsg = (size_t)gpb * lds;
    p.nsg = (int)(lds / sizeof(float));
    int perg = (int)((160 * 1024) / (ldsg + 512));
    perg = perg > 4 / gpb ? 4 / gpb : perg < 1 ? 1 : perg;
    long long nbg = (long long)scf_cu_count() * perg;
    if (nbg > scf_cdiv(ngroups, gpb)) nbg = scf_cdiv(ngroups, gpb);
    LK_LAB_SETUP(p, nbg);
#define SCF_LKG(G_)                                                                                                   \
    case G_: {                                                                                                        \
      static std::atomic<unsigned long long> done{0};                                                                 \
      if (ldsg > 64 * 1024) {                                                                                         \
        const int rc = scf_raise_dynamic_lds(done, (const void*)corr_lookup_kernel<4, SCF_LOOKUP_STORE_MODE, G_>, (int)ldsg); \
        if (rc != SCF_OK) return rc;                                                                                  \
      }                                                                                                               \
      scf_launch((corr_lookup_kernel<4, SCF_LOOKUP_STORE_MODE, G_>), dim3((unsigned)nbg), dim3(256 * G_), ldsg, scf_stream(stream), p); \
      return scf_launch_status();                                                                                     \
    }
    switch (gpb) { SCF_LKG(2) SCF_LKG(3) SCF_LKG(4) default: break; }
#undef SCF_LKG
  }
  LK_LAB_LAUNCH(p, nblk);
#define SCF_LK(R_)                                                                                 \
  case R_:                                                                                         \
    scf_launch((corr_lookup_kernel<R_, SCF_LOOKUP_STORE_MODE>), dim3((unsigned)nblk), dim3(256), lds, scf_stream(stream), p); \
    break;
#define SCF_LK4(SM_)                                                                               \
  case SM_:                                                                                        \
    scf_launch((corr_lookup_kernel<4, SM_ == 5 ? 0 : SM_>), dim3((unsigned)nblk), dim3(256), lds, scf_stream(stream), p); \
    return scf_launch_status();
  if (r == 4)
    switch (g_lookup_store.load(std::memory_order_relaxed)) {   // A/B knob: store policy of the r = 4 kernel
      SCF_LK4(1) SCF_LK4(2) SCF_LK4(3) SCF_LK4(4) SCF_LK4(5)
      default: break;
    }
#undef SCF_LK4
  switch (r) {
    SCF_LK(4) SCF_LK(3) SCF_LK(2) SCF_LK(1)
    default: return SCF_EUNSUPPORTED;
  }
#undef SCF_LK
  return scf_launch_status();
}

extern "C" int scf_corr_lookup_ex(const float* const* levels, const float* flow, float* out, int N,
                                  int h, int w, int r, int L, unsigned tiled_levels, scf_stream_t stream) {
  return lookup_launch(levels, flow, out, N, h, w, r, L, tiled_levels, stream);
}

extern "C" int scf_corr_lookup(const float* const* levels, const float* flow, float* out, int N,
                               int h, int w, int r, int L, scf_stream_t stream) {
  return lookup_launch(levels, flow, out, N, h, w, r, L, 0u, stream);
}
