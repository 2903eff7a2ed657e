// Multi-scale correlation lookup (RAFT/SCFlow "CorrLookup") for gfx950.
//
// Reference semantics: models/utils/corr_lookup.py:102-136 (+ bilinear_sample :31-67).
//
// Roofline: HBM-bound gather.  Per query and level the kernel touches the (2r+2)^2
// footprint of that query's private correlation map once (400 B at r=4) and writes
// (2r+1)^2 outputs (324 B); no byte is shared between queries, so the algorithmic
// traffic is 4*(400+324)+8 = 2904 B/query.
//
// Work decomposition (wave64):
//   block = 256 threads = 4 waves handling the same 32 consecutive queries;
//   wave w owns pyramid level w (levels w, w+4, ... when L > 4).
//   load phase : per query, the wave's 64 lanes fetch the (2r+2)^2 footprint elements in
//                element order with global_load_lds (memory -> LDS, no VGPR staging, so all
//                64 gathers of a wave are in flight at once); exactly the algorithmic bytes
//                are requested; each query's window lands contiguously in LDS at an ODD
//                per-query stride (bank-conflict free for the transposed read that follows);
//                out-of-map taps: the staging area is zero-filled first and their lanes are
//                switched off (zero padding without fetching anything).
//   compute    : lane = (query, half); the two half-waves split the x-offsets; bilinear
//                weights are per (query, level) constants because offsets are integers.
//   store      : out[n, k, y, x]; each half-wave writes 32 consecutive queries of one
//                channel = one full 128-B line.
#include "scf_common.h"

struct LookupParams {
  const float* lvl[SCF_MAX_LEVELS];
  int lh[SCF_MAX_LEVELS];
  int lw[SCF_MAX_LEVELS];
  const float* flow;
  float* out;
  int N, h, w, L;
  int woff[4];            // LDS offset (floats) of each wave's staging region
  int l0_tiled;           // level 0 stored in 8x4-float tiles (scf_corr_build_ex)
  long long total_q;
};

// window read-back + bilinear blend + store for one (wave, level).  SMALL: the level's whole
// map is staged (stride S) and out-of-map taps are masked here; otherwise the zero-padded
// (2r+2)^2 footprint is staged and read unmasked.
template <int R, bool SMALL, int NP>
__device__ __forceinline__ void lookup_emit(const float* f, int lw, int lh, int x0, int y0,
                                            bool flat_x, bool flat_y, float nw, float ne, float sw,
                                            float se, int part, char* obase, unsigned lane_off,
                                            size_t cs, bool qvalid) {
  constexpr int FW = 2 * R + 2, D = 2 * R + 1;
  // The NP lane groups of a wave (64 / NP queries each) split the D x-offsets: group g takes
  // columns [i0(g), i0(g+1)), i0(g) = ceil(g*D / NP).  The loop itself is wave-uniform (NI =
  // the largest share; shorter shares are masked) so that store bases stay in SGPRs and only a
  // 32-bit per-lane offset goes into the address VGPR.
  constexpr int NI = (D + NP - 1) / NP;
  const int i0 = (part * D + NP - 1) / NP;
  const int cnt = ((part + 1) * D + NP - 1) / NP - i0;
  const unsigned lane_off2 = lane_off + (unsigned)((size_t)i0 * D * cs);
  int rowoff[FW];
  unsigned rowok = 0;
  if constexpr (SMALL) {
#pragma unroll
    for (int r = 0; r < FW; ++r) {
      const int yy = flat_y ? 0 : y0 + r;
      const bool ok = (unsigned)yy < (unsigned)lh;
      rowoff[r] = ok ? yy * lw : 0;
      rowok |= (ok ? 1u : 0u) << r;
    }
  }
  const float* fbase = f + (SMALL ? 0 : i0);
  auto column = [&](int it, float (&col)[FW]) {
    if constexpr (SMALL) {
      const int xx = flat_x ? 0 : x0 + i0 + it;
      const bool cok = (unsigned)xx < (unsigned)lw;
      const float* fc = fbase + (cok ? xx : 0);
#pragma unroll
      for (int r = 0; r < FW; ++r) {
        float v = fc[rowoff[r]];
        asm volatile("" : "+v"(v));      // keep the (always in-range) read unconditional
        col[r] = (cok && ((rowok >> r) & 1u)) ? v : 0.f;
      }
    } else {
#pragma unroll
      for (int r = 0; r < FW; ++r) col[r] = fbase[r * FW + it];      // immediate offsets
    }
  };
  float col[2][FW];
  column(0, col[0]);
#pragma unroll
  for (int it = 0; it < NI; ++it) {
    float (&cA)[FW] = col[it & 1];
    float (&cB)[FW] = col[(it & 1) ^ 1];
    column(it + 1, cB);
    if (qvalid && it < cnt) {
      char* oc = obase + (size_t)(it * D) * cs;      // wave-uniform channel base (SGPRs)
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const float v = cA[j] * nw + cB[j] * ne + cA[j + 1] * sw + cB[j + 1] * se;
        *(float*)(oc + (size_t)j * cs + lane_off2) = v;
      }
    }
  }
}

template <int R, bool TILED0, int QB>
__global__ __launch_bounds__(256, QB == 32 ? 4 : 8) void corr_lookup_kernel(LookupParams p) {
  constexpr int FW = 2 * R + 2;       // footprint width
  constexpr int FS = FW * FW;         // footprint size
  constexpr int FSP = FS | 1;         // odd LDS stride per query
  constexpr int D = 2 * R + 1;        // window width
  constexpr int NP = 64 / QB;         // lane groups per wave (QB = queries per block: 32 or 16)
  constexpr int NSET = (FS + 63) / 64;  // wave-loads per query footprint
  constexpr int AUX_NT = 2;           // nt: every footprint byte is read exactly once
  extern __shared__ __attribute__((aligned(16))) float lds_fp[];
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
  const int l32 = lane & (QB - 1), half = lane / QB;     // query within the block, lane group
  const long long gq0 = (long long)blockIdx.x * QB;
  const int hw = p.h * p.w;
  const int ktot = p.L * D * D;
  const int n0 = (int)(gq0 / hw);     // sample of the block's first query (block-uniform)

  // this lane's query (both half-waves hold the same 32 queries)
  const long long gq = gq0 + l32;
  const bool qvalid = gq < p.total_q;
  int n = n0, q = 0;
  float qx = 0.f, qy = 0.f;
  if (qvalid) {
    n = (int)(gq / hw);
    q = (int)(gq - (long long)n * hw);
    const int y = q / p.w, x = q - y * p.w;
    const float* fl = p.flow + (long long)n * 2 * hw + q;
    qx = (float)x + fl[0];
    qy = (float)y + fl[hw];
  }
  // byte offset of this lane's query from the block-uniform base out[n0, k, 0, 0]
  const unsigned lane_off = (unsigned)(((long long)(n - n0) * ktot * hw + q) * 4);
  const size_t cs = (size_t)hw * 4;   // channel stride in bytes
  float* myfp = lds_fp + p.woff[wave];

  // footprint element(s) this lane fetches for EVERY query: e = lane + 64*s
  int erow[NSET], ecol[NSET];
#pragma unroll
  for (int s = 0; s < NSET; ++s) {
    const int e = lane + 64 * s;
    erow[s] = e / FW;
    ecol[s] = e - erow[s] * FW;
  }
  const int nq = (int)((p.total_q - gq0) < QB ? (p.total_q - gq0) : QB);   // block-uniform

  for (int lvl = wave; lvl < p.L; lvl += 4) {
    const int lh = p.lh[lvl], lw = p.lw[lvl];
    const int msz = lh * lw;
    // Reference quirk at degenerate sizes: coordinates are normalised with max(size-1, 1) and
    // grid_sample(align_corners=True) de-normalises with (size-1), so along a size-1 axis
    // EVERY tap lands exactly on index 0 (corr_lookup.py:64-67).
    const bool flat_x = lw == 1, flat_y = lh == 1;
    const float inv = 1.0f / (float)(1 << lvl);
    float cx = flat_x ? (float)R : qx * inv;          // exact power-of-two scaling
    float cy = flat_y ? (float)R : qy * inv;
    cx = fminf(fmaxf(cx, -30000.f), 30000.f);         // far outside any map -> all taps 0
    cy = fminf(fmaxf(cy, -30000.f), 30000.f);
    if (!(cx == cx)) cx = -30000.f;                   // NaN flow: treat as out of range
    if (!(cy == cy)) cy = -30000.f;
    const float x0f = floorf(cx), y0f = floorf(cy);
    const int x0 = (int)x0f - R, y0 = (int)y0f - R;
    const char* lbase = (const char*)(p.lvl[lvl] + gq0 * msz);

    // Staging is LDS-DMA (global_load_lds_dword): memory -> LDS without passing through VGPRs,
    // so ALL of a wave's gathers are in flight together (a register-staged version was
    // latency-bound at 4 queries in flight per wave).  One query per step: the map base and the
    // window origin are wave-uniform (readlane -> SGPRs), each lane adds a 32-bit offset.
    //
    // Small maps (coarse levels: the whole map is no larger than the window footprint) are
    // staged whole instead of as a zero-padded footprint: fewer LDS bytes per query (which is
    // what lets 4 blocks share a CU and the grid finish in ONE wave of blocks at batch 32).
    const bool small = (lh <= FW && lw <= FW);
    const bool tiled = TILED0 && lvl == 0;
    const int S = small ? (msz | 1) : FSP;            // odd per-query LDS stride
    if (small) {
#pragma unroll 4
      for (int qq = 0; qq < QB; ++qq) {
        if (qq < nq) {
          const char* mb = lbase + (size_t)qq * msz * 4;
#pragma unroll
          for (int s = 0; s < NSET; ++s)
            if (lane + 64 * s < msz)
              __builtin_amdgcn_global_load_lds((gptr_t)(mb + (unsigned)(lane + 64 * s) * 4u),
                                               (lptr_t)(myfp + qq * S + 64 * s), 4, 0, AUX_NT);
        }
      }
    } else {
      // zero padding = zero-filled staging area + lanes of out-of-map taps switched off
      for (int i = lane; i < QB * FSP / 4; i += 64)
        ((float __attribute__((ext_vector_type(4)))*)myfp)[i] = 0.f;
      // Per query (computed once, vectorised over the 32 queries in lanes): a bit mask of the
      // in-map window columns (bits 0..FW-1) and rows (bits 16..16+FW-1), and the element
      // offset of the window origin.  Per (lane, set) constants: that element's column/row bit
      // pair and its offset from the origin.  A tap is fetched iff both of its bits are set, and
      // (row-major maps) its address is scalar origin + per-lane constant: 2 VALU per gather.
      auto span = [&](int o, int n) -> unsigned {      // window positions c with 0 <= o + c < n
        const int lo = min(max(-o, 0), FW), hi = max(min(n - o, FW), 0);
        return hi > lo ? (1u << hi) - (1u << lo) : 0u;
      };
      const unsigned qmask = (flat_x ? (1u << FW) - 1u : span(x0, lw)) |
                             ((flat_y ? (1u << FW) - 1u : span(y0, lh)) << 16);
      const int qorg = (flat_y ? 0 : y0 * lw) + (flat_x ? 0 : x0);
      unsigned ebits[NSET], eoff[NSET];
#pragma unroll
      for (int s = 0; s < NSET; ++s) {
        const bool live = NSET * 64 == FS || lane + 64 * s < FS;
        ebits[s] = live ? (1u << ecol[s]) | (1u << (16 + erow[s])) : 0x80000000u;
        eoff[s] = (unsigned)(((flat_y ? 0 : erow[s] * lw) + (flat_x ? 0 : ecol[s])) * 4);
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);             // lgkmcnt(0): zeros land before the DMA data
#pragma unroll 4
      for (int qq = 0; qq < QB; ++qq) {
        if (qq < nq) {
          const unsigned m = (unsigned)__builtin_amdgcn_readlane((int)qmask, qq);
          const char* mb = lbase + (size_t)qq * msz * 4;
          if (!TILED0 || !tiled) {
            const char* org = mb + (long long)__builtin_amdgcn_readlane(qorg, qq) * 4;
#pragma unroll
            for (int s = 0; s < NSET; ++s)
              if ((ebits[s] & m) == ebits[s])
                __builtin_amdgcn_global_load_lds((gptr_t)(org + eoff[s]),
                                                 (lptr_t)(myfp + qq * FSP + 64 * s), 4, 0, AUX_NT);
          } else {
            const int sx0 = __builtin_amdgcn_readlane(x0, qq);
            const int sy0 = __builtin_amdgcn_readlane(y0, qq);
#pragma unroll
            for (int s = 0; s < NSET; ++s) {
              const int xx = sx0 + ecol[s], yy = sy0 + erow[s];
              // 24-bit multiply (full rate); lanes with out-of-range xx/yy are switched off
              const int lin = __mul24(yy >> 2, lw * 4) + ((xx >> 3) << 5) + ((yy & 3) << 3) + (xx & 7);
              if ((ebits[s] & m) == ebits[s])
                __builtin_amdgcn_global_load_lds((gptr_t)(mb + (unsigned)lin * 4u),
                                                 (lptr_t)(myfp + qq * FSP + 64 * s), 4, 0, AUX_NT);
            }
          }
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): the DMA data is in LDS
    __builtin_amdgcn_wave_barrier();

    // ---- read back + blend + store: lane = (query, half); halves split the x-offsets ----
    const float tx = cx - x0f, ty = cy - y0f;
    const float wx0 = (x0f + 1.f) - cx, wy0 = (y0f + 1.f) - cy;   // grid_sample: (x_se - x)
    const float nw = wx0 * wy0, ne = tx * wy0, sw = wx0 * ty, se = tx * ty;
    const float* f = myfp + l32 * S;
    char* obase = (char*)p.out + ((size_t)n0 * ktot + (size_t)lvl * D * D) * cs;
    if (small)
      lookup_emit<R, true, NP>(f, lw, lh, x0, y0, flat_x, flat_y, nw, ne, sw, se, half, obase, lane_off, cs, qvalid);
    else
      lookup_emit<R, false, NP>(f, lw, lh, x0, y0, flat_x, flat_y, nw, ne, sw, se, half, obase, lane_off, cs, qvalid);
    __builtin_amdgcn_wave_barrier();
  }
}

static int lookup_launch(const float* const* levels, const float* flow, float* out, int N, int h, int w,
                         int r, int L, int level0_tiled, scf_stream_t stream) {
  if (level0_tiled && ((w & 7) || (h & 3) || h <= 2 * r + 2 || w <= 2 * r + 2)) return SCF_EUNSUPPORTED;
  if (!levels || !flow || !out || N <= 0 || h <= 0 || w <= 0 || L <= 0) return SCF_EINVAL;
  if (L > SCF_MAX_LEVELS) return SCF_EUNSUPPORTED;
  LookupParams p;
  int lh = h, lw = w;
  for (int l = 0; l < L; ++l) {
    if (!levels[l] || lh <= 0 || lw <= 0) return SCF_EINVAL;
    p.lvl[l] = levels[l];
    p.lh[l] = lh;
    p.lw[l] = lw;
    lh /= 2;
    lw /= 2;
  }
  p.flow = flow;
  p.out = out;
  p.N = N; p.h = h; p.w = w; p.L = L;
  p.total_q = (long long)N * h * w;
  p.l0_tiled = level0_tiled ? 1 : 0;
  // 32 queries per block (full 128-byte store lines).  16 per block (2048 blocks, 8 per CU, 64-byte
  // store segments) was measured slower: 30.5 vs 24.6 us at batch 32.
  constexpr int qb = 32;
  const int nblk = (int)scf_cdiv(p.total_q, qb);
  // per-wave LDS region: wave w stages levels w, w+4, ...; a level whose whole map fits in the
  // (2r+2)^2 footprint is staged whole (stride map|1), otherwise as a footprint (stride FS|1)
  const int FWh = 2 * r + 2, FSPh = (FWh * FWh) | 1;
  int off = 0;
  for (int wv = 0; wv < 4; ++wv) {
    int need = 0;
    for (int l = wv; l < L; l += 4) {
      const bool small = p.lh[l] <= FWh && p.lw[l] <= FWh;
      const int S = small ? ((p.lh[l] * p.lw[l]) | 1) : FSPh;
      need = need > qb * S ? need : qb * S;
    }
    p.woff[wv] = off;
    off += (need + 3) & ~3;                            // 16-byte aligned regions (b128 zero fill)
  }
  const size_t lds = (size_t)off * sizeof(float);
#define SCF_LK2(R_, T_, Q_)                                                                         \
  scf_launch((corr_lookup_kernel<R_, T_, Q_>), dim3(nblk), dim3(256), lds, scf_stream(stream), p)
#define SCF_LK(R_)                                                                                 \
  case R_:                                                                                         \
    if (level0_tiled) { SCF_LK2(R_, true, 32); } else { SCF_LK2(R_, false, 32); }                  \
    break;
  switch (r) {
    SCF_LK(4) SCF_LK(3) SCF_LK(2) SCF_LK(1)
    default: return SCF_EUNSUPPORTED;
  }
#undef SCF_LK2
#undef SCF_LK
  return scf_launch_status();
}

extern "C" int scf_corr_lookup_ex(const float* const* levels, const float* flow, float* out, int N,
                                  int h, int w, int r, int L, int level0_tiled, scf_stream_t stream) {
  return lookup_launch(levels, flow, out, N, h, w, r, L, level0_tiled, stream);
}

extern "C" int scf_corr_lookup_timed(const float* const* levels, const float* flow, float* out, int N,
                                     int h, int w, int r, int L, int level0_tiled, scf_timer_t timer,
                                     scf_stream_t stream) {
  if (!timer) return SCF_EINVAL;
  scf_timer_arm(timer);
  const int rc = lookup_launch(levels, flow, out, N, h, w, r, L, level0_tiled, stream);
  scf_timer_arm(nullptr);
  return rc;
}

extern "C" int scf_corr_lookup(const float* const* levels, const float* flow, float* out, int N,
                               int h, int w, int r, int L, scf_stream_t stream) {
  return lookup_launch(levels, flow, out, N, h, w, r, L, 0, stream);
}
