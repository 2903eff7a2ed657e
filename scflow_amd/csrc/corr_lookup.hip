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
//   load phase : the wave's 64 lanes sweep the 32 x (2r+2)^2 footprint elements in
//                element order -> every wave-load covers contiguous rows of one or two
//                queries' maps (coalesced 40-B runs), exactly the algorithmic bytes are
//                requested, out-of-map taps become 0 (zero padding);
//                values are parked in LDS at an ODD per-query stride (bank-conflict
//                free for the transposed read that follows).
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
  long long total_q;
};

template <int R>
__global__ __launch_bounds__(256, 4) void corr_lookup_kernel(LookupParams p) {
  constexpr int FW = 2 * R + 2;       // footprint width
  constexpr int FS = FW * FW;         // footprint size
  constexpr int FSP = FS | 1;         // odd LDS stride per query
  constexpr int D = 2 * R + 1;        // window width
  constexpr int QB = 32;              // queries per block
  constexpr int NSET = (FS + 63) / 64;  // wave-loads per query footprint
  extern __shared__ float lds_fp[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
  const int l32 = lane & 31, half = lane >> 5;
  const long long gq0 = (long long)blockIdx.x * QB;
  const int hw = p.h * p.w;
  const int ktot = p.L * D * D;

  // this lane's query (both half-waves hold the same 32 queries)
  const long long gq = gq0 + l32;
  const bool qvalid = gq < p.total_q;
  int n = 0, q = 0;
  float qx = 0.f, qy = 0.f;
  if (qvalid) {
    n = (int)(gq / hw);
    q = (int)(gq - (long long)n * hw);
    const int y = q / p.w, x = q - y * p.w;
    const float* fl = p.flow + (long long)n * 2 * hw + q;
    qx = (float)x + fl[0];
    qy = (float)y + fl[hw];
  }
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
    const long long msz = (long long)lh * lw;
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

    // Small maps (coarse levels: the whole map is no larger than the window footprint) are
    // staged whole instead of as a zero-padded footprint: fewer LDS bytes per query (which is
    // what lets 4 blocks share a CU and the grid finish in ONE wave of blocks at batch 32) and
    // the maps of the block's 32 queries are one contiguous, fully coalesced run in memory.
    const bool small = (lh <= FW && lw <= FW);
    const int S = small ? ((int)msz | 1) : FSP;       // odd per-query LDS stride
    if (small) {
      const float* lbase = p.lvl[lvl] + gq0 * msz;
      const int tot = nq * (int)msz;
      for (int f0 = 0; f0 < tot; f0 += 64 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int f = f0 + u * 64 + lane;
          v[u] = __builtin_nontemporal_load(lbase + (f < tot ? f : 0));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int f = f0 + u * 64 + lane;
          if (f < tot) {
            const int qq = f / (int)msz;
            myfp[qq * S + (f - qq * (int)msz)] = v[u];
          }
        }
      }
    } else {
    // ---- load phase: one query per step, lanes <-> footprint elements.  x0/y0 and the map
      //      base are wave-uniform per step (scalar registers); per lane only a bounds test and
      //      a 32-bit offset remain.  8 queries (8*NSET loads) are in flight per wave.
      const float* lbase = p.lvl[lvl] + gq0 * msz;
      constexpr int QU = 4;             // queries whose loads are in flight together (x16 waves/CU)
      for (int qb = 0; qb < QB; qb += QU) {
        float v[QU][NSET];
        unsigned okmask = 0;
        // branch-free: out-of-map / out-of-range taps read element 0 of a valid map and are
        // zeroed afterwards, so all QU*NSET loads issue back to back.
#pragma unroll
        for (int u = 0; u < QU; ++u) {
          const int qq = qb + u;
          const int sx0 = flat_x ? 0 : __builtin_amdgcn_readlane(x0, qq);
          const int sy0 = flat_y ? 0 : __builtin_amdgcn_readlane(y0, qq);
          const bool qok = qq < nq;
          const float* mb = lbase + (qok ? (long long)qq * msz : 0);
#pragma unroll
          for (int s = 0; s < NSET; ++s) {
            const int xx = flat_x ? 0 : sx0 + ecol[s];
            const int yy = flat_y ? 0 : sy0 + erow[s];
            const bool ok = qok && (unsigned)xx < (unsigned)lw && (unsigned)yy < (unsigned)lh &&
                            (NSET * 64 == FS || lane + 64 * s < FS);
            const int idx = ok ? yy * lw + xx : 0;
            v[u][s] = __builtin_nontemporal_load(mb + idx);
            okmask |= (ok ? 1u : 0u) << (u * NSET + s);
          }
        }
#pragma unroll
        for (int u = 0; u < QU; ++u) {
#pragma unroll
          for (int s = 0; s < NSET; ++s) {
            const float val = ((okmask >> (u * NSET + s)) & 1u) ? v[u][s] : 0.f;
            if (NSET * 64 == FS || lane + 64 * s < FS) myfp[(qb + u) * FSP + lane + 64 * s] = val;
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();  // same-wave LDS ops complete in order

    // ---- compute + store: lane = (query, half); halves split the x-offsets ----
    const float tx = cx - x0f, ty = cy - y0f;
    const float wx0 = (x0f + 1.f) - cx, wy0 = (y0f + 1.f) - cy;   // grid_sample: (x_se - x)
    const float nw = wx0 * wy0, ne = tx * wy0, sw = wx0 * ty, se = tx * ty;
    const float* f = myfp + l32 * S;
    const int i0 = half ? (D + 1) / 2 : 0;
    const int i1 = half ? D : (D + 1) / 2;
    // column c of the window (x = x0 + c), rows 0..FW-1
    int rowoff[FW];
    unsigned rowok = 0;
    if (small) {
#pragma unroll
      for (int r = 0; r < FW; ++r) {
        const int yy = flat_y ? 0 : y0 + r;
        const bool ok = (unsigned)yy < (unsigned)lh;
        rowoff[r] = ok ? yy * lw : 0;
        rowok |= (ok ? 1u : 0u) << r;
      }
    }
    auto column = [&](int c, float (&col)[FW]) {
      if (small) {
        const int xx = flat_x ? 0 : x0 + c;
        const bool cok = (unsigned)xx < (unsigned)lw;
        const int xo = cok ? xx : 0;
#pragma unroll
        for (int r = 0; r < FW; ++r) {
          const float v = f[rowoff[r] + xo];
          col[r] = (cok && ((rowok >> r) & 1u)) ? v : 0.f;
        }
      } else {
#pragma unroll
        for (int r = 0; r < FW; ++r) col[r] = f[r * FW + c];
      }
    };
    float colA[FW], colB[FW];
    column(i0, colA);
    // uniform channel base + per-lane query offset: the compiler keeps the base in SGPRs
    float* obase = p.out + ((long long)n * ktot + (long long)lvl * D * D) * hw + q;
    for (int i = i0; i < i1; ++i) {
      column(i + 1, colB);
      if (qvalid) {
        float* oc = obase + (long long)(i * D) * hw;
#pragma unroll
        for (int j = 0; j < D; ++j) {
          const float v = colA[j] * nw + colB[j] * ne + colA[j + 1] * sw + colB[j + 1] * se;
          oc[(long long)j * hw] = v;
        }
      }
#pragma unroll
      for (int r = 0; r < FW; ++r) colA[r] = colB[r];
    }
    __builtin_amdgcn_wave_barrier();
  }
}

extern "C" int scf_corr_lookup(const float* const* levels, const float* flow, float* out, int N,
                               int h, int w, int r, int L, scf_stream_t stream) {
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
  const int nblk = (int)scf_cdiv(p.total_q, 32);
  // per-wave LDS region: wave w stages levels w, w+4, ...; a level whose whole map fits in the
  // (2r+2)^2 footprint is staged whole (stride map|1), otherwise as a footprint (stride FS|1)
  const int FWh = 2 * r + 2, FSPh = (FWh * FWh) | 1;
  int off = 0;
  for (int wv = 0; wv < 4; ++wv) {
    int need = 0;
    for (int l = wv; l < L; l += 4) {
      const bool small = p.lh[l] <= FWh && p.lw[l] <= FWh;
      const int S = small ? ((p.lh[l] * p.lw[l]) | 1) : FSPh;
      need = need > 32 * S ? need : 32 * S;
    }
    p.woff[wv] = off;
    off += need;
  }
  const size_t lds = (size_t)off * sizeof(float);
  switch (r) {
    case 4: hipLaunchKernelGGL(corr_lookup_kernel<4>, dim3(nblk), dim3(256), lds, scf_stream(stream), p); break;
    case 3: hipLaunchKernelGGL(corr_lookup_kernel<3>, dim3(nblk), dim3(256), lds, scf_stream(stream), p); break;
    case 2: hipLaunchKernelGGL(corr_lookup_kernel<2>, dim3(nblk), dim3(256), lds, scf_stream(stream), p); break;
    case 1: hipLaunchKernelGGL(corr_lookup_kernel<1>, dim3(nblk), dim3(256), lds, scf_stream(stream), p); break;
    default: return SCF_EUNSUPPORTED;
  }
  return scf_launch_status();
}
