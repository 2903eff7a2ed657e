// Thin-output convolution (Cout <= 4: the flow / mask prediction layers, xhead predict_layer
// of decoder/raft_decoder.py:256-294).  On the MFMA kernels such a layer pays for a full
// 32-channel fragment (16x-32x wasted matrix work); here it is what it is -- a bandwidth-bound
// reduction over Cin*KH*KW with a handful of FMAs per loaded value -- and runs on the vector
// ALUs straight from NCHW:
//   block = 32 output columns x PY output rows x NCG = 16 channel groups (512 threads);
//   lane <-> column (128-byte coalesced rows), thread = PY vertically adjacent pixels of one
//   channel group (c = cg, cg+NCG, ...): K = 3 loads the centre column of each of the K+PY-1 window rows and
//   takes the neighbours from the adjacent lanes (DPP whole-wave shifts, r5), K = 1 loads what it uses; a
//   block's run time is a chain of dependent memory round trips (22 us per launch with 8 groups x 4 loads in
//   flight, whatever the batch), so the channel loop keeps as many channels in flight as registers allow;
//   weights are staged once per block in LDS and read as wave-uniform broadcasts;
//   the NCG partial sums per pixel are combined through LDS in a fixed order.
#include "scf_common.h"
#include "conv_kernels.h"
#include <type_traits>

#define SCF_THIN_NCG 16
// UNROLL (K = 3): channels whose loads are in flight together per thread -- 2 on full grids (126 registers: two
// blocks per CU; 18.7 -> 13.4 us at batch 32), 4 on grids of at most one block per CU (a lone block is a chain of
// memory round trips: 15.2 -> 7.6 us at batch 1; 16.2 us at batch 32)
// (body as a function of the arguments and the block index: conv_thin_pair_kernel below runs the flow and the mask prediction of an
// iteration -- two different instantiations -- as one launch, r6)
template <int K, int CO, int UNROLL = 2>
__device__ __forceinline__ void conv_thin_body(const ConvK& p, const int bid) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int PY = 2;                        // output rows per thread
  constexpr int NCG = SCF_THIN_NCG, NT = 32 * NCG;
  constexpr int T = K * K, R = K / 2, NR = PY + K - 1;
  const int tid = threadIdx.x, col = tid & 31, cg = tid >> 5;
  const int b = bid;
  const int txi = b % p.tiles_x;
  const int t2 = b / p.tiles_x;
  const int tyi = t2 % p.tiles_y;
  const int n = t2 / p.tiles_y;
  const int x = txi * 32 + col, y0 = tyi * PY;
  const int H = p.H, W = p.W, HW = H * W;

  // weights -> LDS: straight copy of the [c][t][co] packing
  float* wl = lds;
  const int nw4 = (p.Cin * T * CO + 3) / 4;
  for (int e = tid; e < nw4; e += NT)
    reinterpret_cast<float4*>(wl)[e] = reinterpret_cast<const float4*>(p.wthin)[e];
  __syncthreads();

  // per-thread tap geometry (chunk-invariant): offsets of the NR x K window, clamped + masked
  int off[NR][K];
  bool ok[NR][K];
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int d = 0; d < K; ++d) {
      const int iy = y0 - R + r, ix = x - R + d;
      ok[r][d] = iy >= 0 && iy < H && ix >= 0 && ix < W;
      off[r][d] = ok[r][d] ? iy * W + ix : 0;
    }

  float acc[PY][CO];
#pragma unroll
  for (int py = 0; py < PY; ++py)
#pragma unroll
    for (int co = 0; co < CO; ++co) acc[py][co] = 0.f;

  const float* xin = p.in0 + (long long)n * p.in0_ns;
  // generic form: every tap is its own load (K = 1; K = 3 on maps wider than one tile, where a tile's edge columns
  // would need divergent loads inside the unrolled trips: 28.6 us against 40.4 at (8, 256, 60, 80))
  auto run_generic = [&]() {
#pragma unroll 8
    for (int c = cg; c < p.Cin; c += NCG) {
      const float* xc = xin + (long long)c * HW;
      float v[NR][K];
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int d = 0; d < K; ++d) {
          const float t = xc[off[r][d]];
          v[r][d] = ok[r][d] ? t : 0.f;
        }
      const float* wc = wl + c * T * CO;
#pragma unroll
      for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int kx = 0; kx < K; ++kx)
#pragma unroll
          for (int co = 0; co < CO; ++co) {
            const float w = wc[(ky * K + kx) * CO + co];
#pragma unroll
            for (int py = 0; py < PY; ++py) acc[py][co] += v[py + ky][kx] * w;
          }
    }
  };
  if constexpr (K == 3 && UNROLL > 0) {
    // r5: ONE load per window row and channel (the centre column); the left / right neighbours come from the
    // adjacent lanes by whole-wave DPP shifts (lane = column; the two 32-lane halves of a wave are two channel
    // groups, so lanes 0 / 32 and 31 / 63 take zero padding: this instantiation runs maps of ONE tile per row; wider
    // maps take the UNROLL = 0 form, where every tap is its own load).  A third of the load
    // instructions (6 -> 2 per output pixel and channel), so all 16 channels of a group's share are in flight at
    // once: the block is one memory round trip instead of two.
    const bool edge_l = col == 0, edge_r = col == 31;
    int offc[NR];
    bool okc[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) { offc[r] = off[r][1]; okc[r] = ok[r][1]; }
    // the map is one tile wide (the dispatch's condition for this instantiation): a tile's outer neighbours are zero
    // padding, the loop body is branch-free
    {
      // trips of UN channels with a compile-time count (the DPP shifts are convergent operations: a loop with a
      // run-time trip count is not unrolled around them); a channel past Cin re-reads channel cg with zero data
      constexpr int UN = UNROLL;
      for (int c0 = cg; c0 < p.Cin; c0 += UN * NCG) {
        float t[UN][NR];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int c = c0 + u * NCG;
          const float* xc = xin + (long long)(c < p.Cin ? c : cg) * HW;
#pragma unroll
          for (int r = 0; r < NR; ++r) t[u][r] = xc[offc[r]];
        }
        __builtin_amdgcn_sched_barrier(0);        // all UN x NR loads are in flight before the first one is used ...
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int c = c0 + u * NCG;
          const bool cok = c < p.Cin;
          float v[NR][K];
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            v[r][1] = (okc[r] && cok) ? t[u][r] : 0.f;
            // wave_shr:1 (lane i <- lane i - 1, lane 0 <- 0), wave_shl:1 (lane i <- lane i + 1, lane 63 <- 0)
            const int ci = __float_as_int(v[r][1]);
            float l = __int_as_float(__builtin_amdgcn_update_dpp(0, ci, 0x138, 0xf, 0xf, true));
            float rr = __int_as_float(__builtin_amdgcn_update_dpp(0, ci, 0x130, 0xf, 0xf, true));
            l = edge_l ? 0.f : l;           // one tile per row: the tile's outer neighbours are zero padding
            rr = edge_r ? 0.f : rr;
            v[r][0] = l;
            v[r][2] = rr;
          }
          const float* wc = wl + (cok ? c : cg) * T * CO;
#pragma unroll
          for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int kx = 0; kx < K; ++kx)
#pragma unroll
              for (int co = 0; co < CO; ++co) {
                const float w = wc[(ky * K + kx) * CO + co];
#pragma unroll
                for (int py = 0; py < PY; ++py) acc[py][co] += v[py + ky][kx] * w;
              }
          __builtin_amdgcn_sched_barrier(0);      // ... and one channel's weights / shifts at a time (registers)
        }
      }
    }      // (the dispatch sends maps of more than one tile per row to the UNROLL = 0 instantiation)
  } else {
    run_generic();
  }

  // ---- combine the channel groups (fixed order), bias, activation, store ----
  __syncthreads();                                   // weights no longer needed
  float* red = lds;                                  // [cg][py][co][col]
#pragma unroll
  for (int py = 0; py < PY; ++py)
#pragma unroll
    for (int co = 0; co < CO; ++co) red[((cg * PY + py) * CO + co) * 32 + col] = acc[py][co];
  __syncthreads();
  for (int e = tid; e < PY * CO * 32; e += NT) {
    const int c2 = e & 31, q = e >> 5;               // q = py*CO + co
    const int py = q / CO, co = q - py * CO;
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < NCG; ++g) s += red[(g * PY * CO + q) * 32 + c2];
    const int oy = y0 + py, ox = txi * 32 + c2;
    if (co < p.Cout && oy < p.Ho && ox < p.Wo) {
      if (p.bias) s += p.bias[co];
      float* o = p.out + (long long)n * p.out_ns + (long long)co * p.Ho * p.Wo + oy * p.Wo + ox;
      *o = scf_apply_act(s, p.act);
    }
  }
}

template <int K, int CO, int UNROLL = 2>
__global__ __launch_bounds__(32 * SCF_THIN_NCG) void conv_thin_kernel(ConvK p) {
  conv_thin_body<K, CO, UNROLL>(p, (int)blockIdx.x);
}

// r6: the two prediction layers of an iteration (flow: 3x3 256 -> 2, mask: 1x1 256 -> 1; they read disjoint halves of the heads'
// hidden tensor) as ONE launch on grids that leave most of the chip idle (batch 1-4): blocks [0, nba) run a, the rest b
template <int KA, int COA, int UA, int KB, int COB, int UB>
__global__ __launch_bounds__(32 * SCF_THIN_NCG) void conv_thin_pair_kernel(ConvK pa, ConvK pb, int nba) {
  if ((int)blockIdx.x < nba) conv_thin_body<KA, COA, UA>(pa, (int)blockIdx.x);
  else conv_thin_body<KB, COB, UB>(pb, (int)blockIdx.x - nba);
}

template <int K, int CO>
static int launch_thin(const ConvK& k, int nblk, size_t lds_bytes, hipStream_t st) {
  // UNROLL 0 = every tap its own load (K = 1; K = 3 on maps wider than one 32-column tile: its own instantiation, 76
  // registers -- sharing a kernel with the DPP form cost it occupancy: 35.8 vs 28.6 us at (8, 256, 60, 80))
  if (K != 3 || k.tiles_x != 1)
    scf_launch((conv_thin_kernel<K, CO, 0>), dim3(nblk), dim3(32 * SCF_THIN_NCG), lds_bytes, st, k);
  else if (nblk <= scf_cu_count())
    scf_launch((conv_thin_kernel<K, CO, 4>), dim3(nblk), dim3(32 * SCF_THIN_NCG), lds_bytes, st, k);
  else
    scf_launch((conv_thin_kernel<K, CO, 2>), dim3(nblk), dim3(32 * SCF_THIN_NCG), lds_bytes, st, k);
  return scf_launch_status();
}

// SCF_EUNSUPPORTED -> not a thin layer (the caller goes on to the MFMA kernels).
int scf_conv_thin_dispatch(ConvK k, int N, bool dry_run, hipStream_t st, ScfLaunchCap* cap) {
  if (!k.wthin || k.Cout > 4 || k.stride != 1 || k.KH != k.KW || (k.KH != 1 && k.KH != 3)) return SCF_EUNSUPPORTED;
  if (k.pad_h != k.KH / 2 || k.pad_w != k.KW / 2) return SCF_EUNSUPPORTED;
  if (k.in1 || k.w_ns != 0 || k.mode != SCF_CONV_PLAIN || k.res || k.scale || k.act_split > 0 ||
      k.out_tile || k.out_div != 1.0f)
    return SCF_EUNSUPPORTED;
  if (k.Cin < 32) return SCF_EUNSUPPORTED;
  const int CO = k.Cout <= 1 ? 1 : k.Cout <= 2 ? 2 : 4;
  const size_t wbytes = ((size_t)k.Cin * k.T * CO + 3) / 4 * 16;
  const size_t rbytes = (size_t)SCF_THIN_NCG * 2 * CO * 32 * sizeof(float);
  const size_t lds = wbytes > rbytes ? wbytes : rbytes;
  if (lds > 64 * 1024) return SCF_EUNSUPPORTED;
  k.tiles_x = (k.Wo + 31) / 32;
  k.tiles_y = (k.Ho + 1) / 2;
  const long long nblk = (long long)N * k.tiles_x * k.tiles_y;
  if (nblk > 0x7fffffffLL) return SCF_EUNSUPPORTED;
  if (cap) {                // r6: hand the launch back (scf_conv2d_pair); variant = K * 100 + CO * 10 + UNROLL of launch_thin's choice
    cap->k = k; cap->nblk = (int)nblk; cap->ldsb = lds;
    cap->variant = k.KH * 100 + CO * 10 + ((k.KH != 3 || k.tiles_x != 1) ? 0 : nblk <= scf_cu_count() ? 4 : 2);
    return SCF_OK;
  }
  if (dry_run) return SCF_OK;
#define SCF_CASE(K_, C_) if (k.KH == K_ && CO == C_) return launch_thin<K_, C_>(k, (int)nblk, lds, st);
  SCF_CASE(3, 1) SCF_CASE(3, 2) SCF_CASE(3, 4) SCF_CASE(1, 1) SCF_CASE(1, 2) SCF_CASE(1, 4)
#undef SCF_CASE
  return SCF_EUNSUPPORTED;
}

// the one heterogeneous pair the refiner has: flow prediction (3x3, 2 outputs, small-grid unroll) | mask prediction (1x1, 1 output)
int scf_conv_thin_pair_launch(const ScfLaunchCap& a, const ScfLaunchCap& b, hipStream_t st) {
  if (a.variant != 324 || b.variant != 110 || a.nblk <= 0 || b.nblk <= 0) return SCF_EUNSUPPORTED;
  const size_t lds = a.ldsb > b.ldsb ? a.ldsb : b.ldsb;
  scf_launch((conv_thin_pair_kernel<3, 2, 4, 1, 1, 0>), dim3((unsigned)(a.nblk + b.nblk)), dim3(32 * SCF_THIN_NCG), lds, st, a.k, b.k, a.nblk);
  return scf_launch_status();
}
