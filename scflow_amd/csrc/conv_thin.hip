// Thin-output convolution (Cout <= 4: the flow / mask prediction layers, xhead predict_layer
// of decoder/raft_decoder.py:256-294).  On the MFMA kernels such a layer pays for a full
// 32-channel fragment (16x-32x wasted matrix work); here it is what it is -- a bandwidth-bound
// reduction over Cin*KH*KW with a handful of FMAs per loaded value -- and runs on the vector
// ALUs straight from NCHW:
//   block = 32 output columns x PY output rows x NCG = 16 channel groups (512 threads);
//   lane <-> column (128-byte coalesced rows), thread = PY vertically adjacent pixels of one
//   channel group (c = cg, cg+NCG, ...): (K+PY-1)*K loads feed PY*K*K*Cout FMAs per channel,
//   eight channels' loads in flight per thread: a block's run time is a chain of dependent
//   memory round trips (22 us per launch with 8 groups x 4 in flight, whatever the batch), so
//   the channel loop is made as short as the register file allows;
//   weights are staged once per block in LDS and read as wave-uniform broadcasts;
//   the NCG partial sums per pixel are combined through LDS in a fixed order.
#include "scf_common.h"
#include "conv_kernels.h"

#define SCF_THIN_NCG 16
template <int K, int CO>
__global__ __launch_bounds__(32 * SCF_THIN_NCG) void conv_thin_kernel(ConvK p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int PY = 2;                        // output rows per thread
  constexpr int NCG = SCF_THIN_NCG, NT = 32 * NCG;
  constexpr int T = K * K, R = K / 2, NR = PY + K - 1;
  const int tid = threadIdx.x, col = tid & 31, cg = tid >> 5;
  const int b = blockIdx.x;
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

template <int K, int CO>
static int launch_thin(const ConvK& k, int nblk, size_t lds_bytes, hipStream_t st) {
  scf_launch((conv_thin_kernel<K, CO>), dim3(nblk), dim3(32 * SCF_THIN_NCG), lds_bytes, st, k);
  return scf_launch_status();
}

// SCF_EUNSUPPORTED -> not a thin layer (the caller goes on to the MFMA kernels).
int scf_conv_thin_dispatch(ConvK k, int N, bool dry_run, hipStream_t st) {
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
  if (dry_run) return SCF_OK;
#define SCF_CASE(K_, C_) if (k.KH == K_ && CO == C_) return launch_thin<K_, C_>(k, (int)nblk, lds, st);
  SCF_CASE(3, 1) SCF_CASE(3, 2) SCF_CASE(3, 4) SCF_CASE(1, 1) SCF_CASE(1, 2) SCF_CASE(1, 4)
#undef SCF_CASE
  return SCF_EUNSUPPORTED;
}
