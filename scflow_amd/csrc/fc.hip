// The fully connected tail of the pose head (pose_head.py:166-172, 201-211: flatten -> Linear 2048 -> 1024 + ReLU
// -> Linear 1024 -> 256 + ReLU -> rotation_pred | translation_pred) as split-K GEMMs on the matrix cores, with the
// neighbouring element-wise steps folded into the operand loads.
//
// A batch of 32 pairs makes y[n][o] = sum_k W[o][k] x[n][k] a real (if small) GEMM: 32 x 1024 x 2048.  What it
// costs is streaming the weights (8 MB for the first layer) -- the wave-per-feature GEMV of norm.hip re-reads them
// once per 8 samples and needs 16 us per launch; a 32 x 32 output tile per block over the whole K (tried in r2) had
// only 32 blocks to pull 8 MB.  Here a block owns one 32-feature x 32-sample tile of ONE K-slice of 256 features:
// 256 blocks for the first layer, each reading 32 KB of weights and 32 KB of activations exactly once, partial
// sums written per slice; the CONSUMER adds the slices in slice order (+ bias, + ReLU) while it stages its own
// input: no atomics, a fixed summation order, no extra pass.  The producer side folds in the same way: the last
// GroupNorm + ReLU of the pose head (pose_head.py:151-159; a group = 4 channels x 4 x 4 pixels = 64 consecutive
// features of the flattened map) is applied to the staged activation tile in LDS -- a K-slice holds whole groups.
//   block   4 waves; operands staged once in LDS ([32][Ks + 1] floats each: the + 1 makes the 32 rows of an MFMA
//           operand column hit 32 different banks), wave w contracts k in [w Ks / 4, (w + 1) Ks / 4) with
//           v_mfma_f32_32x32x2_f32 (A = W rows, B = sample rows), the four partial tiles are added in wave order
//           through LDS and written with the feature index along the lanes (full 128-byte lines).
//   order   per output: an fma chain over the wave's k range, then w0 + w1 + w2 + w3, then (consumer) slice 0 + 1 + ...
#include "scf_common.h"

typedef float fc_f32x16 __attribute__((ext_vector_type(16)));

#define FC_KS_MAX 256
#define FC_PITCH(ks) ((ks) + 1)

struct FcK {
  const float* x; int parts; long long part_stride;
  const float* x_bias; int x_relu;
  int gn_size, gn_hw; const float* gamma; const float* beta; float eps;      // gn_size = features per group (0: off)
  const float* W; const float* bias; float* y; int O;
  const float* W2; const float* bias2; float* y2; int O2;
  int act, N, K, Ks, slices, tiles1;
};

// GroupNorm + affine + ReLU of one half group (HSZ consecutive floats in LDS), the pair of threads of a group
// exchange their partial sums with one shuffle
template <int HSZ>
__device__ __forceinline__ void fc_group_norm_half(float* xp, int gn_size, float eps, const float* gamma, const float* beta,
                                                   int f0, int gn_hw) {
  float v[HSZ];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < HSZ; ++i) { v[i] = xp[i]; s += v[i]; }
  s += __shfl_xor(s, 1);
  const float mean = s / (float)gn_size;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < HSZ; ++i) { const float a = v[i] - mean; q += a * a; }
  q += __shfl_xor(q, 1);
  const float rstd = 1.0f / sqrtf(q / (float)gn_size + eps);
#pragma unroll
  for (int i = 0; i < HSZ; ++i) {
    const int c = (f0 + i) / gn_hw;
    xp[i] = fmaxf((v[i] - mean) * rstd * gamma[c] + beta[c], 0.f);
  }
}

// KS = the slice width the tile is laid out for (64, 128 or 256 floats per row); the real width p.Ks <= KS
template <int KS>
__global__ __launch_bounds__(256)
void fc_splitk_kernel(FcK p) {
  extern __shared__ __attribute__((aligned(16))) float fc_lds[];
  constexpr int P = FC_PITCH(KS), KQ = KS / 4, NLD = 32 * KQ / 256;     // float4 groups per row, loads per thread
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l32 = lane & 31;
  const int Ks = p.Ks;
  float* Wt = fc_lds;                 // [32][P]
  float* Xt = fc_lds + 32 * P;        // [32][P]
  // which weight matrix / output this feature tile belongs to
  int ot = blockIdx.x;
  const float* W = p.W; const float* bias = p.bias; float* y = p.y; int O = p.O;
  if (ot >= p.tiles1) { ot -= p.tiles1; W = p.W2; bias = p.bias2; y = p.y2; O = p.O2; }
  const int o0 = ot * 32, ks0 = blockIdx.y * Ks, n0 = blockIdx.z * 32;

  // ---- operand tiles: every global load of the block is issued before the first one is used (a block is one
  //      memory round trip, not a chain of them) ----
  float4 wv[NLD], xv[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int e = tid + 256 * i, r = e / KQ, c4 = e - r * KQ;
    wv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (o0 + r < O && 4 * c4 < Ks) wv[i] = *reinterpret_cast<const float4*>(W + (long long)(o0 + r) * p.K + ks0 + 4 * c4);
  }
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int e = tid + 256 * i, r = e / KQ, c4 = e - r * KQ;
    xv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n0 + r < p.N && 4 * c4 < Ks) xv[i] = *reinterpret_cast<const float4*>(p.x + (long long)(n0 + r) * p.K + ks0 + 4 * c4);
  }
  // the producer's other partial buffers, added in slice order.  r6: FOUR slices of loads in flight per round trip (two
  // before): a launch of this kernel is a chain of memory round trips and nothing else -- fc2 reads fc1's eight slices
  // (four round trips -> two), the heads fc2's four (two -> one), fc1 a K-sliced convolution's two or four (-> one).
  // The adds keep their order (slice s, s + 1, s + 2, s + 3), so the sums keep their bits.
  for (int s = 1; s < p.parts; s += 4) {
    float4 u[4][NLD];
    const int left = p.parts - s;                 // 1 ... : slices s .. s + min(left, 4) - 1 exist
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = tid + 256 * i, r = e / KQ, c4 = e - r * KQ;
      const bool in = n0 + r < p.N && 4 * c4 < Ks;
      const float* xp = p.x + (long long)(n0 + r) * p.K + ks0 + 4 * c4 + (long long)s * p.part_stride;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        u[j][i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in && j < left) u[j][i] = *reinterpret_cast<const float4*>(xp + (long long)j * p.part_stride);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < left) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) { xv[i].x += u[j][i].x; xv[i].y += u[j][i].y; xv[i].z += u[j][i].z; xv[i].w += u[j][i].w; }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int e = tid + 256 * i, r = e / KQ, c4 = e - r * KQ;
    float4 v = xv[i];
    if (p.x_bias && 4 * c4 < Ks) {
      const float4 b = *reinterpret_cast<const float4*>(p.x_bias + ks0 + 4 * c4);
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    if (p.x_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (n0 + r >= p.N) v = make_float4(0.f, 0.f, 0.f, 0.f);
    float* dx = Xt + r * P + 4 * c4;
    dx[0] = v.x; dx[1] = v.y; dx[2] = v.z; dx[3] = v.w;
    float* dw = Wt + r * P + 4 * c4;
    dw[0] = wv[i].x; dw[1] = wv[i].y; dw[2] = wv[i].z; dw[3] = wv[i].w;
  }
  __syncthreads();
  // ---- GroupNorm + affine + ReLU on the staged tile: a group = gn_size consecutive features of one sample, two-pass
  //      statistics like group_norm_relu_kernel; a (sample, group) pair is dealt to a pair of threads ----
  if (p.gn_size > 0) {
    const int gpr = Ks / p.gn_size, ngroups = 32 * gpr, hsz = p.gn_size >> 1;
    for (int gi = tid >> 1; gi < ngroups; gi += 128) {
      const int r = gi / gpr, g = gi - r * gpr;
      float* xp = Xt + r * P + g * p.gn_size + (tid & 1) * hsz;
      const int f0 = ks0 + g * p.gn_size + (tid & 1) * hsz;      // global feature index of xp[0]
      if (hsz == 32) {
        fc_group_norm_half<32>(xp, p.gn_size, p.eps, p.gamma, p.beta, f0, p.gn_hw);
      } else {
        float s = 0.f;
        for (int i = 0; i < hsz; ++i) s += xp[i];
        s += __shfl_xor(s, 1);
        const float mean = s / (float)p.gn_size;
        float q = 0.f;
        for (int i = 0; i < hsz; ++i) { const float a = xp[i] - mean; q += a * a; }
        q += __shfl_xor(q, 1);
        const float rstd = 1.0f / sqrtf(q / (float)p.gn_size + p.eps);
        for (int i = 0; i < hsz; ++i) {
          const int c = (f0 + i) / p.gn_hw;
          xp[i] = fmaxf((xp[i] - mean) * rstd * p.gamma[c] + p.beta[c], 0.f);
        }
      }
    }
    __syncthreads();
  }

  // ---- contraction: wave w takes k in [w KS / 4, (w + 1) KS / 4) (columns past Ks hold zeros) ----
  fc_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  constexpr int kw = KS / 4;
  const float* ap = Wt + l32 * P + wave * kw + half;
  const float* bp = Xt + l32 * P + wave * kw + half;
#pragma unroll
  for (int k = 0; k < kw; k += 8) {           // operands of four k-steps requested together
    const float a0 = ap[k], b0 = bp[k], a1 = ap[k + 2], b1 = bp[k + 2], a2 = ap[k + 4], b2 = bp[k + 4], a3 = ap[k + 6], b3 = bp[k + 6];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b3, acc, 0, 0, 0);
  }
  __syncthreads();                    // the operand tiles are dead: their LDS holds the four partial tiles now
  float* red = fc_lds;                // [4 waves][16][64]
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
  // ---- w0 + w1 + w2 + w3, feature index along the lanes ----
  const bool finished = p.slices == 1;
  float* yo = finished ? y : y + (long long)blockIdx.y * p.N * O;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = tid + 256 * i;
    const int n = e >> 5, m = e & 31;
    const int r = 4 * (m >> 3) + (m & 3), ln = ((m >> 2) & 1) * 32 + n;
    float s = ((red[(0 * 16 + r) * 64 + ln] + red[(1 * 16 + r) * 64 + ln]) + red[(2 * 16 + r) * 64 + ln]) + red[(3 * 16 + r) * 64 + ln];
    if (n0 + n < p.N && o0 + m < O) {
      if (finished) {
        if (bias) s += bias[o0 + m];
        s = scf_apply_act(s, p.act);
      }
      yo[(long long)(n0 + n) * O + o0 + m] = s;
    }
  }
}

extern "C" int scf_fc_splitk(const scf_fc_desc* d, scf_stream_t stream) {
  if (!d || !d->x || !d->W || !d->y || d->N <= 0 || d->K <= 0 || d->O <= 0 || d->x_parts < 1 || d->slices < 1)
    return SCF_EINVAL;
  if (d->O2 < 0 || (d->O2 > 0 && (!d->W2 || !d->y2))) return SCF_EINVAL;
  if (d->K % d->slices != 0) return SCF_EUNSUPPORTED;
  const int Ks = d->K / d->slices;
  if (Ks > FC_KS_MAX || (Ks & 7) != 0 || (d->K & 3) != 0) return SCF_EUNSUPPORTED;
  if ((((uintptr_t)d->x | (uintptr_t)d->W | (uintptr_t)d->W2 | (uintptr_t)d->x_bias) & 15) != 0 || (d->x_part_stride & 3) != 0)
    return SCF_EUNSUPPORTED;
  FcK p;
  p.x = d->x; p.parts = d->x_parts; p.part_stride = d->x_part_stride; p.x_bias = d->x_bias; p.x_relu = d->x_relu;
  p.gn_size = 0; p.gn_hw = 1; p.gamma = d->gn_gamma; p.beta = d->gn_beta; p.eps = d->gn_eps;
  if (d->gn_groups > 0) {
    if (!d->gn_gamma || !d->gn_beta || d->gn_hw <= 0 || d->K % d->gn_groups != 0) return SCF_EINVAL;
    p.gn_size = d->K / d->gn_groups; p.gn_hw = d->gn_hw;
    if (Ks % p.gn_size != 0 || (p.gn_size & 1) != 0) return SCF_EUNSUPPORTED;      // a K-slice holds whole groups
  }
  p.W = d->W; p.bias = d->bias; p.y = d->y; p.O = d->O;
  p.W2 = d->W2; p.bias2 = d->bias2; p.y2 = d->y2; p.O2 = d->O2;
  p.act = d->act; p.N = d->N; p.K = d->K; p.Ks = Ks; p.slices = d->slices;
  p.tiles1 = (d->O + 31) / 32;
  if (d->slices > 1 && d->O2 > 0) return SCF_EUNSUPPORTED;       // two heads: finished outputs only
  const int tiles = p.tiles1 + (d->O2 + 31) / 32;
  const int ntiles = (d->N + 31) / 32;
  if (ntiles > 65535) return SCF_EUNSUPPORTED;
  const int KS = Ks <= 64 ? 64 : Ks <= 128 ? 128 : 256;       // tile layout (columns past Ks are zero-filled)
  size_t lds = (size_t)2 * 32 * FC_PITCH(KS) * sizeof(float);
  if (lds < (size_t)4 * 16 * 64 * sizeof(float)) lds = (size_t)4 * 16 * 64 * sizeof(float);
  const dim3 grid((unsigned)tiles, (unsigned)d->slices, (unsigned)ntiles);
  if (KS == 256) {
    static std::atomic<unsigned long long> raised;      // 65.8 KB of dynamic LDS
    const int rc = scf_raise_dynamic_lds(raised, (const void*)fc_splitk_kernel<256>, (int)lds);
    if (rc != SCF_OK) return rc;
    scf_launch(fc_splitk_kernel<256>, grid, dim3(256), lds, scf_stream(stream), p);
  } else if (KS == 128) {
    scf_launch(fc_splitk_kernel<128>, grid, dim3(256), lds, scf_stream(stream), p);
  } else {
    scf_launch(fc_splitk_kernel<64>, grid, dim3(256), lds, scf_stream(stream), p);
  }
  return scf_launch_status();
}
