// Stand-alone probe (not part of the product): how fast can the conv kernel's MFMA phase run
// from LDS alone?  Variant 0 = the shipping loop shape (ds_read_b32 operands read right before
// use); variant 1 = b128 operand reads (4 k-steps per read) with a one-iteration software
// pipeline.  No staging; reports TFLOP/s.  `mfma_probe 1` fills LDS with random operands instead
// of small constants (the rate is data dependent: 149 vs 138 TF/s on <2,2>).
// build: hipcc --offload-arch=gfx950 -O3 mfma_loop_probe.hip -o bin/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int WM, int WN>
__global__ __launch_bounds__(256, 2) void probe_old(float* out, int nchunk, int T, int KW, int PW, int PHW, int rnd) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int KC = 8, NCP = 4, BM = WM * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, half = lane >> 5;
  for (int i = tid; i < 12000; i += 256) { unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15; lds[i] = rnd ? ((float)(h & 0xFFFFFF) / 8388608.0f - 1.0f) : (float)(i & 15) * 0.01f; }
  __syncthreads();
  float* wl = lds; float* pl = lds + KC * T * BM;
  int boff[WN];
  for (int j = 0; j < WN; ++j) boff[j] = ((wave * WN + j)) * PW + l32 + half * PHW;
  f32x16 acc[WM][WN];
  for (int i = 0; i < WM; ++i) for (int j = 0; j < WN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int c = 0; c < nchunk; ++c) {
    __syncthreads();
    for (int t = 0; t < T; ++t) {
      const int ky = t / KW, kx = t - ky * KW;
      const float* wt = wl + (t * KC + half) * BM + l32;
      const float* pt = pl + ky * PW + kx;
#pragma unroll
      for (int cc = 0; cc < NCP; ++cc) {
        const int cp = 2 * cc;
        float a[WM], b[WN];
#pragma unroll
        for (int i = 0; i < WM; ++i) a[i] = wt[cp * BM + i * 32];
#pragma unroll
        for (int j = 0; j < WN; ++j) b[j] = pt[cp * PHW + boff[j]];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < WM; ++i) for (int j = 0; j < WN; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + tid] = s;
}

// b128 operands: weights [t][h][m][4], patch [h][PH][PW][4]; one-iteration software pipeline
template <int WM, int WN>
__global__ __launch_bounds__(256, 2) void probe_new(float* out, int nchunk, int T, int KW, int PW, int PHW, int rnd) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int KC = 8, BM = WM * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, half = lane >> 5;
  for (int i = tid; i < 12000; i += 256) { unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15; lds[i] = rnd ? ((float)(h & 0xFFFFFF) / 8388608.0f - 1.0f) : (float)(i & 15) * 0.01f; }
  __syncthreads();
  const f32x4* wl = (const f32x4*)lds;                       // [t][h][BM]
  const f32x4* pl = (const f32x4*)(lds + KC * T * BM);       // [h][PHW]
  int boff[WN];
  for (int j = 0; j < WN; ++j) boff[j] = ((wave * WN + j)) * PW + l32 + half * PHW;
  f32x16 acc[WM][WN];
  for (int i = 0; i < WM; ++i) for (int j = 0; j < WN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int c = 0; c < nchunk; ++c) {
    __syncthreads();
    f32x4 a[2][WM], b[2][WN];
    auto load = [&](int t, f32x4 (&aa)[WM], f32x4 (&bb)[WN]) {
      const int ky = t / KW, kx = t - ky * KW;
      const f32x4* wt = wl + (t * 2 + half) * BM + l32;
      const f32x4* pt = pl + ky * PW + kx;
#pragma unroll
      for (int i = 0; i < WM; ++i) aa[i] = wt[i * 32];
#pragma unroll
      for (int j = 0; j < WN; ++j) bb[j] = pt[boff[j]];
    };
    load(0, a[0], b[0]);
    for (int t = 0; t < T; t += 2) {
      load(t + 1 < T ? t + 1 : t, a[1], b[1]);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][i][s], b[0][j][s], acc[i][j], 0, 0, 0);
      if (t + 1 < T) {
        load(t + 2 < T ? t + 2 : t + 1, a[0], b[0]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1][i][s], b[1][j][s], acc[i][j], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < WM; ++i) for (int j = 0; j < WN; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + tid] = s;
}

static int g_rnd = 0;
// b128 weights [t][h][m][4], plain patch [c][PH][PW] read as 4 x ds_read_b32 (channel 2s+h at
// k-step s); one-iteration software pipeline
template <int WM, int WN>
__global__ __launch_bounds__(256, 2) void probe_mix(float* out, int nchunk, int T, int KW, int PW, int PHW, int rnd) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int KC = 8, BM = WM * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, half = lane >> 5;
  for (int i = tid; i < 12000; i += 256) { unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15; lds[i] = rnd ? ((float)(h & 0xFFFFFF) / 8388608.0f - 1.0f) : (float)(i & 15) * 0.01f; }
  __syncthreads();
  const f32x4* wl = (const f32x4*)lds;                       // [t][h][BM]
  const float* pl = lds + KC * T * BM;                          // [c][PHW]
  int boff[WN];
  for (int j = 0; j < WN; ++j) boff[j] = ((wave * WN + j)) * PW + l32 + half * PHW;
  f32x16 acc[WM][WN];
  for (int i = 0; i < WM; ++i) for (int j = 0; j < WN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int c = 0; c < nchunk; ++c) {
    __syncthreads();
    f32x4 a[2][WM], b[2][WN];
    auto load = [&](int t, f32x4 (&aa)[WM], f32x4 (&bb)[WN]) {
      const int ky = t / KW, kx = t - ky * KW;
      const f32x4* wt = wl + (t * 2 + half) * BM + l32;
      const float* pt = pl + ky * PW + kx;
#pragma unroll
      for (int i = 0; i < WM; ++i) aa[i] = wt[i * 32];
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int s = 0; s < 4; ++s) bb[j][s] = pt[boff[j] + s * 2 * PHW];
    };
    load(0, a[0], b[0]);
    for (int t = 0; t < T; t += 2) {
      load(t + 1 < T ? t + 1 : t, a[1], b[1]);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][i][s], b[0][j][s], acc[i][j], 0, 0, 0);
      if (t + 1 < T) {
        load(t + 2 < T ? t + 2 : t + 1, a[0], b[0]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1][i][s], b[1][j][s], acc[i][j], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < WM; ++i) for (int j = 0; j < WN; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <typename K>
void run(const char* name, K kern, int wm, int wn, int T, int KW, int blocks_per_cu) {
  const int nblk = 256 * blocks_per_cu * 4, nchunk = 32, PW = 34, PHW = 34 * 6;
  float* out; hipMalloc(&out, (size_t)nblk * 256 * 4);
  const size_t ldsb = 60000;   // 2 blocks per CU, like the 228-VGPR conv kernel
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), ldsb, 0, out, nchunk, T, KW, PW, PHW, g_rnd);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double fl = (double)nblk * 4 * nchunk * T * 4 * wm * wn * 4096.0;
    if (rep == 2) printf("%-28s T=%d  %8.3f ms  %7.1f TF/s\n", name, T, ms, fl / ms / 1e9);
  }
  hipFree(out);
}

int main(int argc, char** argv) {
  g_rnd = argc > 1 ? atoi(argv[1]) : 0;
  printf("rnd=%d\n", g_rnd);
  run("old <4,1>", probe_old<4, 1>, 4, 1, 9, 3, 2);
  run("new <4,1>", probe_new<4, 1>, 4, 1, 9, 3, 2);
  run("mix <2,2>", probe_mix<2, 2>, 2, 2, 9, 3, 2);
  run("mix <2,1>", probe_mix<2, 1>, 2, 1, 9, 3, 2);
  run("mix <1,1>", probe_mix<1, 1>, 1, 1, 9, 3, 2);
  run("old <2,2>", probe_old<2, 2>, 2, 2, 9, 3, 2);
  run("new <2,2>", probe_new<2, 2>, 2, 2, 9, 3, 2);
  run("old <2,1>", probe_old<2, 1>, 2, 1, 9, 3, 2);
  run("new <2,1>", probe_new<2, 1>, 2, 1, 9, 3, 2);
  run("old <1,1>", probe_old<1, 1>, 1, 1, 9, 3, 2);
  run("new <1,1>", probe_new<1, 1>, 1, 1, 9, 3, 2);
  run("old <4,1> T5", probe_old<4, 1>, 4, 1, 5, 5, 2);
  run("new <4,1> T5", probe_new<4, 1>, 4, 1, 5, 5, 2);
  return 0;
}
