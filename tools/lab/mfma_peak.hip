// Lab: sustained fp32 MFMA rate of the chip with nothing else going on (no memory traffic):
// the practical ceiling under the power/clock management for v_mfma_f32_32x32x2_f32 streams.
//   hipcc --offload-arch=gfx950 -O3 tools/lab/mfma_peak.hip -o tools/lab/bin/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a, float b) {
  if (a == 0.f) {                      // "random data" mode: per-lane operands that change every k-step
    unsigned h = (threadIdx.x + 1) * 2654435761u + blockIdx.x * 40503u;
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    a = (float)(int)(h & 0xFFFF) * (1.0f / 65536.0f) - 0.5f;
    b = (float)(int)(h >> 16) * (1.0f / 65536.0f) - 0.5f;
  }
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    a = -a * 1.0001f; b = b * 0.9999f + 1e-3f;     // operands keep changing (toggle activity)
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.f) out[threadIdx.x] = s;
}
template <int NACC>
void run(int blocks_per_cu, int ms_target, float a0 = 1.0f) {
  int dev; hipGetDevice(&dev); hipDeviceProp_t pr; hipGetDeviceProperties(&pr, dev);
  const int grid = pr.multiProcessorCount * blocks_per_cu;
  float* out; hipMalloc(&out, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int iters = 20000;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    mfma_loop<NACC><<<grid, 256>>>(out, iters, a0, 2.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)grid * 4 * iters * 8 * NACC * 2.0 * 32 * 32 * 2;
    printf("%s NACC=%d blocks/CU=%d iters=%d: %.2f ms  %.1f TFLOP/s\n", a0 == 0.f ? "random-data" : "constant   ", NACC, blocks_per_cu, iters, ms, fl / ms * 1e-9);
    if (ms < ms_target) iters *= 4;
  }
  hipFree(out);
}
int main() {
  run<4>(2, 50);
  run<4>(2, 50, 0.f);
  run<4>(1, 50, 0.f);
  return 0;
}
