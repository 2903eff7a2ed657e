// Lab: sustained fp32 MFMA rate of the chip with nothing else going on (no memory traffic):
// the practical ceiling under the power/clock management for v_mfma_f32_32x32x2_f32 streams.
//   hipcc --offload-arch=gfx950 -O3 tools/lab/mfma_peak.hip -o tools/lab/bin/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.f) out[threadIdx.x] = s;
}
template <int NACC>
void run(int blocks_per_cu, int ms_target) {
  int dev; hipGetDevice(&dev); hipDeviceProp_t pr; hipGetDeviceProperties(&pr, dev);
  const int grid = pr.multiProcessorCount * blocks_per_cu;
  float* out; hipMalloc(&out, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int iters = 20000;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    mfma_loop<NACC><<<grid, 256>>>(out, iters, 1.0f, 2.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)grid * 4 * iters * 8 * NACC * 2.0 * 32 * 32 * 2;
    printf("NACC=%d blocks/CU=%d iters=%d: %.2f ms  %.1f TFLOP/s\n", NACC, blocks_per_cu, iters, ms, fl / ms * 1e-9);
    if (ms < ms_target) iters *= 4;
  }
  hipFree(out);
}
int main() {
  run<4>(1, 50);
  run<4>(2, 50);
  run<2>(2, 50);
  run<1>(2, 50);
  return 0;
}
