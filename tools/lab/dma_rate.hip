// Lab: issue rate / throughput of global_load_lds_dword[x4] (LDS-DMA) per wave and per CU.
//   hipcc --offload-arch=gfx950 -O3 -w tools/lab/dma_rate.hip -o tools/lab/bin/dma_rate
// Each wave issues NI DMA instructions back to back into its own LDS area (source: a small,
// L2-resident buffer), stamps s_memtime after the issue and after vmcnt(0); repeated `iters` times.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int X4, int NI>
__global__ __launch_bounds__(256) void dma_loop(const float* src, unsigned long long* stamps, int iters, int src_mask) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)lds + wave * NI * ((X4 ? 1024 : 256));
  unsigned voff[NI];
  for (int i = 0; i < NI; ++i)
    voff[i] = (((blockIdx.x * 4 + wave) * NI + i) * 64 + lane) * (X4 ? 16 : 4) & src_mask;
  unsigned long long t_issue = 0, t_done = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if (X4 == 3) asm volatile("v_cmpx_ne_u32_e32 -1, %1\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0\n\ts_mov_b64 exec, -1" : : "s"(src), "v"(voff[i]), "s"(lbase + i * 1024) : "memory", "vcc");
      else if (X4 == 2) asm volatile("v_cmp_ne_u32_e32 vcc, -1, %1\n\ts_mov_b64 exec, vcc\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0\n\ts_mov_b64 exec, -1" : : "s"(src), "v"(voff[i]), "s"(lbase + i * 1024) : "memory", "vcc");
      else if (X4) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" : : "s"(src), "v"(voff[i]), "s"(lbase + i * 1024) : "memory");
      else asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %0" : : "s"(src), "v"(voff[i]), "s"(lbase + i * 256) : "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    __builtin_amdgcn_s_waitcnt(0x0F70);
    const unsigned long long t2 = __builtin_readcyclecounter();
    t_issue += t1 - t0; t_done += t2 - t0;
  }
  if (lane == 0) {
    stamps[(blockIdx.x * 4 + wave) * 2] = t_issue;
    stamps[(blockIdx.x * 4 + wave) * 2 + 1] = t_done;
  }
}
template <int X4, int NI>
int run(const float* src, unsigned long long* stamps, int blocks_per_cu, int ncu, int src_mask) {
  const int grid = ncu * blocks_per_cu, iters = 2000;
  const size_t ldsb = 4 * NI * ((X4 ? 1024 : 256));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_loop<X4, NI>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  dma_loop<X4, NI><<<grid, 256, ldsb>>>(src, stamps, 10, src_mask);
  CK(hipEventRecord(e0));
  dma_loop<X4, NI><<<grid, 256, ldsb>>>(src, stamps, iters, src_mask);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(grid * 8);
  CK(hipMemcpy(h.data(), stamps, grid * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double ti = 0, td = 0;
  for (int i = 0; i < grid * 4; ++i) { ti += h[2 * i]; td += h[2 * i + 1]; }
  ti /= (double)grid * 4 * iters * NI; td /= (double)grid * 4 * iters * NI;
  const double bytes = (double)grid * 4 * iters * NI * ((X4 ? 1024 : 256));
  printf("%s NI=%2d blocks/CU=%d (waves/CU=%d): issue %.1f ticks/instr, issue+land %.1f ticks/instr, %.2f us/iter, %.1f GB/s total, %.1f B/ns/CU\n",
         X4 ? "x4   " : "dword", NI, blocks_per_cu, blocks_per_cu * 4, ti, td, ms * 1e3 / iters, bytes / ms * 1e-6, bytes / ms * 1e-6 / ncu);
  return 0;
}
int main() {
  int dev; CK(hipGetDevice(&dev)); hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, dev));
  const int ncu = pr.multiProcessorCount;
  float* src; CK(hipMalloc(&src, 64 << 20)); CK(hipMemset(src, 0, 64 << 20));
  unsigned long long* stamps; CK(hipMalloc(&stamps, 1 << 20));
  printf("readcyclecounter ticks (s_memtime); CUs=%d\n", ncu);
  const int mask = (8 << 20) - 1;      // 8 MiB source window (L2/MALL resident)
  run<1, 16>(src, stamps, 1, ncu, mask);
  run<1, 16>(src, stamps, 2, ncu, mask);
  run<0, 16>(src, stamps, 1, ncu, mask);
  run<0, 16>(src, stamps, 2, ncu, mask);
  printf("-- with the v_cmp / exec-mask sequence of the conv kernel\n");
  run<2, 16>(src, stamps, 1, ncu, mask);
  run<2, 16>(src, stamps, 2, ncu, mask);
  printf("-- v_cmpx variant\n");
  run<3, 16>(src, stamps, 1, ncu, mask);
  run<3, 16>(src, stamps, 2, ncu, mask);
  printf("-- 64 MiB source window (HBM / Infinity Cache)\n");
  run<1, 16>(src, stamps, 1, ncu, (64 << 20) - 1);
  run<1, 16>(src, stamps, 2, ncu, (64 << 20) - 1);
  run<2, 16>(src, stamps, 2, ncu, (64 << 20) - 1);
  printf("-- 1 MiB source window (all blocks read the same lines: XCD L2 hits)\n");
  run<1, 16>(src, stamps, 2, ncu, (1 << 20) - 1);
  run<2, 16>(src, stamps, 2, ncu, (1 << 20) - 1);
  return 0;
}
