// Stand-alone probe (not part of the product): issue cost of LDS-DMA (global_load_lds) per wave
// instruction, for the variants used / considered by conv_dma.hip.  8 waves per CU (2 blocks of
// 256 threads), every wave issues REPS x 16 DMA instructions from an L2-resident buffer and
// reports shader cycles per instruction (s_memtime, includes the vmcnt(0) drain per 16).
// build: hipcc --offload-arch=gfx950 -O3 dma_issue_probe.hip -o bin/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(const float* src, unsigned long long* out, int reps) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned ldsbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(lds + wave * 64 * 4 * 16);
  // per-lane byte offsets: MODE 0/1/3: 4 channel planes x 16 px (like the interleaved patch);
  // MODE 2/4: contiguous
  unsigned off[16];
  for (int u = 0; u < 16; ++u) {
    const int e = tid + u * 256;
    off[u] = (MODE == 2 || MODE == 4) ? (unsigned)e * (MODE == 4 ? 16u : 4u)
                                     : (unsigned)(((e & 3) * 16384 + (e >> 2)) * 4);
  }
  const unsigned long long mask = (MODE == 1) ? 0x0FFFFFFFFFFFFFF0ull : ~0ull;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (MODE == 3) {   // compiler builtin, no exec games
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)src + off[u]),
                                         (__attribute__((address_space(3))) void*)(lds + wave * 1024 + u * 64), 4, 0, 0);
      } else if (MODE == 4) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0"
                     : : "s"(src), "v"(off[u]), "s"(ldsbase + u * 1024) : "memory");
      } else {
        asm volatile("s_mov_b64 exec, %3\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %0\n\ts_mov_b64 exec, -1"
                     : : "s"(src), "v"(off[u]), "s"(ldsbase + u * 256), "s"(mask) : "memory");
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) atomicAdd(out, t1 - t0);
}

static int g_threads = 256;
template <int MODE>
void run(const char* name, const float* src, unsigned long long* out) {
  const int reps = 64, nblk = 512;
  hipMemset(out, 0, 8);
  hipLaunchKernelGGL(probe<MODE>, dim3(nblk), dim3(g_threads), 70000, 0, src, out, reps);
  hipDeviceSynchronize();
  unsigned long long h = 0;
  hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
  printf("%-44s %7.1f cycles per wave-instruction\n", name, (double)h / (nblk * (g_threads / 64.0)) / (reps * 16.0));
}

int main(int argc, char** argv) {
  g_threads = argc > 1 ? atoi(argv[1]) : 256;   // 64: one issuing wave per block (2 per CU)
  printf("threads per block %d\n", g_threads);
  float* src; unsigned long long* out;
  hipMalloc(&src, 64 << 20); hipMemset(src, 0, 64 << 20); hipMalloc(&out, 8);
  hipFuncSetAttribute((const void*)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 70000);
  hipFuncSetAttribute((const void*)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 70000);
  hipFuncSetAttribute((const void*)probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 70000);
  hipFuncSetAttribute((const void*)probe<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 70000);
  hipFuncSetAttribute((const void*)probe<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 70000);
  for (int k = 0; k < 2; ++k) {
    run<0>("b32, 4-plane gather, exec=all (asm)", src, out);
    run<1>("b32, 4-plane gather, 56 of 64 lanes (asm)", src, out);
    run<2>("b32, contiguous, exec=all (asm)", src, out);
    run<3>("b32, 4-plane gather, compiler builtin", src, out);
    run<4>("b128, contiguous (asm)", src, out);
  }
  return 0;
}
