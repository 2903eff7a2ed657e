// Lab: what does a non-MFMA instruction cost the matrix pipe of its SIMD?
//   hipcc --offload-arch=gfx950 -O3 tools/lab/coissue.hip -o tools/lab/bin/coissue
// The convolution kernels' phase ablations (conv_wino.hip) say "MFMA time + everything else", whatever the
// occupancy.  This measures the mechanism in isolation, with the shader clock (s_memtime) around the MFMA stream:
//   A  PARTNER mode: 8 waves per CU = 2 per SIMD; waves 0-3 stream N dependent-free MFMAs, waves 4-7 stream
//      instructions of ONE kind (v_fma, v_pk_fma, ds_read_b64, conflicting ds_read2_b32, LDS-DMA, s_nop, MFMA)
//      for as long as the MFMA waves run.  Reported: cycles per MFMA of the MFMA waves (64 = the pipe is theirs).
//   B  SAME-WAVE mode: one wave per SIMD runs [MFMA, k x X] groups: cycles per group - 64 = what k instructions
//      of kind X add to an MFMA when they come from the same wave.
//   C  TWO-WAVE INTERLEAVE: both waves of a SIMD run [MFMA, k x X] (the convolution kernels' situation):
//      cycles per (MFMA of either wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int rsrc4 __attribute__((ext_vector_type(4)));

enum { X_NONE = 0, X_FMA, X_PKFMA, X_DSREAD64, X_DSREAD2_CONFLICT, X_DMA, X_SNOP, X_MFMA, X_DSREAD128, X_COUNT };
static const char* xname[] = {"nothing", "v_fma_f32", "v_pk_fma_f32", "ds_read_b64", "ds_read2_b32 4-way conflict",
                              "buffer_load_dwordx4 lds", "s_nop 0", "v_mfma (partner)", "ds_read_b128"};

__device__ __forceinline__ unsigned long long clk() { return __builtin_amdgcn_s_memtime(); }

template <int X>
__device__ __forceinline__ void do_x(float& v0, float& v1, f32x2& p0, f32x2& p1, const float* lds, unsigned ldsa,
                                     rsrc4 rs, unsigned voff, f32x16& acc2, float a, float b) {
  if constexpr (X == X_FMA) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v0) : "v"(v1), "v"(a)); }
  else if constexpr (X == X_PKFMA) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p0) : "v"(p1), "v"(p1)); }
  else if constexpr (X == X_DSREAD64) { f32x2 t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"(ldsa)); asm volatile("" :: "v"(t)); }
  else if constexpr (X == X_DSREAD128) { float __attribute__((ext_vector_type(4))) t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(ldsa & ~15u)); asm volatile("" :: "v"(t)); }
  else if constexpr (X == X_DSREAD2_CONFLICT) { f32x2 t; asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1" : "=v"(t) : "v"(ldsa)); asm volatile("" :: "v"(t)); }
  else if constexpr (X == X_DMA) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane(ldsa) & 0xF000u;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %0, 0 offen lds" :: "s"(rs), "v"(voff), "s"(m0v) : "memory");
  }
  else if constexpr (X == X_SNOP) { asm volatile("s_nop 0"); }
  else if constexpr (X == X_MFMA) { acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc2, 0, 0, 0); }
}

// mode 0: partner (waves 4-7 run X only, waves 0-3 MFMA only); mode 1: same wave [MFMA, k x X], 4 waves;
// mode 2: all 8 waves run [MFMA, k x X]
template <int X, int K, int G = 1>
__global__ __launch_bounds__(512, 1) void co_kernel(unsigned long long* out, int iters, int mode, const float* gbuf, float a, float b) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (mode == 1 && wave >= 4) return;
  for (int i = tid; i < 8192; i += 512) lds[i] = (float)i;
  if (tid < 4) reinterpret_cast<volatile int*>(lds + 12000)[tid] = 0;
  __syncthreads();
  f32x16 acc[4], acc2;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
  float v0 = a, v1 = b;
  f32x2 p0 = {a, b}, p1 = {b, a};
  // conflict pattern: 64 lanes, dword index = 2 * (lane & 15) + 32 * (lane >> 4) + 3 -> 16 odd banks, 4 lanes each
  const unsigned ldsa = X == X_DSREAD2_CONFLICT ? (unsigned)((2 * (lane & 15) + 32 * (lane >> 4) + 3) * 4) + (unsigned)(wave * 2048)
                                                : (unsigned)(lane * 16 + wave * 2048);
  rsrc4 rs;
  {
    const unsigned long long ga = (unsigned long long)gbuf;
    rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)ga);
    rs[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(ga >> 32) & 0xFFFFu));
    rs[2] = 1 << 20; rs[3] = 0x00020000;
  }
  const unsigned voff = (unsigned)(lane * 16 + (wave & 3) * 1024 + (blockIdx.x & 63) * 4096);
  const bool mf = mode != 0 || wave < 4;
  const unsigned long long t0 = clk();
  volatile int* flags = reinterpret_cast<volatile int*>(lds + 12000);      // [4]: MFMA wave w is done
  int xcount = 0;
  if (mode == 0 && !mf) {
    // the partner streams X for as long as the MFMA wave of its SIMD runs (wave w and w + 4 share a SIMD)
    do {
#pragma unroll
      for (int u = 0; u < 16; ++u) do_x<X>(v0, v1, p0, p1, lds, ldsa, rs, voff, acc2, a, b);
      if (X == X_DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ++xcount;
    } while (flags[wave - 4] == 0 && xcount < 1000000);
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; u += G) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          acc[u + g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u + g], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (mode != 0) {
#pragma unroll
          for (int k = 0; k < K * G; ++k) do_x<X>(v0, v1, p0, p1, lds, ldsa, rs, voff, acc2, a, b);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (X == X_DMA && mode != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  const unsigned long long t1 = clk();
  if (mode == 0 && mf && lane == 0) flags[wave] = 1;
  float s = v0 + p0[0] + p0[1];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int r = 0; r < 16; ++r) s += acc2[r];
  if (s == 12345.f) out[4096 + tid] = 1;
  if (lane == 0) out[blockIdx.x * 8 + wave] = (mode == 0 && !mf) ? (unsigned long long)xcount : t1 - t0;
}

template <int X, int K, int G = 1>
void run(int mode, const char* what) {
  int dev; hipGetDevice(&dev); hipDeviceProp_t pr; hipGetDeviceProperties(&pr, dev);
  const int grid = pr.multiProcessorCount;
  unsigned long long* out; hipMalloc(&out, (grid * 8 + 8192) * 8);
  float* gbuf; hipMalloc(&gbuf, 1 << 20); hipMemset(gbuf, 0, 1 << 20);
  const int iters = 2000;
  hipFuncSetAttribute((const void*)co_kernel<X, K, G>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  std::vector<unsigned long long> h(grid * 8);
  double best = 1e30, bestx = 0, besto = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(out, 0, grid * 8 * 8);
    co_kernel<X, K, G><<<grid, 512, 48 * 1024>>>(out, iters, mode, gbuf, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), out, grid * 8 * 8, hipMemcpyDeviceToHost);
    // median over blocks: the slower and the faster of the two waves of SIMD 0 (waves 0 and 4), and the partner's count
    std::vector<double> m, x, o;
    for (int b = 0; b < grid; ++b) {
      const double w0 = (double)h[b * 8], w4 = (double)h[b * 8 + 4];
      m.push_back(mode == 2 ? std::max(w0, w4) : w0); o.push_back(std::min(w0, w4)); x.push_back(w4);
    }
    std::sort(m.begin(), m.end()); std::sort(x.begin(), x.end()); std::sort(o.begin(), o.end());
    if (m[grid / 2] < best) { best = m[grid / 2]; bestx = x[grid / 2]; besto = o[grid / 2]; }
  }
  const double n = iters * 4.0;
  if (mode == 0)
    printf("partner    %-28s : %6.1f clk per MFMA of the MFMA wave; the partner issued %6.2f X per MFMA\n", xname[X], best / n, bestx * 16.0 / n);
  else if (mode == 1)
    printf("one wave   [%d MFMA + %2d x %-28s] : %6.1f clk per MFMA  (%+5.1f; %4.1f per X)\n", G, K * G, xname[X], best / n, best / n - 64.0,
           K ? (best / n - 64.0) / K : 0.0);
  else
    printf("two waves  [%d MFMA + %2d x %-28s] : %6.1f clk per MFMA of the SIMD (both waves done; the first after %5.1f%% of that)  (%+5.1f; %4.1f per X)\n",
           G, K * G, xname[X], best / (2 * n), 100.0 * besto / best, best / (2 * n) - 64.0, K ? (best / (2 * n) - 64.0) / K : 0.0);
  hipFree(out); hipFree(gbuf);
}


// r5: BURST structure and wave priorities.  Every wave runs [NB MFMAs back to back][KB x X back to back]; two waves per
// SIMD.  PRIO 0: no s_setprio; 1: priority 3 during the X burst, 0 during the MFMA burst (the wave doing "everything else"
// goes first, the MFMA stream fills what is left); 2: the reverse.  If a high-priority partner can issue under an MFMA
// stream, two waves in anti-phase keep the pipe full with ANY amount of other work up to the MFMA burst's length.
// Partner mode with priorities (mode 0): the X-only wave runs at priority PRIO_X, the MFMA wave at PRIO_M.
template <int X, int NB, int KB, int PRIO>
__global__ __launch_bounds__(512, 1) void burst_kernel(unsigned long long* out, int iters, const float* gbuf, float a, float b, int stagger) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 8192; i += 512) lds[i] = (float)i;
  __syncthreads();
  f32x16 acc[4], acc2;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
  float v0 = a, v1 = b;
  f32x2 p0 = {a, b}, p1 = {b, a};
  const unsigned ldsa = (unsigned)(lane * 16 + wave * 2048);
  rsrc4 rs;
  {
    const unsigned long long ga = (unsigned long long)gbuf;
    rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)ga);
    rs[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(ga >> 32) & 0xFFFFu));
    rs[2] = 1 << 20; rs[3] = 0x00020000;
  }
  const unsigned voff = (unsigned)(lane * 16 + (wave & 3) * 1024 + (blockIdx.x & 63) * 4096);
  if (stagger && wave >= 4) {           // the second wave of every SIMD starts half a period later
    for (int u = 0; u < NB / 2; ++u) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc2, 0, 0, 0);
  }
  const unsigned long long t0 = clk();
  for (int it = 0; it < iters; ++it) {
    if (PRIO == 1) __builtin_amdgcn_s_setprio(3);
    if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int k = 0; k < KB; ++k) do_x<X>(v0, v1, p0, p1, lds, ldsa, rs, voff, acc2, a, b);
    __builtin_amdgcn_sched_barrier(0);
    if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
    if (PRIO == 2) __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = clk();
  float s = v0 + p0[0] + p0[1];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int r = 0; r < 16; ++r) s += acc2[r];
  if (s == 12345.f) out[4096 + tid] = 1;
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int X, int NB, int KB, int PRIO>
void run_burst(int stagger) {
  int dev; hipGetDevice(&dev); hipDeviceProp_t pr; hipGetDeviceProperties(&pr, dev);
  const int grid = pr.multiProcessorCount;
  unsigned long long* out; hipMalloc(&out, (grid * 8 + 8192) * 8);
  float* gbuf; hipMalloc(&gbuf, 1 << 20); hipMemset(gbuf, 0, 1 << 20);
  const int iters = 500;
  hipFuncSetAttribute((const void*)burst_kernel<X, NB, KB, PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  std::vector<unsigned long long> h(grid * 8);
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(out, 0, grid * 8 * 8);
    burst_kernel<X, NB, KB, PRIO><<<grid, 512, 48 * 1024>>>(out, iters, gbuf, 1.0f, 0.5f, stagger);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), out, grid * 8 * 8, hipMemcpyDeviceToHost);
    std::vector<double> m;
    for (int b = 0; b < grid; ++b) m.push_back(std::max((double)h[b * 8], (double)h[b * 8 + 4]));
    std::sort(m.begin(), m.end());
    best = std::min(best, m[grid / 2]);
  }
  const double n = iters * (double)NB;
  printf("burst      [%2d x %-24s | %2d MFMA] prio %d%s : %6.1f clk per MFMA of the SIMD (two waves)  (%+5.1f; %4.2f per X)\n", KB, xname[X], NB, PRIO,
         stagger ? " staggered" : "          ", best / (2 * n), best / (2 * n) - 64.0, KB ? (best / (2 * n) - 64.0) * NB / KB : 0.0);
  hipFree(out); hipFree(gbuf);
}

// partner mode with priorities: waves 0-3 stream MFMAs at priority PM, waves 4-7 stream X at priority PX
template <int X, int PM, int PX>
__global__ __launch_bounds__(512, 1) void partner_prio_kernel(unsigned long long* out, int iters, const float* gbuf, float a, float b) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 8192; i += 512) lds[i] = (float)i;
  if (tid < 4) reinterpret_cast<volatile int*>(lds + 12000)[tid] = 0;
  __syncthreads();
  f32x16 acc[4], acc2;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
  float v0 = a, v1 = b;
  f32x2 p0 = {a, b}, p1 = {b, a};
  const unsigned ldsa = (unsigned)(lane * 16 + wave * 2048);
  rsrc4 rs;
  {
    const unsigned long long ga = (unsigned long long)gbuf;
    rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)ga);
    rs[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(ga >> 32) & 0xFFFFu));
    rs[2] = 1 << 20; rs[3] = 0x00020000;
  }
  const unsigned voff = (unsigned)(lane * 16 + (wave & 3) * 1024 + (blockIdx.x & 63) * 4096);
  volatile int* flags = reinterpret_cast<volatile int*>(lds + 12000);
  const bool mf = wave < 4;
  if (mf) __builtin_amdgcn_s_setprio(PM); else __builtin_amdgcn_s_setprio(PX);
  const unsigned long long t0 = clk();
  int xcount = 0;
  if (!mf) {
    do {
#pragma unroll
      for (int u = 0; u < 16; ++u) do_x<X>(v0, v1, p0, p1, lds, ldsa, rs, voff, acc2, a, b);
      if (X == X_DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ++xcount;
    } while (flags[wave - 4] == 0 && xcount < 1000000);
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const unsigned long long t1 = clk();
  if (mf && lane == 0) flags[wave] = 1;
  float s = v0 + p0[0] + p0[1];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int r = 0; r < 16; ++r) s += acc2[r];
  if (s == 12345.f) out[4096 + tid] = 1;
  if (lane == 0) out[blockIdx.x * 8 + wave] = mf ? t1 - t0 : (unsigned long long)xcount;
}
template <int X, int PM, int PX>
void run_partner_prio() {
  int dev; hipGetDevice(&dev); hipDeviceProp_t pr; hipGetDeviceProperties(&pr, dev);
  const int grid = pr.multiProcessorCount;
  unsigned long long* out; hipMalloc(&out, (grid * 8 + 8192) * 8);
  float* gbuf; hipMalloc(&gbuf, 1 << 20); hipMemset(gbuf, 0, 1 << 20);
  const int iters = 2000;
  hipFuncSetAttribute((const void*)partner_prio_kernel<X, PM, PX>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  std::vector<unsigned long long> h(grid * 8);
  hipMemset(out, 0, grid * 8 * 8);
  partner_prio_kernel<X, PM, PX><<<grid, 512, 48 * 1024>>>(out, iters, gbuf, 1.0f, 0.5f);
  hipDeviceSynchronize();
  hipMemcpy(h.data(), out, grid * 8 * 8, hipMemcpyDeviceToHost);
  std::vector<double> m, x;
  for (int b = 0; b < grid; ++b) { m.push_back((double)h[b * 8]); x.push_back((double)h[b * 8 + 4]); }
  std::sort(m.begin(), m.end()); std::sort(x.begin(), x.end());
  const double n = iters * 4.0;
  printf("partner    %-28s MFMA wave prio %d, X wave prio %d : %6.1f clk per MFMA; the partner issued %6.2f X per MFMA\n", xname[X], PM, PX,
         m[grid / 2] / n, x[grid / 2] * 16.0 / n);
  hipFree(out); hipFree(gbuf);
}

void run_r5() {
  run_partner_prio<X_FMA, 0, 3>(); run_partner_prio<X_FMA, 3, 0>(); run_partner_prio<X_FMA, 0, 0>();
  run_partner_prio<X_DSREAD64, 0, 3>(); run_partner_prio<X_DMA, 0, 3>(); run_partner_prio<X_SNOP, 0, 3>();
  // the Winograd kernels' chunk: 16 MFMAs + ~24 ... 60 vector instructions + ~16 LDS reads per wave
  run_burst<X_FMA, 16, 24, 0>(0); run_burst<X_FMA, 16, 24, 1>(0); run_burst<X_FMA, 16, 24, 2>(0);
  run_burst<X_FMA, 16, 24, 0>(1); run_burst<X_FMA, 16, 24, 1>(1);
  run_burst<X_FMA, 16, 56, 0>(0); run_burst<X_FMA, 16, 56, 1>(0); run_burst<X_FMA, 16, 56, 0>(1); run_burst<X_FMA, 16, 56, 1>(1);
  run_burst<X_DSREAD64, 16, 16, 0>(0); run_burst<X_DSREAD64, 16, 16, 1>(0); run_burst<X_DSREAD64, 16, 16, 1>(1);
  run_burst<X_DMA, 16, 5, 0>(0); run_burst<X_DMA, 16, 5, 1>(0); run_burst<X_DMA, 16, 5, 1>(1);
  run_burst<X_FMA, 8, 12, 0>(0); run_burst<X_FMA, 8, 12, 1>(0); run_burst<X_FMA, 4, 6, 0>(0); run_burst<X_FMA, 4, 6, 1>(0);
  run_burst<X_FMA, 32, 48, 0>(0); run_burst<X_FMA, 32, 48, 1>(0); run_burst<X_FMA, 32, 112, 1>(0); run_burst<X_FMA, 32, 112, 1>(1);
}

int main(int argc, char** argv) {
  if (argc > 1 && argv[1][0] == '5') { run_r5(); return 0; }

  run<X_NONE, 0>(1, ""); run<X_NONE, 0>(2, "");
  run<X_FMA, 0>(0, ""); run<X_DSREAD64, 0>(0, ""); run<X_SNOP, 0>(0, ""); run<X_MFMA, 0>(0, "");
  // one wave per SIMD
  run<X_FMA, 1>(1, ""); run<X_FMA, 2>(1, ""); run<X_FMA, 4>(1, ""); run<X_FMA, 8>(1, ""); run<X_PKFMA, 1>(1, ""); run<X_PKFMA, 4>(1, "");
  run<X_DSREAD64, 1>(1, ""); run<X_DSREAD64, 2>(1, ""); run<X_DSREAD64, 4>(1, ""); run<X_DSREAD2_CONFLICT, 2>(1, ""); run<X_DSREAD128, 2>(1, "");
  run<X_DMA, 1>(1, ""); run<X_SNOP, 1>(1, ""); run<X_SNOP, 4>(1, "");
  run<X_FMA, 1, 2>(1, ""); run<X_FMA, 4, 2>(1, ""); run<X_FMA, 4, 4>(1, ""); run<X_DSREAD64, 2, 2>(1, ""); run<X_DSREAD64, 2, 4>(1, "");
  // two waves per SIMD, aggregate
  run<X_FMA, 1>(2, ""); run<X_FMA, 2>(2, ""); run<X_FMA, 4>(2, ""); run<X_FMA, 8>(2, ""); run<X_PKFMA, 1>(2, ""); run<X_PKFMA, 4>(2, "");
  run<X_DSREAD64, 1>(2, ""); run<X_DSREAD64, 2>(2, ""); run<X_DSREAD64, 4>(2, ""); run<X_DSREAD2_CONFLICT, 2>(2, ""); run<X_DSREAD128, 2>(2, "");
  run<X_DMA, 1>(2, ""); run<X_SNOP, 1>(2, ""); run<X_SNOP, 4>(2, "");
  run<X_FMA, 1, 2>(2, ""); run<X_FMA, 4, 2>(2, ""); run<X_FMA, 4, 4>(2, ""); run<X_DSREAD64, 2, 2>(2, ""); run<X_DSREAD64, 2, 4>(2, ""); run<X_PKFMA, 2, 2>(2, "");
  return 0;
}
