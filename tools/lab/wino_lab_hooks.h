// Timeline instrumentation of conv_wino_kernel for tools/lab/wino_trace.py (never part of the library
// build; scflow_amd/csrc/conv_wino.hip includes this file only under -DSCF_WINO_LAB).  Stamps are shader-clock
// reads buffered in LDS (a global store would disturb the kernel's own vmcnt accounting) and copied out once
// per block: [block][wave][128] u32; slot 0 entry, 1 prologue done, 2 + 8c + k = chunk c (c < 15): k = 0 DMA
// issued, 1..4 MFMA groups issued, 5 own copies landed, 6 barrier passed; 122 loop done, 123 exchange written,
// 126 = HW_ID, 127 = XCC_ID.
#pragma once
__device__ unsigned* scf_wino_trace_ptr = nullptr;
__device__ int scf_wino_trace_nblk = 0;
extern "C" int scf_wino_trace_set(unsigned* p, int nblk) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(scf_wino_trace_nblk), &nblk, sizeof(nblk)) != hipSuccess) return -3;
  return hipMemcpyToSymbol(HIP_SYMBOL(scf_wino_trace_ptr), &p, sizeof(p)) == hipSuccess ? 0 : -3;
}
#define WN_TRACE_BYTES 2048
#define WN_T(slot)                                                                    \
  do {                                                                                \
    if (q.lab & 32) {                                                                 \
      const unsigned t_ = (unsigned)__builtin_readcyclecounter();                     \
      if ((threadIdx.x & 63) == 0) wn_trace[(threadIdx.x >> 6) * 128 + (slot)] = t_;  \
    }                                                                                 \
  } while (0)
// 100 MHz wall clock next to the shader clock: slots 118 / 119 at entry, 120 / 121 at the dump (low 32 bits each)
#define WN_T_RT(slot)                                                                 \
  do {                                                                                \
    if (q.lab & 32) {                                                                 \
      const unsigned r_ = (unsigned)__builtin_amdgcn_s_memrealtime();                 \
      const unsigned t_ = (unsigned)__builtin_readcyclecounter();                     \
      if ((threadIdx.x & 63) == 0) { wn_trace[(threadIdx.x >> 6) * 128 + (slot)] = r_; wn_trace[(threadIdx.x >> 6) * 128 + (slot) + 1] = t_; } \
    }                                                                                 \
  } while (0)
#define WN_T_CHUNK(c, k) do { if ((c) < 15) WN_T(2 + 8 * (c) + (k)); } while (0)
#define WN_T_DUMP()                                                                                     \
  do {                                                                                                  \
    WN_T_RT(120);                                                                                       \
    if ((q.lab & 32) && scf_wino_trace_ptr && (int)blockIdx.x < scf_wino_trace_nblk) {                  \
      if ((threadIdx.x & 63) == 0) {                                                                    \
        wn_trace[(threadIdx.x >> 6) * 128 + 126] = __builtin_amdgcn_s_getreg(4 | (31 << 11));           \
        wn_trace[(threadIdx.x >> 6) * 128 + 127] = __builtin_amdgcn_s_getreg(20 | (31 << 11));          \
      }                                                                                                 \
      __syncthreads();                                                                                  \
      for (int i_ = threadIdx.x; i_ < 512; i_ += 256) scf_wino_trace_ptr[(size_t)blockIdx.x * 512 + i_] = wn_trace[i_]; \
    }                                                                                                   \
  } while (0)
