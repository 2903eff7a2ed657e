// Per-chunk timeline instrumentation of conv_dma_kernel for tools/lab/conv_trace*.py (never part of the
// library build): s_memrealtime stamps of the first blocks, shader clock, HW_ID / XCC_ID.
// scflow_amd/csrc/conv_dma.hip includes this file only when compiled with -DSCF_CONV_LAB
// (tools/lab/build_exp.sh "-DSCF_CONV_LAB").
#pragma once
// -DSCF_CONV_LAB_MASK=m: compile-time phase ablation instead of the timeline (tools/lab/build_conv_masks.sh,
// conv_phases.py): bit 0 no copies in the chunk loop, bit 1 no MFMAs, bit 2 no epilogue (results are wrong)
#ifdef SCF_CONV_LAB_MASK
#define CLAB(bit) ((SCF_CONV_LAB_MASK >> (bit)) & 1)
#define CTRACE(slot) do { } while (0)
#else
#define CLAB(bit) 0
__device__ unsigned long long* scf_conv_trace_ptr = nullptr;
__device__ int scf_conv_trace_nblk = 0;
extern "C" int scf_conv_trace_set(unsigned long long* p, int nblk) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(scf_conv_trace_nblk), &nblk, sizeof(nblk)) != hipSuccess) return -3;
  return hipMemcpyToSymbol(HIP_SYMBOL(scf_conv_trace_ptr), &p, sizeof(p)) == hipSuccess ? 0 : -3;
}
// [block][wave][128]: slot 0 entry, 1 setup, 2 prologue, 3 end, 4 + 4c.. chunk c (wait, barrier, stage, mfma),
// slot 127 = HW_ID | XCC_ID << 32
#define CTRACE(slot)                                                                              \
  do {                                                                                            \
    if (scf_conv_trace_ptr && (int)blockIdx.x < scf_conv_trace_nblk && (threadIdx.x & 63) == 0 && (slot) < 125) { \
      unsigned long long* tp_ = scf_conv_trace_ptr + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 128; \
      tp_[slot] = __builtin_amdgcn_s_memrealtime();                                               \
      if ((slot) == 0 || (slot) == 3) tp_[(slot) == 0 ? 125 : 126] = __builtin_readcyclecounter();  /* shader clock */ \
      if ((slot) == 0) tp_[127] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) | \
                                  ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32); \
    }                                                                                             \
  } while (0)
#endif
