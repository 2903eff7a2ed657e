// LDS-DMA helpers shared by the gfx950 kernels that stage operands memory -> LDS without passing
// through VGPRs (`buffer_load_dword[x4] voff, rsrc, 0 offen lds`: 4 / 16 bytes per lane from
// rsrc.base + voff to LDS byte M0 + lane * size).  The raw buffer descriptor's range check does the
// masking: a lane whose offset is >= num_records writes ZEROS to its LDS cell (checked on gfx950:
// tools/lab/buf_lds_test.hip) -- zero padding, out-of-image positions and short last chunks need no
// EXEC mask and no pre-zeroed LDS.  The compiler does not count these loads: callers wait (vmcnt)
// themselves.
#pragma once
#include "scf_common.h"

typedef int scf_rsrc4 __attribute__((ext_vector_type(4)));
#define SCF_BUF_OOB 0x80000000u        // an offset past every descriptor range: the lane's cell is zeroed

__device__ __forceinline__ unsigned scf_lds_addr(const void* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

// base and bytes must be wave-uniform
__device__ __forceinline__ scf_rsrc4 scf_make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  scf_rsrc4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xFFFFu));   // stride 0: raw buffer
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);                             // num_records (bytes)
  r[3] = 0x00020000;
  return r;
}
__device__ __forceinline__ void scf_bdma_b128(scf_rsrc4 rsrc, unsigned voff, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %0, 0 offen lds"
               : : "s"(rsrc), "v"(voff), "s"(lds_base) : "memory");
}
__device__ __forceinline__ void scf_bdma_b32(scf_rsrc4 rsrc, unsigned voff, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %1, %0, 0 offen lds"
               : : "s"(rsrc), "v"(voff), "s"(lds_base) : "memory");
}

// s_waitcnt vmcnt(n) for a wave-uniform RUN-TIME n (the instruction takes an immediate)
template <int V>
__device__ __forceinline__ void scf_wait_vmcnt_imm() {
  __builtin_amdgcn_s_waitcnt(0x0F70 | (V & 15) | ((V >> 4) << 14));
}
__device__ __forceinline__ void scf_wait_vmcnt_le(int n) {
#define SCF_W4(b) case b: scf_wait_vmcnt_imm<b>(); break; case b + 1: scf_wait_vmcnt_imm<b + 1>(); break; \
                  case b + 2: scf_wait_vmcnt_imm<b + 2>(); break; case b + 3: scf_wait_vmcnt_imm<b + 3>(); break;
  switch (n) {
    SCF_W4(0) SCF_W4(4) SCF_W4(8) SCF_W4(12) SCF_W4(16) SCF_W4(20) SCF_W4(24) SCF_W4(28)
    SCF_W4(32) SCF_W4(36) SCF_W4(40) SCF_W4(44) SCF_W4(48) SCF_W4(52) SCF_W4(56) SCF_W4(60)
    default: scf_wait_vmcnt_imm<0>(); break;
  }
#undef SCF_W4
}
