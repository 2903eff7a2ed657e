// Instrumentation of the product lookup kernel for tools/lab/lookup_lab.hip (never part of the
// library build): s_memrealtime stamps per wave, two ablation switches (no gathers / no stores), a
// grid override and a store-policy sweep.  scflow_amd/csrc/corr_lookup.hip includes this file only
// when it is compiled with -DSCF_LOOKUP_LAB.
#pragma once

static unsigned long long* scf_lab_trace = nullptr;
static int scf_lab_skip_dma = 0, scf_lab_skip_store = 0, scf_lab_rotate = -1, scf_lab_grid = 0, scf_lab_store_mode = 0;

#define LK_LAB_PARAMS                                                                              \
  unsigned long long* trace; /* [groups][4 waves][8] s_memrealtime stamps */                        \
  int skip_dma, skip_store;

#define LK_TRACE(slot)                                                                             \
  do {                                                                                             \
    if (p.trace && (threadIdx.x & 63) == 0)                                                        \
      p.trace[((size_t)g * 4 + (threadIdx.x >> 6)) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)

// after the emit of one (wave, level): emitted, stores acknowledged, (level, HW_ID)
#define LK_TRACE_END(lvl)                                                                          \
  do {                                                                                             \
    LK_TRACE(5);                                                                                   \
    __builtin_amdgcn_s_waitcnt(0x0F70);                                                            \
    LK_TRACE(6);                                                                                   \
    if (p.trace && lane == 0)                                                                      \
      p.trace[((size_t)g * 4 + wave) * 8 + 7] =                                                    \
          ((unsigned long long)(lvl) << 32) | __builtin_amdgcn_s_getreg(4 | (31 << 11));           \
  } while (0)

#define LK_SKIP_DMA (p.skip_dma != 0)
#define LK_SKIP_STORE (p.skip_store != 0)

// launch-side: hand the switches to the kernel, optional grid override, store policy A/B (r = 4)
#define LK_LAB_LAUNCH(p, nblk)                                                                     \
  do {                                                                                             \
    p.trace = scf_lab_trace; p.skip_dma = scf_lab_skip_dma; p.skip_store = scf_lab_skip_store;     \
    if (scf_lab_grid > 0) nblk = scf_lab_grid;                                                     \
    if (r == 4 && scf_lab_store_mode > 0) {                                                        \
      switch (scf_lab_store_mode) {                                                                \
        case 1: scf_launch((corr_lookup_kernel<4, 1>), dim3((unsigned)nblk), dim3(256), lds, scf_stream(stream), p); break; \
        case 2: scf_launch((corr_lookup_kernel<4, 2>), dim3((unsigned)nblk), dim3(256), lds, scf_stream(stream), p); break; \
        case 3: scf_launch((corr_lookup_kernel<4, 3>), dim3((unsigned)nblk), dim3(256), lds, scf_stream(stream), p); break; \
        default: scf_launch((corr_lookup_kernel<4, 4>), dim3((unsigned)nblk), dim3(256), lds, scf_stream(stream), p); break; \
      }                                                                                            \
      return scf_launch_status();                                                                  \
    }                                                                                              \
  } while (0)
