// Instrumentation of the product lookup kernel for tools/lab/lookup_lab.hip (never part of the
// library build): s_memrealtime stamps per wave, two ablation switches (no gathers / no stores), a
// grid override and a store-policy sweep.  scflow_amd/csrc/corr_lookup.hip includes this file only
// when it is compiled with -DSCF_LOOKUP_LAB.
#pragma once

static unsigned long long* scf_lab_trace = nullptr;
static int scf_lab_skip_dma = 0, scf_lab_skip_store = 0, scf_lab_rotate = -1, scf_lab_grid = 0, scf_lab_store_mode = 0;
static int scf_lab_stagger = 0, scf_lab_early = 1;   // odd block slots sleep stagger * 3.9 us / 8; whole-map DMAs before the flow wait
static int scf_lab_pipe = -1;     // -1: the library's own choice; else the SCF_TUNE_LOOKUP_PIPE value

#define LK_LAB_PARAMS                                                                              \
  unsigned long long* trace; /* [groups][4 waves][8] s_memrealtime stamps */                        \
  int skip_dma, skip_store, stagger, early;

#define LK_TRACE(slot)                                                                             \
  do {                                                                                             \
    if (p.trace && (threadIdx.x & 63) == 0)                                                        \
      p.trace[((size_t)g * 4 + ((threadIdx.x >> 6) & 3)) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
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

// pipelined kernel: stamps are indexed by (group, level) so that the analysis is the same for both kernels
#define LK_TRACE_U(g, lvl, slot)                                                                   \
  do {                                                                                             \
    if (p.trace && (threadIdx.x & 63) == 0)                                                        \
      p.trace[((size_t)(g) * 4 + (lvl)) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime();          \
  } while (0)
#define LK_TRACE_END_U(g, lvl)                                                                     \
  do {                                                                                             \
    LK_TRACE_U(g, lvl, 5);                                                                         \
    if (issued == 0) {                                                                             \
      __builtin_amdgcn_s_waitcnt(0x0F70);                                                          \
      LK_TRACE_U(g, lvl, 6);                                                                       \
    }                                                                                              \
    if (p.trace && lane == 0)                                                                      \
      p.trace[((size_t)(g) * 4 + (lvl)) * 8 + 7] =                                                 \
          ((unsigned long long)(lvl) << 32) | __builtin_amdgcn_s_getreg(4 | (31 << 11));           \
  } while (0)
#define LK_LAB_PIPE_MODE(v) (scf_lab_pipe >= 0 ? scf_lab_pipe : (v))
#define LK_LAB_SETUP(p, nblk)                                                                      \
  do {                                                                                             \
    p.trace = scf_lab_trace; p.skip_dma = scf_lab_skip_dma; p.skip_store = scf_lab_skip_store;     \
    p.stagger = scf_lab_stagger; p.early = scf_lab_early;                                          \
    if (scf_lab_grid > 0) nblk = scf_lab_grid;                                                     \
  } while (0)
#define LK_EARLY_MAPS (p.early != 0)
// blocks whose slot on their XCD is odd start late: stagger x s_sleep 16 (~0.5 us each)
#define LK_LAB_STAGGER                                                                             \
  do {                                                                                             \
    if (p.stagger > 0 && ((blockIdx.x >> 3) & 1))                                                  \
      for (int i_ = 0; i_ < p.stagger; ++i_) __builtin_amdgcn_s_sleep(16);                         \
  } while (0)

#define LK_SKIP_DMA (p.skip_dma != 0)
#define LK_SKIP_STORE (p.skip_store != 0)

// launch-side: hand the switches to the kernel, optional grid override, store policy A/B (r = 4)
#define LK_LAB_LAUNCH(p, nblk)                                                                     \
  do {                                                                                             \
    p.trace = scf_lab_trace; p.skip_dma = scf_lab_skip_dma; p.skip_store = scf_lab_skip_store;     \
    p.stagger = scf_lab_stagger; p.early = scf_lab_early;                                          \
    if (scf_lab_grid > 0) nblk = scf_lab_grid;                                                     \
    if (r == 4 && scf_lab_store_mode > 0) {                                                        \
      switch (scf_lab_store_mode) {                                                                \
        case 1: scf_launch((corr_lookup_kernel<4, 1>), dim3((unsigned)nblk), dim3(256), lds, scf_stream(stream), p); break; \
        case 2: scf_launch((corr_lookup_kernel<4, 2>), dim3((unsigned)nblk), dim3(256), lds, scf_stream(stream), p); break; \
        case 3: scf_launch((corr_lookup_kernel<4, 3>), dim3((unsigned)nblk), dim3(256), lds, scf_stream(stream), p); break; \
        case 5: scf_launch((corr_lookup_kernel<4, 0>), dim3((unsigned)nblk), dim3(256), lds, scf_stream(stream), p); break; \
        default: scf_launch((corr_lookup_kernel<4, 4>), dim3((unsigned)nblk), dim3(256), lds, scf_stream(stream), p); break; \
      }                                                                                            \
      return scf_launch_status();                                                                  \
    }                                                                                              \
  } while (0)
