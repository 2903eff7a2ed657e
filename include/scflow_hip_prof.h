/*
 * scflow_hip_prof.h -- measurement aids of libscflow_hip.so.  NOT part of the operator ABI a
 * reference maintainer binds (include/scflow_hip.h): bench.py, tools/ and the tests use these to
 * time single launches and to inspect the convolution tile selection.  Same conventions as
 * scflow_hip.h (plain C, SCF_* return codes).
 */
#ifndef SCFLOW_HIP_PROF_H
#define SCFLOW_HIP_PROF_H

#include "scflow_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* A timer = HIP start / stop events bound to ONE kernel launch (hipExtLaunchKernel: the
 * dispatch's own begin / end timestamps, what a kernel trace reports; a pair of recorded events
 * around a launch additionally contains ~3 us of dispatch).  A timer is reusable after its launch
 * has completed; read it after synchronising the stream. */
typedef void* scf_timer_t;
int scf_timer_create(scf_timer_t* timer);
int scf_timer_destroy(scf_timer_t timer);
int scf_timer_elapsed_us(scf_timer_t timer, float* microseconds);
/* attach the timer to the NEXT kernel this thread launches through the library (the convolution
 * of scf_conv2d, the contraction of scf_corr_build*, the lookup, ...); NULL disarms */
int scf_timer_arm(scf_timer_t timer);

/* dry run of scf_conv2d's tile selection: info[4] = {WM, WN, grid blocks, MFMAs per wave per
 * staged chunk (negative: the LDS-DMA / split-fp16 kernel runs)}; SCF_EUNSUPPORTED when the
 * packing's KC does not fit this shape.  No launch: works without a GPU. */
int scf_conv2d_query(const scf_conv_desc* desc, int32_t* info);

/* Dispatch log: which kernel family ran each convolution launch.  scf_conv_log_enable(capacity > 0) starts
 * (and clears) a process-wide log of up to `capacity` records; every successful scf_conv2d launch -- also
 * those issued inside scf_sepconv_gru[_ctx] and scf_scflow_iteration -- appends one; scf_conv_log_enable(0)
 * stops and frees it.  scf_conv_log_read copies up to max_entries records (out may be NULL) and returns the
 * number recorded.  The tests use it to assert that a parity case really exercised the kernel it names
 * (kernel selection depends on the grid size and the device: a threshold change must not silently move a
 * golden test onto other arithmetic). */
enum {
  SCF_KERNEL_THIN = 1,        /* conv_thin_kernel: Cout <= 4, vector ALU                           */
  SCF_KERNEL_TAPS = 2,        /* conv_taps_kernel: Cin <= 4, contraction over taps                 */
  SCF_KERNEL_WINO = 3,        /* conv_wino_kernel: Winograd F(2x2, 3x3), wave pair per fragment    */
  SCF_KERNEL_WINO1D = 4,      /* conv_wino1d_kernel: Winograd F(2, 5)                              */
  SCF_KERNEL_F16X3 = 5,       /* conv_f16x3_kernel: split-fp16 3xMFMA                              */
  SCF_KERNEL_DMA = 6,         /* conv_dma_kernel: direct, LDS-DMA staged (pixel-split or K-split)  */
  SCF_KERNEL_MFMA = 7,        /* conv_mfma_kernel: direct, register staged                         */
  SCF_KERNEL_MFMA_KSPLIT = 8, /* conv_mfma_ksplit_kernel                                           */
  SCF_KERNEL_WINO_Q = 9,      /* conv_wino_q_kernel: Winograd F(2x2, 3x3), one transform row x two channel
                                 fragments per wave (even fragment counts)                           */
  SCF_KERNEL_WINO1D4 = 10     /* conv_wino1d4_kernel: Winograd F(4, 5)                             */
};
typedef struct scf_conv_log_entry {
  int32_t kernel;             /* SCF_KERNEL_*                                                      */
  int32_t Cin, Cout, KH, KW, stride, Ho, Wo, N;
  int32_t mode;               /* SCF_CONV_*                                                        */
} scf_conv_log_entry;
int scf_conv_log_enable(int capacity);
/* measurement knobs (A/B runs of kernel variants from bench.py / tools): returns the previous value, or
 * SCF_EINVAL for an unknown key.  0 always means "the dispatch's own choice". */
enum {
  SCF_TUNE_WINO_VARIANT = 1,  /* F(2x2,3x3): 1 = pair kernel, 2 = quarter-domain kernel (4 waves; the default where it fits).  3 (8 waves),
                                 4 (four ring slots, 80 KB of LDS) and 5 (4 + burst form of a chunk) are compiled into -DSCF_WINO_LAB
                                 builds only: the product library answers SCF_EINVAL for them */
  SCF_TUNE_DMA_FORCE_KSPLIT = 2, /* 1: the LDS-DMA kernel takes its K-split tile (32 channels x 32 pixels per block) on every grid */
  SCF_TUNE_DMA_KSPLIT_GROUPS = 3, /* 1: K-split blocks keep one wave group (no intra-block split of the chunk chain) */
  SCF_TUNE_WINO1D4 = 4,       /* 1 (default): 1x5 / 5x1 layers that carry an F(4, 5) packing use it on large grids; 0: F(2, 5);
                                 2: F(4, 5) on every grid it supports (tests of small ragged shapes) */
  SCF_TUNE_CONV_AUTOSLICE = 7, /* 1 (default): small-grid convolutions on a stream with a registered workspace (scf_conv_workspace) are
                                 split into K slices + a combine launch; 0: never */
  SCF_TUNE_LOOKUP_STORE = 6,  /* correlation lookup (r = 4, one group per block): cache policy of the output stores, 0 = the build's,
                                 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt, 5 plain */
  SCF_TUNE_ITER_MERGE = 8,    /* scf_scflow_iteration: 1 (default) = merged launches (1/8 flow + its copy, both up-samplings, pose update +
                                 re-projection), 0 = one launch each (the r4 sequence; same results) */
  SCF_TUNE_WINO1D4_HALF = 9,  /* F(4, 5): 1 = half-domain kernel (a wave holds 4 of the 8 positions for two channel fragments: half the
                                 input-transform work per MFMA, one exchange per block), 0 (default) = the full-domain kernel */
  SCF_TUNE_CONV_PAIR = 10,    /* scf_conv2d_pair: 0 (default) = one launch when both layers take the same small-grid instantiation and their
                                 grids are resident on the chip together, 1 = always two launches */
  SCF_TUNE_LOOKUP_PIPE = 5    /* correlation lookup: 0 = the dispatch's own choice, 1 = one group of 32 queries per block (the r3
                                 kernel), 2 / 3 = the pipelined kernel with two / three groups per block wherever it fits,
                                 4 / 5 / 6 = two / four / three groups per 512- / 1024- / 768-thread block (same waves, fewer
                                 workgroups; 6 is the form measured on configs[4]: 32.5 vs 30.3 us, notebook R5.6) */
};
int scf_tune(int key, int value);
int scf_conv_log_read(scf_conv_log_entry* out, int max_entries);

#ifdef __cplusplus
}
#endif
#endif /* SCFLOW_HIP_PROF_H */
