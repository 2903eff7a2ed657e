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

#ifdef __cplusplus
}
#endif
#endif /* SCFLOW_HIP_PROF_H */
