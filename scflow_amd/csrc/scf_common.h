// Shared helpers for the gfx950 kernels of libscflow_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <atomic>

#include "../../include/scflow_hip.h"
#include "../../include/scflow_hip_prof.h"

#define SCF_WAVE 64

static inline int scf_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SCF_OK : SCF_ELAUNCH;
}

static inline hipStream_t scf_stream(scf_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t scf_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// CUs of the current device (cached per device; 256 = MI355X when no device answers, e.g. a dry
// run of the tile selection on a host without a GPU)
static inline int scf_cu_count() {
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev] = n;
  }
  return cus[dev];
}

// Opt a kernel in to more than 64 KiB of dynamic LDS, once per (kernel, device): `done` is that kernel's own
// flag word (bit d = raised on device d).  Two host threads meeting here both call hipFuncSetAttribute -- the call
// is idempotent -- and both set the bit: no unsynchronised flag, no lock on the launch path.
static inline int scf_raise_dynamic_lds(std::atomic<unsigned long long>& done, const void* fn, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return SCF_ELAUNCH;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return SCF_OK;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return SCF_ELAUNCH;
  done.fetch_or(bit, std::memory_order_release);
  return SCF_OK;
}

// A timer = two HIP events bound to ONE kernel launch (hipExtLaunchKernel start / stop events:
// the dispatch's own begin / end timestamps, i.e. what a kernel trace reports; a pair of
// recorded events around a launch additionally contains ~3 us of dispatch).  scf_timer_arm()
// attaches a timer to the NEXT kernel this thread launches through scf_launch().
struct ScfTimer { hipEvent_t start, stop; };
ScfTimer*& scf_armed_timer();          // thread-local slot, defined in capi.hip

template <typename F, typename... Args>
static inline void scf_launch(F kernel, dim3 grid, dim3 block, size_t lds, hipStream_t st, Args... args) {
  ScfTimer*& slot = scf_armed_timer();
  if (slot) {
    ScfTimer* t = slot;
    slot = nullptr;
    hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)lds, st, t->start, t->stop, 0, args...);
  } else {
    hipLaunchKernelGGL(kernel, grid, block, lds, st, args...);
  }
}

// XCD-aware block remap (bijective for any grid size): consecutive LOGICAL ids run on the
// same XCD so that blocks sharing an input halo / a weight slab hit the same L2.
// Hardware places physical block b on XCD b % 8 (observed, used for speed only).
__device__ __forceinline__ int scf_xcd_remap(int b, int nblk) {
  const int nx = 8;
  int q = nblk / nx, r = nblk % nx;
  int xcd = b % nx, slot = b / nx;
  int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

// sigmoid / tanh on the hardware transcendental units (v_exp_f32, v_rcp_f32: ~1 ulp each) for the
// fused convolution epilogues: ~6 vector instructions instead of ~35 for the libm expansions --
// vector issue slots are the scarce resource next to a co-resident wave's MFMA stream.
// |error| <= ~3e-7 (sigmoid, relative) / ~1.5e-7 (tanh, absolute).
__device__ __forceinline__ float scf_fast_sigmoid(float v) {
  return __builtin_amdgcn_rcpf(1.f + __expf(-v));
}
__device__ __forceinline__ float scf_fast_tanh(float v) {
  const float t = __expf(-2.f * fabsf(v));            // in (0, 1]: never overflows
  const float r = (1.f - t) * __builtin_amdgcn_rcpf(1.f + t);
  return copysignf(r, v);
}

__device__ __forceinline__ float scf_apply_act(float v, int act) {
  switch (act) {
    case SCF_ACT_RELU: return v > 0.f ? v : 0.f;
    case SCF_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    case SCF_ACT_TANH: return tanhf(v);
    default: return v;
  }
}

// Output stores.  A plain store leaves its line dirty in this XCD's L2 until the release at the end of the kernel writes
// the whole L2 back -- inside the kernel's duration; a write-through store (sc1) drains while the kernel runs (the lookup uses
// them: -2.5 us per launch).  SCF_ST_SC1 (bit mask, lab builds: -DSCF_ST_SC1=m) selects further kernels: 1 bilinear resize,
// 2 instance norm, 4 the convolutions' affine epilogue (bias / ReLU), 8 the F(2x2, 3x3) kernels' epilogues, 16 the F(4, 5)
// gate epilogues, 32 the convolutions' BN / residual epilogue.  Measured on the batch-32 step with identical results checked
// (tools/lab/lib_ab.py, five MI355X boxes, mask 55 against 0): -0.07, +0.10, +0.35, +-0, -0.07 ms -- box to box it helps or
// hurts, on the slowest box the most; the product keeps plain stores (0).
#ifndef SCF_ST_SC1
#define SCF_ST_SC1 0
#endif
typedef float scf_st_f32x4 __attribute__((ext_vector_type(4)));
template <bool SC1>
__device__ __forceinline__ void scf_store4(float* p, float a, float b, float c, float d) {
  if constexpr (SC1) {
    const scf_st_f32x4 v = {a, b, c, d};
    // s_nop 1: a VMEM store of more than 64 bits must not be followed within two wait states by a VALU write of its data
    // registers (gfx940+); the compiler's hazard recogniser pads its own stores but cannot see inside an asm statement --
    // without the nop the next instruction overwrote the data before the store had read it (GRU state: inf).
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
  } else {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
  }
}
typedef float scf_st_f32x2 __attribute__((ext_vector_type(2)));
template <bool SC1>
__device__ __forceinline__ void scf_store2(float* p, float a, float b) {
  if constexpr (SC1) {
    const scf_st_f32x2 v = {a, b};
    asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
  } else {
    *reinterpret_cast<scf_st_f32x2*>(p) = scf_st_f32x2{a, b};
  }
}
template <bool SC1>
__device__ __forceinline__ void scf_store1(float* p, float a) {
  if constexpr (SC1) asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p), "v"(a) : "memory");
  else *p = a;
}

// ---- launches shared between translation units (not part of the C ABI) ----
// resample.hip: bilinear resize (align_corners) with an optional second job of the same geometry and an optional second,
// sample-strided destination of job 0
int scf_resize_bilinear_jobs(const float* a, const float* b, float* out, int64_t planes, float mul,
                             const float* a1, const float* b1, float* out1, int64_t planes1, float mul1,
                             float* out2, int out2_group, int64_t out2_gstride,
                             int Hin, int Win, int Hout, int Wout, scf_stream_t stream);
// pose.hip: scf_pose_update + scf_reproject_flow as one launch (every block of a sample recomputes that sample's pose)
int scf_pose_update_reproject(const float* rot_all, const float* trans_all, const int64_t* label, int num_class,
                              int label_mode, const float* R_in, const float* t_in, float* d_rot, float* d_trans,
                              float* R_out, float* t_out, const float* depth, const float* K, const float* R0,
                              const float* t0, float* flow, int N, int H, int W, float invalid_num, scf_stream_t stream);
