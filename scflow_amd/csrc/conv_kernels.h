// Shared between the fp32 (conv_mfma.hip) and split-fp16 (conv_f16x3.hip) convolution kernels.
#pragma once
#include "scf_common.h"

struct ConvK {
  const float* in0; const float* in1;
  int C0, Cin;
  long long in0_ns, in1_ns;
  int H, W, Ho, Wo;
  const float* wp; long long w_ns;
  const void* wp16;          // split-fp16 packing (conv_f16x3.hip) or nullptr
  const float* wthin;               // [Cin][T][CO] packing (conv_thin.hip) or nullptr
  const float* wp4; int G4, Mld4;   // LDS-DMA packing (conv_dma.hip) or nullptr
  const float* wp4s; int G4s;       // its small-grid variant (more channels per chunk) or nullptr
  const float* wp4t; int G4t;       // tiny-grid variant for 3x3 layers (32-channel chunks) or nullptr
  int Mld, Cout, Krows;
  int KH, KW, T, stride, pad_h, pad_w, KC, nchunk;
  int fc_log2, tiles_x, tiles_y, mblocks, PH, PW, PWin;
  int px_off;                // PX4 patch staging: columns between the 16-byte-aligned row start and ix0
  int in_step;               // 2: a 1x1 / stride-2 layer run as a dense 1x1 over every second input row and column (conv_dma.hip)
  int wvec;
  int out_tile;              // 1: 8x4-float tiled output planes (correlation level 0)
  int ksplit;                // 32-pixel tile, the 4 waves split the k-steps (small grids)
  float* out; long long out_ns;
  const float* bias; const float* scale; const float* shift;
  const float* res; long long res_ns;
  float out_div;
  int out_div_pow2;          // out_div is an exact power of two
  int act, act2, act_split, mode;
  const float* gru_h; long long gru_h_ns;
  float* gru_aux; long long gru_aux_ns;
  const float* gru_z; long long gru_z_ns;
  int kslices;               // > 1: K split across groups of blocks (conv_dma.hip), partial tensors slice_ns floats apart
  long long slice_ns;
  int slice_blocks;          // blocks per slice
};

// Per-sample base pointers of every tensor the epilogue touches.  The sample index is uniform
// per block, so these live in SGPRs and each access is "scalar base + 32-bit lane offset"
// (co*HWo + pix): one VGPR of addressing per output row, shared by all tensors.
struct ConvEpi {
  float* out; const float* res; const float* gru_h; float* gru_aux; const float* gru_z;
  int HWo;
};

__device__ __forceinline__ ConvEpi scf_conv_epi(const ConvK& p, int n) {
  ConvEpi e;
  e.out = p.out + (long long)n * p.out_ns;
  e.res = p.res ? p.res + (long long)n * p.res_ns : nullptr;
  e.gru_h = p.gru_h ? p.gru_h + (long long)n * p.gru_h_ns : nullptr;
  e.gru_aux = p.gru_aux ? p.gru_aux + (long long)n * p.gru_aux_ns : nullptr;
  e.gru_z = p.gru_z ? p.gru_z + (long long)n * p.gru_z_ns : nullptr;
  e.HWo = p.Ho * p.Wo;
  return e;
}

// ---------------------------------------------------------------------------------------------
// Fused epilogue.  One 32x32 accumulator fragment = 16 values per lane: channel rows
// co0 + (r&3) + 8*(r>>2) + 4*half of one pixel, handled as 4 groups of 4 consecutive rows.
// Order of operations: /div, +bias, BN scale/shift, +residual, activation, GRU gating.
//
// The epilogue KIND is uniform per launch and selected once per fragment, so that the path a
// launch actually executes is a short straight run of code: the common affine kind (bias / BN /
// residual / ReLU-or-none) costs ~5 instructions per value; the transcendental kinds (sigmoid /
// tanh heads, the two GRU gates) keep their own specialised bodies.  (A single body with
// per-value switches made the epilogue ~25 % of some layers' time and ~100 KB of code.)
// ---------------------------------------------------------------------------------------------
typedef float scf_f32x16 __attribute__((ext_vector_type(16)));
typedef float scf_f32x4 __attribute__((ext_vector_type(4)));

enum { SCF_EPI_AFFINE = 0, SCF_EPI_GENERAL = 1, SCF_EPI_GRU_ZR = 2, SCF_EPI_GRU_Q = 3 };

__device__ __forceinline__ int scf_conv_epi_kind(const ConvK& p) {
  if (p.mode == SCF_CONV_GRU_ZR) return SCF_EPI_GRU_ZR;
  if (p.mode == SCF_CONV_GRU_Q) return SCF_EPI_GRU_Q;
  const bool lin1 = p.act == SCF_ACT_NONE || p.act == SCF_ACT_RELU;
  const bool lin2 = p.act_split <= 0 || p.act2 == SCF_ACT_NONE || p.act2 == SCF_ACT_RELU;
  return (lin1 && lin2) ? SCF_EPI_AFFINE : SCF_EPI_GENERAL;
}

// affine kind, 4 rows cb..cb+3 (cb % 4 == 0) of one pixel
__device__ __forceinline__ void scf_epi_affine_group(const ConvK& p, const ConvEpi& e,
                                                     const float (&acc)[4], int cb, int pix,
                                                     bool use_div) {
  if (cb >= p.Cout) return;
  const bool full = cb + 3 < p.Cout;
  float v[4] = {acc[0], acc[1], acc[2], acc[3]};
  if (use_div) {
    // a power-of-two divisor (sqrt(C) for C = 64, 256, 1024: the SCFlow feature widths) is an
    // exact multiply by its reciprocal; anything else takes the correctly rounded division
    if (p.out_div_pow2) {
      const float rd = 1.0f / p.out_div;
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = v[q] * rd;
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = v[q] / p.out_div;
    }
  }
  int off = cb * e.HWo + pix;
  float r[4] = {0.f, 0.f, 0.f, 0.f};
  if (e.res) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (full || cb + q < p.Cout) r[q] = e.res[off + q * e.HWo];
  }
  if (p.bias) {
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] += p.bias[(full || cb + q < p.Cout) ? cb + q : cb];
  }
  if (p.scale) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = (full || cb + q < p.Cout) ? cb + q : cb;
      v[q] = v[q] * p.scale[c] + p.shift[c];
    }
  }
  if (e.res) {
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] += r[q];
  }
  // ReLU as compare+select: a NaN accumulator becomes 0 (the reference's F.relu would propagate
  // it; activations are finite by construction).  Per-row choice under act_split
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int a = (p.act_split > 0 && cb + q >= p.act_split) ? p.act2 : p.act;
    if (a == SCF_ACT_RELU) v[q] = v[q] > 0.f ? v[q] : 0.f;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (full || cb + q < p.Cout) e.out[off + q * e.HWo] = v[q];
}

// every other kind: one group of 4 consecutive channel rows cb..cb+3 of one pixel.  The
// auxiliary operands of the group (residual / GRU z, h) are gathered first so that their loads
// are in flight together instead of one dependent round trip per value.
template <int KIND>
__device__ __forceinline__ void scf_epi_general_group(const ConvK& p, const ConvEpi& e,
                                                      const float (&acc)[4], int cb, int pix,
                                                      bool use_div) {
  const bool need_aux = (KIND == SCF_EPI_GENERAL) ? (e.res != nullptr) : true;
  const int hc = p.Cout >> 1;
  float aux0[4] = {0.f, 0.f, 0.f, 0.f}, aux1[4] = {0.f, 0.f, 0.f, 0.f};
  float aux2[4] = {0.f, 0.f, 0.f, 0.f};    // GRU kinds: pre-activation term (res), e.g. the context part
  if (need_aux) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int co = cb + q;
      if (co < p.Cout) {
        const int off = co * e.HWo + pix;
        if (KIND == SCF_EPI_GENERAL) {
          aux0[q] = e.res[off];
        } else if (KIND == SCF_EPI_GRU_ZR) {
          if (co >= hc) aux0[q] = e.gru_h[off - hc * e.HWo];
          if (e.res) aux2[q] = e.res[off];
        } else {
          aux0[q] = e.gru_h[off];
          aux1[q] = e.gru_z[off];
          if (e.res) aux2[q] = e.res[off];
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int co = cb + q;
    if (co < p.Cout) {
      const int off = co * e.HWo + pix;
      float v = acc[q];
      if (use_div) v = v / p.out_div;
      if (p.bias) v += p.bias[co];
      if (KIND == SCF_EPI_GENERAL) {
        if (p.scale) v = v * p.scale[co] + p.shift[co];
        v += aux0[q];
        const int a = (p.act_split > 0 && co >= p.act_split) ? p.act2 : p.act;
        e.out[off] = scf_apply_act(v, a);
      } else if (KIND == SCF_EPI_GRU_ZR) {
        const float sg = scf_fast_sigmoid(v + aux2[q]);
        if (co < hc) e.out[off] = sg;
        else e.gru_aux[off - hc * e.HWo] = sg * aux0[q];
      } else {
        const float qv = scf_fast_tanh(v + aux2[q]);
        e.out[off] = (1.f - aux1[q]) * aux0[q] + aux1[q] * qv;
      }
    }
  }
}

// One whole 32x32 fragment (16 rows of one pixel) of a non-affine kind: every load (bias and the
// auxiliary operands: residual / GRU h, z) is issued before the first store, so the in-order
// vmcnt counter makes the wave wait for outstanding stores once per fragment instead of once
// per 4-row group.  cb = first channel of this lane's rows (co0 + 4*half).
template <int KIND>
__device__ __forceinline__ void scf_epi_general_frag(const ConvK& p, const ConvEpi& e,
                                                     const scf_f32x16& acc, int cb, int pix,
                                                     bool use_div) {
  const int hc = p.Cout >> 1;
  float bv[16], a0[16], a1[16];      // GRU kinds: bv also takes the pre-activation term (res)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = cb + 8 * (r >> 2) + (r & 3);
    const bool ok = co < p.Cout;
    const int cc = ok ? co : 0;
    const int off = cc * e.HWo + pix;
    bv[r] = p.bias ? p.bias[cc] : 0.f;
    a0[r] = 0.f;
    a1[r] = 0.f;
    if (KIND == SCF_EPI_GENERAL) {
      if (e.res) a0[r] = e.res[off];
    } else if (KIND == SCF_EPI_GRU_ZR) {
      if (ok && co >= hc) a0[r] = e.gru_h[off - hc * e.HWo];
      if (e.res) bv[r] += e.res[off];
    } else {
      a0[r] = e.gru_h[off];
      a1[r] = e.gru_z[off];
      if (e.res) bv[r] += e.res[off];
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = cb + 8 * (r >> 2) + (r & 3);
    if (co < p.Cout) {
      const int off = co * e.HWo + pix;
      float v = acc[r];
      if (use_div) v = v / p.out_div;
      v += bv[r];
      if (KIND == SCF_EPI_GENERAL) {
        if (p.scale) v = v * p.scale[co] + p.shift[co];
        v += a0[r];
        const int a = (p.act_split > 0 && co >= p.act_split) ? p.act2 : p.act;
        e.out[off] = scf_apply_act(v, a);
      } else if (KIND == SCF_EPI_GRU_ZR) {
        const float sg = scf_fast_sigmoid(v);
        if (co < hc) e.out[off] = sg;
        else e.gru_aux[off - hc * e.HWo] = sg * a0[r];
      } else {
        const float qv = scf_fast_tanh(v);
        e.out[off] = (1.f - a1[r]) * a0[r] + a1[r] * qv;
      }
    }
  }
}

// one group of 4 rows, kind chosen at run time (K-split kernel)
__device__ __forceinline__ void scf_conv_epilogue_group(const ConvK& p, const ConvEpi& e,
                                                        const float (&acc)[4], int cb, int pix,
                                                        bool use_div) {
  switch (scf_conv_epi_kind(p)) {
    case SCF_EPI_AFFINE: scf_epi_affine_group(p, e, acc, cb, pix, use_div); break;
    case SCF_EPI_GENERAL: scf_epi_general_group<SCF_EPI_GENERAL>(p, e, acc, cb, pix, use_div); break;
    case SCF_EPI_GRU_ZR: scf_epi_general_group<SCF_EPI_GRU_ZR>(p, e, acc, cb, pix, use_div); break;
    default: scf_epi_general_group<SCF_EPI_GRU_Q>(p, e, acc, cb, pix, use_div); break;
  }
}

// All fragments of a wave: acc[i][j] covers channels m0 + 32 i .. +31 at pixel pix[j]
// (pix[j] < 0: outside the image).  The kind switch sits OUTSIDE the fragment loops.
template <int WM, int WN, typename ACC>
__device__ __forceinline__ void scf_conv_epilogue_tile(const ConvK& p, const ConvEpi& e,
                                                       const ACC& acc, int m0, int half,
                                                       const int (&pix)[WN], bool use_div) {
  const int kind = scf_conv_epi_kind(p);
  // Fast path (most launches: bias, optional ReLU, whole channel fragments): every load of the
  // tile is issued BEFORE the first store.  Loads and stores share the in-order vmcnt counter on
  // gfx9, so a bias load issued after a store cannot be consumed until that store has been
  // acknowledged by memory (~2k cycles) -- per 4-row group in the generic path below.
  // r6: the same schedule for the transcendental activations and for a split at a fragment boundary (the context
  // encoder's head: tanh | relu at channel 128, raft_encoder.py / scflow_refiner.py:104-110) -- the activation is then
  // uniform per 32-channel fragment and runs on the hardware exp / rcp units (scf_fast_tanh / scf_fast_sigmoid, |error|
  // <= 3e-7).  On the per-group generic path below (libm tanhf, a store-acknowledge wait per 4 rows) that launch cost
  // 54 us against 29 us for the same layer with a plain epilogue (tools/lab/r6_quick.py ctxsplit).
  if (kind == SCF_EPI_GENERAL && !e.res && !p.scale && !use_div && m0 + WM * 32 <= p.Cout &&
      ((uintptr_t)p.bias & 15) == 0 && (p.act_split <= 0 || (p.act_split & 31) == 0)) {
    scf_f32x4 bv[WM][4];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        bv[i][g] = p.bias ? *reinterpret_cast<const scf_f32x4*>(p.bias + m0 + i * 32 + 8 * g + 4 * half)
                          : scf_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      if (pix[j] < 0) continue;
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const int a = (p.act_split > 0 && m0 + i * 32 >= p.act_split) ? p.act2 : p.act;      // wave-uniform
        float* o = e.out + (m0 + i * 32 + 4 * half) * e.HWo + pix[j];
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r] + bv[i][r >> 2][r & 3];
        if (a == SCF_ACT_TANH) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = scf_fast_tanh(v[r]);
        } else if (a == SCF_ACT_SIGMOID) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = scf_fast_sigmoid(v[r]);
        } else if (a == SCF_ACT_RELU) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) scf_store1<(SCF_ST_SC1 & 4) != 0>(o + (8 * (r >> 2) + (r & 3)) * e.HWo, v[r]);
      }
    }
    return;
  }
  if (kind == SCF_EPI_AFFINE && !e.res && !p.scale && p.act_split <= 0 && !use_div &&
      m0 + WM * 32 <= p.Cout && ((uintptr_t)p.bias & 15) == 0) {
    scf_f32x4 bv[WM][4];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        bv[i][g] = p.bias ? *reinterpret_cast<const scf_f32x4*>(p.bias + m0 + i * 32 + 8 * g + 4 * half)
                          : scf_f32x4{0.f, 0.f, 0.f, 0.f};
    const bool relu = p.act == SCF_ACT_RELU;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      if (pix[j] < 0) continue;
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        float* o = e.out + (m0 + i * 32 + 4 * half) * e.HWo + pix[j];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[i][j][r] + bv[i][r >> 2][r & 3];
          if (relu) v = v > 0.f ? v : 0.f;
          scf_store1<(SCF_ST_SC1 & 4) != 0>(o + (8 * (r >> 2) + (r & 3)) * e.HWo, v);
        }
      }
    }
    return;
  }
  // Same idea for BN scale/shift and residual adds (context-encoder blocks), one fragment at a
  // time: all of a fragment's loads, then its 16 stores (one store-acknowledge wait per fragment
  // instead of one per 4-row group).  The scheduling barriers keep the other fragments' loads
  // from being hoisted on top (registers).
  if (kind == SCF_EPI_AFFINE && p.act_split <= 0 && !use_div && m0 + WM * 32 <= p.Cout &&
      (((uintptr_t)p.bias | (uintptr_t)p.scale | (uintptr_t)p.shift) & 15) == 0) {
    const scf_f32x4 zero4 = {0.f, 0.f, 0.f, 0.f}, one4 = {1.f, 1.f, 1.f, 1.f};
    const bool relu = p.act == SCF_ACT_RELU;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      if (pix[j] < 0) continue;
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const int c0 = m0 + i * 32 + 4 * half;
        const int o0 = c0 * e.HWo + pix[j];
        scf_f32x4 bv[4], sc[4], sh[4];
        float rs[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bv[g] = p.bias ? *reinterpret_cast<const scf_f32x4*>(p.bias + c0 + 8 * g) : zero4;
          sc[g] = p.scale ? *reinterpret_cast<const scf_f32x4*>(p.scale + c0 + 8 * g) : one4;
          sh[g] = p.scale ? *reinterpret_cast<const scf_f32x4*>(p.shift + c0 + 8 * g) : zero4;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) rs[r] = e.res ? e.res[o0 + (8 * (r >> 2) + (r & 3)) * e.HWo] : 0.f;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[i][j][r] + bv[r >> 2][r & 3];
          if (p.scale) v = v * sc[r >> 2][r & 3] + sh[r >> 2][r & 3];
          v += rs[r];
          if (relu) v = v > 0.f ? v : 0.f;
          scf_store1<(SCF_ST_SC1 & 32) != 0>(e.out + o0 + (8 * (r >> 2) + (r & 3)) * e.HWo, v);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    return;
  }
#define SCF_EPI_LOOP(CALL)                                                        \
  _Pragma("unroll") for (int j = 0; j < WN; ++j) {                                \
    if (pix[j] >= 0) {                                                            \
      _Pragma("unroll") for (int i = 0; i < WM; ++i) {                            \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                           \
          const float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], \
                              acc[i][j][4 * g + 3]};                              \
          CALL(p, e, v, m0 + i * 32 + 8 * g + 4 * half, pix[j], use_div);         \
        }                                                                         \
      }                                                                           \
    }                                                                             \
  }
  if (kind == SCF_EPI_AFFINE) { SCF_EPI_LOOP(scf_epi_affine_group) }
#undef SCF_EPI_LOOP
#define SCF_EPI_FRAGS(KIND_)                                                      \
  _Pragma("unroll") for (int j = 0; j < WN; ++j) {                                \
    if (pix[j] >= 0) {                                                            \
      _Pragma("unroll") for (int i = 0; i < WM; ++i)                              \
        scf_epi_general_frag<KIND_>(p, e, acc[i][j], m0 + i * 32 + 4 * half, pix[j], use_div); \
    }                                                                             \
  }
  else if (kind == SCF_EPI_GENERAL) { SCF_EPI_FRAGS(SCF_EPI_GENERAL) }
  else if (kind == SCF_EPI_GRU_ZR) { SCF_EPI_FRAGS(SCF_EPI_GRU_ZR) }
  else { SCF_EPI_FRAGS(SCF_EPI_GRU_Q) }
#undef SCF_EPI_FRAGS
}

// conv_f16x3.hip: tile selection + launch of the split-fp16 kernel (SCF_EUNSUPPORTED -> caller
// falls back to fp32).  info (optional): {WM, WN, blocks, MFMAs per wave per chunk}.
int scf_conv_f16x3_dispatch(ConvK k, int N, bool dry_run, int* info, hipStream_t st);

// r6: what a dispatch WOULD launch (filled instead of launching when a capture is passed): two captures of the same kernel
// instantiation run as ONE launch (scf_conv_*_pair_launch: blocks [0, a.nblk) = layer a, the rest layer b).
struct ScfLaunchCap {
  ConvK k;
  int nblk = 0;
  size_t ldsb = 0;
  int variant = -1;           // kernel instantiation id (family-specific); -1 = not pairable (pixel-split tiles, persistent, ...)
  const float* wt = nullptr;  // taps: the packing
  int Kp = 0, PWp = 0;        // taps
  alignas(8) unsigned char aux[96] = {};      // Winograd: the kernel's second argument (WinoK)
};
// conv_taps.hip: Cin <= 4 layers, contraction over taps x channels (wt = the [Kp][Mld] taps packing).
int scf_conv_taps_dispatch(ConvK k, const float* wt, int N, bool dry_run, int* info, hipStream_t st, ScfLaunchCap* cap = nullptr);
int scf_conv_taps_pair_launch(const ScfLaunchCap& a, const ScfLaunchCap& b, hipStream_t st);      // SCF_EUNSUPPORTED: not the same instantiation
// conv_wino.hip: 3x3 / stride-1 / pad-1 layers with an affine epilogue in the Winograd F(2x2, 3x3) form
// (wu = the G g G^T packing).  info: {16, fragments per block, blocks, LDS bytes}.
int scf_conv_wino_dispatch(ConvK k, const float* wu, int N, bool dry_run, int* info, hipStream_t st, int* which = nullptr, ScfLaunchCap* cap = nullptr);
int scf_conv_wino_pair_launch(const ScfLaunchCap& a, const ScfLaunchCap& b, hipStream_t st);      // quarter-domain kernel only
// conv_wino1d.hip: 1x5 / 5x1 stride-1 'same' layers, any epilogue kind, in the Winograd F(2, 5) form (wu = the G g packing).
int scf_conv_wino1d_dispatch(ConvK k, const float* wu, int N, bool dry_run, int* info, hipStream_t st);
// conv_wino1d4.hip: the same layers in the F(4, 5) form (wu = its own G g packing), grids of more than CUs / 2 blocks only.
int scf_conv_wino1d4_dispatch(ConvK k, const float* wu, int N, bool any_grid, bool dry_run, int* info, hipStream_t st);
// conv_thin.hip: Cout <= 4 layers on the vector ALUs.
int scf_conv_thin_dispatch(ConvK k, int N, bool dry_run, hipStream_t st, ScfLaunchCap* cap = nullptr);
int scf_conv_thin_pair_launch(const ScfLaunchCap& a, const ScfLaunchCap& b, hipStream_t st);
// conv_dma.hip: same contract for the stride-1 LDS-DMA fp32 kernel.
int scf_conv_dma_dispatch(ConvK k, int N, bool dry_run, int* info, hipStream_t st, ScfLaunchCap* cap = nullptr);
int scf_conv_dma_pair_launch(const ScfLaunchCap& a, const ScfLaunchCap& b, hipStream_t st);
int scf_conv_dma_taps_pair_launch(const ScfLaunchCap& dma, const ScfLaunchCap& taps, hipStream_t st);      // K-split layer | thin-input layer
