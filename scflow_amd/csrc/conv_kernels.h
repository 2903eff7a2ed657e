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
  int Mld, Cout, Krows;
  int KH, KW, T, stride, pad_h, pad_w, KC, nchunk;
  int fc_log2, tiles_x, tiles_y, mblocks, PH, PW;
  int wvec;
  int out_tile;              // 1: 8x4-float tiled output planes (correlation level 0)
  int ksplit;                // 32-pixel tile, the 4 waves split the k-steps (small grids)
  float* out; long long out_ns;
  const float* bias; const float* scale; const float* shift;
  const float* res; long long res_ns;
  float out_div;
  int act, act2, act_split, mode;
  const float* gru_h; long long gru_h_ns;
  float* gru_aux; long long gru_aux_ns;
  const float* gru_z; long long gru_z_ns;
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

// One 32x32 accumulator fragment (16 values per lane: channel rows co0 + (r&3) + 8*(r>>2) +
// 4*half, one pixel), handled in 4 groups of 4 rows: the auxiliary operands of a group
// (residual / GRU z, h) are gathered first so that their loads are in flight together instead
// of one dependent round trip per value.
// Order: /div, +bias, BN scale/shift, +residual, activation, GRU gating.
typedef float scf_f32x16 __attribute__((ext_vector_type(16)));

// one group of 4 consecutive channel rows cb..cb+3 of one pixel
__device__ __forceinline__ void scf_conv_epilogue_group(const ConvK& p, const ConvEpi& e,
                                                        const float (&acc)[4], int cb, int pix,
                                                        bool use_div) {
  const bool need_aux = (p.mode == SCF_CONV_PLAIN) ? (e.res != nullptr) : true;
  const int hc = p.Cout >> 1;
  float aux0[4] = {0.f, 0.f, 0.f, 0.f}, aux1[4] = {0.f, 0.f, 0.f, 0.f};
  if (need_aux) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int co = cb + q;
      if (co < p.Cout) {
        const int off = co * e.HWo + pix;
        if (p.mode == SCF_CONV_PLAIN) {
          aux0[q] = e.res[off];
        } else if (p.mode == SCF_CONV_GRU_ZR) {
          if (co >= hc) aux0[q] = e.gru_h[off - hc * e.HWo];
        } else {
          aux0[q] = e.gru_h[off];
          aux1[q] = e.gru_z[off];
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
      if (p.mode == SCF_CONV_PLAIN) {
        if (p.scale) v = v * p.scale[co] + p.shift[co];
        v += aux0[q];
        const int a = (p.act_split > 0 && co >= p.act_split) ? p.act2 : p.act;
        e.out[off] = scf_apply_act(v, a);
      } else if (p.mode == SCF_CONV_GRU_ZR) {
        const float sg = 1.f / (1.f + expf(-v));
        if (co < hc) e.out[off] = sg;
        else e.gru_aux[off - hc * e.HWo] = sg * aux0[q];
      } else {
        const float qv = tanhf(v);
        e.out[off] = (1.f - aux1[q]) * aux0[q] + aux1[q] * qv;
      }
    }
  }
}

__device__ __forceinline__ void scf_conv_epilogue_frag(const ConvK& p, const ConvEpi& e,
                                                       const scf_f32x16 acc, int co0, int half,
                                                       int pix, bool use_div) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float v[4] = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    scf_conv_epilogue_group(p, e, v, co0 + 8 * g + 4 * half, pix, use_div);
  }
}

// conv_f16x3.hip: tile selection + launch of the split-fp16 kernel (SCF_EUNSUPPORTED -> caller
// falls back to fp32).  info (optional): {WM, WN, blocks, MFMAs per wave per chunk}.
int scf_conv_f16x3_dispatch(ConvK k, int N, bool dry_run, int* info, hipStream_t st);
