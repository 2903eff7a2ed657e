// 3x3 / stride-1 / pad-1 convolution in the Winograd F(2x2, 3x3) form, fp32 throughout:
//     Y = A^T [ (G g G^T) . (B^T d B) ] A        (Lavin & Gray; the reference reaches the same
// algorithm through cuDNN for these layers: ConvModule / nn.Conv2d 3x3 in resnet.py:75-86,
// raft_decoder.py:141-160, 430-440, scflow_decoder.py:100-105.)
// 16 multiplies per 2x2 outputs and channel pair instead of 36: 2.25x fewer matrix-core flops than
// the direct kernels of conv_dma.hip, on the layers that are matrix-core bound there.
//
//   GEMM view   for each of the 16 transform positions xi:  M[xi][cout][tile] = sum_cin U[xi][cout][cin] V[xi][cin][tile]
//               v_mfma_f32_32x32x2_f32, M = 32 output channels, N = 32 tiles (2 x 2 outputs each), k = 2 channels.
//   wave        ONE 32 x 32 fragment for ALL 16 xi = 256 accumulator registers (one wave per SIMD, the
//               accumulators live in AGPRs), so the output transform A^T M A is lane-local: no exchange.
//   block       CW x TW waves: CW channel fragments x TW tile groups (stacked vertically in the image).
//   U           = G g G^T, transformed and packed on the host (scf_pack_conv_weight_wino) in the exact
//               LDS image [chunk][xi][fragment][k-half][cout][2]: one ds_read_b64 per xi feeds both
//               k-steps of a 4-channel chunk; staged by LDS-DMA into a 3-deep ring.
//   V           = B^T d B, computed IN the kernel: the raw input patch of the next chunk is staged by
//               LDS-DMA (descriptor range check = zero padding), each thread transforms one
//               (tile, channel) 4 x 4 window (32 adds) and writes its 16 values to the V double buffer,
//               interleaved with the MFMAs of the current chunk.
//   pipeline    per chunk: issue DMA of chunk c+2 (U) / c+3 (patch); transform patch c+1 -> V; MFMAs of
//               chunk c; one barrier.
//   epilogue    lane-local output transform, then the affine epilogue (bias, BN scale/shift, residual,
//               ReLU) and float2 stores: 16 lanes cover one full 128-byte line of an output row.
//
// Arithmetic: fp32 adds / fmas only; the transforms re-associate the sum, so results differ from the
// direct kernel by a few ulp of the accumulated magnitude (measured in tests/test_gpu_ops.py).
#include <stdlib.h>
#include <string.h>
#include "scf_common.h"
#include "conv_kernels.h"
#include "scf_dma.h"

typedef float wn_f32x16 __attribute__((ext_vector_type(16)));
typedef float wn_f32x2 __attribute__((ext_vector_type(2)));

#define WN_NPI(TW) ((TW) == 2 ? 6 : 11)   // patch DMA instructions per wave per chunk (256 floats each per block): fixed per
                                         // block shape, lanes past the patch write zeros into the slot's padding
#define WN_KC 4             // channels per chunk

struct WinoK {
  const float* wu;          // [nchunk][16][F][2][32][2]
  int F;                    // channel fragments in the packing
  int txl;                  // log2(tiles per row of a wave's 32-tile group)
  int PH, PW, PWp, PPL;     // patch rows, columns, row pitch, plane stride (floats)
  int nchunk;
  int sx, sy;               // strips per image
  int mblocks;
#ifdef SCF_WINO_LAB
  int lab;                  // tools/lab/wino_phases.py: bit 0 no MFMAs, 1 no transform, 2 no DMA in the loop, 3 no stores, 4 no barrier
#endif
};
#ifdef SCF_WINO_LAB
#define WN_LAB(bit) (q.lab & (1 << (bit)))
#else
#define WN_LAB(bit) 0
#endif

// floor(e / d) for 0 <= e < 2^20, 0 < d < 2^12 without the integer-division expansion
__device__ __forceinline__ int wn_div(int e, int d, float rd) {
  int q = (int)((float)e * rd);
  const int r = e - q * d;
  q += (r >= d) ? 1 : 0;
  q -= (r < 0) ? 1 : 0;
  return q;
}

template <int CW, int TW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void conv_wino_kernel(ConvK p, WinoK q) {
  static_assert(CW * TW == 4, "four waves per block");
  extern __shared__ __attribute__((aligned(16))) float wn_lds[];
  constexpr int USLOT = CW * 2048, VSLOT = TW * 2048;           // floats per ring slot
  constexpr int NUI = (CW == 1) ? 8 : CW * 2;                   // U DMA instructions per wave per chunk
  constexpr int NPI = WN_NPI(TW);                               // patch DMA instructions per wave per chunk
  constexpr int PSLOT = NPI * 256;
  constexpr int GRP = NUI + NPI;                                // DMA instructions per wave per group
  constexpr int TQ = (TW * 128 + 255) / 256;                    // windows per thread per chunk
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cw = wave / TW, tw = wave % TW;
  const int half = lane >> 5, l32 = lane & 31;

  int lb = scf_xcd_remap(blockIdx.x, gridDim.x);
  const int mb = __builtin_amdgcn_readfirstlane(lb % q.mblocks);
  lb /= q.mblocks;
  const int xs = __builtin_amdgcn_readfirstlane(lb % q.sx);
  lb /= q.sx;
  const int ys = __builtin_amdgcn_readfirstlane(lb % q.sy);
  const int n = __builtin_amdgcn_readfirstlane(lb / q.sy);
  const int TXW = 1 << q.txl, TYW = 32 >> q.txl;
  const int y0 = ys * (2 * TW * TYW), x0 = xs * (2 * TXW);      // first output pixel of the block
  const int f0 = mb * CW;
  const int HW = p.H * p.W;

  float* Us = wn_lds;
  float* Vs = Us + 3 * USLOT;
  float* Ps = Vs + 2 * VSLOT;
  const unsigned u_lds = scf_lds_addr(Us), p_lds = scf_lds_addr(Ps);

  // ---- chunk-invariant DMA offsets --------------------------------------------------------------
  unsigned pvo[NPI];                            // patch: byte offset inside the chunk's 4 channel planes
  {
    const float rPPL = 1.0f / (float)q.PPL, rPW = 1.0f / (float)q.PWp;
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
      const int e = i * 256 + tid;
      const int c = wn_div(e, q.PPL, rPPL), r = e - c * q.PPL;
      const int py = wn_div(r, q.PWp, rPW), px = r - py * q.PWp;
      const int iy = y0 - 1 + py, ix = x0 - 1 + px;
      const bool ok = c < WN_KC && py < q.PH && px < q.PW && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      pvo[i] = ok ? (unsigned)((c * HW + iy * p.W + ix) * 4) : SCF_BUF_OOB;
    }
  }
  unsigned uvo[NUI], uld[NUI];                  // U: byte offset inside the chunk's slab / inside the ring slot
#pragma unroll
  for (int i = 0; i < NUI; ++i) {
    const int j = wave + 4 * i;
    if (CW == 1) {
      const int xi = j >> 1, part = j & 1;
      uvo[i] = (unsigned)((xi * q.F + f0) * 512 + part * 256 + lane * 4);
      uld[i] = (unsigned)(xi * 512 + part * 256);
    } else {
      constexpr int IPX = CW / 2 > 0 ? CW / 2 : 1;
      const int xi = j / IPX, part = j % IPX;
      uvo[i] = (unsigned)((xi * q.F + f0) * 512 + part * 1024 + lane * 16);
      uld[i] = (unsigned)(xi * CW * 512 + part * 1024);
    }
  }
  const unsigned u_chunk_bytes = (unsigned)(16 * q.F * 512);
  const unsigned u_total = (unsigned)q.nchunk * u_chunk_bytes;

  auto issue_u = [&](int k, int slot) {          // U[k] -> ring slot (chunks past the end: zeros)
    const unsigned done = (unsigned)k * u_chunk_bytes;
    const scf_rsrc4 rs = scf_make_rsrc((const char*)q.wu + done, k < q.nchunk ? u_total - done : 0u);
    const unsigned dst = u_lds + (unsigned)(slot * USLOT * 4);
#pragma unroll
    for (int i = 0; i < NUI; ++i) {
      if (CW == 1) scf_bdma_b32(rs, uvo[i], dst + uld[i]);
      else scf_bdma_b128(rs, uvo[i], dst + uld[i]);
    }
  };
  auto issue_p = [&](int k, int slot) {          // patch[k] -> ring slot
    const int c0 = k * WN_KC;
    const float* base;
    int left;
    if (c0 < p.C0) { base = p.in0 + (long long)n * p.in0_ns + (long long)c0 * HW; left = p.C0 - c0; }
    else { base = (p.in1 ? p.in1 + (long long)n * p.in1_ns : p.in0) + (long long)(c0 - p.C0) * HW; left = p.in1 ? p.Cin - c0 : 0; }
    if (left > WN_KC) left = WN_KC;
    if (left < 0) left = 0;
    const scf_rsrc4 rs = scf_make_rsrc(base, (unsigned)(left * HW * 4));
    const unsigned dst = p_lds + (unsigned)((slot * PSLOT + wave * 64) * 4);
#pragma unroll
    for (int i = 0; i < NPI; ++i) scf_bdma_b32(rs, pvo[i], dst + (unsigned)(i * 1024));
  };

  // ---- input transform: window (tile, channel) of a patch slot -> 16 values of a V slot, in three
  //      pieces that the main loop places between groups of MFMAs (a piece runs in their shadow) ------
  int poff[TQ];
#pragma unroll
  for (int u = 0; u < TQ; ++u) {
    const int qi = tid + 256 * u;
    const int s = qi & 1, t32 = (qi >> 1) & 31, kh = (qi >> 6) & 1, twq = qi >> 7;
    const int ty = twq * TYW + (t32 >> q.txl), tx = t32 & (TXW - 1);
    poff[u] = (2 * s + kh) * q.PPL + 2 * ty * q.PWp + 2 * tx;
  }
  float d[TQ][4][4], w[TQ][4][4];
  auto tr_load = [&](const float* ps) {
#pragma unroll
    for (int u = 0; u < TQ; ++u) {
      const float* s0 = ps + poff[u];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const wn_f32x2 lo = *reinterpret_cast<const wn_f32x2*>(s0 + i * q.PWp);
        const wn_f32x2 hi = *reinterpret_cast<const wn_f32x2*>(s0 + i * q.PWp + 2);
        d[u][i][0] = lo[0]; d[u][i][1] = lo[1]; d[u][i][2] = hi[0]; d[u][i][3] = hi[1];
      }
    }
  };
  auto tr_rows = [&](int u) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      w[u][0][j] = d[u][0][j] - d[u][2][j];
      w[u][1][j] = d[u][1][j] + d[u][2][j];
      w[u][2][j] = d[u][2][j] - d[u][1][j];
      w[u][3][j] = d[u][1][j] - d[u][3][j];
    }
  };
  auto tr_cols = [&](float* vs, int u, int i0) {     // rows i0, i0 + 1 of the result
    if (TW * 128 >= 256 || tid + 256 * u < TW * 128) {
      float* o = vs + tid + 256 * u;
#pragma unroll
      for (int i = i0; i < i0 + 2; ++i) {
        o[(4 * i + 0) * (TW * 128)] = w[u][i][0] - w[u][i][2];
        o[(4 * i + 1) * (TW * 128)] = w[u][i][1] + w[u][i][2];
        o[(4 * i + 2) * (TW * 128)] = w[u][i][2] - w[u][i][1];
        o[(4 * i + 3) * (TW * 128)] = w[u][i][1] - w[u][i][3];
      }
    }
  };

  wn_f32x16 acc[16];
#pragma unroll
  for (int x = 0; x < 16; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;

  // ---- prologue: patch[0]; (U[0], patch[1]); (U[1], patch[2]) -----------------------------------------
  issue_p(0, 0);
  issue_u(0, 0); issue_p(1, 1);
  issue_u(1, 1); issue_p(2, 2);
  scf_wait_vmcnt_imm<GRP>();           // all but the last group have landed
  __syncthreads();
  tr_load(Ps);
#pragma unroll
  for (int u = 0; u < TQ; ++u) { tr_rows(u); tr_cols(Vs, u, 0); tr_cols(Vs, u, 2); }
  __syncthreads();

  const float* ua = Us + cw * 128 + lane * 2;
  const float* vb = Vs + tw * 128 + lane * 2;
  int us = 0, ps1 = 1, vs0 = 0;        // ring slots of U[c], patch[c + 1], V[c]
  for (int c = 0; c < q.nchunk; ++c) {
    {
      const int un = us == 0 ? 2 : us - 1;          // (c + 2) % 3
      const int pn = ps1 == 0 ? 2 : ps1 - 1;        // (c + 3) % 3
      if (!WN_LAB(2)) { issue_u(c + 2, un); issue_p(c + 3, pn); }
    }
    wn_f32x2 a[16], b[16];
    const float* uc = ua + us * USLOT;
    const float* vc = vb + vs0 * VSLOT;
    float* vnext = Vs + (vs0 ^ 1) * VSLOT;
#pragma unroll
    for (int x = 0; x < 16; ++x) {
      a[x] = *reinterpret_cast<const wn_f32x2*>(uc + x * (CW * 128));
      b[x] = *reinterpret_cast<const wn_f32x2*>(vc + x * (TW * 128));
    }
    if (!WN_LAB(1)) tr_load(Ps + ps1 * PSLOT);       // patch[c + 1]; past the last chunk: zeros, result unused
    __builtin_amdgcn_sched_barrier(0);
#define WN_MFMA(X0, X1, S)                                                                            \
    if (!WN_LAB(0)) _Pragma("unroll") for (int x = X0; x < X1; ++x)                                   \
      acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[x][S], b[x][S], acc[x], 0, 0, 0);               \
    __builtin_amdgcn_sched_barrier(0);
    WN_MFMA(0, 4, 0)
    if (!WN_LAB(1)) tr_rows(0);
    __builtin_amdgcn_sched_barrier(0);
    WN_MFMA(4, 8, 0)
    if (!WN_LAB(1)) tr_cols(vnext, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    WN_MFMA(8, 12, 0)
    if (!WN_LAB(1)) tr_cols(vnext, 0, 2);
    __builtin_amdgcn_sched_barrier(0);
    WN_MFMA(12, 16, 0)
    if (TQ > 1 && !WN_LAB(1)) { tr_rows(TQ - 1); __builtin_amdgcn_sched_barrier(0); }
    WN_MFMA(0, 4, 1)
    if (TQ > 1 && !WN_LAB(1)) { tr_cols(vnext, TQ - 1, 0); __builtin_amdgcn_sched_barrier(0); }
    WN_MFMA(4, 8, 1)
    if (TQ > 1 && !WN_LAB(1)) { tr_cols(vnext, TQ - 1, 2); __builtin_amdgcn_sched_barrier(0); }
    WN_MFMA(8, 16, 1)
#undef WN_MFMA
    scf_wait_vmcnt_imm<GRP>();
    if (!WN_LAB(4)) __syncthreads();
    us = us == 2 ? 0 : us + 1;
    ps1 = ps1 == 2 ? 0 : ps1 + 1;
    vs0 ^= 1;
  }
  scf_wait_vmcnt_imm<0>();             // the zero-filled groups past the end

  // ---- output transform + affine epilogue ------------------------------------------------------------
  const ConvEpi e = scf_conv_epi(p, n);
  const int ty = tw * TYW + (l32 >> q.txl), tx = l32 & (TXW - 1);
  const int oy = y0 + 2 * ty, ox = x0 + 2 * tx;
  if (ox >= p.Wo || oy >= p.Ho || WN_LAB(3)) return;
  const bool row1 = oy + 1 < p.Ho;
  const bool relu = p.act == SCF_ACT_RELU;
  const int cb = (f0 + cw) * 32 + 4 * half;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = cb + 8 * (r >> 2) + (r & 3);
    if (co >= p.Cout) continue;
    float t0[4], t1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t0[j] = (acc[j][r] + acc[4 + j][r]) + acc[8 + j][r];
      t1[j] = (acc[4 + j][r] - acc[8 + j][r]) - acc[12 + j][r];
    }
    float y[2][2];
    y[0][0] = (t0[0] + t0[1]) + t0[2];
    y[0][1] = (t0[1] - t0[2]) - t0[3];
    y[1][0] = (t1[0] + t1[1]) + t1[2];
    y[1][1] = (t1[1] - t1[2]) - t1[3];
    const float bv = p.bias ? p.bias[co] : 0.f;
    const float sc = p.scale ? p.scale[co] : 1.f, sh = p.scale ? p.shift[co] : 0.f;
    const int off = co * e.HWo + oy * p.Wo + ox;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i == 1 && !row1) break;
      wn_f32x2 v = {y[i][0] + bv, y[i][1] + bv};
      if (p.scale) { v[0] = v[0] * sc + sh; v[1] = v[1] * sc + sh; }
      if (e.res) {
        const wn_f32x2 rr = *reinterpret_cast<const wn_f32x2*>(e.res + off + i * p.Wo);
        v[0] += rr[0]; v[1] += rr[1];
      }
      if (relu) { v[0] = v[0] > 0.f ? v[0] : 0.f; v[1] = v[1] > 0.f ? v[1] : 0.f; }
      *reinterpret_cast<wn_f32x2*>(e.out + off + i * p.Wo) = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Host side: packing and launch
// ---------------------------------------------------------------------------------------------------
extern "C" int64_t scf_pack_conv_weight_wino_size(int32_t cout, int32_t cin) {
  if (cout <= 0 || cin <= 0) return 0;
  const int64_t F = (cout + 31) / 32, nchunk = (cin + WN_KC - 1) / WN_KC;
  return nchunk * 16 * F * 128;
}

extern "C" int scf_pack_conv_weight_wino(const float* w, int32_t cout, int32_t cin, float* out) {
  if (!w || !out || cout <= 0 || cin <= 0) return SCF_EINVAL;
  const int F = (cout + 31) / 32;
  memset(out, 0, sizeof(float) * (size_t)scf_pack_conv_weight_wino_size(cout, cin));
  static const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci) {
      const float* g = w + ((size_t)co * cin + ci) * 9;
      double t[4][3];
      for (int i = 0; i < 4; ++i)
        for (int b = 0; b < 3; ++b) t[i][b] = G[i][0] * g[b] + G[i][1] * g[3 + b] + G[i][2] * g[6 + b];
      const int chunk = ci / WN_KC, cl = ci % WN_KC, s = cl >> 1, kh = cl & 1;
      const int frag = co / 32, m = co % 32;
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
          const double u = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
          out[(((size_t)chunk * 16 + (4 * i + j)) * F + frag) * 128 + kh * 64 + m * 2 + s] = (float)u;
        }
    }
  return SCF_OK;
}

static int wino_lds_attr(const void* fn, size_t bytes) {
  return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess ? SCF_OK : SCF_ELAUNCH;
}

// Tile selection + launch; SCF_EUNSUPPORTED -> the caller goes on to the direct kernels.
// info (optional): {CW, TW, blocks, LDS bytes}.
int scf_conv_wino_dispatch(ConvK k, const float* wu, int N, bool dry_run, int* info, hipStream_t st) {
  if (!wu || k.KH != 3 || k.KW != 3 || k.stride != 1 || k.pad_h != 1 || k.pad_w != 1) return SCF_EUNSUPPORTED;
  if (k.w_ns != 0 || k.out_tile || k.mode != SCF_CONV_PLAIN || k.out_div != 1.0f || k.act_split > 0) return SCF_EUNSUPPORTED;
  if (k.act != SCF_ACT_NONE && k.act != SCF_ACT_RELU) return SCF_EUNSUPPORTED;
  if ((k.Wo & 1) || (k.out_ns & 1) || ((uintptr_t)k.out & 7) || (k.res && ((k.res_ns & 1) || ((uintptr_t)k.res & 7)))) return SCF_EUNSUPPORTED;
  if (k.in1 && (k.C0 % WN_KC) != 0) return SCF_EUNSUPPORTED;
  if (((uintptr_t)wu & 15) || (long long)WN_KC * k.H * k.W * 4 >= 0x7fffffffLL) return SCF_EUNSUPPORTED;
  const int F = (k.Cout + 31) / 32;
  int CW, TW;
  if (F % 2 == 0) { CW = 2; TW = 2; } else { CW = 1; TW = 4; }
  // tiles per row of a wave's group: 16 unless a narrower group wastes clearly fewer columns
  const int tcols = (k.Wo + 1) / 2;
  int txl = 4;
  {
    int best = (tcols + 15) / 16 * 16;
    if (tcols < 16) { txl = 0; while ((1 << txl) < tcols) ++txl; if (txl < 2) txl = 2; best = 0; }
    for (int l = 3; l >= 2 && best; --l) {
      const int wpad = (tcols + (1 << l) - 1) >> l << l;
      if (wpad * 10 <= best * 9) { best = wpad; txl = l; }
    }
  }
  const int TXW = 1 << txl, TYW = 32 >> txl;
  WinoK q;
  q.wu = wu; q.F = F; q.txl = txl;
  q.PH = 2 * TW * TYW + 2; q.PW = 2 * TXW + 2; q.PWp = q.PW;
  q.PPL = q.PH * q.PWp;
  q.PPL += ((16 - (q.PPL & 31)) + 32) & 31;            // plane stride = 16 mod 32 floats: the two channel planes a half-wave reads hit disjoint banks
  const int npi = WN_NPI(TW);
  if (WN_KC * q.PPL > npi * 256) return SCF_EUNSUPPORTED;
  q.nchunk = (k.Cin + WN_KC - 1) / WN_KC;
  q.sx = (k.Wo + 2 * TXW - 1) / (2 * TXW);
  q.sy = (k.Ho + 2 * TW * TYW - 1) / (2 * TW * TYW);
  q.mblocks = F / CW;
  const long long nblk = (long long)N * q.sx * q.sy * q.mblocks;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return SCF_EUNSUPPORTED;
  const size_t ldsb = (size_t)(3 * CW * 2048 + 2 * TW * 2048 + 3 * npi * 256) * sizeof(float);
  if (ldsb > 160 * 1024) return SCF_EUNSUPPORTED;
#ifdef SCF_WINO_LAB
  q.lab = getenv("SCF_WINO_LAB") ? atoi(getenv("SCF_WINO_LAB")) : 0;
#endif
  if (info) { info[0] = CW; info[1] = TW; info[2] = (int)nblk; info[3] = (int)ldsb; }
  if (dry_run) return SCF_OK;
  static bool raised[64][2] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SCF_ELAUNCH;
  const int cfg = CW == 2 ? 0 : 1;
  if (!raised[dev][cfg]) {
    const int rc = cfg == 0 ? wino_lds_attr((const void*)conv_wino_kernel<2, 2>, 160 * 1024)
                            : wino_lds_attr((const void*)conv_wino_kernel<1, 4>, 160 * 1024);
    if (rc != SCF_OK) return rc;
    raised[dev][cfg] = true;
  }
  if (cfg == 0) scf_launch((conv_wino_kernel<2, 2>), dim3((unsigned)nblk), dim3(256), ldsb, st, k, q);
  else scf_launch((conv_wino_kernel<1, 4>), dim3((unsigned)nblk), dim3(256), ldsb, st, k, q);
  return scf_launch_status();
}
