/*
 * scflow_hip.h -- C ABI of libscflow_hip.so: hand-written gfx950 (MI355X) kernels for
 * the SCFlow recurrent flow/pose refinement hot path.
 *
 * Conventions (all entry points)
 *   - plain C, no C++/torch types; every pointer is a DEVICE pointer to fp32 data laid
 *     out exactly as the reference's contiguous NCHW torch tensors unless stated;
 *   - pointers are borrowed: nothing is allocated, freed or retained;
 *   - work is enqueued asynchronously on `stream` (a hipStream_t; NULL = default
 *     stream); the call returns after launch, it never synchronises;
 *   - return value: SCF_OK (0) or a negative SCF_E* code; scf_error_string() decodes;
 *   - thread-safety: calls are independent; concurrent calls on different streams OK.
 *
 * The reference is pure Python on torch (no FFI of its own).  Each entry point names the
 * reference operator (file:line under the reference checkout) whose arithmetic it
 * replaces; INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 */
#ifndef SCFLOW_HIP_H
#define SCFLOW_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* scf_stream_t; /* bit-compatible with hipStream_t */

enum {
  SCF_OK = 0,
  SCF_EINVAL = -1,      /* bad argument (null pointer, non-positive size, ...) */
  SCF_EUNSUPPORTED = -2,/* shape/config outside what the kernels implement */
  SCF_ELAUNCH = -3,     /* HIP reported a launch error */
  SCF_ENODEVICE = -4    /* no gfx950 device visible */
};

/* Pyramid levels an entry point accepts.  Level l of an h x w map is (h >> l) x (w >> l) and the
 * volume holds (h w)^2 floats per pair, so 12 levels already means a pyramid of more than 288 GB:
 * the bound never binds on this device. */
#define SCF_MAX_LEVELS 12

/* activation codes used by scf_conv2d / scf_linear */
enum { SCF_ACT_NONE = 0, SCF_ACT_RELU = 1, SCF_ACT_SIGMOID = 2, SCF_ACT_TANH = 3 };

/* fused epilogues of scf_conv2d */
enum {
  SCF_CONV_PLAIN = 0,
  SCF_CONV_GRU_ZR = 1, /* rows [0,Cout/2): z=sigmoid -> out; rows [Cout/2,Cout): r=sigmoid, aux = r*h */
  SCF_CONV_GRU_Q = 2   /* q=tanh; out = (1-z)*h + z*q                                   */
};

/* ABI version of this header: SCF_ABI_MAJOR changes whenever a struct layout or a signature changes
 * (scf_conv_desc has grown twice), the minor part when entry points are added.  scf_version()
 * returns the number the LIBRARY was built with: a C caller compares scf_version() / 100 with
 * SCF_ABI_MAJOR before its first call; structs additionally carry no size field,
 * so a mismatch must be refused, not worked around. */
#define SCF_ABI_MAJOR 5
#define SCF_VERSION (SCF_ABI_MAJOR * 100 + 2)   /* .1: label_mode is a bit set (SCF_POSE_*); .2: scf_conv2d_pair, overlap_* = 2 */
int scf_version(void);
const char* scf_error_string(int code);
/* number of HIP devices visible (>=0) or SCF_ENODEVICE */
int scf_device_count(void);

/* ---------------------------------------------------------------------------------
 * Correlation volume + pyramid.            replaces CorrelationPyramid.forward
 *                                          models/decoder/raft_decoder.py:35-58
 * level0[n, i, j] = sum_c feat1[n,c,i] * feat2[n,c,j] / sqrt(C)   (i, j in [0, h*w))
 * level(l+1) = 2x2/stride-2 average pool of level l over the target (j) dims.
 * levels[l] : (N*h*w, 1, h>>l, w>>l) contiguous, l < L (host array of device ptrs).
 * MFMA (v_mfma_f32_32x32x2_f32) contraction; exact fp32 fma chain over c.
 * --------------------------------------------------------------------------------- */
int scf_corr_build(const float* feat1, const float* feat2, float* const* levels,
                   int N, int C, int h, int w, int L, scf_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Multi-scale correlation lookup.          replaces CorrLookup.forward
 *                                          models/utils/corr_lookup.py:102-136
 * out[n, 81*l + 9*a + b, y, x] = bilinear(level l of query (n,y,x)) sampled at
 *   ((x + flow[n,0,y,x]) / 2^l + a - r, (y + flow[n,1,y,x]) / 2^l + b - r),
 * zero padding, align_corners=True semantics; out is (N, L*(2r+1)^2, h, w).
 * HBM-bound gather: 2904 B/query algorithmic traffic at r=4, L=4.
 * --------------------------------------------------------------------------------- */
int scf_corr_lookup(const float* const* levels, const float* flow, float* out,
                    int N, int h, int w, int r, int L, scf_stream_t stream);

/* The same pair with a lookup-friendly layout for the levels the caller names: bit l of
 * `tiled_levels` set = every query's level-l map is stored in 8(x) x 4(y)-float tiles of one
 * 128-byte line each -- tile-major, row-major inside a tile, the map padded to a multiple of 4 rows
 * and 8 columns (padding floats are never read and may hold anything).  A (2r+2)^2 lookup window
 * then touches ~(1 + (2r+1)/8)(1 + (2r+1)/4) = 6.9 lines at r = 4 instead of one or two lines per
 * window row of a row-major map.  Level 0 is written by the correlation GEMM whose fragments are
 * whole tiles: bit 0 needs w % 8 == 0 and h % 4 == 0 (SCF_EUNSUPPORTED otherwise).
 *   scf_corr_level_floats      floats per query of level `level` in the given layout
 *                              (levels[l] holds N*h*w maps of that many floats)
 *   scf_corr_preferred_layout  the mask the lookup kernel is fastest with (tiles for every level
 *                              whose rows are at least 24 floats long and that does not fit the
 *                              lookup window whole)
 * tiled_levels = 0 is exactly scf_corr_build / scf_corr_lookup.  Both calls of a pair must be given
 * the same mask.
 * scf_corr_lookup[_ex] accepts ANY radius >= 1, level count <= SCF_MAX_LEVELS and map size
 * (CorrLookup's constructor arguments, corr_lookup.py:91-102): r <= 4 with maps of at most 32767
 * floats run on the LDS-DMA kernel, everything else on a plain gather kernel with the same
 * arithmetic.  A query whose flow is NaN / inf yields NaN in all its taps, as torch's
 * grid_sample does on the reference's CPU path. */
int64_t scf_corr_level_floats(int h, int w, int level, int tiled);
unsigned scf_corr_preferred_layout(int h, int w, int r, int L);
int scf_corr_build_ex(const float* feat1, const float* feat2, float* const* levels, int N, int C,
                      int h, int w, int L, unsigned tiled_levels, scf_stream_t stream);
int scf_corr_lookup_ex(const float* const* levels, const float* flow, float* out, int N, int h,
                       int w, int r, int L, unsigned tiled_levels, scf_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Direct convolution as implicit GEMM on MFMA with fused epilogue.
 * replaces every torch conv2d (+bias +BN(eval) +residual +activation +GRU gating) on
 * the path: raft_encoder.py:286-314, resnet.py:67-94, raft_decoder.py:152-166,
 * :235-253, :292-294, scflow_decoder.py:216-217, pose_head.py:148-160.
 *
 * Input = channel-concatenation of up to two NCHW segments (avoids torch.cat).
 * Weights are pre-packed by scf_pack_conv_weight_size/scf_pack layout rules:
 *   wp[((chunk*T + t)*KC + cl) * Mld + co],  chunk = ci / KC, cl = ci % KC, t = ky*KW+kx,
 *   rows for ci >= Cin are zero, Mld = row stride (>= Cout, multiple of 4).
 * Epilogue order: v = acc / out_div (+bias) -> v*scale+shift -> +res -> act -> mode.
 * --------------------------------------------------------------------------------- */
typedef struct scf_conv_desc {
  const float* in0; const float* in1;   /* input segments (in1 may be NULL)            */
  int32_t C0, C1;                       /* channels of each segment                    */
  int64_t in0_nstride, in1_nstride;     /* floats between consecutive samples          */
  int32_t N, H, W;                      /* batch, input height/width                   */
  const float* wp;                      /* packed weights                              */
  int64_t w_nstride;                    /* 0 = shared weights; else per-sample stride  */
  int32_t Mld;                          /* packed row stride                           */
  int32_t Cout;
  int32_t KH, KW, stride, pad_h, pad_w;
  int32_t KC;                           /* channel chunk used at packing time (2, 8, 32) */
  float* out; int64_t out_nstride;
  const float* bias;                    /* [Cout] or NULL                              */
  const float* scale; const float* shift; /* [Cout] each or both NULL (BN eval)        */
  const float* res; int64_t res_nstride;  /* residual added before act (GRU modes: before
                                             the gate's sigmoid / tanh), or NULL       */
  float out_div;                        /* accumulator divided by this (1 = off)       */
  int32_t act, act2, act_split;         /* act for co < act_split, act2 otherwise;
                                           act_split <= 0 -> act everywhere            */
  int32_t mode;                         /* SCF_CONV_*                                  */
  const float* gru_h; int64_t gru_h_nstride; /* hidden state (ZR, Q)                   */
  float* gru_aux; int64_t gru_aux_nstride;   /* ZR: r*h destination                    */
  const float* gru_z; int64_t gru_z_nstride; /* Q: z                                   */
  const void* wp_f16;                   /* optional split-fp16 packing of the same weights:
                                           [(chunk16*T + tap)*2 + k8][plane hi|lo][Mld][8] halves;
                                           non-NULL selects the 3xMFMA fp16 kernel where the
                                           shape fits (fp32-class accuracy)   */
  const float* wp_a4;                   /* optional second packing of the same weights for the
                                           LDS-DMA kernel (stride 1): [chunk][tap][g][h][Mld_a4][4]
                                           floats, channel = chunk*8G + 8g + 2s + h at float s      */
  int32_t a4_groups;                    /* G in {1,2,4}: 8G channels per staged chunk              */
  int32_t a4_mld;                       /* Cout rounded up to 32                                    */
  const float* wp_a4s;                  /* optional: the same a4 packing with MORE channels per chunk
                                           (a4s_groups = 4 for 1x1 / 1x5 / 5x1, 2 for 3x3), used on
                                           SMALL grids (batch 1): there one block runs per CU, a chunk's
                                           fixed cost (barrier, staging issue) dominates 8-channel chunks,
                                           and LDS is free for a deep ring of bigger ones                */
  int32_t a4s_groups;
  const float* wp_thin;                 /* optional third packing for Cout <= 4 layers (vector-ALU
                                           kernel): [Cin][KH*KW][CO] floats, CO = 1, 2 or 4 (Cout
                                           rounded up), zero padded                                */
  int32_t out_tile8x4;                  /* 1: store every output plane in 8(x) x 4(y)-float tiles
                                           of 128 B (tile-major, row-major inside) instead of
                                           row-major; needs Wo % 8 == 0 and Ho % 4 == 0          */
  const float* wp_taps;                 /* optional packing for thin INPUTS (Cin <= 4: the 7x7 stems, the
                                           2 -> 128 7x7 and 1 -> 64 3x3 first layers): [Kp][Mld] floats,
                                           row k = ci * KH*KW + t, Kp = Cin*KH*KW rounded up to a multiple of
                                           8, zero padded; selects the kernel that contracts over taps x channels
                                           as one dense K dimension                                  */
  const float* wp_a4t;                  /* optional: the a4 packing with a4t_groups = 4 (32-channel chunks) for
                                           3x3 layers, used on TINY grids (no more K-split blocks than CUs:
                                           every block is alone on its CU, a launch is a chain of one memory
                                           round trip per chunk, so half as many chunks is half the chain)    */
  int32_t a4t_groups;
  const float* wp_wino1d;               /* optional: G g of a 1x5 / 5x1 stride-1 'same' layer (scf_pack_conv_weight_wino1d);
                                           selects the one-dimensional Winograd F(2, 5) fp32 kernel (every epilogue
                                           kind incl. the GRU gates) on grids of >= CUs / 2 blocks (128 on the
                                           MI355X); same contract as wp_wino */
  const float* wp_wino;                 /* optional: G g G^T of a 3x3 / stride-1 / pad-1 layer
                                           (scf_pack_conv_weight_wino); selects the Winograd F(2x2, 3x3)
                                           fp32 kernel for plain / affine epilogues (bias, BN, residual,
                                           ReLU) on grids of >= CUs / 2 blocks (smaller grids: the direct kernels).
                                           Same fp32 arithmetic, re-associated sums: results differ from the direct
                                           kernels by a few ulp of sum |w||x|.  Kernel selection therefore depends on
                                           N and on the device: leave both Winograd packings NULL for results that
                                           do not */
  const float* wp_wino1d4;              /* optional: the F(4, 5) packing of the same 1x5 / 5x1 layer (scf_pack_conv_weight_wino1d4):
                                           four outputs per 8 multiplies; taken before wp_wino1d on grids of more than CUs / 2 of its
                                           blocks (64 channels x 256 pixels); same contract as wp_wino */
  int32_t k_slices;                     /* 0 / 1: off.  S > 1 (r5): the contraction over the input channels is split across
                                           S groups of BLOCKS: slice s contracts its share of the channel chunks and stores the
                                           raw partial sums at out + s * out_slice_stride; the CONSUMER adds the S partial
                                           tensors in slice order (scf_group_norm_relu_parts, scf_fc_splitk's x_parts) -- no
                                           atomics, a fixed summation order.  Small grids only (a block there is a chain of one
                                           memory round trip per chunk: S slices = 1 / S of the chain on S times the blocks).
                                           Plain epilogue required: no bias / scale / res / act / GRU mode / out_div / tiled
                                           output; LDS-DMA kernel only (wp_a4 or wp_a4s), else SCF_EUNSUPPORTED */
  int64_t out_slice_stride;             /* floats between consecutive partial tensors (>= N * out_nstride) */
} scf_conv_desc;

int scf_conv2d(const scf_conv_desc* desc, scf_stream_t stream);

/* Optional scratch for small grids (r5): register `floats` floats of device memory for launches on `stream` (borrowed until
 * replaced or cleared with ptr = NULL, floats = 0; one workspace per stream -- concurrent streams must not share one).  A
 * convolution whose every block would be alone on its CU with a chain of >= 4 staged chunks (batch 1 ... 4: one memory
 * round trip per chunk, whatever it computes) is then split into up to 4 K slices that write partial tensors to the
 * workspace, followed by one combine launch that adds them in slice order and applies the descriptor's whole epilogue (any
 * kind, GRU gates included).  Same result as the single launch up to the re-association of the partial sums;
 * deterministic; applies to every entry point that launches convolutions on that stream (scf_sepconv_gru*,
 * scf_scflow_iteration).  Needs N * Cout * Ho * Wo * slices floats; launches that need more stay unsliced.
 * The registry is keyed by the raw stream handle: clear the entry (ptr = NULL) BEFORE destroying the stream, a later
 * stream that gets the same handle would inherit it. */
int scf_conv_workspace(scf_stream_t stream, float* ptr, int64_t floats);

/* r6: two INDEPENDENT convolutions (no data flows between them) as ONE launch where both fall to the same small-grid kernel
 * instantiation (the K-split LDS-DMA tile, the thin-input kernel): blocks [0, nA) run a, the rest b.  Sub-chip grids
 * (batch 1-4: a layer is 8-64 blocks on 256 CUs) then run side by side without a second stream -- on this runtime a hipGraph
 * replay pays ~1.2 us per node once the graph holds a parallel branch.  Any other pair: the two launches one after the
 * other.  Results are those of scf_conv2d(a) and scf_conv2d(b), bit for bit, in both cases. */
int scf_conv2d_pair(const scf_conv_desc* a, const scf_conv_desc* b, scf_stream_t stream);

/* Host-side weight packers (plain CPU loops, run once per checkpoint): w is a HOST pointer to a
 * contiguous (Cout, Cin, KH, KW) fp32 tensor -- a torch Conv2d weight as stored in the reference's
 * state_dict -- and out a HOST buffer of scf_pack_conv_weight*_size() floats; copy the result
 * to the device and pass it as scf_conv_desc.wp (+ KC, Mld = Cout rounded up to 32) or
 * scf_conv_desc.wp_a4 (+ a4_groups, a4_mld = Cout rounded up to 32).
 *   KC packing : out[((chunk*T + t)*KC + cl)*Mld + co] = w[co][chunk*KC + cl][t], zeros elsewhere
 *   a4 packing : out[((((chunk*T + t)*G + g)*2 + h)*Mld + co)*4 + s] = w[co][chunk*8G + 8g + 2s + h][t] */
int64_t scf_pack_conv_weight_size(int Cout, int Cin, int KH, int KW, int KC);
int scf_pack_conv_weight(const float* w, int Cout, int Cin, int KH, int KW, int KC, float* out);
int64_t scf_pack_conv_weight_a4_size(int Cout, int Cin, int KH, int KW, int groups);
int scf_pack_conv_weight_a4(const float* w, int Cout, int Cin, int KH, int KW, int groups, float* out);
/*   taps packing (scf_conv_desc.wp_taps, Cin <= 4): out[(ci*KH*KW + t)*Mld + co] = w[co][ci][t], Mld = Cout
 *   rounded up to 32, rows rounded up to a multiple of 8, zeros elsewhere */
int64_t scf_pack_conv_weight_taps_size(int Cout, int Cin, int KH, int KW);
int scf_pack_conv_weight_taps(const float* w, int Cout, int Cin, int KH, int KW, float* out);
/*   Winograd packing (scf_conv_desc.wp_wino, 3x3 only): U[i][j] = (G g G^T)[i][j] per (co, ci), computed
 *   in double and rounded once;  out[((chunk*F + co/32)*16 + 4*pi(i) + j)*128 + (cl & 1)*64 + (co % 32)*2 + (cl >> 1)],
 *   pi = (0, 1, 3, 2): the rows of the transform domain are stored in the order 0, 1, 3, 2,
 *   with ci = 4*chunk + cl, F = Cout rounded up to 32, / 32; zeros elsewhere */
int64_t scf_pack_conv_weight_wino1d_size(int32_t Cout, int32_t Cin);
int scf_pack_conv_weight_wino1d(const float* w, int32_t Cout, int32_t Cin, float* out);   /* w: (Cout, Cin, 5) taps;
     out[((chunk*F + co/32)*6 + i)*256 + (cl & 1)*128 + (co % 32)*4 + (cl >> 1)] = (G g)[i], ci = 8*chunk + cl,
     G = the 6 x 5 matrix of the points 0, 1, -1, 2, -2, infinity */
int64_t scf_pack_conv_weight_wino1d4_size(int32_t Cout, int32_t Cin);
int scf_pack_conv_weight_wino1d4(const float* w, int32_t Cout, int32_t Cin, float* out);  /* w: (Cout, Cin, 5) taps;
     out[((chunk*F + co/32)*8 + i)*128 + (cl & 1)*64 + (co % 32)*2 + (cl >> 1)] = (G g)[i], ci = 4*chunk + cl,
     G = the 8 x 5 matrix of the points 0, 1, -1, 2, -2, 1/2, -1/2, infinity, row 0 negated */
int64_t scf_pack_conv_weight_wino_size(int32_t Cout, int32_t Cin);
int scf_pack_conv_weight_wino(const float* w, int32_t Cout, int32_t Cin, float* out);

/* ---------------------------------------------------------------------------------
 * Convolutional GRU update, whole cell.     replaces ConvGRU.forward
 *                                           models/decoder/raft_decoder.py:235-253
 * hx = [h (Ch channels) | x (Cx channels)] of one sample-strided NCHW buffer; for every pass
 * (SeqConv: a (1,5) then a (5,1) convolution triple, :180-181; Conv: one 3x3 triple)
 *     z = sigmoid(conv_z(hx)),  r = sigmoid(conv_r(hx)),  q = tanh(conv_q([r*h | x])),
 *     h <- (1 - z)*h + z*q                                              (in place in hx)
 * as two launches: one 2*Ch-row convolution whose epilogue emits z and r*h, one Ch-row
 * convolution whose epilogue applies tanh and the state update.  z and rh are caller-provided
 * dense (N, Ch, H, W) scratch buffers.
 * Weights per pass, DEVICE pointers in the packings above:
 *   wp_zr : KC = 8 packing of the (2*Ch, Ch+Cx, KH, KW) tensor cat([conv_z.weight, conv_r.weight], 0)
 *   wp_q  : KC = 8 packing of conv_q.weight (Ch, Ch+Cx, KH, KW);  bias_zr [2*Ch], bias_q [Ch]
 *   wp_*_a4 (+ a4_groups: 2 for (1,5)/(5,1), 1 for 3x3) select the LDS-DMA kernel (optional, faster)
 *   wp_*_f16 select the split-fp16 3xMFMA kernel (optional, see scf_conv_desc.wp_f16)
 *   wp_*_a4s (+ a4s_groups) : small-grid a4 packings (see scf_conv_desc.wp_a4s), optional
 *   wp_*_k32 : KC = 32 packings of the same tensors (optional): used instead of the KC = 8 ones
 *              when the grid is so small (batch 1) that the register-staged kernel with its
 *              smallest tile runs and an 8-channel chunk is too short to hide its prefetch
 * --------------------------------------------------------------------------------- */
typedef struct scf_gru_pass {
  int32_t KH, KW, pad_h, pad_w;
  const float* wp_zr; const float* bias_zr;
  const float* wp_q; const float* bias_q;
  const float* wp_zr_a4; const float* wp_q_a4; int32_t a4_groups;
  const void* wp_zr_f16; const void* wp_q_f16;
  const float* wp_zr_k32; const float* wp_q_k32;
  const float* wp_zr_a4s; const float* wp_q_a4s; int32_t a4s_groups;
  const float* wp_zr_a4t; const float* wp_q_a4t; int32_t a4t_groups;   /* 3x3 passes: tiny-grid packings (scf_conv_desc.wp_a4t), optional */
  const float* wp_zr_wino1d; const float* wp_q_wino1d;   /* 1x5 / 5x1 passes: F(2, 5) packings (scf_conv_desc.wp_wino1d), optional */
  const float* wp_zr_wino1d4; const float* wp_q_wino1d4; /* 1x5 / 5x1 passes: F(4, 5) packings (scf_conv_desc.wp_wino1d4), optional */
} scf_gru_pass;

int scf_sepconv_gru(float* hx, int64_t hx_nstride, int N, int Ch, int Cx, int H, int W,
                    const scf_gru_pass* passes, int npass, float* z, float* rh,
                    scf_stream_t stream);

/* The same update (ConvGRU.forward, raft_decoder.py:235-253) with the iteration-invariant part
 * of x hoisted out of the refinement loop.  hx = [h (Ch) | c (Cc) | x' (Cx)], where c -- the
 * context features, scflow_decoder.py:189-190 / raft_decoder.py:430 -- is the same in every
 * iteration: conv([h | c | x']) = conv([h | x']) + conv_c(c).  The caller computes, once per
 * pair and per pass i, ctx[i] = conv_c(c) + bias as a plain scf_conv2d over c whose weight is the
 * c columns of conv_z | conv_r | conv_q stacked into 3 Ch output rows: (N, 3 Ch, H, W) with
 * sample stride ctx_nstride, channels [0, 2 Ch) for z | r and [2 Ch, 3 Ch) for q.  `passes` then
 * holds packings over the remaining Ch + Cx input channels, with bias_zr = bias_q = NULL (folded
 * into ctx).  Results equal scf_sepconv_gru's up to fp32 summation order. */
int scf_sepconv_gru_ctx(float* hx, int64_t hx_nstride, int N, int Ch, int Cc, int Cx, int H, int W,
                        const scf_gru_pass* passes, int npass, const float* const* ctx,
                        int64_t ctx_nstride, float* z, float* rh, scf_stream_t stream);

/* ---------------------------------------------------------------------------------
 * One whole refinement iteration.           replaces the loop body of SCFlowDecoder.forward
 *                                           models/decoder/scflow_decoder.py:196-243
 * The launch sequence of an iteration -- 1/8 flow, lookup, motion encoder, SepConvGRU, flow / mask
 * heads, delta-flow / mask encoders, full-resolution outputs, pose head, pose update, pose-induced
 * flow: ~33 launches -- behind ONE call, so that a caller without hipGraph capture is bound by the
 * launch API and not by its interpreter.  Nothing new is computed: every step is one of the
 * operator entry points of this header, issued in a fixed order (results are bit-identical to
 * issuing them one by one).  The caller fills the struct ONCE per decoder pass (all scratch buffers
 * and convolution descriptors, weights in the packings of scf_conv_desc) and changes only the
 * "per iteration" pointers between calls.  struct_size = sizeof(scf_scflow_iter) is checked.
 * overlap_* (small batches): that branch is issued on side_stream between event fork / join points,
 * beside the main stream's work; the call itself never synchronises (hipGraph-capturable).
 * --------------------------------------------------------------------------------- */
typedef struct scf_iter_gn {          /* GroupNorm(G, eps, affine) + ReLU after a pose-head convolution */
  const float* gamma; const float* beta; float* out;
  int32_t C, HW, G; float eps;
} scf_iter_gn;

typedef struct scf_scflow_iter {
  int32_t struct_size;
  int32_t N, H, W, h, w;                     /* batch; full-resolution and 1/8-resolution sizes        */
  /* correlation pyramid (scf_corr_build_ex) and lookup */
  int32_t L, radius; uint32_t tiled_levels; int32_t corr_channels;   /* L * (2 radius + 1)^2            */
  const float* levels[SCF_MAX_LEVELS];
  float* flow_lr;                            /* (N, 2, h, w) scratch                                    */
  float* corr;                               /* (N, corr_channels, h, w) scratch                        */
  /* decoder switches mask_flow / mask_corr (:199-205) */
  int32_t mask_flow, mask_corr;
  const float* mask_prev;                    /* (N, 1, h, w): previous iteration's mask (ones before the first) */
  float* flow_masked;                        /* (N, 2, h, w) scratch, mask_flow only                    */
  /* motion encoder (in / out pointers set; flow0.in0 is replaced by the call) */
  scf_conv_desc flow0, flow1, corr0, corr1, outn;
  float* flow_copy_dst;                      /* hx[:, Ch + Cc + 126 ...]: the 2 flow channels of x      */
  /* SepConvGRU: hx = [h (Ch) | context (Cc) | motion features (Cx)] */
  float* hx; int64_t hx_nstride; int32_t Ch, Cc, Cx, npass;
  scf_gru_pass gru[2];
  const float* ctx[2]; int64_t ctx_nstride;  /* hoisted context terms (scf_sepconv_gru_ctx) or NULLs    */
  float* z; float* rh;
  /* heads and their encoders */
  scf_conv_desc heads, fpred, mpred, menc0, menc1, denc0, denc1;
  /* pose head: 3 x (conv -> GroupNorm + ReLU), 2 FC layers, rotation / translation heads */
  scf_conv_desc pose[3];
  scf_iter_gn gn[3];
  const float* fc1_w; const float* fc1_b; float* fc1_out; int32_t fc1_K, fc1_O;
  const float* fc2_w; const float* fc2_b; float* fc2_out; int32_t fc2_O;
  /* fc_fused != 0: the tail runs as three scf_fc_splitk launches -- the third GroupNorm + ReLU is applied by
   * fc1's operand load (gn[2].out unused), fc1_out / fc2_out are (fc1_slices, N, fc1_O) / (fc2_slices, N, fc2_O)
   * partial-sum buffers whose bias + ReLU the next layer's load applies.  0: scf_linear launches as before. */
  int32_t fc_fused, fc1_slices, fc2_slices;
  const float* rot_w; const float* rot_b; float* rot_all; int32_t rot_O;
  const float* trans_w; const float* trans_b; float* trans_all; int32_t trans_O;
  const int64_t* label; int32_t num_class, label_mode;
  /* pose-induced flow */
  const float* depth; const float* K; const float* R0; const float* t0; float invalid_flow_num;
  /* ---- per iteration ---- */
  const float* flow_in;                      /* (N, 2, H, W): the previous pose-induced flow (init_flow first) */
  const float* R_in; const float* t_in;
  float* flow_out; float* flow_pred; float* mask_up;       /* (N,2,H,W), (N,2,H,W), (N,1,H,W)           */
  float* R_out; float* t_out; float* d_rot; float* d_trans;
  /* ---- two-stream overlap ---- */
  scf_stream_t side_stream; int32_t overlap_flow, overlap_mask, overlap_up;
  void* lookup_timer;                        /* optional scf_timer_t (scflow_hip_prof.h) bound to the lookup launch */
} scf_scflow_iter;

int scf_scflow_iteration(const scf_scflow_iter* iter, scf_stream_t stream);

/* ---------------------------------------------------------------------------------
 * InstanceNorm2d(eps, affine=False) [+ residual] [+ ReLU] over N*C planes of HW floats.
 * replaces F.instance_norm at resnet.py:75-86 / raft_encoder.py:300-302 (norm_cfg IN).
 * out = relu?( (x - mean) * rsqrt(var_biased + eps) + res? ).  In-place allowed.
 * --------------------------------------------------------------------------------- */
int scf_instance_norm(const float* x, const float* res, float* out, int64_t planes,
                      int HW, float eps, int relu, scf_stream_t stream);

/* GroupNorm(G, eps, affine) + ReLU on (N, C, HW).     replaces pose_head.py:151-159 */
int scf_group_norm_relu(const float* x, const float* gamma, const float* beta, float* out,
                        int N, int C, int HW, int G, float eps, scf_stream_t stream);
/* the same on an input that arrives as `parts` partial tensors part_stride floats apart (a convolution launched
 * with scf_conv_desc.k_slices = parts): every element is the sum of its parts in part order */
int scf_group_norm_relu_parts(const float* x, int parts, int64_t part_stride, const float* gamma,
                              const float* beta, float* out, int N, int C, int HW, int G, float eps,
                              scf_stream_t stream);

/* y[n, o] = act(sum_k W[o,k] x[n,k] + b[o]);  W row-major (O, K). replaces nn.Linear
 * at pose_head.py:166-172, 203-206.                                                 */
int scf_linear(const float* x, const float* W, const float* b, float* y, int N, int K,
               int O, int act, scf_stream_t stream);

/* Two linear layers over the same input in one launch: y1 = act(W1 x + b1), y2 = act(W2 x + b2)
 * (rotation_pred and translation_pred, pose_head.py:203-206).  Same arithmetic as two scf_linear
 * calls.                                                                            */
int scf_linear_pair(const float* x, const float* W1, const float* b1, float* y1, int O1,
                    const float* W2, const float* b2, float* y2, int O2, int N, int K, int act,
                    scf_stream_t stream);

/* ---------------------------------------------------------------------------------
 * nn.Linear as a split-K GEMM on the matrix cores, with its element-wise neighbours folded into the operand
 * loads: the pose head's flatten -> fc1 -> ReLU -> fc2 -> ReLU -> rotation_pred | translation_pred
 * (pose_head.py:166-172, 201-211) in three launches that read every weight once per batch.
 *   input    x[n][k] = f( sum_{s < x_parts} x[s][n][k] + x_bias[k] ), f = ReLU if x_relu: the partial sums a
 *            previous scf_fc_splitk wrote (x_part_stride floats between parts; x_parts = 1, x_bias = NULL: a plain
 *            (N, K) tensor);
 *            gn_groups > 0: GroupNorm(gn_groups, gn_eps, affine) + ReLU over the K features of each sample first
 *            (group = K / gn_groups consecutive features, channel of feature k = k / gn_hw: the flattened
 *            (C, h, w) map of pose_head.py:151-159 with gn_hw = h w) -- replaces scf_group_norm_relu on that map;
 *   output   slices == 1: y[n][o] = act(sum_k W[o][k] x[n][k] + bias[o]), and the same for the optional second
 *            matrix (W2, bias2, y2, O2: rotation_pred and translation_pred read the same features);
 *            slices  > 1: y = (slices, N, O) partial sums over K / slices features each, no bias / act -- the
 *            consumer adds them in slice order (x_parts, x_bias, x_relu of the next call).
 * K % slices == 0, K / slices <= 256 and a multiple of 8 (and of the group size), 16-byte aligned rows.
 * W row-major (O, K) like nn.Linear.weight.  Sums are fp32 fma chains in a fixed order (deterministic).
 * --------------------------------------------------------------------------------- */
typedef struct scf_fc_desc {
  const float* x; int32_t x_parts; int64_t x_part_stride;
  const float* x_bias; int32_t x_relu;
  int32_t gn_groups, gn_hw; const float* gn_gamma; const float* gn_beta; float gn_eps;
  int32_t N, K;
  const float* W; const float* bias; float* y; int32_t O;
  const float* W2; const float* bias2; float* y2; int32_t O2;
  int32_t act;                          /* SCF_ACT_* of the finished outputs (slices == 1)  */
  int32_t slices;
} scf_fc_desc;
int scf_fc_splitk(const scf_fc_desc* desc, scf_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Pose head tail + pose update.  replaces pose_head.py:207-210 (class select) and
 * get_pose_from_delta_pose, models/utils/pose.py:124-149 (+ :153-169 ortho6d).
 * rot_all (N, num_class*6), trans_all (N, num_class*3) are the two linear heads' outputs.
 * label_mode is a bit set (any other bit: SCF_EINVAL):
 *   0                          the reference's inference path: every sample is decoded with class label[0]
 *                              (index_select(...)[:, 0], pose_head.py:209-210), depth_transform='exp' (pose.py:137-138)
 *   SCF_POSE_LABEL_PER_SAMPLE  sample n uses label[n]
 *   SCF_POSE_DEPTH_LINEAR      the other depth_transform branch, pose.py:139-141: t_z' = t_z * (d_z + 1)
 * Outputs: d_rot (N,6), d_trans (N,3), R_out (N,3,3), t_out (N,3).  R_out/t_out may alias R_in/t_in.
 * A label outside [0, num_class) is CLAMPED (a kernel cannot raise; the reference's index_select
 * does): validate labels on the host (SCFlowRefiner.forward_single_pass does).
 * --------------------------------------------------------------------------------- */
#define SCF_POSE_LABEL_PER_SAMPLE 1
#define SCF_POSE_DEPTH_LINEAR 2
int scf_pose_update(const float* rot_all, const float* trans_all, const int64_t* label,
                    int num_class, int label_mode, const float* R_in, const float* t_in,
                    float* d_rot, float* d_trans, float* R_out, float* t_out, int N,
                    scf_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Pose-induced flow (dense form of cal_3d_2d_corr + get_flow_from_delta_pose_and_points,
 * models/utils/pose.py:44-64, 66-88).  For every pixel with depth > 0:
 *   P = R0^-1 (K^-1 [x y 1]^T d - t0);  p = K (R P + t);  flow = (p_x/p_z - x, p_y/p_z - y)
 * else flow = invalid_num.  depth (N,H,W); K,R0,R (N,3,3); t0,t (N,3); flow (N,2,H,W).
 * --------------------------------------------------------------------------------- */
int scf_reproject_flow(const float* depth, const float* K, const float* R0, const float* t0,
                       const float* R, const float* t, float* flow, int N, int H, int W,
                       float invalid_num, scf_stream_t stream);

/* BaseDataset.eval_pose_error (datasets/base_dataset.py:378-424; project_3d_point,
 * datasets/pose.py:18-78) for the samples sample_idx[0..nsel) that share one vertex set
 * verts (nv,3): err3d[s] = ADD (symmetric = 0) or ADD-S (closest predicted point, symmetric
 * != 0), err2d[s] = mean reprojection distance with x / (z + 1e-8).  float64 throughout, like
 * the reference's numpy arrays; gt_r/pred_r/K (N,3,3), gt_t/pred_t (N,3), outputs (N). */
int scf_pose_error(const double* verts, int nv, const double* gt_r, const double* gt_t,
                   const double* pred_r, const double* pred_t, const double* K,
                   const int* sample_idx, int nsel, int symmetric, double* err3d, double* err2d,
                   scf_stream_t stream);

/* filter_flow_by_mask (models/utils/flow.py:6-26), in place on flow (N,2,H,W): a vector is set
 * to invalid_num when both components are >= invalid_num or when mask (N,H,W), sampled
 * bilinearly (zeros padding) at the vector's end point, is < 0.9.  The end point is normalised
 * with (size-1) (coords_grid, warp.py:9-29) and de-normalised per align_corners, as the
 * reference does (its default align_corners=0 therefore samples at (x+fx)*W/(W-1) - 0.5). */
int scf_filter_flow_by_mask(float* flow, const float* mask, int N, int H, int W,
                            float invalid_num, int align_corners, scf_stream_t stream);

/* cal_epe (models/utils/flow.py:64-88), all three reductions from one pass over flow_tgt / flow_pred
 * (N,2,H,W) and the optional mask (N,H,W; NULL = none):
 *   valid = sqrt(tgt_x^2 + tgt_y^2) < max_flow [&& mask >= 0.5],  err = |flow_tgt - flow_pred|_2
 *   err_map      (N,H,W) or NULL      reduction='none':  err * valid
 *   mean, ratios (N), (nthr,N) / NULL reduction='mean':  sum(err * valid) / (count(valid) + 1e-10) per sample;
 *                                     ratios[t][n] = count(err < threshs[t] among the INVALID pixels) / that
 *                                     total -- the reference overwrites the valid pixels with 1e8 before the
 *                                     comparison (flow.py:79), reproduced as-is; fix_threshold_quirk != 0
 *                                     counts the valid pixels instead
 *   total_mean, total_ratios (1), (nthr) / NULL   reduction='total_mean': the same over the whole batch,
 *                                     ratios over the valid pixels (flow.py:84-87)
 * threshs is a HOST array of nthr <= 8 thresholds.  Squares, adds and square roots are separately rounded
 * fp32 operations as in torch; error sums are accumulated in fp64 in a fixed order and rounded once.
 * workspace: scf_cal_epe_workspace_bytes(N, H, W) bytes of device memory (block partials). */
int64_t scf_cal_epe_workspace_bytes(int N, int H, int W);
int scf_cal_epe(const float* flow_tgt, const float* flow_pred, const float* mask, int N, int H, int W,
                float max_flow, const float* threshs, int nthr, int fix_threshold_quirk, float* err_map,
                float* mean, float* ratios, float* total_mean, float* total_ratios, void* workspace,
                scf_stream_t stream);

/* object-frame points of every pixel (dense cal_3d_2d_corr): pts (N,3,H,W), 0 where
 * depth <= 0.  Test/diagnostic entry; scf_reproject_flow recomputes them on the fly. */
int scf_unproject_depth(const float* depth, const float* K, const float* R0, const float* t0,
                        float* pts, int N, int H, int W, scf_stream_t stream);

/* out = mul * bilinear_resize(a + b?) with align_corners=True (F.interpolate semantics,
 * scflow_decoder.py:188-197, 222-227).  a, b: (planes, Hin, Win); out (planes, Hout, Wout) */
int scf_resize_bilinear(const float* a, const float* b, float* out, int64_t planes, int Hin,
                        int Win, int Hout, int Wout, float mul, scf_stream_t stream);

/* RAFT convex up-sampling (x8, 3x3 neighbourhood).  replaces RAFTDecoder._upsample
 * models/decoder/raft_decoder.py:381-416 and RAFTDecoderMask.upsample_flow/upsample_mask
 * raft_decoder_mask.py:104-160:  out[n,c,8y+sy,8x+sx] = sum_k softmax_k(mask_mul *
 * mask[n, k*64+sy*8+sx, y, x]) * x_mul * x[n, c, y+k/3-1, x+k%3-1] (zero padded).
 * x (N,C,h,w); mask (N,9*64,h,w); out (N,C,8h,8w).                                     */
int scf_convex_upsample(const float* x, const float* mask, float* out, int N, int C, int h,
                        int w, int scale, float x_mul, float mask_mul, scf_stream_t stream);

/* 2x2 stride-2 average pool over (planes, Hin, Win) -> (planes, Hin/2, Win/2)          */
int scf_avgpool2x2(const float* x, float* out, int64_t planes, int Hin, int Win,
                   scf_stream_t stream);

/* out[n, c, :] = x[n, c, :] * mask[n, 0, :]  (mask (N, 1, HW) dense; x / out sample-strided): the
 * occlusion masking of the looked-up correlation / of the flow, scflow_decoder.py:199-205
 * (constructor switches mask_corr / mask_flow; both False in configs/refine_models/scflow.py). */
int scf_mul_mask(const float* x, int64_t x_nstride, const float* mask, float* out,
                 int64_t out_nstride, int N, int C, int HW, scf_stream_t stream);

/* elementwise helpers used for glue (split tanh/relu of the context features etc.)   */
int scf_copy_strided(const float* src, int64_t src_nstride, float* dst, int64_t dst_nstride,
                     int N, int64_t count, scf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SCFLOW_HIP_H */
