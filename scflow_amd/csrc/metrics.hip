// cal_epe (reference models/utils/flow.py:64-88): the end-point-error metric of the evaluation side
// (SURVEY.md 8(f) row 3), all three reductions from ONE pass over the two flow fields:
//     mag   = sqrt(tgt_x^2 + tgt_y^2)            valid = mag < max_flow  [and mask >= 0.5]
//     err   = sqrt((tgt_x - pred_x)^2 + (tgt_y - pred_y)^2)
//     'none'        err * valid                                                        (N, H, W)
//     'mean'        sum(err * valid) / (count(valid) + 1e-10) per sample; the '<t>px' ratios count --
//                   as the reference does, flow.py:79 overwrites the VALID pixels with 1e8 first --
//                   the INVALID pixels with err < t (fix_threshold_quirk: the valid ones)
//     'total_mean'  the same over the whole batch, ratios over the valid pixels
// Arithmetic is the reference's, operation by operation: every square, add and sqrt is a separately
// rounded fp32 operation (no fma contraction: torch evaluates `x ** 2`, `sum(dim=1)`, `sqrt` as three
// kernels), counts are exact integers, the error sums are accumulated in fp64 in a FIXED order (block
// partials, then a fixed-order combine) and rounded to fp32 once -- the value torch's pairwise fp32 sum
// approximates; the final divisions are fp32 like torch's (`int64 count + 1e-10` is a float32 tensor).
#include "scf_common.h"

#define EPE_MAX_THR 8
#define EPE_THREADS 256
#define EPE_PIX_PER_BLOCK 4096          // 16 pixels per thread

struct EpeK {
  const float* tgt; const float* pred; const float* mask;
  float* err_map;
  int HW; int nthr; int blocks_per_sample;
  float max_flow;
  float thr[EPE_MAX_THR];
};

// workspace per (sample, block): [sum_err (double)] [cnt_valid, cnt_valid_lt[8], cnt_invalid_lt[8]] as 64-bit words
#define EPE_WS_WORDS (2 + 2 * EPE_MAX_THR)

__global__ __launch_bounds__(EPE_THREADS)
void cal_epe_partial_kernel(EpeK k, unsigned long long* ws) {
  // every operation below is the separately rounded fp32 operation torch performs: no fma contraction, and sqrtf /
  // the divisions are the correctly rounded ones (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt; the
  // __fsqrt_rn / __fdiv_rn intrinsics are the 1-ulp NATIVE instructions in this toolchain and must not be used)
#pragma clang fp contract(off)
  // one flat grid (sample-major): any N with N * blocks_per_sample < 2^31
  const int n = (int)(blockIdx.x / (unsigned)k.blocks_per_sample), b = (int)(blockIdx.x - (unsigned)n * (unsigned)k.blocks_per_sample);
  const int tid = threadIdx.x;
  const float* tx = k.tgt + (long long)n * 2 * k.HW;
  const float* ty = tx + k.HW;
  const float* px = k.pred + (long long)n * 2 * k.HW;
  const float* py = px + k.HW;
  const float* mk = k.mask ? k.mask + (long long)n * k.HW : nullptr;
  float* em = k.err_map ? k.err_map + (long long)n * k.HW : nullptr;
  double sum = 0.0;
  unsigned cv = 0, cvl[EPE_MAX_THR], cil[EPE_MAX_THR];
#pragma unroll
  for (int t = 0; t < EPE_MAX_THR; ++t) cvl[t] = cil[t] = 0;
  const int p0 = b * EPE_PIX_PER_BLOCK;
  // a thread walks pixels p0 + tid, p0 + tid + 256, ...: coalesced, and the order of a thread's adds is fixed
  for (int i = tid; i < EPE_PIX_PER_BLOCK; i += EPE_THREADS) {
    const int p = p0 + i;
    if (p >= k.HW) break;
    const float a = tx[p], c = ty[p];
    const float mag = sqrtf(a * a + c * c);          // contraction is off in this function: mul, mul, add, sqrt
    const float dx = a - px[p], dy = c - py[p];
    const float err = sqrtf(dx * dx + dy * dy);
    bool valid = mag < k.max_flow;
    if (mk) valid = valid && (mk[p] >= 0.5f);
    const float ev = err * (valid ? 1.0f : 0.0f);                  // NaN * 0 stays NaN, as in torch
    if (em) em[p] = ev;
    sum += (double)ev;
    cv += valid ? 1u : 0u;
#pragma unroll
    for (int t = 0; t < EPE_MAX_THR; ++t) {
      if (t < k.nthr) {
        const bool lt = err < k.thr[t];
        cvl[t] += (valid && lt) ? 1u : 0u;
        cil[t] += (!valid && lt) ? 1u : 0u;
      }
    }
  }
  // fixed-order block reduction: shuffle tree inside a wave, then the 4 waves in order through LDS
  __shared__ double s_sum[EPE_THREADS / 64];
  __shared__ unsigned s_cnt[EPE_THREADS / 64][1 + 2 * EPE_MAX_THR];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sum += __shfl_down(sum, o, 64);
    cv += __shfl_down(cv, o, 64);
#pragma unroll
    for (int t = 0; t < EPE_MAX_THR; ++t) {
      cvl[t] += __shfl_down(cvl[t], o, 64);
      cil[t] += __shfl_down(cil[t], o, 64);
    }
  }
  const int wave = tid >> 6, lane = tid & 63;
  if (lane == 0) {
    s_sum[wave] = sum;
    s_cnt[wave][0] = cv;
#pragma unroll
    for (int t = 0; t < EPE_MAX_THR; ++t) { s_cnt[wave][1 + t] = cvl[t]; s_cnt[wave][1 + EPE_MAX_THR + t] = cil[t]; }
  }
  __syncthreads();
  if (tid == 0) {
    unsigned long long* o = ws + ((long long)n * k.blocks_per_sample + b) * EPE_WS_WORDS;
    double s = s_sum[0];
    for (int w = 1; w < EPE_THREADS / 64; ++w) s += s_sum[w];
    o[0] = (unsigned long long)__double_as_longlong(s);
    for (int j = 0; j < 1 + 2 * EPE_MAX_THR; ++j) {
      unsigned long long c = 0;
      for (int w = 0; w < EPE_THREADS / 64; ++w) c += s_cnt[w][j];
      o[1 + j] = c;
    }
  }
}

// thread n combines the block partials of sample n in block order ('mean'); 'total_mean' is a strided fixed-order
// sum per thread + a fixed LDS tree over the block's 256 threads
__global__ void cal_epe_final_kernel(const unsigned long long* ws, int N, int bps, int nthr, int fix_quirk,
                                     float* mean, float* ratios, float* total_mean, float* total_ratios) {
#pragma clang fp contract(off)
  if (mean || ratios) {
    for (int n = (int)threadIdx.x; n < N; n += (int)blockDim.x) {
      double s = 0.0;
      unsigned long long c[1 + 2 * EPE_MAX_THR];
      for (int j = 0; j < 1 + 2 * EPE_MAX_THR; ++j) c[j] = 0;
      for (int b = 0; b < bps; ++b) {
        const unsigned long long* o = ws + ((long long)n * bps + b) * EPE_WS_WORDS;
        s += __longlong_as_double((long long)o[0]);
        for (int j = 0; j < 1 + 2 * EPE_MAX_THR; ++j) c[j] += o[1 + j];
      }
      const float total = ((float)(long long)c[0] + 1e-10f);
      if (mean) mean[n] = ((float)s / total);
      if (ratios)
        for (int t = 0; t < nthr; ++t)
          ratios[(long long)t * N + n] =
              ((float)(long long)(fix_quirk ? c[1 + t] : c[1 + EPE_MAX_THR + t]) / total);
    }
  }
  if (total_mean || total_ratios) {
    // 'total_mean': thread t adds the partials t, t + T, t + 2T, ... in that order (fp64 sums, exact integer
    // counts), then the T thread totals are folded by a fixed binary tree through LDS: the order depends on the
    // launch shape only, never on timing
    __shared__ double t_sum[256];
    __shared__ unsigned long long t_cnt[256][1 + EPE_MAX_THR];
    const int T = (int)blockDim.x, t = (int)threadIdx.x;
    double s = 0.0;
    unsigned long long c[1 + EPE_MAX_THR];
    for (int j = 0; j < 1 + EPE_MAX_THR; ++j) c[j] = 0;
    for (long long i = t; i < (long long)N * bps; i += T) {
      const unsigned long long* o = ws + i * EPE_WS_WORDS;
      s += __longlong_as_double((long long)o[0]);
      for (int j = 0; j < 1 + EPE_MAX_THR; ++j) c[j] += o[1 + j];
    }
    t_sum[t] = s;
    for (int j = 0; j < 1 + EPE_MAX_THR; ++j) t_cnt[t][j] = c[j];
    __syncthreads();
    for (int o = T >> 1; o > 0; o >>= 1) {
      if (t < o) {
        t_sum[t] += t_sum[t + o];
        for (int j = 0; j < 1 + EPE_MAX_THR; ++j) t_cnt[t][j] += t_cnt[t + o][j];
      }
      __syncthreads();
    }
    if (t == 0) {
      const float total = ((float)(long long)t_cnt[0][0] + 1e-10f);
      if (total_mean) total_mean[0] = ((float)t_sum[0] / total);
      if (total_ratios)
        for (int q = 0; q < nthr; ++q) total_ratios[q] = ((float)(long long)t_cnt[0][1 + q] / total);
    }
  }
}

static int epe_blocks(int HW) { return (HW + EPE_PIX_PER_BLOCK - 1) / EPE_PIX_PER_BLOCK; }

extern "C" int64_t scf_cal_epe_workspace_bytes(int N, int H, int W) {
  if (N <= 0 || H <= 0 || W <= 0 || (int64_t)H * W > 0x7fffffffLL) return SCF_EINVAL;
  return (int64_t)N * epe_blocks(H * W) * EPE_WS_WORDS * 8;
}

extern "C" int scf_cal_epe(const float* flow_tgt, const float* flow_pred, const float* mask, int N, int H, int W,
                           float max_flow, const float* threshs, int nthr, int fix_threshold_quirk, float* err_map,
                           float* mean, float* ratios, float* total_mean, float* total_ratios, void* workspace,
                           scf_stream_t stream) {
  if (!flow_tgt || !flow_pred || !workspace || N <= 0 || H <= 0 || W <= 0 || nthr < 0 || nthr > EPE_MAX_THR ||
      (nthr && !threshs))
    return SCF_EINVAL;
  if ((int64_t)H * W > 0x7fffffffLL || (int64_t)N * epe_blocks(H * W) > 0x7fffffffLL) return SCF_EUNSUPPORTED;
  const bool reduce = mean || ratios || total_mean || total_ratios;
  if (!reduce && !err_map) return SCF_EINVAL;
  EpeK k;
  k.tgt = flow_tgt; k.pred = flow_pred; k.mask = mask; k.err_map = err_map;
  k.HW = H * W; k.nthr = nthr; k.blocks_per_sample = epe_blocks(k.HW); k.max_flow = max_flow;
  for (int t = 0; t < EPE_MAX_THR; ++t) k.thr[t] = t < nthr ? threshs[t] : 0.f;
  hipStream_t st = scf_stream(stream);
  unsigned long long* ws = static_cast<unsigned long long*>(workspace);
  scf_launch(cal_epe_partial_kernel, dim3((unsigned)((int64_t)k.blocks_per_sample * N)), dim3(EPE_THREADS), 0, st, k, ws);
  if (scf_launch_status() != SCF_OK) return SCF_ELAUNCH;
  if (!reduce) return SCF_OK;
  scf_launch(cal_epe_final_kernel, dim3(1), dim3(256), 0, st, (const unsigned long long*)ws, N, k.blocks_per_sample,
             nthr, fix_threshold_quirk, mean, ratios, total_mean, total_ratios);
  return scf_launch_status();
}
