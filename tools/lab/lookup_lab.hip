// Stand-alone laboratory for the correlation-lookup kernel (not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Itools/lab tools/lab/lookup_lab.hip -o tools/lab/bin/lookup_lab
// Compiles the PRODUCT kernel source with SCF_LOOKUP_LAB (lookup_lab_hooks.h: s_memrealtime stamps per
// wave + two ablation switches) and times it against streaming ceilings on a pyramid that cannot sit in the
// 256 MiB Infinity Cache (two batch-B halves used alternately + a 1 GiB flush between launches).
#include "../../scflow_amd/csrc/capi.hip"
#define SCF_LOOKUP_LAB 1
#include "../../scflow_amd/csrc/corr_lookup.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(float* p, size_t n, unsigned seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) {
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (float)(h & 0xffff) * (1.0f / 65536.f) - 0.5f;
  }
}
__global__ void flush_kernel(float4* p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t st = (size_t)gridDim.x * blockDim.x;
  float acc = 0.f;
  for (; i < n; i += st) { float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 123.456f) p[0].x = acc;      // never true: the flush only READS (no dirty lines left behind)
}
// flush that leaves the Infinity Cache full of DIRTY lines (what the convolutions of a step leave behind)
__global__ void dirty_kernel(float4* p, size_t n, float v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t st = (size_t)gridDim.x * blockDim.x;
  const float4 w = {v, v, v, v};
  for (; i < n; i += st) p[i] = w;
}
// matrix-pipe burn: pulls the shader clock down to what it is inside the step (power management)
__global__ __launch_bounds__(256) void burn_kernel(float* out, int iters) {
  typedef float v16 __attribute__((ext_vector_type(16)));
  v16 acc = {0};
  float a = (float)threadIdx.x * 1e-3f, b = 1.0f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  if (acc[0] == 123.f) out[0] = acc[1];
}
// streaming ceiling: each block reads rd_bytes and writes wr_bytes (float4, coalesced)
__global__ __launch_bounds__(256) void stream_kernel(const float4* src, float4* dst, int rd4, int wr4) {
  const float4* s = src + (size_t)blockIdx.x * rd4;
  float4* d = dst + (size_t)blockIdx.x * wr4;
  float4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < rd4; i += 256) {
    float4 v = s[i];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  for (int i = threadIdx.x; i < wr4; i += 256) d[i] = acc;
}
// same bytes, but reads issued as float4 loads all in flight first (unrolled 8), then stores
__global__ __launch_bounds__(256) void stream2_kernel(const float4* src, float4* dst, int rd4, int wr4) {
  const float4* s = src + (size_t)blockIdx.x * rd4;
  float4* d = dst + (size_t)blockIdx.x * wr4;
  float4 acc = {0.f, 0.f, 0.f, 0.f};
  int i = threadIdx.x;
  for (; i + 7 * 256 < rd4; i += 8 * 256) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = s[i + u * 256];
#pragma unroll
    for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  for (; i < rd4; i += 256) { float4 v = s[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
  for (int j = threadIdx.x; j < wr4; j += 256) d[j] = acc;
}

struct Timer { hipEvent_t a, b; };

static double median(std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IOLBF, 0);              // a device fault must not swallow the lines before it
  const int B = argc > 1 ? atoi(argv[1]) : 32;      // pairs per launch
  const int reps = argc > 2 ? atoi(argv[2]) : 20;
  const int h = argc > 3 ? atoi(argv[3]) : 32, w = argc > 4 ? atoi(argv[4]) : 32;
  const int r = 4, L = 4, hw = h * w;
  const int NH = 2;                                  // alternating halves
  const size_t Q = (size_t)B * hw;
  float* lv[NH][4];
  for (int hf = 0; hf < NH; ++hf)
    for (int l = 0; l < L; ++l) {
      const size_t n = Q * (size_t)(((h >> l) + 3) / 4 * 4) * (((w >> l) + 7) / 8 * 8);      // room for either layout
      CK(hipMalloc(&lv[hf][l], n * 4));
      fill_kernel<<<2048, 256>>>(lv[hf][l], n, 17u * l + hf);
    }
  float *flow, *out;
  CK(hipMalloc(&flow, Q * 2 * 4));
  CK(hipMalloc(&out, Q * 324 * 4));
  {
    std::vector<float> hf(Q * 2);
    std::mt19937 g(1);
    std::normal_distribution<float> nd(0.f, 3.f);
    for (auto& v : hf) v = nd(g);
    CK(hipMemcpy(flow, hf.data(), hf.size() * 4, hipMemcpyHostToDevice));
  }
  const size_t FL = (size_t)1 << 28;                 // 256 Mi float4 = ... 4 GiB is too much: use 64 Mi float4 = 1 GiB
  float4* flushbuf;
  const size_t fl4 = (size_t)64 << 20;
  CK(hipMalloc(&flushbuf, fl4 * 16));
  CK(hipMemset(flushbuf, 0, fl4 * 16));
  (void)FL;
  const int nblk = (int)((Q + 31) / 32);
  unsigned long long* trace;
  CK(hipMalloc(&trace, (size_t)nblk * 4 * 8 * 8));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const unsigned tiled = getenv("LAB_ROWMAJOR") ? 0u : scf_corr_preferred_layout(h, w, r, L);      // tile mask

  auto run = [&](const char* name, int skip_dma, int skip_store, bool with_trace, bool flush, int rotate = -1, int grid = 0, int pipe = 1) {
    scf_lab_skip_dma = skip_dma; scf_lab_skip_store = skip_store; scf_lab_rotate = rotate; scf_lab_grid = grid; scf_lab_pipe = pipe;
    printf("[run] %s\n", name);
    std::vector<float> us;
    std::vector<unsigned long long> tr((size_t)nblk * 32);
    for (int it = 0; it < reps + 3; ++it) {
      if (flush) {
        if (getenv("LAB_DIRTY")) dirty_kernel<<<4096, 256, 0, st>>>(flushbuf, fl4 / 2, (float)it);   // 512 MiB written
        else flush_kernel<<<4096, 256, 0, st>>>(flushbuf, fl4);
      }
      if (getenv("LAB_BURN")) burn_kernel<<<1024, 256, 0, st>>>((float*)flushbuf, atoi(getenv("LAB_BURN")));
      scf_lab_trace = with_trace ? trace : nullptr;
      if (with_trace) CK(hipMemsetAsync(trace, 0, (size_t)nblk * 32 * 8, st));
      scf_timer_t tm;
      scf_timer_create(&tm);
      const float* lvp[4] = {lv[it % NH][0], lv[it % NH][1], lv[it % NH][2], lv[it % NH][3]};
      scf_timer_arm(tm);
      int rc = scf_corr_lookup_ex(lvp, flow, out, B, h, w, r, L, tiled, st);
      scf_timer_arm(nullptr);
      if (rc) { printf("lookup rc %d\n", rc); exit(1); }
      CK(hipStreamSynchronize(st));
      float u = 0; scf_timer_elapsed_us(tm, &u); scf_timer_destroy(tm);
      if (it >= 3) us.push_back(u);
      if (with_trace && it == reps + 2) CK(hipMemcpy(tr.data(), trace, tr.size() * 8, hipMemcpyDeviceToHost));
    }
    const double m = median(us);
    printf("%-28s median %7.2f us  min %7.2f  max %7.2f   %.0f GB/s algorithmic\n", name, m,
           *std::min_element(us.begin(), us.end()), *std::max_element(us.begin(), us.end()), 2904.0 * Q / m / 1e3);
    if (with_trace) {
      // per level (wave): distribution of every stamp relative to the earliest kernel entry (10 ns ticks)
      unsigned long long t0 = ~0ull;
      for (size_t i = 0; i < (size_t)nblk * 4; ++i) if (tr[i * 8]) t0 = std::min(t0, tr[i * 8]);
      const char* nm[7] = {"entry", "flow", "pre-dma", "issued", "landed", "emitted", "st-ack"};
      for (int lv = 0; lv < 4; ++lv) {
        printf("  level %d:", lv);
        for (int s = 0; s < 7; ++s) {
          std::vector<float> v;
          for (int b = 0; b < nblk; ++b)
            for (int wv = 0; wv < 4; ++wv) {
              if ((int)(tr[((size_t)b * 4 + wv) * 8 + 7] >> 32) != lv) continue;
              const unsigned long long t = tr[((size_t)b * 4 + wv) * 8 + s];
              if (t) v.push_back((float)(t - t0) * 0.01f);
            }
          if (v.empty()) { printf(" %s -", nm[s]); continue; }
          std::sort(v.begin(), v.end());
          printf(" %s %.1f/%.1f/%.1f/%.1f", nm[s], v[v.size() / 20], v[v.size() / 2], v[v.size() - 1 - v.size() / 20], v.back());
        }
        printf("   (p5/p50/p95/max us)\n");
      }
      // timeline: per 0.5 us bin, the units with gathers in flight (pre-dma .. landed) and the units emitting
      // (landed .. emitted); "both" = bins in which at least 10 % of the units are in each phase
      {
        const int NB = 64;
        int gat[NB] = {0}, emi[NB] = {0};
        float tend = 0;
        const size_t units = (size_t)nblk * 4;
        for (size_t i = 0; i < units; ++i) {
          const unsigned long long* e = &tr[i * 8];
          if (!e[2] || !e[4] || !e[5]) continue;
          const float a = (float)(e[2] - t0) * 0.01f, b = (float)(e[4] - t0) * 0.01f, c = (float)(e[5] - t0) * 0.01f;
          tend = std::max(tend, e[6] ? (float)(e[6] - t0) * 0.01f : c);
          for (int k = 0; k < NB; ++k) {
            const float lo = 0.5f * k, hi = lo + 0.5f;
            if (a < hi && b > lo) gat[k]++;
            if (b < hi && c > lo) emi[k]++;
          }
        }
        int both = 0, nbins = (int)(tend / 0.5f) + 1;
        printf("  timeline (0.5 us bins, %% of units gathering | emitting):\n   ");
        for (int k = 0; k < nbins && k < NB; ++k) {
          const int pg = (int)(100.0 * gat[k] / units), pe = (int)(100.0 * emi[k] / units);
          if (pg >= 10 && pe >= 10) ++both;
          printf(" %d|%d", pg, pe);
        }
        printf("\n  last stamp %.1f us; bins with >= 10 %% of the units in BOTH phases: %d of %d (%.0f %% of the wave-active time, %.0f %% of the kernel's %.2f us)\n",
               tend, both, nbins, 100.0 * both / nbins, 100.0 * both * 0.5 / m, m);
      }
      // hardware placement: HW_ID of (group, wave): SIMD = bits 5:4, TG_ID = bits 19:16, CU = 11:8
      int simd_of_wave[4][4] = {{0}}, tg_hist[16] = {0}, lvl_on_simd[4][4] = {{0}};
      for (int b = 0; b < nblk; ++b)
        for (int wv = 0; wv < 4; ++wv) {
          const unsigned long long e = tr[((size_t)b * 4 + wv) * 8 + 7];
          const unsigned hwid = (unsigned)e;
          const int lv = (int)(e >> 32) & 3;
          simd_of_wave[wv][(hwid >> 4) & 3]++;
          lvl_on_simd[(hwid >> 4) & 3][lv]++;
          if (wv == 0) tg_hist[(hwid >> 16) & 15]++;
        }
      printf("  wave->SIMD counts:");
      for (int wv = 0; wv < 4; ++wv) printf(" w%d[%d %d %d %d]", wv, simd_of_wave[wv][0], simd_of_wave[wv][1], simd_of_wave[wv][2], simd_of_wave[wv][3]);
      printf("\n  levels per SIMD:");
      for (int sd = 0; sd < 4; ++sd) printf(" simd%d[%d %d %d %d]", sd, lvl_on_simd[sd][0], lvl_on_simd[sd][1], lvl_on_simd[sd][2], lvl_on_simd[sd][3]);
      printf("\n  TG_ID histogram:");
      for (int t = 0; t < 16; ++t) printf(" %d", tg_hist[t]);
      printf("\n");
    }
    return m;
  };

  printf("lookup lab: B=%d h=%d w=%d queries=%zu blocks=%d tiled=%d pyramid=%.0f MiB per half\n", B, h, w, Q, nblk,
         (int)tiled, Q * (double)(hw + hw / 4 + hw / 16 + hw / 64) * 4 / 1048576.0);
  if (getenv("LAB_ABL")) {          // ablations of the pipelined kernel only (fault hunt)
    scf_lab_store_mode = 0;
    const int which = atoi(getenv("LAB_ABL"));
    const bool tr = getenv("LAB_TRACE") != nullptr;
    if (which & 1) run("v9 G=2: no stores (cold)", 0, 1, tr, true, -1, 0, 2);
    if (which & 2) run("v9 G=2: no gathers (cold)", 1, 0, tr, true, -1, 0, 2);
    if (which & 4) run("v9 G=2: neither (cold)", 1, 1, tr, true, -1, 0, 2);
    if (which & 8) run("v9 G=2: no gathers (warm)", 1, 0, tr, false, -1, 0, 2);
    printf("done\n");
    return 0;
  }
  if (getenv("LAB_POL")) {          // store-policy sweep of v8 (the tap policy is this binary's -DSCF_LOOKUP_TAP_POL)
    const char* nm[6] = {"", "nt", "sc1", "sc0 sc1", "sc1 nt", "plain"};
    for (int rep = 0; rep < 2; ++rep)
      for (int m = 1; m < 6; ++m) {
        scf_lab_store_mode = m;
        char buf[64];
        snprintf(buf, sizeof buf, "v8 stores %s cold", nm[m]);
        run(buf, 0, 0, false, true, -1, 0, 1);
      }
    return 0;
  }
  if (getenv("LAB_QUICK")) {        // the A/B set of round 5 (cold = what the pipeline sees)
    const bool tr = getenv("LAB_TRACE") != nullptr;
    scf_lab_store_mode = 0;
    for (int rep = 0; rep < 2; ++rep) {
      scf_lab_early = 0; scf_lab_stagger = 0;
      run("v8 late maps warm", 0, 0, false, false, -1, 0, 1);
      run("v8 late maps cold", 0, 0, tr && rep, true, -1, 0, 1);
      scf_lab_early = 1;
      run("v8 early maps warm", 0, 0, false, false, -1, 0, 1);
      run("v8 early maps cold", 0, 0, tr && rep, true, -1, 0, 1);
      scf_lab_stagger = 4;
      run("v8 early maps, odd slots +2 us, cold", 0, 0, tr && rep, true, -1, 0, 1);
      scf_lab_stagger = 8;
      run("v8 early maps, odd slots +4 us, cold", 0, 0, tr && rep, true, -1, 0, 1);
      scf_lab_stagger = 0;
      run("v9 pipe G=2 warm", 0, 0, false, false, -1, 0, 2);
      run("v9 pipe G=2 cold", 0, 0, tr && rep, true, -1, 0, 2);
      run("v8, 2 groups per block, warm", 0, 0, false, false, -1, 0, 4);
      run("v8, 2 groups per block, cold", 0, 0, tr && rep, true, -1, 0, 4);
      run("v8, 4 groups per block, warm", 0, 0, false, false, -1, 0, 5);
      run("v8, 4 groups per block, cold", 0, 0, tr && rep, true, -1, 0, 5);
    }
    return 0;
  }
  {
    const char* nm[5] = {"plain", "nt", "sc1", "sc0 sc1", "sc1 nt"};
    const int modes[3] = {0, 2, 1};
    for (int rep = 0; rep < 2; ++rep)
      for (int mi = 0; mi < (getenv("LAB_STORE_SWEEP") ? 3 : 2); ++mi) {
        const int m = modes[mi];
        scf_lab_store_mode = m;
        char buf[64];
        snprintf(buf, sizeof buf, "stores %s, warm", nm[m]);
        run(buf, 0, 0, false, false);
        snprintf(buf, sizeof buf, "stores %s, cold", nm[m]);
        run(buf, 0, 0, rep == 1, true);
      }
    scf_lab_store_mode = 2;
    run("sc1: no stores (cold)", 0, 1, true, true);
    run("sc1: no gathers (cold)", 1, 0, true, true);
    run("sc1: neither (cold)", 1, 1, true, true);
    for (int rep = 0; rep < 2; ++rep) {
      run("v8 (1 group/block) warm", 0, 0, false, false, -1, 0, 1);
      run("v8 (1 group/block) cold", 0, 0, rep == 1, true, -1, 0, 1);
      run("v9 pipe G=2 warm", 0, 0, false, false, -1, 0, 2);
      run("v9 pipe G=2 cold", 0, 0, rep == 1, true, -1, 0, 2);
      run("v9 pipe G=3 warm", 0, 0, false, false, -1, 0, 3);
      run("v9 pipe G=3 cold", 0, 0, rep == 1, true, -1, 0, 3);
    }
    run("v9 G=2: no stores (cold)", 0, 1, true, true, -1, 0, 2);
    run("v9 G=2: no gathers (cold)", 1, 0, true, true, -1, 0, 2);
    run("v9 G=2: neither (cold)", 1, 1, true, true, -1, 0, 2);
    if (nblk > 768) run("sc1 cold, grid 768", 0, 0, false, true, -1, 768);
    if (nblk > 512) run("sc1 cold, grid 512", 0, 0, false, true, -1, 512);
    scf_lab_store_mode = 0;
  }
  for (int pipe = 1; pipe <= 3; ++pipe) {   // checksum of the output of a full run (compare across kernel versions: same inputs)
    scf_lab_skip_dma = scf_lab_skip_store = 0; scf_lab_rotate = -1; scf_lab_grid = 0; scf_lab_trace = nullptr; scf_lab_pipe = pipe;
    CK(hipMemset(out, 0xff, Q * 324 * 4));
    const float* lvp[4] = {lv[0][0], lv[0][1], lv[0][2], lv[0][3]};
    scf_corr_lookup_ex(lvp, flow, out, B, h, w, r, L, tiled, st);
    CK(hipStreamSynchronize(st));
    std::vector<float> ho(Q * 324);
    CK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
    double s1 = 0, s2 = 0; unsigned long long hsh = 1469598103934665603ull; size_t nan = 0;
    for (size_t i = 0; i < ho.size(); ++i) {
      const float v = ho[i];
      if (!(v == v)) { ++nan; continue; }
      s1 += v; s2 += (double)v * v * (double)((i % 977) + 1);
      unsigned u; memcpy(&u, &v, 4); if (u == 0x80000000u) u = 0;      // -0 == +0
      hsh = (hsh ^ u) * 1099511628211ull;
    }
    printf("checksum (pipe mode %d): sum %.9e  wsum2 %.9e  nan %zu  fnv %016llx\n", pipe, s1, s2, nan, hsh);
  }

  // streaming ceilings with the same per-launch bytes and the same grid
  {
    const int rd4 = 1608 * 32 / 16, wr4 = 1296 * 32 / 16;     // bytes per block of 32 queries / 16
    float4 *src, *dst;
    CK(hipMalloc(&src, (size_t)nblk * rd4 * 16 * 2));
    CK(hipMalloc(&dst, (size_t)nblk * wr4 * 16));
    CK(hipMemset(src, 0, (size_t)nblk * rd4 * 16 * 2));
    for (int variant = 0; variant < 2; ++variant) {
      std::vector<float> us;
      for (int it = 0; it < reps + 3; ++it) {
        flush_kernel<<<4096, 256, 0, st>>>(flushbuf, fl4);
        hipEvent_t a, b;
        CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        const float4* s = src + (size_t)(it & 1) * nblk * rd4;
        if (variant == 0)
          hipExtLaunchKernelGGL(stream_kernel, dim3(nblk), dim3(256), 0, st, a, b, 0, s, dst, rd4, wr4);
        else
          hipExtLaunchKernelGGL(stream2_kernel, dim3(nblk), dim3(256), 0, st, a, b, 0, s, dst, rd4, wr4);
        CK(hipStreamSynchronize(st));
        float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
        if (it >= 3) us.push_back(ms * 1e3f);
        CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
      }
      const double m = median(us);
      printf("stream ceiling v%d (%d blocks, %d B read + %d B written per block): median %.2f us = %.0f GB/s\n", variant,
             nblk, rd4 * 16, wr4 * 16, m, (double)nblk * (rd4 + wr4) * 16 / m / 1e3);
    }
  }
  return 0;
}
