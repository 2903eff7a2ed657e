// Lab: what does a device-wide barrier inside one launch cost on MI355X, next to what it would replace -- the boundary
// between two dependent kernels of a hipGraph?  (Round 5: is a fused, cooperative pose-head tail worth building.)
//   hipcc --offload-arch=gfx950 -O3 tools/lab/grid_barrier.hip -o tools/lab/bin/grid_barrier
// Variants of the barrier (256 threads per block, grid <= resident capacity):
//   0  __threadfence + atomicAdd (device scope) + spin on a volatile load          (the portable form)
//   1  the same, but the spin is an atomic read (atomicAdd(p, 0))
//   2  hand-written: buffer_wbl2 sc1; s_waitcnt; global_atomic_add sc1 ... spin global_load_dword sc0 sc1; buffer_inv sc1
// Each block also WRITES 1 KB before a barrier and READS another block's 1 KB after it (checked): the barrier has to
// publish data across XCDs, not only count.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__device__ __forceinline__ void bar_portable(unsigned* ctr, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    atomicAdd(ctr, 1u);
    for (int sp = 0; sp < (1 << 20) && *(volatile unsigned*)ctr < target; ++sp) __builtin_amdgcn_s_sleep(1);   // bounded: a lab must not hang the box
    __threadfence();
  }
  __syncthreads();
}
__device__ __forceinline__ void bar_atomic_spin(unsigned* ctr, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    atomicAdd(ctr, 1u);
    for (int sp = 0; sp < (1 << 20) && atomicAdd(ctr, 0u) < target; ++sp) __builtin_amdgcn_s_sleep(1);
    __threadfence();
  }
  __syncthreads();
}
__device__ __forceinline__ void bar_asm(unsigned* ctr, unsigned target) {
  // release: every wave writes its dirty lines back to memory scope before the count; acquire: invalidate after
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\tbuffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned one = 1u, v;
    asm volatile("global_atomic_add %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : : "v"(ctr), "v"(one) : "memory");
    for (int sp = 0; sp < (1 << 20); ++sp) {
      asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(ctr) : "memory");
      if (v >= target) break;
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  asm volatile("buffer_inv sc1" ::: "memory");
}

template <int MODE>
__global__ __launch_bounds__(256) void bar_kernel(unsigned* ctr, float* buf, int nbar, unsigned base, int* bad) {
  const int nb = gridDim.x, b = blockIdx.x;
  int wrong = 0;
  for (int i = 0; i < nbar; ++i) {
    buf[(size_t)(i & 1) * nb * 256 + (size_t)b * 256 + threadIdx.x] = (float)(i * 1000 + b);
    const unsigned target = base + (unsigned)(i + 1) * nb;
    if (MODE == 0) bar_portable(ctr, target);
    else if (MODE == 1) bar_atomic_spin(ctr, target);
    else bar_asm(ctr, target);
    const int ob = (b + nb / 2 + 1) % nb;      // a block on another XCD
    float v;
    if (MODE == 2) {
      const float* q = buf + (size_t)(i & 1) * nb * 256 + (size_t)ob * 256 + threadIdx.x;
      asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(q) : "memory");
    } else {
      v = __builtin_nontemporal_load(buf + (size_t)(i & 1) * nb * 256 + (size_t)ob * 256 + threadIdx.x);
    }
    if (v != (float)(i * 1000 + ob)) ++wrong;
  }
  if (wrong) atomicAdd(bad, wrong);
}

__global__ void tiny_kernel(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ __launch_bounds__(256) void tiny_wide_kernel(float* p) { p[(size_t)blockIdx.x * 256 + threadIdx.x] += 1.f; }

template <int MODE>
void run(int grid, int nbar) {
  unsigned* ctr; float* buf; int* bad;
  hipMalloc(&ctr, 256); hipMemset(ctr, 0, 256);
  hipMalloc(&buf, (size_t)2 * grid * 256 * 4); hipMemset(buf, 0, (size_t)2 * grid * 256 * 4);
  hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  unsigned base = 0;
  std::vector<float> ts;
  float t1 = 0.f;
  for (int rep = 0; rep < 12; ++rep) {
    const int nb = (rep & 1) ? nbar : 1;       // alternate 1 barrier / nbar barriers: the slope is the barrier
    hipEventRecord(e0);
    bar_kernel<MODE><<<grid, 256>>>(ctr, buf, nb, base, bad);
    hipEventRecord(e1); hipEventSynchronize(e1);
    base += (unsigned)nb * grid;
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 2) { if (rep & 1) ts.push_back(ms); else t1 = ms; }
  }
  std::sort(ts.begin(), ts.end());
  int hb = 0; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
  printf("mode %d grid %4d: 1 barrier launch %.1f us, %d barriers %.1f us -> %.2f us per barrier (+write/read of 1 KB per block), wrong reads %d\n",
         MODE, grid, t1 * 1e3, nbar, ts[ts.size() / 2] * 1e3, (ts[ts.size() / 2] - t1) * 1e3 / (nbar - 1), hb);
  hipFree(ctr); hipFree(buf); hipFree(bad);
}

void graph_chain(int nk, bool wide, int grid) {
  float* p; hipMalloc(&p, (size_t)grid * 256 * 4); hipMemset(p, 0, (size_t)grid * 256 * 4);
  hipStream_t st; hipStreamCreate(&st);
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < nk; ++i) {
    if (wide) tiny_wide_kernel<<<grid, 256, 0, st>>>(p); else tiny_kernel<<<1, 64, 0, st>>>(p);
  }
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> ts;
  for (int rep = 0; rep < 8; ++rep) {
    hipEventRecord(e0, st);
    hipGraphLaunch(ge, st);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 2) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  printf("graph of %d dependent %s kernels: %.1f us -> %.2f us per kernel\n", nk, wide ? "256-block" : "1-block", ts[ts.size() / 2] * 1e3,
         ts[ts.size() / 2] * 1e3 / nk);
  // the same chain launched eagerly
  ts.clear();
  for (int rep = 0; rep < 8; ++rep) {
    hipEventRecord(e0, st);
    for (int i = 0; i < nk; ++i) {
      if (wide) tiny_wide_kernel<<<grid, 256, 0, st>>>(p); else tiny_kernel<<<1, 64, 0, st>>>(p);
    }
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 2) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  printf("eager   %d dependent %s kernels: %.1f us -> %.2f us per kernel\n", nk, wide ? "256-block" : "1-block", ts[ts.size() / 2] * 1e3,
         ts[ts.size() / 2] * 1e3 / nk);
  hipFree(p);
}

int main() {
  for (int grid : {64, 256, 512}) {
    run<0>(grid, 101);
    run<1>(grid, 101);
    run<2>(grid, 101);
  }
  graph_chain(100, false, 1);
  graph_chain(100, true, 256);
  graph_chain(100, true, 1024);
  return 0;
}
