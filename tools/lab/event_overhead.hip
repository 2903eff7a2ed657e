// Lab: what do the two HIP events of hipExtLaunchKernelGGL(start, stop) measure?  A kernel of known length (every wave
// spins on s_memrealtime, 100 MHz, for a requested time; the kernel's own extent = max(end) - min(begin) over all waves,
// collected with atomics) timed (a) with two distinct events, (b) with ONE event passed as start and stop (both then refer
// to the dispatch itself: hipEventElapsedTime returns its end - begin stamps), (c) hipEventRecord before / after.
//   hipcc --offload-arch=gfx950 -O3 tools/lab/event_overhead.hip -o tools/lab/bin/event_overhead
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(1024) void spin_kernel(unsigned long long* lo, unsigned long long* hi, int ticks, float* sink) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long t = t0;
  while ((long long)(t - t0) < ticks) { __builtin_amdgcn_s_sleep(2); t = __builtin_amdgcn_s_memrealtime(); }
  if ((threadIdx.x & 63) == 0) { atomicMin(lo, t0); atomicMax(hi, t); }
  if (ticks < 0) sink[threadIdx.x] = (float)t;
}
__global__ void filler_kernel(float* p, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = p[i] * 1.0001f + 1.f;
}

int main() {
  unsigned long long *lo, *hi; float* sink; float* buf;
  hipMalloc(&lo, 8); hipMalloc(&hi, 8); hipMalloc(&sink, 4096); hipMalloc(&buf, 64 << 20);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1, es, r0, r1;
  hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&es); hipEventCreate(&r0); hipEventCreate(&r1);
  for (int ticks : {0, 500, 1800}) {           // 0 / 5 / 18 us of spinning per wave
    for (int mode = 0; mode < 3; ++mode) {
      std::vector<float> ev, own;
      for (int rep = 0; rep < 40; ++rep) {
        const unsigned long long big = ~0ull, zero = 0;
        hipMemcpyAsync(lo, &big, 8, hipMemcpyHostToDevice, st); hipMemcpyAsync(hi, &zero, 8, hipMemcpyHostToDevice, st);
        filler_kernel<<<2048, 256, 0, st>>>(buf, 16 << 20);        // the launch follows other work, as in the step
        float ms = 0.f;
        if (mode == 0) {
          hipExtLaunchKernelGGL(spin_kernel, dim3(256), dim3(1024), 0, st, e0, e1, 0, lo, hi, ticks, sink);
          hipStreamSynchronize(st); hipEventElapsedTime(&ms, e0, e1);
        } else if (mode == 1) {
          hipExtLaunchKernelGGL(spin_kernel, dim3(256), dim3(1024), 0, st, es, es, 0, lo, hi, ticks, sink);
          hipStreamSynchronize(st); hipEventElapsedTime(&ms, es, es);
        } else {
          hipEventRecord(r0, st);
          hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(1024), 0, st, lo, hi, ticks, sink);
          hipEventRecord(r1, st);
          hipStreamSynchronize(st); hipEventElapsedTime(&ms, r0, r1);
        }
        unsigned long long a, b;
        hipMemcpy(&a, lo, 8, hipMemcpyDeviceToHost); hipMemcpy(&b, hi, 8, hipMemcpyDeviceToHost);
        if (rep >= 8) { ev.push_back(ms * 1e3f); own.push_back((float)(b - a) * 0.01f); }
      }
      std::sort(ev.begin(), ev.end()); std::sort(own.begin(), own.end());
      const char* names[3] = {"hipExtLaunch(start, stop) two events", "hipExtLaunch(e, e) one event       ", "hipEventRecord before / after       "};
      printf("spin %5.1f us  %s: events %6.2f us (median; min %6.2f)   first wave begin -> last wave end %6.2f us\n", ticks * 0.01f, names[mode],
             ev[ev.size() / 2], ev[0], own[own.size() / 2]);
    }
  }
  return 0;
}
