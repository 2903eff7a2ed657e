// Lab: what do the two HIP events of hipExtLaunchKernelGGL(start, stop) measure?  A kernel of known length (every wave
// spins on s_memrealtime, 100 MHz, for a requested time; the kernel's own extent = max(end) - min(begin) over all waves,
// collected with atomics) follows a filler kernel 40 times back to back WITHOUT host synchronisation (the queue stays full,
// as in the refinement step); all timers are read at the end.  Modes:
//   0  hipExtLaunchKernelGGL(start, stop), events from hipEventCreate                    (what scf_timer did until r5)
//   1  the same with hipEventDisableSystemFence events (no system-scope release folded into the dispatch's end)
//   2  hipEventRecord before / after
//   3  stop event bound to the launch + hipEventRecord of a second event right behind it: elapsed(stop, second) =
//      second.end - stop.START = the dispatch's own begin -> the marker behind it
//   hipcc --offload-arch=gfx950 -O3 tools/lab/event_overhead.hip -o tools/lab/bin/event_overhead
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(1024) void spin_kernel(unsigned long long* lohi, int slot, int ticks, float* sink) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long t = t0;
  while ((long long)(t - t0) < ticks) { __builtin_amdgcn_s_sleep(2); t = __builtin_amdgcn_s_memrealtime(); }
  if ((threadIdx.x & 63) == 0) { atomicMin(lohi + 2 * slot, t0); atomicMax(lohi + 2 * slot + 1, t); }
  if (ticks < 0) sink[threadIdx.x] = (float)t;
}
__global__ void filler_kernel(float* p, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = p[i] * 1.0001f + 1.f;
}

int main() {
  const int NREP = 40;
  unsigned long long* lohi; float* sink; float* buf;
  hipMalloc(&lohi, 16 * NREP); hipMalloc(&sink, 4096); hipMalloc(&buf, 16 << 20);
  hipStream_t st; hipStreamCreate(&st);
  std::vector<hipEvent_t> ea(NREP), eb(NREP);
  for (int ticks : {0, 500, 1800}) {           // 0 / 5 / 18 us of spinning per wave
    for (int mode = 0; mode < 4; ++mode) {
      for (int i = 0; i < NREP; ++i) {
        if (mode == 1) { hipEventCreateWithFlags(&ea[i], hipEventDisableSystemFence); hipEventCreateWithFlags(&eb[i], hipEventDisableSystemFence); }
        else { hipEventCreate(&ea[i]); hipEventCreate(&eb[i]); }
      }
      std::vector<unsigned long long> init(2 * NREP);
      for (int i = 0; i < NREP; ++i) { init[2 * i] = ~0ull; init[2 * i + 1] = 0; }
      hipMemcpy(lohi, init.data(), 16 * NREP, hipMemcpyHostToDevice);
      hipStreamSynchronize(st);
      for (int rep = 0; rep < NREP; ++rep) {
        filler_kernel<<<1024, 256, 0, st>>>(buf, 1 << 20);        // ~4 us: the launch follows other work, as in the step
        if (mode <= 1) {
          hipExtLaunchKernelGGL(spin_kernel, dim3(256), dim3(1024), 0, st, ea[rep], eb[rep], 0, lohi, rep, ticks, sink);
        } else if (mode == 2) {
          hipEventRecord(ea[rep], st);
          hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(1024), 0, st, lohi, rep, ticks, sink);
          hipEventRecord(eb[rep], st);
        } else {
          hipExtLaunchKernelGGL(spin_kernel, dim3(256), dim3(1024), 0, st, nullptr, ea[rep], 0, lohi, rep, ticks, sink);
          hipEventRecord(eb[rep], st);
        }
        filler_kernel<<<1024, 256, 0, st>>>(buf, 1 << 20);
      }
      hipStreamSynchronize(st);
      std::vector<unsigned long long> h(2 * NREP);
      hipMemcpy(h.data(), lohi, 16 * NREP, hipMemcpyDeviceToHost);
      std::vector<float> ev, own;
      for (int rep = 8; rep < NREP; ++rep) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, ea[rep], eb[rep]);
        ev.push_back(ms * 1e3f); own.push_back((float)(h[2 * rep + 1] - h[2 * rep]) * 0.01f);
      }
      std::sort(ev.begin(), ev.end()); std::sort(own.begin(), own.end());
      const char* names[4] = {"hipExtLaunch(start, stop), default events      ", "hipExtLaunch(start, stop), DisableSystemFence  ",
                              "hipEventRecord before / after                  ", "hipExtLaunch(-, stop) + hipEventRecord behind   "};
      printf("spin %5.1f us  %s: events %6.2f us (median; min %6.2f)   first wave begin -> last wave end %6.2f us\n", ticks * 0.01f, names[mode],
             ev[ev.size() / 2], ev[0], own[own.size() / 2]);
      for (int i = 0; i < NREP; ++i) { hipEventDestroy(ea[i]); hipEventDestroy(eb[i]); }
    }
  }
  return 0;
}
