// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access types of the lookup kernel
// (VERDICT r1 item 3): dword global_load_lds GATHERS with `nt`, and dword stores with `sc1`.
//   hipcc --offload-arch=gfx950 -O3 tools/lab/fetch_calib.hip -o tools/lab/bin/fetch_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d DIR -o f -- tools/lab/bin/fetch_calib
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d DIR -o w -- tools/lab/bin/fetch_calib
// Every kernel touches a KNOWN set of bytes of a 1 GiB buffer (4x the Infinity Cache) exactly once:
//   calib_f4_stream      float4 loads, contiguous          -> the guide's reference pattern (reports 1/2)
//   calib_dma_contig     dword LDS-DMA nt, contiguous      -> counting factor of the lookup's load type
//   calib_dma_sector64   dword LDS-DMA nt, ONE dword per 64-byte sector, every sector
//   calib_dma_line128    dword LDS-DMA nt, ONE dword per 128-byte line, every line
//   calib_dma_half128    dword LDS-DMA nt, 40 contiguous bytes at the start of every 128-byte line
//                        (the lookup's row-major level-0 access shape)
//   calib_store_plain / calib_store_sc1   dword stores, contiguous 128-byte lines per half-wave
// tools/summarize_pmc.py --calib turns the two CSVs into bytes-per-reported-byte for each pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void dma_dword_nt(const void* sbase, unsigned voff, unsigned lds, unsigned long long mask) {
  asm volatile("s_mov_b64 exec, %3\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %0 nt\n\ts_mov_b64 exec, -1"
               : : "s"(sbase), "v"(voff), "s"(lds), "s"(mask) : "memory");
}
__device__ __forceinline__ const char* sptr(const char* p) {
  const unsigned long long u = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)u);          // unsigned: no sign extension
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return (const char*)(((unsigned long long)hi << 32) | lo);
}

__global__ __launch_bounds__(256) void calib_f4_stream(const float4* p, size_t n4, float* sink) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = p[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) *sink = acc;
}

// every wave: NI instructions per iteration, lane byte offset = lane*LANE_STRIDE (+ instr*64*LANE_STRIDE);
// LANES of the 64 lanes are active and read their dword at lane offset
template <int LANE_STRIDE, int LANES_PER_GROUP, int GROUP>
__global__ __launch_bounds__(256) void calib_dma(const char* buf, size_t bytes, float* sink) {
  __shared__ float lds[4 * 64 * 8];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t per_instr = (size_t)64 * LANE_STRIDE;
  const size_t ninstr = bytes / per_instr;
  const size_t w = (size_t)blockIdx.x * 4 + wave, nw = (size_t)gridDim.x * 4;
  // lanes: GROUP consecutive lanes form a group, the first LANES_PER_GROUP of them are active
  const unsigned long long mask = __ballot((lane % GROUP) < LANES_PER_GROUP);
  const unsigned voff = (unsigned)lane * LANE_STRIDE;
  const unsigned l0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds + wave * 64 * 8 * 4;
  for (size_t i = w * 8; i + 8 <= ninstr; i += nw * 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) dma_dword_nt(sptr(buf + (i + u) * per_instr), voff, l0 + u * 256, mask);
    __builtin_amdgcn_s_waitcnt(0x0F70);
  }
  __syncthreads();
  if (lds[threadIdx.x] == 123.456f) *sink = 1.f;
}

template <int MODE>
__global__ __launch_bounds__(256) void calib_store(char* buf, size_t bytes) {
  const size_t n = bytes / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v = (float)i;
    if (MODE == 0) ((float*)buf)[i] = v;
    else asm volatile("global_store_dword %0, %1, off sc1" : : "v"((float*)buf + i), "v"(v) : "memory");
  }
}

int main() {
  const size_t bytes = (size_t)1 << 30;
  char* buf; float* sink;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(buf, 0, bytes));
  CK(hipDeviceSynchronize());
  for (int rep = 0; rep < 2; ++rep) {
    calib_f4_stream<<<4096, 256>>>((const float4*)buf, bytes / 16, sink);
    calib_dma<4, 1, 1><<<2048, 256>>>(buf, bytes, sink);        // contig: every lane, 4-byte stride
    calib_dma<64, 1, 1><<<2048, 256>>>(buf, bytes, sink);       // one dword per 64-byte sector
    calib_dma<128, 1, 1><<<2048, 256>>>(buf, bytes, sink);      // one dword per 128-byte line
    calib_dma<4, 10, 32><<<2048, 256>>>(buf, bytes, sink);      // 40 contiguous bytes per 128-byte line
    calib_store<0><<<4096, 256>>>(buf, bytes);
    calib_store<1><<<4096, 256>>>(buf, bytes);
    CK(hipDeviceSynchronize());
  }
  printf("fetch_calib: buffer %zu bytes; patterns: f4_stream reads all; dma<4,1,1> all; dma<64,1,1> 1/16 (every 64-B sector); "
         "dma<128,1,1> 1/32 (every 128-B line); dma<4,10,32> 40 B of every 128-B line; stores write all\n", bytes);
  return 0;
}
