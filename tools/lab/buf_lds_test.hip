// Lab: semantics of `buffer_load_dwordx4 ... offen lds` for out-of-range lanes (raw buffer, stride 0):
// does the LDS destination of an out-of-range lane receive zeros, or is it left untouched?
//   hipcc --offload-arch=gfx950 -O3 -w tools/lab/buf_lds_test.hip -o tools/lab/bin/buf_lds_test
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* src, float* out, int nbytes) {
  __shared__ __attribute__((aligned(16))) float lds[256];
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) lds[i] = -7.0f;        // sentinel
  __syncthreads();
  const unsigned long long a = (unsigned long long)src;
  i32x4 rsrc;
  rsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  rsrc[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xFFFFu));
  rsrc[2] = __builtin_amdgcn_readfirstlane(nbytes);
  rsrc[3] = 0x00020000;
  // lanes 0..31 in range; 32..47 past num_records; 48..63 offset 0xFFFFFFF0 (wrap check)
  unsigned voff = lane * 16;
  if (lane >= 48) voff = 0xFFFFFFF0u;
  const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)lds;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %0, 0 offen lds\n\ts_waitcnt vmcnt(0)"
               : : "s"(rsrc), "v"(voff), "s"(lbase) : "memory");
  __syncthreads();
  for (int i = lane; i < 256; i += 64) out[i] = lds[i];
}
int main() {
  float *src, *out; CK(hipMalloc(&src, 4096)); CK(hipMalloc(&out, 1024));
  float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (float)(i + 1);
  CK(hipMemcpy(src, h, 4096, hipMemcpyHostToDevice));
  k<<<1, 64>>>(src, out, 32 * 16);
  CK(hipDeviceSynchronize());
  float o[256]; CK(hipMemcpy(o, out, 1024, hipMemcpyDeviceToHost));
  for (int l = 0; l < 64; l += 1) if (l < 2 || (l >= 30 && l < 34) || (l >= 46 && l < 50) || l == 63)
    printf("lane %2d -> lds %g %g %g %g\n", l, o[4 * l], o[4 * l + 1], o[4 * l + 2], o[4 * l + 3]);
  return 0;
}
