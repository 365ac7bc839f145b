// Hardware semantics probe for global_load_lds (LDS-DMA): destination = wave-uniform base + lane*size, EXEC-masked lanes skip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* __restrict__ src, float* __restrict__ out, int n) {
  __shared__ __attribute__((aligned(16))) float buf[2][1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) { buf[0][i] = -7.f; buf[1][i] = -7.f; }
  __syncthreads();
  for (int i = 0; i < 4; ++i) {
    const int idx = (wave * 4 + i) * 64 + lane;
    if ((idx % 3) != 0)  // divergent: lanes with idx%3==0 are masked off and write a marker with a normal LDS store
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (n - 1 - idx)),
                                       (__attribute__((address_space(3))) void*)(&buf[0][(wave * 4 + i) * 64]), 4, 0, 0);
    else
      buf[0][idx] = 0.f;
  }
  const int c = wave * 64 + lane;  // 16-byte pieces: lane -> 4 floats
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 4 * c),
                                   (__attribute__((address_space(3))) void*)(&buf[1][wave * 256]), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) { out[i] = buf[0][i]; out[1024 + i] = buf[1][i]; }
}
int main() {
  const int n = 1024;
  std::vector<float> h(n), o(2048);
  for (int i = 0; i < n; ++i) h[i] = (float)i;
  float *d, *dout;
  hipMalloc(&d, n * 4); hipMalloc(&dout, 2048 * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, dout, n);
  hipMemcpy(o.data(), dout, 2048 * 4, hipMemcpyDeviceToHost);
  int bad0 = 0, bad1 = 0;
  for (int i = 0; i < 1024; ++i) {
    const float exp0 = (i % 3) != 0 ? (float)(n - 1 - i) : 0.f;
    if (o[i] != exp0) { if (bad0 < 5) printf("buf0[%d]=%g exp %g\n", i, o[i], exp0); ++bad0; }
    if (o[1024 + i] != (float)i) { if (bad1 < 5) printf("buf1[%d]=%g exp %d\n", i, o[1024 + i], i); ++bad1; }
  }
  printf("dma_lds_test: dword+mask bad=%d, dwordx4 bad=%d\n", bad0, bad1);
  return bad0 + bad1 ? 1 : 0;
}
