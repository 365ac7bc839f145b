// What `buffer_load_dwordx4 ... lds` (LDS-DMA) writes for lanes whose offset is out of range, and for a 16-byte piece that straddles
// num_records.  hipcc --offload-arch=gfx950 -O2 tools/hwtests/lds_dma_oob.hip -o /tmp/lds_dma_oob && /tmp/lds_dma_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;
__global__ void k(const float* x, float* y, int recs, int soff) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  for (int i = threadIdx.x; i < 256; i += 64) smem[i] = -7.f;  // sentinel
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, recs, 0x00020000);
  int voff = threadIdx.x * 16;
  if ((threadIdx.x & 3) == 1) voff = (int)0x80000000u;  // far out of range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)smem, 16, voff, soff, 0, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) y[i] = smem[i];
}
int main() {
  float *x, *y;
  hipMalloc(&x, 4096); hipMalloc(&y, 1024);
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = (float)(i + 1);
  hipMemcpy(x, h.data(), 4096, hipMemcpyHostToDevice);
  for (int t = 0; t < 3; ++t) {  // 1000 bytes: lane 62's piece (bytes 992..1007) straddles the end; soffset 512: is it range-checked?
    const int recs = t == 0 ? 4096 : 1000, soff = t == 2 ? 512 : 0;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, x, y, recs, soff);
    std::vector<float> o(256);
    hipMemcpy(o.data(), y, 1024, hipMemcpyDeviceToHost);
    printf("num_records %d soffset %d\n", recs, soff);
    for (int l : {0, 1, 2, 5, 28, 30, 31, 32, 61, 62, 63}) printf("  lane %2d: %g %g %g %g\n", l, o[4 * l], o[4 * l + 1], o[4 * l + 2], o[4 * l + 3]);
  }
  return 0;
}
