// Does VALU work of a second wave overlap the fp32 MFMAs of the first wave on the same SIMD?  8 waves per workgroup, one
// workgroup per CU (LDS-limited): waves 0-3 run `na` MFMA iterations (8 independent accumulators), waves 4-7 run `nb` iterations
// of 8 independent v_fma chains (or LDS reads with mode 2).  Times: MFMA only, VALU only, both.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(512) k(float* out, int na, int nb, int mode) {
  extern __shared__ float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  lds[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  float r = 0.f;
  if (wave < 4) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float a = (float)lane * 1e-3f, b = 1.0f + (float)lane * 1e-4f;
    for (int it = 0; it < na; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else if (mode == 1) {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = (float)(lane + i);
    for (int it = 0; it < nb; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 0.999f, 0.5f);
    for (int i = 0; i < 8; ++i) r += v[i];
  } else if (mode == 2) {
    float s = 0.f;
    for (int it = 0; it < nb; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) s += lds[(lane * 4 + i * 64 + it) & 8191];
    r = s;
  }
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = r;
}
static float run(float* d, int na, int nb, int mode) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(256), dim3(512), 100 * 1024, 0, d, na, nb, mode);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), 100 * 1024, 0, d, na, nb, mode);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}
int main() {
  float* d; hipMalloc(&d, 256 * 512 * 4);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
  const int NA = 4000, NB = 8000;  // 32000 MFMAs (32 cyc each) ~ 1.02 M cycles; 64000 v_fma (4 cyc each) ~ 0.26 M cycles
  printf("MFMA only            : %.3f ms\n", run(d, NA, 0, 1));
  printf("VALU only            : %.3f ms\n", run(d, 0, NB, 1));
  printf("MFMA + VALU          : %.3f ms\n", run(d, NA, NB, 1));
  printf("VALU x4 only         : %.3f ms\n", run(d, 0, 4 * NB, 1));
  printf("MFMA + VALU x4       : %.3f ms\n", run(d, NA, 4 * NB, 1));
  printf("LDS reads only       : %.3f ms\n", run(d, 0, NB, 2));
  printf("MFMA + LDS reads     : %.3f ms\n", run(d, NA, NB, 2));
  return 0;
}
