// How long does a wave's VALU work take when the other wave on its SIMD issues fp32 MFMAs back to back?
// Part 1: workgroup = 8 waves (one per CU): waves 0-3 run MFMAs until waves 4-7 have finished `nb` rounds of VALU work and
// raised a flag in LDS; the VALU waves time themselves with s_memtime.  chains = independent v_fma chains (1 = fully dependent).
// prio: s_setprio level of the VALU waves (MFMA waves stay at 0).
// Part 2: ONE wave per SIMD interleaves J v_fma after every MFMA in its own instruction stream; ticks per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CH, int PRIO>
__global__ void __launch_bounds__(512) k(float* out, long long* cyc, int nb, int with_mfma) {
  __shared__ volatile int done;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x == 0) done = 0;
  __syncthreads();
  float r = 0.f;
  if (wave < 4) {
    if (with_mfma) {
      f32x4 acc[8];
      for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float a = (float)lane * 1e-3f, b = 1.0f + (float)lane * 1e-4f;
      int guard = 0;
      while (done < 4 && guard < 200000) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        ++guard;
      }
      for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
  } else {
    __builtin_amdgcn_s_setprio(PRIO);
    float v[CH];
    for (int i = 0; i < CH; ++i) v[i] = (float)(lane + i);
    const long long t0 = clock64();
    for (int it = 0; it < nb; ++it)
#pragma unroll
      for (int rep = 0; rep < 8 / CH; ++rep)
#pragma unroll
        for (int i = 0; i < CH; ++i) v[i] = __builtin_fmaf(v[i], 0.999f, 0.5f);
    for (int i = 0; i < CH; ++i) r += v[i];
    const long long t1 = clock64();
    if (lane == 0) {
      atomicAdd((int*)&done, 1);
      if (blockIdx.x == 0 && wave == 4) cyc[0] = t1 - t0;
    }
  }
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = r;
}
template <int J>
__global__ void __launch_bounds__(256) k2(float* out, long long* cyc, int n) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = (float)(lane + i);
  const float a = (float)lane * 1e-3f, b = 1.0f + (float)lane * 1e-4f;
  const long long t0 = clock64();
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < J; ++j) v[(i + j) & 7] = __builtin_fmaf(v[(i + j) & 7], 0.999f, 0.5f);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, J, 0);  // J VALU
    }
  }
  const long long t1 = clock64();
  float r = 0.f;
  for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int CH, int PRIO>
static void run(float* d, long long* c, int nb) {
  for (int m = 0; m < 2; ++m) {
    hipLaunchKernelGGL((k<CH, PRIO>), dim3(256), dim3(512), 0, 0, d, c, nb, m);
    (void)hipDeviceSynchronize();
    long long h; (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("chains=%d prio=%d %-12s: %8lld ticks for %d v_fma = %.2f ticks/instr\n", CH, PRIO, m ? "under MFMA" : "alone", h, nb * 8, (double)h / (nb * 8));
  }
}
template <int J>
static void run2(float* d, long long* c) {
  const int n = 2000;
  hipLaunchKernelGGL((k2<J>), dim3(256), dim3(256), 0, 0, d, c, n);
  (void)hipDeviceSynchronize();
  long long h; (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("same wave, %d v_fma after each MFMA: %.2f ticks per MFMA\n", J, (double)h / (n * 8));
}
int main() {
  float* d; long long* c; (void)hipMalloc(&d, 256 * 512 * 4); (void)hipMalloc(&c, 8);
  run<1, 0>(d, c, 2000); run<4, 0>(d, c, 2000); run<8, 0>(d, c, 2000);
  run<1, 3>(d, c, 2000); run<4, 3>(d, c, 2000); run<8, 3>(d, c, 2000);
  run2<0>(d, c); run2<1>(d, c); run2<2>(d, c); run2<4>(d, c); run2<6>(d, c); run2<8>(d, c);
  return 0;
}
