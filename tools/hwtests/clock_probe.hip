// What clock does a lightly loaded chip run at, and what does a v_mfma_f32_16x16x4_f32 cost with 1 / 2 / 4 / 8 independent
// accumulators per wave?  One wave per SIMD (256-thread workgroups), grids of 8 .. 2048 workgroups; per launch the first wave
// reports shader cycles (s_getreg SHADER_CYCLES, 20 bit) and the constant 100 MHz counter (s_memtime) around 2048 MFMA groups.
// hipcc --offload-arch=gfx950 -O3 tools/hwtests/clock_probe.hip -o tools/hwtests/clock_probe && ./tools/hwtests/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CH>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* t, int iters) {
  f32x4 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float a = (float)threadIdx.x, b = 1.0f / (1 + threadIdx.x);
  const unsigned long long m0 = __builtin_amdgcn_s_memtime();
  const unsigned c0 = __builtin_amdgcn_s_getreg((20 - 1) << 11 | 29);  // HW_REG_SHADER_CYCLES, 20 bits
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8 / CH; ++r)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
  }
  const unsigned c1 = __builtin_amdgcn_s_getreg((20 - 1) << 11 | 29);
  const unsigned long long m1 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
  for (int c = 0; c < CH; ++c) r += acc[c][0];
  if (r == 12345.f) out[0] = r;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    t[0] = m1 - m0;
    t[1] = (c1 - c0) & 0xFFFFF;
  }
}

template <int CH>
void run(int grid, float* out, unsigned long long* t) {
  const int iters = 256;  // x 8 MFMAs = 2048 per wave: ~65 k cycles, inside the 20-bit counter
  for (int w = 0; w < 50; ++w) hipLaunchKernelGGL(k<CH>, dim3(grid), dim3(256), 0, 0, out, t, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int w = 0; w < 200; ++w) hipLaunchKernelGGL(k<CH>, dim3(grid), dim3(256), 0, 0, out, t, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2];
  hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
  printf("grid %5d  chains %d: %6.2f us/launch | in-kernel: %6.2f us (100 MHz counter), %6llu shader cycles -> %.2f GHz, %.1f cycles per MFMA\n",
         grid, CH, ms * 1e3 / 200, h[0] / 100.0, h[1], h[1] / (h[0] / 100.0) / 1e3, h[1] / 2048.0);
}

int main() {
  float* out;
  unsigned long long* t;
  hipMalloc(&out, 4096);
  hipMalloc(&t, 64);
  for (int grid : {8, 24, 256, 2048}) {
    run<1>(grid, out, t);
    run<2>(grid, out, t);
    run<4>(grid, out, t);
    run<8>(grid, out, t);
  }
  return 0;
}
