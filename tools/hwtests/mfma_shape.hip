// fp32 MFMA shape experiment for the Winograd kernel's inner loop: the same FLOPs per wave and the same
// output tile (32 out-channels x 32 tiles of one component pair, K = 8 channels per "chunk") issued as
//   A: v_mfma_f32_16x16x4_f32, 2 x 2 tiles  -- per K=4: 2 A fragments + 2 B fragments (4 operand registers), 4 MFMAs of 32 cycles
//   B: v_mfma_f32_32x32x2_f32, one tile     -- per K=4: 2 A + 2 B fragments (4 operand registers), 2 MFMAs of 64 cycles
//   C: the product kernel's arrangement, NIW = 2: 2 x 1 tiles of 16x16x4 (3 operand registers per 2 MFMAs), half the tile per wave
// with every operand re-read from LDS by ds_read_b128 inside the loop (as the kernel does), 8 waves per workgroup, one workgroup
// per CU, random operands.  Second part: J independent v_pk_add_f32 per MFMA group in the same wave (the kernel's staging /
// epilogue work is vector work that has to fit somewhere).  Prints time per launch and TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE, int J>
__global__ void __launch_bounds__(512) k(float* out, const float* in, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = in[i];
  __syncthreads();
  const float* base = lds + wave * 1024 + lane * 4;
  f32x2 v[8];
  for (int j = 0; j < 8; ++j) v[j] = f32x2{(float)lane, (float)j};
  float r = 0.f;
  if (MODE == 0) {  // 16x16x4, 2 x 2 tiles, 8 components' worth of accumulators = 8 x 4 tiles
    f32x4 acc[8][4];
    for (int c = 0; c < 8; ++c) for (int t = 0; t < 4; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(base + ((c + it) & 7) * 256);           // 2 A fragments x 2 k-steps
        const f32x4 b = *reinterpret_cast<const f32x4*>(base + 8192 + ((c + it) & 7) * 256);    // 2 B fragments x 2 k-steps
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int t = 0; t < 4; ++t)
            acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks * 2 + (t >> 1)], b[ks * 2 + (t & 1)], acc[c][t], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < J; ++j) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(v[(j + 1) & 7]));
      }
    }
    for (int c = 0; c < 8; ++c) for (int t = 0; t < 4; ++t) r += acc[c][t][0] + acc[c][t][3];
  } else if (MODE == 1) {  // 32x32x2, one tile per component
    f32x16 acc[8];
    for (int c = 0; c < 8; ++c) for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(base + ((c + it) & 7) * 256);
        const f32x4 b = *reinterpret_cast<const f32x4*>(base + 8192 + ((c + it) & 7) * 256);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks], b[ks], acc[c], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < J; ++j) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(v[(j + 1) & 7]));
      }
    }
    for (int c = 0; c < 8; ++c) r += acc[c][0] + acc[c][15];
  } else {  // the product arrangement: NIW = 2 -> 2 x 1 tiles, 16 components, K = 8 per pass: 1 B + 2 A reads per component pair
    f32x4 acc[16][2];
    for (int c = 0; c < 16; ++c) for (int t = 0; t < 2; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int cp = 0; cp < 8; ++cp) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(base + ((cp + it) & 7) * 256);
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(base + 8192 + ((cp + it) & 7) * 256);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(base + 8192 + 2048 + ((cp + it) & 7) * 256);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int par = 0; par < 2; ++par) {
            acc[2 * cp + par][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[ks * 2 + par], b[ks * 2 + par], acc[2 * cp + par][0], 0, 0, 0);
            acc[2 * cp + par][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[ks * 2 + par], b[ks * 2 + par], acc[2 * cp + par][1], 0, 0, 0);
          }
#pragma unroll
        for (int j = 0; j < J; ++j) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(v[(j + 1) & 7]));
      }
    }
    for (int c = 0; c < 16; ++c) r += acc[c][0][0] + acc[c][1][3];
  }
  for (int j = 0; j < 8; ++j) r += v[j][0];
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = r;
}

template <int MODE, int J>
static void run(float* d, const float* in, const char* what) {
  const int iters = 400;
  hipFuncSetAttribute((const void*)k<MODE, J>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<MODE, J>), dim3(256), dim3(512), 100 * 1024, 0, d, in, iters);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<MODE, J>), dim3(256), dim3(512), 100 * 1024, 0, d, in, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
  // FLOPs per wave and iteration: MODE 0/1: 8 comps x 32x32 x K=8 x 2 = 131072; MODE 2: 16 comps x 32x16 x K=8 x 2 = 131072
  const double flop = 131072.0 * iters * 8 * 256;
  printf("%-58s J=%d : %.3f ms  %.1f TFLOP/s\n", what, J, ms, flop / ms / 1e9);
}
int main() {
  float *d, *in; hipMalloc(&d, 256 * 512 * 4); hipMalloc(&in, 16384 * 4);
  float* h = (float*)malloc(16384 * 4);
  srand(1);
  for (int i = 0; i < 16384; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
  hipMemcpy(in, h, 16384 * 4, hipMemcpyHostToDevice);
  run<0, 0>(d, in, "16x16x4, 2x2 tiles (4 operand regs / 4 MFMAs)");
  run<1, 0>(d, in, "32x32x2, one tile  (4 operand regs / 2 MFMAs)");
  run<2, 0>(d, in, "16x16x4, 2x1 tiles as in wino3x3_mfma<2,..> (3 / 2)");
  run<0, 2>(d, in, "16x16x4, 2x2 tiles"); run<1, 2>(d, in, "32x32x2"); run<2, 1>(d, in, "16x16x4, 2x1 tiles (J per 4 MFMAs)");
  run<0, 6>(d, in, "16x16x4, 2x2 tiles"); run<1, 6>(d, in, "32x32x2"); run<2, 3>(d, in, "16x16x4, 2x1 tiles (J per 4 MFMAs)");
  run<0, 12>(d, in, "16x16x4, 2x2 tiles"); run<1, 12>(d, in, "32x32x2"); run<2, 6>(d, in, "16x16x4, 2x1 tiles (J per 4 MFMAs)");
  return 0;
}
