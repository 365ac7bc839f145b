// Issue cost (cycles per instruction, one wave per SIMD, 8 independent chains) of the VALU instructions the Winograd staging
// and epilogue are made of.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ void __launch_bounds__(256) k(float* out, long long* cyc, int n) {
  const int lane = threadIdx.x & 63;
  f32x2 v[8];
  for (int i = 0; i < 8; ++i) v[i] = f32x2{(float)(lane + i), (float)(lane - i)};
  const f32x2 c = f32x2{0.999f, 1.001f};
  const long long t0 = clock64();
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i][0]) : "v"(c[0]));
      if (OP == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
      if (OP == 2) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,0]" : "+v"(v[i]) : "v"(c));
      if (OP == 3) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[i][0]) : "v"(v[i][1]));
      if (OP == 4) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[i][0]) : "v"(v[i][1]));
      if (OP == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i][0]) : "v"(c[0]));
      if (OP == 6) asm volatile("v_mov_b32 %0, %1" : "+v"(v[i][0]) : "v"(v[i][1]));
      if (OP == 7) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
      if (OP == 8) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i][0]) : "v"(c[0]));
      if (OP == 9) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i][0]) : "v"(c[0]));
      if (OP == 10) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(v[i]) : "v"(c));
      if (OP == 11) asm volatile("v_mov_b64 %0, %1" : "+v"(v[i]) : "v"(c));
    }
  }
  const long long t1 = clock64();
  float r = 0.f;
  for (int i = 0; i < 8; ++i) r += v[i][0] + v[i][1];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int OP>
static void run(float* d, long long* c, const char* name) {
  const int n = 4000;
  hipLaunchKernelGGL((k<OP>), dim3(256), dim3(256), 0, 0, d, c, n);
  (void)hipDeviceSynchronize();
  long long h; (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("%-44s %.2f cycles/instr\n", name, (double)h / (n * 8));
}
int main() {
  float* d; long long* c; (void)hipMalloc(&d, 256 * 256 * 4); (void)hipMalloc(&c, 8);
  run<0>(d, c, "v_add_f32"); run<1>(d, c, "v_pk_add_f32"); run<2>(d, c, "v_pk_add_f32 op_sel/neg");
  run<3>(d, c, "v_mov_b32_dpp row_shr:1"); run<4>(d, c, "v_mov_b32_dpp wave_shr:1"); run<5>(d, c, "v_cndmask_b32");
  run<6>(d, c, "v_mov_b32"); run<7>(d, c, "v_pk_mul_f32"); run<8>(d, c, "v_fma_f32"); run<9>(d, c, "v_max_f32");
  run<10>(d, c, "v_lshl_add_u64"); run<11>(d, c, "v_mov_b64");
  return 0;
}
