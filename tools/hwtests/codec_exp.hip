// Stand-alone timing harness for the codec forward pass: compiles musicgan_amd/csrc/codec.hip in place, so a locally modified
// copy can be A/B-timed in one gpurun call.  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DCODEC_EXP_NAME='"BASE"' codec_exp.hip
#include "../../musicgan_amd/csrc/core.hip"
#include "../../musicgan_amd/csrc/codec.hip"
#include <vector>
int main() {
  const int T = 103360;
  float *c, *bark, *m, *p; void* ws;
  const size_t n = (size_t)512 * T;
  hipMalloc(&c, n * 8); hipMalloc(&bark, 512 * 4);
  const int S = (T - 1) / 512;
  hipMalloc(&m, (size_t)S * 512 * 512 * 4); hipMalloc(&p, (size_t)S * 512 * 512 * 4);
  const size_t wsb = mg_codec_fwd_ws_bytes(T);
  hipMalloc(&ws, wsb);
  std::vector<float> h(n * 2);
  unsigned s = 12345u;
  for (size_t i = 0; i < n * 2; ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)(s >> 8) / 8388608.f - 1.f; }
  hipMemcpy(c, h.data(), n * 8, hipMemcpyHostToDevice);
  for (int i = 0; i < 512; ++i) h[i] = 1.f;
  hipMemcpy(bark, h.data(), 512 * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  if (mg_codec_fwd(c, bark, m, p, ws, wsb, T, 512, nullptr) != 0) { printf("err %s\n", mg_last_error()); return 1; }
  hipEventRecord(e0, nullptr);
  for (int i = 0; i < 5; ++i) mg_codec_fwd(c, bark, m, p, ws, wsb, T, 512, nullptr);
  hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-20s codec fwd %.3f ms\n", CODEC_EXP_NAME, ms / 5);
  return 0;
}
