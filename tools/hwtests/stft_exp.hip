// Stand-alone timing harness for the STFT kernel (compiles musicgan_amd/csrc/stft.hip in place, so -D switches can be A/B-timed).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DSTFT_EXP_NAME='"BASE"' stft_exp.hip -o stft_exp_BASE
#include "../../musicgan_amd/csrc/core.hip"
#include "../../musicgan_amd/csrc/stft.hip"
#include <vector>
int main() {
  const long long L = 44100ll * 600; const int T = 1 + (int)(L / 256);
  float *wav, *out;
  (void)hipMalloc(&wav, L * 4); (void)hipMalloc(&out, (size_t)512 * T * 8);
  std::vector<float> h(L);
  for (long long i = 0; i < L; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  (void)hipMemcpy(wav, h.data(), L * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) if (mg_stft_1024(wav, out, nullptr, L, nullptr) != 0) { printf("%s\n", mg_last_error()); return 1; }
  (void)hipEventRecord(e0, nullptr);
  for (int i = 0; i < 20; ++i) mg_stft_1024(wav, out, nullptr, L, nullptr);
  (void)hipEventRecord(e1, nullptr); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 20;
  printf("%-20s %.4f ms  %.1f M frames/s  %.2f TB/s algorithmic\n", STFT_EXP_NAME, ms, T / ms / 1e3, T * 5120.0 / ms / 1e9);
  return 0;
}
