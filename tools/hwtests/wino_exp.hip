// Stand-alone timing harness for the Winograd conv kernel: compiles musicgan_amd/csrc/wino3x3.hip in place (so a locally
// modified copy, or one built with extra -D switches, can be A/B-timed in a single gpurun call without touching the library)
// and times mg_wino3x3 on random data with HIP events.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DWINO_EXP_NAME='"BASE"' wino_exp.hip -o wino_exp_BASE
//   ./wino_exp_BASE N Cin Cout H W flags
#include "../../musicgan_amd/csrc/core.hip"
#include "../../musicgan_amd/csrc/wino3x3.hip"
#include <vector>
int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 64, Cin = argc > 2 ? atoi(argv[2]) : 48, Cout = argc > 3 ? atoi(argv[3]) : 64;
  const int H = argc > 4 ? atoi(argv[4]) : 128, W = argc > 5 ? atoi(argv[5]) : 128, flags = argc > 6 ? atoi(argv[6]) : 1;
  const size_t nx = (size_t)N * Cin * H * W, ny = (size_t)N * Cout * H * W;
  float *x, *w, *up, *y, *p, *b, *aux;
  hipMalloc(&x, nx * 4); hipMalloc(&y, ny * 4); hipMalloc(&p, ny * 4); hipMalloc(&aux, ny * 4); hipMemset(aux, 0x3f, ny * 4); hipMalloc(&w, (size_t)Cout * Cin * 9 * 4); hipMalloc(&b, Cout * 4);
  hipMalloc(&up, mg_wino3x3_packed_floats(Cin, Cout) * 4);
  std::vector<float> h(nx);
  for (size_t i = 0; i < nx; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  hipMemcpy(x, h.data(), nx * 4, hipMemcpyHostToDevice);
  hipMemcpy(w, h.data(), (size_t)Cout * Cin * 9 * 4, hipMemcpyHostToDevice);
  hipMemcpy(b, h.data(), Cout * 4, hipMemcpyHostToDevice);
  if (mg_wino3x3_pack(w, up, Cout, Cin, 0, nullptr) != 0) { printf("pack: %s\n", mg_last_error()); return 1; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i)
    if (mg_wino3x3(x, up, (flags & 4) ? nullptr : b, (flags & 4) ? aux : nullptr, y, p, nullptr, N, Cin, Cout, H, W, flags, 0.2f, nullptr) != 0) { printf("run: %s\n", mg_last_error()); return 1; }
  hipEventRecord(e0, nullptr);
  const int it = 20;
  for (int i = 0; i < it; ++i) mg_wino3x3(x, up, (flags & 4) ? nullptr : b, (flags & 4) ? aux : nullptr, y, p, nullptr, N, Cin, Cout, H, W, flags, 0.2f, nullptr);
  hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
  const double fl = 2.0 * 9 * Cin * Cout * (double)H * W * N;
  printf("%-28s N=%d %d->%d @%dx%d flags=%d: %.3f ms  alg %.1f TF  mfma %.1f TF\n", WINO_EXP_NAME, N, Cin, Cout, H, W, flags, ms, fl / ms / 1e9, fl / 2.25 / ms / 1e9);
  return 0;
}
