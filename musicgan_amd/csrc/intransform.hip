// The per-batch input transform in front of the training step, fused on the device:
//   ChannelMinMaxNorm -> ChangeRange(-1, 1) -> Resize(S)      /root/reference/music_gan/audio/transforms.py:4-40,
//   /root/reference/music_gan/utils.py:70-86 (Grower.__get_transform), applied at train.py:138-140 on the CPU per batch.
// Resize is torchvision's tensor path = aten upsample_bilinear2d_aa (bilinear, antialias, align_corners = False), restated
// from its published definition: per output index i, centre = scale (i + 1/2), support = scale (for scale >= 1, else 1),
// taps j in [xmin, xmin + xsize), weight = max(0, 1 - |(j - centre + 1/2) / max(scale, 1)|) normalised by their sum; the
// horizontal pass runs first, then the vertical one, with a float32 intermediate (the CPU kernel's separable order).
// Input: the (N, 2, H, W) batch as the DataLoader delivers it -- float64 (the dataset's storage type, cast to float32 on read,
// exactly x.to(th.float)) or float32.  Three launches: partial min/max per plane, horizontal pass (applies the normalisation to
// every tap as the reference does before resizing), vertical pass.  HBM-bound: the batch is read twice, everything else is small.
#include "mg_common.h"

namespace {

constexpr int MMP = 16;  // min/max partial blocks per (n, c) plane

template <typename T>
__global__ void __launch_bounds__(256) it_minmax_part(const T* __restrict__ x, float* __restrict__ part, size_t plane) {
  __shared__ float red[16];
  const int nc = blockIdx.x, p = blockIdx.y;
  const T* xp = x + (size_t)nc * plane;
  const size_t per = (plane + MMP - 1) / MMP;
  const size_t lo = (size_t)p * per, hi = lo + per < plane ? lo + per : plane;
  float mn = INFINITY, mx = -INFINITY;
  for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
    const float v = (float)xp[i];
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, d));
    mx = fmaxf(mx, __shfl_xor(mx, d));
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    red[wave] = mn;
    red[8 + wave] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k) {
      mn = fminf(mn, red[k]);
      mx = fmaxf(mx, red[8 + k]);
    }
    part[((size_t)nc * MMP + p) * 2] = mn;
    part[((size_t)nc * MMP + p) * 2 + 1] = mx;
  }
}

// weights of output index i along an axis of `in` samples resized to `out`: returns xmin, xsize and 1 / (sum of raw weights)
__device__ __forceinline__ void aa_span(int i, int in, float scale, int& xmin, int& xsize, float& centre, float& invscale,
                                        float& inv_total) {
  const float support = scale >= 1.f ? scale : 1.f;
  centre = scale * ((float)i + 0.5f);
  invscale = scale >= 1.f ? 1.f / scale : 1.f;
  const long long a = (long long)(centre - support + 0.5f);
  xmin = a > 0 ? (int)a : 0;
  const long long b = (long long)(centre + support + 0.5f);
  xsize = (int)(b < in ? b : in) - xmin;
  float total = 0.f;
  for (int j = 0; j < xsize; ++j) {
    float t = ((float)(j + xmin) - centre + 0.5f) * invscale;
    t = t < 0.f ? -t : t;
    total += t < 1.f ? 1.f - t : 0.f;
  }
  inv_total = total != 0.f ? total : 1.f;  // the caller divides (weights are w / total, as aten stores them)
}

// horizontal pass: tmp[nc][y][j] = sum_b w[j][b] * norm(x[nc][y][xmin_j + b]),  norm(v) = (v - mn) / (mx - mn + eps) * 2 - 1
template <typename T>
__global__ void __launch_bounds__(256) it_resize_h(const T* __restrict__ x, const float* __restrict__ part, float* __restrict__ tmp,
                                                   int H, int W, int S, float eps, size_t total) {
  const float scale = (float)W / (float)S;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int j = (int)(e % S);
    const size_t r = e / S;
    const int y = (int)(r % H);
    const int nc = (int)(r / H);
    float mn = INFINITY, mx = -INFINITY;
#pragma unroll
    for (int p = 0; p < MMP; ++p) {
      mn = fminf(mn, part[((size_t)nc * MMP + p) * 2]);
      mx = fmaxf(mx, part[((size_t)nc * MMP + p) * 2 + 1]);
    }
    const float den = mx - mn + eps;
    int xmin, xsize;
    float centre, invscale, tw;
    aa_span(j, W, scale, xmin, xsize, centre, invscale, tw);
    const T* row = x + ((size_t)nc * H + y) * W + xmin;
    float acc = 0.f;
    for (int b = 0; b < xsize; ++b) {
      float t = ((float)(b + xmin) - centre + 0.5f) * invscale;
      t = t < 0.f ? -t : t;
      const float w = (t < 1.f ? 1.f - t : 0.f) / tw;
      const float v = ((float)row[b] - mn) / den * 2.f + -1.f;
      acc += v * w;
    }
    tmp[e] = acc;
  }
}

// vertical pass: out[nc][i][j] = sum_a w[i][a] * tmp[nc][ymin_i + a][j]
__global__ void __launch_bounds__(256) it_resize_v(const float* __restrict__ tmp, float* __restrict__ out, int H, int S,
                                                   size_t total) {
  const float scale = (float)H / (float)S;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int j = (int)(e % S);
    const size_t r = e / S;
    const int i = (int)(r % S);
    const int nc = (int)(r / S);
    int ymin, ysize;
    float centre, invscale, tw;
    aa_span(i, H, scale, ymin, ysize, centre, invscale, tw);
    const float* col = tmp + ((size_t)nc * H + ymin) * S + j;
    float acc = 0.f;
    for (int a = 0; a < ysize; ++a) {
      float t = ((float)(a + ymin) - centre + 0.5f) * invscale;
      t = t < 0.f ? -t : t;
      const float w = (t < 1.f ? 1.f - t : 0.f) / tw;
      acc += col[(size_t)a * S] * w;
    }
    out[e] = acc;
  }
}

template <typename T>
int run_it(const T* x, float* out, float* part, float* tmp, int N, int H, int W, int S, float eps, hipStream_t s) {
  const size_t plane = (size_t)H * W;
  hipLaunchKernelGGL(it_minmax_part<T>, dim3(N * 2, MMP), dim3(256), 0, s, x, part, plane);
  const size_t t1 = (size_t)N * 2 * H * S, t2 = (size_t)N * 2 * S * S;
  int b1 = (int)((t1 + 255) / 256), b2 = (int)((t2 + 255) / 256);
  if (b1 > 16384) b1 = 16384;
  if (b2 > 16384) b2 = 16384;
  hipLaunchKernelGGL(it_resize_h<T>, dim3(b1), dim3(256), 0, s, x, part, tmp, H, W, S, eps, t1);
  hipLaunchKernelGGL(it_resize_v, dim3(b2), dim3(256), 0, s, tmp, out, H, S, t2);
  MG_CHECK_LAUNCH("mg_input_transform");
  return MG_OK;
}

}  // namespace

extern "C" size_t mg_input_transform_ws_bytes(int N, int H, int W, int S) {
  (void)W;
  return ((size_t)N * 2 * MMP * 2 + (size_t)N * 2 * H * S) * sizeof(float);
}

extern "C" int mg_input_transform(const void* x, int x_is_f64, float* out, void* ws, size_t ws_bytes, int N, int H, int W, int S,
                                  float eps, mg_stream_t stream) {
  MG_CHECK_ARG(x && out && ws && N > 0 && H > 0 && W > 0 && S > 0, "mg_input_transform: bad arguments");
  MG_CHECK_ARG(S <= H && S <= W, "mg_input_transform: only down-sampling (S=%d from %dx%d) is implemented", S, H, W);
  if (ws_bytes < mg_input_transform_ws_bytes(N, H, W, S)) {
    mg_set_error("mg_input_transform: workspace too small");
    return MG_EWORKSPACE;
  }
  float* part = reinterpret_cast<float*>(ws);
  float* tmp = part + (size_t)N * 2 * MMP * 2;
  hipStream_t s = (hipStream_t)stream;
  if (x_is_f64) return run_it(reinterpret_cast<const double*>(x), out, part, tmp, N, H, W, S, eps, s);
  return run_it(reinterpret_cast<const float*>(x), out, part, tmp, N, H, W, S, eps, s);
}
