// HBM-bound element-wise / reduction kernels of the ProGAN step: PixelNorm fwd/bwd, nearest x2 up-sampling and 2x2 average
// pooling (both directions), LeakyReLU backward, fade-in blends, Linear(160->1), gradient-penalty helpers, fused Adam.
// Reference ops: /root/reference/music_gan/networks/layers.py:11-17, generator.py:26-29,124, discriminator.py:24,103-124,
// 166-184, train.py:64-70.  One thread owns 4 consecutive floats wherever the shape allows (16-byte accesses).
#include <cstdlib>

#include <cstdint>

#include "mg_common.h"

namespace {

constexpr float PN_EPS = 1e-8f;

inline int ew_grid(size_t items) {
  size_t b = (items + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

template <int V>
__device__ __forceinline__ void ld(const float* p, float (&o)[V]) {
  if constexpr (V == 4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
    o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = t[3];
  } else {
#pragma unroll
    for (int v = 0; v < V; ++v) o[v] = p[v];
  }
}
template <int V>
__device__ __forceinline__ void st(float* p, const float (&o)[V]) {
  if constexpr (V == 4) {
    *reinterpret_cast<f32x4*>(p) = f32x4{o[0], o[1], o[2], o[3]};
  } else {
#pragma unroll
    for (int v = 0; v < V; ++v) p[v] = o[v];
  }
}

// ---------------------------------------------------------------- PixelNorm
template <int V>
__global__ void __launch_bounds__(256) pixelnorm_fwd_k(const float* __restrict__ y, float* __restrict__ p,
                                                       float* __restrict__ rn, int N, int C, int HW) {
  const int q = HW / V;
  const size_t total = (size_t)N * q;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / q);
    const int px = (int)(i - (size_t)n * q) * V;
    const float* yp = y + (size_t)n * C * HW + px;
    float ss[V];
#pragma unroll
    for (int v = 0; v < V; ++v) ss[v] = 0.f;
    for (int c = 0; c < C; ++c) {
      float t[V];
      ld<V>(yp + (size_t)c * HW, t);
#pragma unroll
      for (int v = 0; v < V; ++v) ss[v] = fmaf(t[v], t[v], ss[v]);
    }
    float r[V];
#pragma unroll
    for (int v = 0; v < V; ++v) r[v] = 1.0f / sqrtf(ss[v] / (float)C + PN_EPS);
    if (rn) st<V>(rn + (size_t)n * HW + px, r);
    float* pp = p + (size_t)n * C * HW + px;
    for (int c = 0; c < C; ++c) {
      float t[V];
      ld<V>(yp + (size_t)c * HW, t);
#pragma unroll
      for (int v = 0; v < V; ++v) t[v] *= r[v];
      st<V>(pp + (size_t)c * HW, t);
    }
  }
}

// Small maps (the first generator blocks: 2x2 .. 16x16, 32-160 channels): one thread per pixel walking all channels twice is a
// 2 x 160-deep chain of dependent-latency loads in 1-8 workgroups (33-54 us measured for 4-130 KB of data).  Here a workgroup
// takes 64 pixels (lane = pixel: coalesced), its 4 waves split the channels and keep their values in registers (<= 40 each),
// the partial sums of squares meet in LDS, and each wave scales and writes what it holds: one pass, 4x the threads, no chain.
constexpr int PN_SMALL_MAXC = 160;
__global__ void __launch_bounds__(256) pixelnorm_fwd_small_k(const float* __restrict__ y, float* __restrict__ p,
                                                             float* __restrict__ rn, int N, int C, int HW) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t i = (size_t)blockIdx.x * 64 + lane;
  const bool ok = i < (size_t)N * HW;
  const int n = ok ? (int)(i / HW) : 0;
  const int px = ok ? (int)(i - (size_t)n * HW) : 0;
  const float* yp = y + (size_t)n * C * HW + px;
  float v[PN_SMALL_MAXC / 4];
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < PN_SMALL_MAXC / 4; ++k) {
    const int c = wave + 4 * k;
    v[k] = (ok && c < C) ? yp[(size_t)c * HW] : 0.f;
    ss = fmaf(v[k], v[k], ss);
  }
  part[wave][lane] = ss;
  __syncthreads();
  const float r = 1.0f / sqrtf((((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane]) / (float)C + PN_EPS);
  if (ok) {
    if (rn && wave == 0) rn[(size_t)n * HW + px] = r;
    float* pp = p + (size_t)n * C * HW + px;
#pragma unroll
    for (int k = 0; k < PN_SMALL_MAXC / 4; ++k) {
      const int c = wave + 4 * k;
      if (c < C) pp[(size_t)c * HW] = v[k] * r;
    }
  }
}

// Backward through PixelNorm and the LeakyReLU in front of it.  from_p == 0: `y` is the post-LeakyReLU activation, p = y*rn.
// from_p != 0: `y` IS the normalised output p (the pre-norm activation is never stored: sign(p) == sign(y) since rn > 0).
template <int V>
__global__ void __launch_bounds__(256) pixelnorm_lrelu_bwd_k(const float* __restrict__ gp, const float* __restrict__ y,
                                                             const float* __restrict__ rn, float* __restrict__ gpre,
                                                             int N, int C, int HW, float slope, int from_p) {
  const int q = HW / V;
  const size_t total = (size_t)N * q;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / q);
    const int px = (int)(i - (size_t)n * q) * V;
    const size_t base = (size_t)n * C * HW + px;
    float r[V], dot[V], pr[V];
    ld<V>(rn + (size_t)n * HW + px, r);
#pragma unroll
    for (int v = 0; v < V; ++v) {
      dot[v] = 0.f;
      pr[v] = from_p ? 1.f : r[v];
    }
    for (int c = 0; c < C; ++c) {
      float g[V], t[V];
      ld<V>(gp + base + (size_t)c * HW, g);
      ld<V>(y + base + (size_t)c * HW, t);
#pragma unroll
      for (int v = 0; v < V; ++v) dot[v] = fmaf(g[v], t[v] * pr[v], dot[v]);
    }
#pragma unroll
    for (int v = 0; v < V; ++v) dot[v] /= (float)C;
    for (int c = 0; c < C; ++c) {
      float g[V], t[V], o[V];
      ld<V>(gp + base + (size_t)c * HW, g);
      ld<V>(y + base + (size_t)c * HW, t);
#pragma unroll
      for (int v = 0; v < V; ++v) o[v] = mg_lrelu_mask(t[v], slope) * r[v] * (g[v] - t[v] * pr[v] * dot[v]);
      st<V>(gpre + base + (size_t)c * HW, o);
    }
  }
}

// Same arithmetic, one HBM pass instead of two over gp and y: the first sweep parks both in LDS (a thread only ever reads back
// its own column: no barrier, no bank conflict -- LDS as a second register file), the second sweep works from there.
// One pixel per thread; NT threads x C channels x 8 B of LDS (C <= 128; wider layers keep the two-pass kernel).
template <int NT>
__global__ void __launch_bounds__(NT) pixelnorm_lrelu_bwd_lds_k(const float* __restrict__ gp, const float* __restrict__ y,
                                                                const float* __restrict__ rn, float* __restrict__ gpre, int N,
                                                                int C, int HW, float slope, int from_p) {
  extern __shared__ __attribute__((aligned(16))) float park[];  // [2][C][NT]
  float* gs = park;
  float* ts = park + (size_t)C * NT;
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  const size_t total = (size_t)N * HW;
  if (i >= total) return;
  const int n = (int)(i / HW);
  const int px = (int)(i - (size_t)n * HW);
  const size_t base = (size_t)n * C * HW + px;
  const float r = rn[(size_t)n * HW + px];
  const float pr = from_p ? 1.f : r;
  float dot = 0.f;
#pragma unroll 8
  for (int c = 0; c < C; ++c) {
    const float g = gp[base + (size_t)c * HW], t = y[base + (size_t)c * HW];
    gs[c * NT + threadIdx.x] = g;
    ts[c * NT + threadIdx.x] = t;
    dot = fmaf(g, t * pr, dot);
  }
  dot /= (float)C;
#pragma unroll 8
  for (int c = 0; c < C; ++c) {
    const float g = gs[c * NT + threadIdx.x], t = ts[c * NT + threadIdx.x];
    gpre[base + (size_t)c * HW] = mg_lrelu_mask(t, slope) * r * (g - t * pr * dot);
  }
}

// Small maps (<= 16 k pixels): the same split as pixelnorm_fwd_small_k -- a workgroup takes 64 pixels (lane = pixel), its 4 waves
// split the channels and keep both operands of their channels in registers (<= 40 each), the partial dot products meet in LDS,
// each wave writes what it holds.  One pass, 4x the threads of the kernel above, and every load of a wave is in flight at once
// (one thread per pixel walking 96-128 channels was 11-14 us for a few KB).
__global__ void __launch_bounds__(256) pixelnorm_lrelu_bwd_small_k(const float* __restrict__ gp, const float* __restrict__ y,
                                                                   const float* __restrict__ rn, float* __restrict__ gpre,
                                                                   int N, int C, int HW, float slope, int from_p) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t i = (size_t)blockIdx.x * 64 + lane;
  const bool ok = i < (size_t)N * HW;
  const int n = ok ? (int)(i / HW) : 0;
  const int px = ok ? (int)(i - (size_t)n * HW) : 0;
  const size_t base = (size_t)n * C * HW + px;
  const float r = ok ? rn[(size_t)n * HW + px] : 1.f;
  const float pr = from_p ? 1.f : r;
  float gv[PN_SMALL_MAXC / 4], tv[PN_SMALL_MAXC / 4];
#pragma unroll
  for (int k = 0; k < PN_SMALL_MAXC / 4; ++k) {
    const int c = wave + 4 * k;
    const bool live = ok && c < C;
    gv[k] = live ? gp[base + (size_t)c * HW] : 0.f;
    tv[k] = live ? y[base + (size_t)c * HW] : 0.f;
  }
  float dot = 0.f;
#pragma unroll
  for (int k = 0; k < PN_SMALL_MAXC / 4; ++k) dot = fmaf(gv[k], tv[k] * pr, dot);
  part[wave][lane] = dot;
  __syncthreads();
  dot = (((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane]) / (float)C;
  if (ok) {
#pragma unroll
    for (int k = 0; k < PN_SMALL_MAXC / 4; ++k) {
      const int c = wave + 4 * k;
      if (c < C) gpre[base + (size_t)c * HW] = mg_lrelu_mask(tv[k], slope) * r * (gv[k] - tv[k] * pr * dot);
    }
  }
}

// ---------------------------------------------------------------- up-sampling / pooling
__global__ void __launch_bounds__(256) upsample2x_fwd_k(const float* __restrict__ x, float* __restrict__ y, size_t total,
                                                        int Hin, int Win) {
  // one thread per INPUT element: writes its 2x2 block
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int w = (int)(i % Win);
    const size_t r = i / Win;
    const int h = (int)(r % Hin);
    const size_t nc = r / Hin;
    const float v = x[i];
    float* o = y + (nc * (2 * Hin) + 2 * h) * (size_t)(2 * Win) + 2 * w;
    *reinterpret_cast<float2*>(o) = make_float2(v, v);
    *reinterpret_cast<float2*>(o + 2 * Win) = make_float2(v, v);
  }
}

__global__ void __launch_bounds__(256) upsample2x_bwd_k(const float* __restrict__ gy, float* __restrict__ gx,
                                                        size_t total, int Hin, int Win) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int w = (int)(i % Win);
    const size_t r = i / Win;
    const int h = (int)(r % Hin);
    const size_t nc = r / Hin;
    const float* s = gy + (nc * (2 * Hin) + 2 * h) * (size_t)(2 * Win) + 2 * w;
    const float2 a = *reinterpret_cast<const float2*>(s);
    const float2 b = *reinterpret_cast<const float2*>(s + 2 * Win);
    gx[i] = (a.x + a.y) + (b.x + b.y);
  }
}

__global__ void __launch_bounds__(256) avgpool2_fwd_k(const float* __restrict__ x, float* __restrict__ y, size_t total,
                                                      int Ho, int Wo) {
  // one thread per OUTPUT element
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int w = (int)(i % Wo);
    const size_t r = i / Wo;
    const int h = (int)(r % Ho);
    const size_t nc = r / Ho;
    const float* s = x + (nc * (2 * Ho) + 2 * h) * (size_t)(2 * Wo) + 2 * w;
    const float2 a = *reinterpret_cast<const float2*>(s);
    const float2 b = *reinterpret_cast<const float2*>(s + 2 * Wo);
    y[i] = ((a.x + a.y) + (b.x + b.y)) * 0.25f;
  }
}

__global__ void __launch_bounds__(256) avgpool2_bwd_k(const float* __restrict__ gy, const float* __restrict__ act,
                                                      float* __restrict__ gx, size_t total, int Ho, int Wo,
                                                      float slope) {
  // one thread per pooled (gy) element: writes the 2x2 block of gx
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int w = (int)(i % Wo);
    const size_t r = i / Wo;
    const int h = (int)(r % Ho);
    const size_t nc = r / Ho;
    const float g = gy[i] * 0.25f;
    const size_t o = (nc * (2 * Ho) + 2 * h) * (size_t)(2 * Wo) + 2 * w;
    float2 m0 = make_float2(1.f, 1.f), m1 = make_float2(1.f, 1.f);
    if (act) {
      const float2 a0 = *reinterpret_cast<const float2*>(act + o);
      const float2 a1 = *reinterpret_cast<const float2*>(act + o + 2 * Wo);
      m0 = make_float2(mg_lrelu_mask(a0.x, slope), mg_lrelu_mask(a0.y, slope));
      m1 = make_float2(mg_lrelu_mask(a1.x, slope), mg_lrelu_mask(a1.y, slope));
    }
    *reinterpret_cast<float2*>(gx + o) = make_float2(g * m0.x, g * m0.y);
    *reinterpret_cast<float2*>(gx + o + 2 * Wo) = make_float2(g * m1.x, g * m1.y);
  }
}

// the same with the mask as tile bytes (MG_CONV_MASK_OUT of mg_wino3x3: bit 2i+j of byte [nc][h][w] <-> act[2h+i][2w+j] > 0).
// One thread per PAIR of pooled elements: 8-byte gradient load, 2-byte mask load, two 16-byte stores.
__global__ void __launch_bounds__(256) avgpool2_bwd_bytes_k(const float* __restrict__ gy, const unsigned char* __restrict__ mask,
                                                            float* __restrict__ gx, size_t pairs, int Ho, int Wo2,
                                                            float slope) {
  const float qh = 0.25f, ql = 0.25f * slope;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += (size_t)gridDim.x * blockDim.x) {
    const int w2 = (int)(i % Wo2);
    const size_t r = i / Wo2;
    const int h = (int)(r % Ho);
    const size_t nc = r / Ho;
    const float2 g = *reinterpret_cast<const float2*>(gy + 2 * i);
    const unsigned m = *reinterpret_cast<const unsigned short*>(mask + 2 * i);
    const size_t o = (nc * (2 * Ho) + 2 * h) * (size_t)(4 * Wo2) + 4 * w2;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      float4 v;
      v.x = g.x * (((m >> (2 * rr)) & 1u) ? qh : ql);
      v.y = g.x * (((m >> (2 * rr + 1)) & 1u) ? qh : ql);
      v.z = g.y * (((m >> (8 + 2 * rr)) & 1u) ? qh : ql);
      v.w = g.y * (((m >> (9 + 2 * rr)) & 1u) ? qh : ql);
      *reinterpret_cast<float4*>(gx + o + (size_t)rr * (4 * Wo2)) = v;
    }
  }
}

// `coef` (optional, all three fade-in kernels): the two coefficients in DEVICE memory instead of launch arguments, so that a
// captured HIP graph of an update stays valid while alpha moves through the fade-in (the values, hence the results, are the same)
__global__ void __launch_bounds__(256) blend_up_k(float a, const float* __restrict__ x, float b,
                                                  const float* __restrict__ ylow, float* __restrict__ out, size_t total,
                                                  int Ho, int Wo, const float* __restrict__ coef) {
  if (coef) { a = coef[0]; b = coef[1]; }
  // one thread per LOW-res element: out 2x2 block = a*x + b*ylow
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int w = (int)(i % Wo);
    const size_t r = i / Wo;
    const int h = (int)(r % Ho);
    const size_t nc = r / Ho;
    const float yl = b * ylow[i];
    const size_t o = (nc * (2 * Ho) + 2 * h) * (size_t)(2 * Wo) + 2 * w;
    const float2 x0 = *reinterpret_cast<const float2*>(x + o);
    const float2 x1 = *reinterpret_cast<const float2*>(x + o + 2 * Wo);
    *reinterpret_cast<float2*>(out + o) = make_float2(fmaf(a, x0.x, yl), fmaf(a, x0.y, yl));
    *reinterpret_cast<float2*>(out + o + 2 * Wo) = make_float2(fmaf(a, x1.x, yl), fmaf(a, x1.y, yl));
  }
}

// ---------------------------------------------------------------- flat element-wise
template <int V>
__global__ void __launch_bounds__(256) lrelu_bwd_k(const float* __restrict__ g, const float* __restrict__ act,
                                                   float* __restrict__ out, size_t nq, float slope) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += (size_t)gridDim.x * blockDim.x) {
    float a[V], b[V];
    ld<V>(g + i * V, a);
    ld<V>(act + i * V, b);
#pragma unroll
    for (int v = 0; v < V; ++v) a[v] *= mg_lrelu_mask(b[v], slope);
    st<V>(out + i * V, a);
  }
}

// backward of the fade-in blend alpha*a + (1-alpha)*o followed by the two LeakyReLUs that produced a and o: one pass over g
template <int V>
__global__ void __launch_bounds__(256) blend_lrelu_bwd_k(const float* __restrict__ g, const float* __restrict__ act_a,
                                                         const float* __restrict__ act_o, float ca, float co,
                                                         float* __restrict__ out_a, float* __restrict__ out_o, size_t nq,
                                                         float slope, const float* __restrict__ coef) {
  if (coef) { ca = coef[0]; co = coef[1]; }
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += (size_t)gridDim.x * blockDim.x) {
    float gv[V], a[V], o[V];
    ld<V>(g + i * V, gv);
    ld<V>(act_a + i * V, a);
    ld<V>(act_o + i * V, o);
#pragma unroll
    for (int v = 0; v < V; ++v) {
      a[v] = (ca * gv[v]) * mg_lrelu_mask(a[v], slope);
      o[v] = (co * gv[v]) * mg_lrelu_mask(o[v], slope);
    }
    st<V>(out_a + i * V, a);
    st<V>(out_o + i * V, o);
  }
}

template <int V>
__global__ void __launch_bounds__(256) axpby_k(float a, const float* __restrict__ x, float b,
                                               const float* __restrict__ y, float* __restrict__ out, size_t nq,
                                               const float* __restrict__ coef) {
  if (coef) { a = coef[0]; if (y) b = coef[1]; }
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += (size_t)gridDim.x * blockDim.x) {
    float xv[V], yv[V];
    ld<V>(x + i * V, xv);
    if (y) {
      ld<V>(y + i * V, yv);
#pragma unroll
      for (int v = 0; v < V; ++v) xv[v] = a * xv[v] + b * yv[v];
    } else {
#pragma unroll
      for (int v = 0; v < V; ++v) xv[v] = a * xv[v];
    }
    st<V>(out + i * V, xv);
  }
}

template <int V>
__global__ void __launch_bounds__(256) gp_interp_k(const float* __restrict__ xr, const float* __restrict__ xf,
                                                   const float* __restrict__ eps, float* __restrict__ out, int N,
                                                   size_t chwq) {
  const size_t total = (size_t)N * chwq;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / chwq);
    const float e = eps[n];
    float a[V], b[V];
    ld<V>(xr + i * V, a);
    ld<V>(xf + i * V, b);
#pragma unroll
    for (int v = 0; v < V; ++v) a[v] = e * a[v] + (1.f - e) * b[v];
    st<V>(out + i * V, a);
  }
}

template <int V>
__global__ void __launch_bounds__(256) scale_per_sample_k(const float* __restrict__ g, const float* __restrict__ coef,
                                                          float* __restrict__ out, int N, size_t chwq) {
  const size_t total = (size_t)N * chwq;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / chwq);
    const float c = coef[n];
    float a[V];
    ld<V>(g + i * V, a);
#pragma unroll
    for (int v = 0; v < V; ++v) a[v] *= c;
    st<V>(out + i * V, a);
  }
}

__device__ __forceinline__ float block_sum_1024(float v, float* red) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float s = 0.f;
  const int nw = (blockDim.x + 63) >> 6;
  for (int k = 0; k < nw; ++k) s += red[k];
  return s;
}

// One workgroup per sample (a fixed summation order: the penalty must not depend on how many workgroups the chip offers).  With
// the reference's batch of 6 that is 6 workgroups for 2 x 512 x 512 floats each at level 7: what matters is memory-level
// parallelism inside the workgroup -- 16-byte loads, four of them in flight per thread, four independent partial sums -- not
// occupancy (a scalar one-load-one-fma chain took 138 us there).
__global__ void __launch_bounds__(1024) sumsq_per_sample_k(const float* __restrict__ g, float* __restrict__ out,
                                                           size_t chw) {
  __shared__ float red[16];
  const float* p = g + (size_t)blockIdx.x * chw;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if ((chw & 3) == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
    const f32x4* p4 = reinterpret_cast<const f32x4*>(p);
    const size_t n4 = chw / 4;
    size_t i = threadIdx.x;
    for (; i + 3 * 1024 < n4; i += 4 * 1024) {
      const f32x4 a = p4[i], b = p4[i + 1024], c = p4[i + 2048], d = p4[i + 3072];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s0 = fmaf(a[e], a[e], s0);
        s1 = fmaf(b[e], b[e], s1);
        s2 = fmaf(c[e], c[e], s2);
        s3 = fmaf(d[e], d[e], s3);
      }
    }
    for (; i < n4; i += 1024) {
      const f32x4 a = p4[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) s0 = fmaf(a[e], a[e], s0);
    }
  } else {
    for (size_t i = threadIdx.x; i < chw; i += blockDim.x) s0 = fmaf(p[i], p[i], s0);
  }
  const float s = block_sum_1024((s0 + s1) + (s2 + s3), red);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}

__global__ void gp_finish_k(const float* __restrict__ sumsq, float* __restrict__ penalty, float* __restrict__ coef,
                            int N, float factor, float upstream) {
  __shared__ float red[16];
  float s = 0.f;
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    const float nrm = sqrtf(sumsq[n]);
    const float d = nrm - 1.f;
    s += d * d;
    if (coef) coef[n] = nrm > 0.f ? upstream * factor * 2.f * d / ((float)N * nrm) : 0.f;
  }
  s = block_sum_1024(s, red);
  if (threadIdx.x == 0 && penalty) penalty[0] = factor * s / (float)N;
}

// gp_finish_k and scale_per_sample_k in one launch: every thread derives its sample's coefficient from sumsq[n] itself (a square root
// and a division per 4 elements), workgroup 0 also sums the penalty.
template <int V>
__global__ void __launch_bounds__(256) gp_apply_k(const float* __restrict__ g, const float* __restrict__ sumsq,
                                                  float* __restrict__ penalty, float* __restrict__ out, int N, size_t chwq,
                                                  float factor, float upstream) {
  __shared__ float red[16];
  const size_t total = (size_t)N * chwq;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / chwq);
    const float nrm = sqrtf(sumsq[n]);
    const float c = nrm > 0.f ? upstream * factor * 2.f * (nrm - 1.f) / ((float)N * nrm) : 0.f;
    float a[V];
    ld<V>(g + i * V, a);
#pragma unroll
    for (int v = 0; v < V; ++v) a[v] *= c;
    st<V>(out + i * V, a);
  }
  if (blockIdx.x == 0 && penalty != nullptr) {
    float s = 0.f;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
      const float d = sqrtf(sumsq[n]) - 1.f;
      s += d * d;
    }
    s = block_sum_1024(s, red);
    if (threadIdx.x == 0) penalty[0] = factor * s / (float)N;
  }
}

// Means of consecutive groups of n critic scores + the Wasserstein loss built from them, one launch (criterion.py:12-18):
//   out[g] = mean(x[g*n .. g*n+n)),  out[groups] = groups >= 2 ? out[1] - out[0]  (= -(mean D(real) - mean D(fake)))  :  -out[0]
__global__ void __launch_bounds__(256) group_means_k(const float* __restrict__ x, int groups, int n, float* __restrict__ out) {
  __shared__ double part[4];
  __shared__ float means[8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int g = 0; g < groups; ++g) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)x[(size_t)g * n + i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if (lane == 0) part[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float m = (float)((((part[0] + part[1]) + part[2]) + part[3]) / (double)n);
      means[g] = m;
      out[g] = m;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[groups] = groups >= 2 ? means[1] - means[0] : -means[0];
}

__global__ void __launch_bounds__(1024) channel_sum_k(const float* __restrict__ x, float* __restrict__ out, int N, int C,
                                                      int HW, int accumulate) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  float s = 0.f;
  for (int n = 0; n < N; ++n) {
    const float* p = x + ((size_t)n * C + c) * HW;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) s += p[i];
  }
  s = block_sum_1024(s, red);
  if (threadIdx.x == 0) out[c] = accumulate ? out[c] + s : s;
}

// ---------------------------------------------------------------- Linear(K -> 1)
__global__ void __launch_bounds__(64) linear1_fwd_k(const float* __restrict__ x, const float* __restrict__ w,
                                                    const float* __restrict__ b, float* __restrict__ y, int K) {
  const int n = blockIdx.x;
  double s = 0.0;  // K = 160 exact products summed in fp64: the critic score feeds a difference of batch means
  for (int k = threadIdx.x; k < K; k += 64) s += (double)w[k] * (double)x[(size_t)n * K + k];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
  if (threadIdx.x == 0) y[n] = (float)(s + (b ? (double)b[0] : 0.0));
}

__global__ void __launch_bounds__(256) linear1_bwd_k(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ gy, float* __restrict__ gx,
                                                     float* __restrict__ gw, float* __restrict__ gb, int N, int K,
                                                     int accumulate, int bias_n) {
  // The classifier weight gradient is a cancellation: -mean(real features) + mean(fake features) + penalty term leaves
  // ~1e-6 from ~0.05-sized summands.  Summing the few hundred terms in fp64 costs nothing (K = 160 threads x N <= 600 fmas) and
  // removes the order-dependent fp32 round-off of the intermediate sums (each product is exact in fp64).
  // one wave per feature k (4 per workgroup), its 64 lanes stride over the samples; lane partials and the xor-tree are in fp64 and in
  // a fixed order (deterministic).  (A single workgroup walking all N samples per thread took 35-95 us: 100+ dependent loads.)
  const int lane = threadIdx.x & 63;
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k < K) {
    const float wk = w[k];
    double s = 0.0;
    for (int n = lane; n < N; n += 64) {
      const float g = gy[n];
      if (gx) gx[(size_t)n * K + k] = g * wk;
      if (gw) s += (double)g * (double)x[(size_t)n * K + k];
    }
    if (gw) {
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
      if (lane == 0) gw[k] = accumulate ? (float)((double)gw[k] + s) : (float)s;
    }
  }
  if (gb && blockIdx.x == 0 && threadIdx.x < 64) {
    double s = 0.0;
    for (int n = lane; n < bias_n; n += 64) s += (double)gy[n];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if (lane == 0) gb[0] = accumulate ? (float)((double)gb[0] + s) : (float)s;
  }
}

// ---------------------------------------------------------------- Adam
// The tensor records travel in the kernel-argument segment (ADAM_CHUNK per launch): no descriptor upload, hence no host
// synchronisation per optimizer step (a pageable host-to-device copy waits for the stream to drain) and nothing to keep alive.
constexpr int ADAM_CHUNK = 48;
struct AdamChunk {
  mg_adam_tensor_t t[ADAM_CHUNK];
};

__global__ void __launch_bounds__(256) adam_k(const AdamChunk desc, float beta1, float beta2, float eps, float grad_scale) {
  const mg_adam_tensor_t d = desc.t[blockIdx.y];
  const float step_size = d.step_size, bc2s = d.bc2_sqrt;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.numel; i += (int64_t)gridDim.x * blockDim.x) {
    const float g = d.grad[i] * grad_scale;
    float m = d.exp_avg[i], v = d.exp_avg_sq[i];
    m = m + (g - m) * (1.f - beta1);           // exp_avg.lerp_(grad, 1-beta1)
    v = v * beta2 + (1.f - beta2) * g * g;     // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1-beta2)
    const float denom = sqrtf(v) / bc2s + eps;
    d.param[i] = d.param[i] - step_size * (m / denom);
    d.exp_avg[i] = m;
    d.exp_avg_sq[i] = v;
  }
}

// Capturable variant (HIP graphs): the step count lives in device memory, the bias corrections are formed in the kernel, and a
// second one-wave launch behind it advances the counters -- nothing step-dependent is baked into the launch arguments.
constexpr int ADAM_DEV_CHUNK = 56;
struct AdamDevChunk {
  mg_adam_tensor_dev_t t[ADAM_DEV_CHUNK];
};

__global__ void __launch_bounds__(256) adam_dev_k(const AdamDevChunk desc, float lr, float beta1, float beta2, float eps,
                                                  float grad_scale) {
  const mg_adam_tensor_dev_t d = desc.t[blockIdx.y];
  const float t = (float)(*d.step + 1);  // count AFTER this update
  const float bc1 = 1.f - (beta1 > 0.f ? powf(beta1, t) : 0.f);
  const float bc2s = sqrtf(1.f - powf(beta2, t));
  const float step_size = lr / bc1;
  auto update = [&](float g, float& p, float& m, float& v) {
    g *= grad_scale;
    m = m + (g - m) * (1.f - beta1);
    v = v * beta2 + (1.f - beta2) * g * g;
    const float denom = sqrtf(v) / bc2s + eps;
    p = p - step_size * (m / denom);
  };
  // 16-byte path when the four arrays allow it (a tensor's slice of a flat gradient bucket may start at any multiple of 4 bytes)
  const bool vec = ((d.numel & 3) == 0) &&
                   (((uintptr_t)d.param | (uintptr_t)d.grad | (uintptr_t)d.exp_avg | (uintptr_t)d.exp_avg_sq) & 15) == 0;
  if (vec) {
    const int64_t nq = d.numel >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += (int64_t)gridDim.x * blockDim.x) {
      const f32x4 g4 = reinterpret_cast<const f32x4*>(d.grad)[i];
      f32x4 p4 = reinterpret_cast<f32x4*>(d.param)[i], m4 = reinterpret_cast<f32x4*>(d.exp_avg)[i],
            v4 = reinterpret_cast<f32x4*>(d.exp_avg_sq)[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float p = p4[e], m = m4[e], v = v4[e];
        update(g4[e], p, m, v);
        p4[e] = p; m4[e] = m; v4[e] = v;
      }
      reinterpret_cast<f32x4*>(d.param)[i] = p4;
      reinterpret_cast<f32x4*>(d.exp_avg)[i] = m4;
      reinterpret_cast<f32x4*>(d.exp_avg_sq)[i] = v4;
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.numel; i += (int64_t)gridDim.x * blockDim.x) {
    float p = d.param[i], m = d.exp_avg[i], v = d.exp_avg_sq[i];
    update(d.grad[i], p, m, v);
    d.param[i] = p;
    d.exp_avg[i] = m;
    d.exp_avg_sq[i] = v;
  }
}

__global__ void __launch_bounds__(64) adam_tick_k(const AdamDevChunk desc, int n) {
  if ((int)threadIdx.x < n) *desc.t[threadIdx.x].step += 1;
}

}  // namespace

#define EW_LAUNCH(kernel, grid, block, ...)                                                   \
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, (hipStream_t)stream, __VA_ARGS__)

extern "C" int mg_pixelnorm_fwd(const float* y, float* p, float* rn, int N, int C, int HW, mg_stream_t stream) {
  MG_CHECK_ARG(y && p && N > 0 && C > 0 && HW > 0, "mg_pixelnorm_fwd: bad arguments");
  if ((size_t)N * HW <= (1u << 17) && C <= PN_SMALL_MAXC)
    EW_LAUNCH(pixelnorm_fwd_small_k, (unsigned)(((size_t)N * HW + 63) / 64), 256, y, p, rn, N, C, HW);
  else if ((HW & 3) == 0) EW_LAUNCH(pixelnorm_fwd_k<4>, ew_grid((size_t)N * HW / 4), 256, y, p, rn, N, C, HW);
  else EW_LAUNCH(pixelnorm_fwd_k<1>, ew_grid((size_t)N * HW), 256, y, p, rn, N, C, HW);
  MG_CHECK_LAUNCH("mg_pixelnorm_fwd");
  return MG_OK;
}

extern "C" int mg_pixelnorm_lrelu_bwd(const float* gp, const float* y, const float* rn, float* gpre, int N, int C, int HW,
                                      float slope, int from_p, mg_stream_t stream) {
  MG_CHECK_ARG(gp && y && rn && gpre && N > 0 && C > 0 && HW > 0, "mg_pixelnorm_lrelu_bwd: bad arguments");
  const size_t px_total = (size_t)N * HW;
  if (px_total <= (1u << 14) && C <= PN_SMALL_MAXC && getenv("MG_PN_BWD_NOLDS") == nullptr) {
    EW_LAUNCH(pixelnorm_lrelu_bwd_small_k, (unsigned)((px_total + 63) / 64), 256, gp, y, rn, gpre, N, C, HW, slope, from_p);
    MG_CHECK_LAUNCH("mg_pixelnorm_lrelu_bwd");
    return MG_OK;
  }
  if (C <= 128 && getenv("MG_PN_BWD_NOLDS") == nullptr) {  // up to 128 KB of LDS per workgroup
    static MgPerDevice once;  // the LDS limit is a per-device function attribute
    if (mg_first_use_on_device(once)) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pixelnorm_lrelu_bwd_lds_k<256>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pixelnorm_lrelu_bwd_lds_k<128>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    if (C <= 64) {
      hipLaunchKernelGGL(pixelnorm_lrelu_bwd_lds_k<256>, dim3((unsigned)((px_total + 255) / 256)), dim3(256),
                         (size_t)2 * C * 256 * sizeof(float), (hipStream_t)stream, gp, y, rn, gpre, N, C, HW, slope, from_p);
    } else {
      hipLaunchKernelGGL(pixelnorm_lrelu_bwd_lds_k<128>, dim3((unsigned)((px_total + 127) / 128)), dim3(128),
                         (size_t)2 * C * 128 * sizeof(float), (hipStream_t)stream, gp, y, rn, gpre, N, C, HW, slope, from_p);
    }
    MG_CHECK_LAUNCH("mg_pixelnorm_lrelu_bwd");
    return MG_OK;
  }
  if ((HW & 3) == 0)
    EW_LAUNCH(pixelnorm_lrelu_bwd_k<4>, ew_grid((size_t)N * HW / 4), 256, gp, y, rn, gpre, N, C, HW, slope, from_p);
  else EW_LAUNCH(pixelnorm_lrelu_bwd_k<1>, ew_grid((size_t)N * HW), 256, gp, y, rn, gpre, N, C, HW, slope, from_p);
  MG_CHECK_LAUNCH("mg_pixelnorm_lrelu_bwd");
  return MG_OK;
}

extern "C" int mg_upsample2x_fwd(const float* x, float* y, int NC, int Hin, int Win, mg_stream_t stream) {
  MG_CHECK_ARG(x && y && NC > 0 && Hin > 0 && Win > 0, "mg_upsample2x_fwd: bad arguments");
  const size_t total = (size_t)NC * Hin * Win;
  EW_LAUNCH(upsample2x_fwd_k, ew_grid(total), 256, x, y, total, Hin, Win);
  MG_CHECK_LAUNCH("mg_upsample2x_fwd");
  return MG_OK;
}

extern "C" int mg_upsample2x_bwd(const float* gy, float* gx, int NC, int Hin, int Win, mg_stream_t stream) {
  MG_CHECK_ARG(gy && gx && NC > 0 && Hin > 0 && Win > 0, "mg_upsample2x_bwd: bad arguments");
  const size_t total = (size_t)NC * Hin * Win;
  EW_LAUNCH(upsample2x_bwd_k, ew_grid(total), 256, gy, gx, total, Hin, Win);
  MG_CHECK_LAUNCH("mg_upsample2x_bwd");
  return MG_OK;
}

extern "C" int mg_avgpool2_fwd(const float* x, float* y, int NC, int H, int W, mg_stream_t stream) {
  MG_CHECK_ARG(x && y && NC > 0 && H > 0 && W > 0 && (H % 2 == 0) && (W % 2 == 0), "mg_avgpool2_fwd: bad arguments");
  const size_t total = (size_t)NC * (H / 2) * (W / 2);
  EW_LAUNCH(avgpool2_fwd_k, ew_grid(total), 256, x, y, total, H / 2, W / 2);
  MG_CHECK_LAUNCH("mg_avgpool2_fwd");
  return MG_OK;
}

extern "C" int mg_avgpool2_bwd(const float* gy, const float* act, float* gx, int NC, int H, int W, float slope,
                               mg_stream_t stream) {
  MG_CHECK_ARG(gy && gx && NC > 0 && H > 0 && W > 0 && (H % 2 == 0) && (W % 2 == 0), "mg_avgpool2_bwd: bad arguments");
  const size_t total = (size_t)NC * (H / 2) * (W / 2);
  EW_LAUNCH(avgpool2_bwd_k, ew_grid(total), 256, gy, act, gx, total, H / 2, W / 2, slope);
  MG_CHECK_LAUNCH("mg_avgpool2_bwd");
  return MG_OK;
}

extern "C" int mg_avgpool2_bwd_tilemask(const float* gy, const unsigned char* mask, float* gx, int NC, int H, int W, float slope,
                                        mg_stream_t stream) {
  MG_CHECK_ARG(gy && mask && gx && NC > 0 && H > 0 && W > 0 && (H % 2 == 0) && (W % 4 == 0),
               "mg_avgpool2_bwd_tilemask: bad arguments (W must be a multiple of 4)");
  const size_t pairs = (size_t)NC * (H / 2) * (W / 4);
  EW_LAUNCH(avgpool2_bwd_bytes_k, ew_grid(pairs), 256, gy, mask, gx, pairs, H / 2, W / 4, slope);
  MG_CHECK_LAUNCH("mg_avgpool2_bwd_tilemask");
  return MG_OK;
}

static int blend_up_impl(float a, const float* x, float b, const float* y, float* out, int NC, int H, int W, const float* coef,
                         mg_stream_t stream) {
  MG_CHECK_ARG(x && y && out && NC > 0 && H > 0 && W > 0 && (H % 2 == 0) && (W % 2 == 0), "mg_blend_up: bad arguments");
  const size_t total = (size_t)NC * (H / 2) * (W / 2);
  EW_LAUNCH(blend_up_k, ew_grid(total), 256, a, x, b, y, out, total, H / 2, W / 2, coef);
  MG_CHECK_LAUNCH("mg_blend_up");
  return MG_OK;
}
extern "C" int mg_blend_up(float a, const float* x, float b, const float* y, float* out, int NC, int H, int W,
                           mg_stream_t stream) {
  return blend_up_impl(a, x, b, y, out, NC, H, W, nullptr, stream);
}
extern "C" int mg_blend_up_dev(const float* coef, const float* x, const float* y, float* out, int NC, int H, int W,
                               mg_stream_t stream) {
  MG_CHECK_ARG(coef, "mg_blend_up_dev: coef is NULL");
  return blend_up_impl(0.f, x, 0.f, y, out, NC, H, W, coef, stream);
}

extern "C" int mg_lrelu_bwd(const float* g, const float* act, float* out, size_t n, float slope, mg_stream_t stream) {
  MG_CHECK_ARG(g && act && out && n > 0, "mg_lrelu_bwd: bad arguments");
  if ((n & 3) == 0) EW_LAUNCH(lrelu_bwd_k<4>, ew_grid(n / 4), 256, g, act, out, n / 4, slope);
  else EW_LAUNCH(lrelu_bwd_k<1>, ew_grid(n), 256, g, act, out, n, slope);
  MG_CHECK_LAUNCH("mg_lrelu_bwd");
  return MG_OK;
}

static int blend_lrelu_bwd_impl(const float* g, const float* act_a, const float* act_o, float ca, float co, float* out_a,
                                float* out_o, size_t n, float slope, const float* coef, mg_stream_t stream) {
  MG_CHECK_ARG(g && act_a && act_o && out_a && out_o && n > 0, "mg_blend_lrelu_bwd: bad arguments");
  if ((n & 3) == 0)
    EW_LAUNCH(blend_lrelu_bwd_k<4>, ew_grid(n / 4), 256, g, act_a, act_o, ca, co, out_a, out_o, n / 4, slope, coef);
  else EW_LAUNCH(blend_lrelu_bwd_k<1>, ew_grid(n), 256, g, act_a, act_o, ca, co, out_a, out_o, n, slope, coef);
  MG_CHECK_LAUNCH("mg_blend_lrelu_bwd");
  return MG_OK;
}
extern "C" int mg_blend_lrelu_bwd(const float* g, const float* act_a, const float* act_o, float ca, float co, float* out_a,
                                  float* out_o, size_t n, float slope, mg_stream_t stream) {
  return blend_lrelu_bwd_impl(g, act_a, act_o, ca, co, out_a, out_o, n, slope, nullptr, stream);
}
extern "C" int mg_blend_lrelu_bwd_dev(const float* g, const float* act_a, const float* act_o, const float* coef, float* out_a,
                                      float* out_o, size_t n, float slope, mg_stream_t stream) {
  MG_CHECK_ARG(coef, "mg_blend_lrelu_bwd_dev: coef is NULL");
  return blend_lrelu_bwd_impl(g, act_a, act_o, 0.f, 0.f, out_a, out_o, n, slope, coef, stream);
}

static int axpby_impl(float a, const float* x, float b, const float* y, float* out, size_t n, const float* coef,
                      mg_stream_t stream) {
  MG_CHECK_ARG(x && out && n > 0, "mg_axpby: bad arguments");
  if ((n & 3) == 0) EW_LAUNCH(axpby_k<4>, ew_grid(n / 4), 256, a, x, b, y, out, n / 4, coef);
  else EW_LAUNCH(axpby_k<1>, ew_grid(n), 256, a, x, b, y, out, n, coef);
  MG_CHECK_LAUNCH("mg_axpby");
  return MG_OK;
}
extern "C" int mg_axpby(float a, const float* x, float b, const float* y, float* out, size_t n, mg_stream_t stream) {
  return axpby_impl(a, x, b, y, out, n, nullptr, stream);
}
extern "C" int mg_axpby_dev(const float* coef, const float* x, const float* y, float* out, size_t n, mg_stream_t stream) {
  MG_CHECK_ARG(coef, "mg_axpby_dev: coef is NULL");
  return axpby_impl(0.f, x, 0.f, y, out, n, coef, stream);
}

extern "C" int mg_adam_step_dev(const mg_adam_tensor_dev_t* desc, int n_tensors, float lr, float beta1, float beta2, float eps,
                                float grad_scale, mg_stream_t stream) {
  MG_CHECK_ARG(desc && n_tensors > 0, "mg_adam_step_dev: bad arguments");
  for (int first = 0; first < n_tensors; first += ADAM_DEV_CHUNK) {
    const int n = n_tensors - first < ADAM_DEV_CHUNK ? n_tensors - first : ADAM_DEV_CHUNK;
    AdamDevChunk c;
    for (int i = 0; i < n; ++i) {
      c.t[i] = desc[first + i];
      MG_CHECK_ARG(c.t[i].param && c.t[i].grad && c.t[i].exp_avg && c.t[i].exp_avg_sq && c.t[i].step && c.t[i].numel >= 0,
                   "mg_adam_step_dev: record %d has a null pointer", first + i);
    }
    hipLaunchKernelGGL(adam_dev_k, dim3(64, n), dim3(256), 0, (hipStream_t)stream, c, lr, beta1, beta2, eps, grad_scale);
    MG_CHECK_LAUNCH("mg_adam_step_dev");
    hipLaunchKernelGGL(adam_tick_k, dim3(1), dim3(64), 0, (hipStream_t)stream, c, n);  // behind every reader of the counters
    MG_CHECK_LAUNCH("mg_adam_step_dev(tick)");
  }
  return MG_OK;
}

extern "C" int mg_linear1_fwd(const float* x, const float* w, const float* b, float* y, int N, int K,
                              mg_stream_t stream) {
  MG_CHECK_ARG(x && w && y && N > 0 && K > 0, "mg_linear1_fwd: bad arguments");
  EW_LAUNCH(linear1_fwd_k, N, 64, x, w, b, y, K);
  MG_CHECK_LAUNCH("mg_linear1_fwd");
  return MG_OK;
}

extern "C" int mg_linear1_bwd(const float* x, const float* w, const float* gy, float* gx, float* gw, float* gb, int N,
                              int K, int accumulate, int bias_n, mg_stream_t stream) {
  MG_CHECK_ARG(w && gy && N > 0 && K > 0 && (!gw || x), "mg_linear1_bwd: bad arguments");
  EW_LAUNCH(linear1_bwd_k, (K + 3) / 4, 256, x, w, gy, gx, gw, gb, N, K, accumulate, (bias_n <= 0 || bias_n > N) ? N : bias_n);
  MG_CHECK_LAUNCH("mg_linear1_bwd");
  return MG_OK;
}

extern "C" int mg_gp_interp(const float* x_real, const float* x_fake, const float* eps, float* out, int N, size_t chw,
                            mg_stream_t stream) {
  MG_CHECK_ARG(x_real && x_fake && eps && out && N > 0 && chw > 0, "mg_gp_interp: bad arguments");
  if ((chw & 3) == 0) EW_LAUNCH(gp_interp_k<4>, ew_grid(N * chw / 4), 256, x_real, x_fake, eps, out, N, chw / 4);
  else EW_LAUNCH(gp_interp_k<1>, ew_grid(N * chw), 256, x_real, x_fake, eps, out, N, chw);
  MG_CHECK_LAUNCH("mg_gp_interp");
  return MG_OK;
}

extern "C" int mg_sumsq_per_sample(const float* g, float* out, int N, size_t chw, mg_stream_t stream) {
  MG_CHECK_ARG(g && out && N > 0 && chw > 0, "mg_sumsq_per_sample: bad arguments");
  EW_LAUNCH(sumsq_per_sample_k, N, 1024, g, out, chw);
  MG_CHECK_LAUNCH("mg_sumsq_per_sample");
  return MG_OK;
}

extern "C" int mg_scale_per_sample(const float* g, const float* coef, float* out, int N, size_t chw,
                                   mg_stream_t stream) {
  MG_CHECK_ARG(g && coef && out && N > 0 && chw > 0, "mg_scale_per_sample: bad arguments");
  if ((chw & 3) == 0) EW_LAUNCH(scale_per_sample_k<4>, ew_grid(N * chw / 4), 256, g, coef, out, N, chw / 4);
  else EW_LAUNCH(scale_per_sample_k<1>, ew_grid(N * chw), 256, g, coef, out, N, chw);
  MG_CHECK_LAUNCH("mg_scale_per_sample");
  return MG_OK;
}

extern "C" int mg_gp_finish(const float* sumsq, float* penalty, float* coef, int N, float factor, float upstream,
                            mg_stream_t stream) {
  MG_CHECK_ARG(sumsq && N > 0, "mg_gp_finish: bad arguments");
  EW_LAUNCH(gp_finish_k, 1, 256, sumsq, penalty, coef, N, factor, upstream);
  MG_CHECK_LAUNCH("mg_gp_finish");
  return MG_OK;
}

extern "C" int mg_gp_apply(const float* g, const float* sumsq, float* penalty, float* out, int N, size_t chw, float factor,
                           float upstream, mg_stream_t stream) {
  MG_CHECK_ARG(g && sumsq && out && N > 0 && chw > 0, "mg_gp_apply: bad arguments");
  if ((chw & 3) == 0) EW_LAUNCH(gp_apply_k<4>, ew_grid(N * chw / 4), 256, g, sumsq, penalty, out, N, chw / 4, factor, upstream);
  else EW_LAUNCH(gp_apply_k<1>, ew_grid(N * chw), 256, g, sumsq, penalty, out, N, chw, factor, upstream);
  MG_CHECK_LAUNCH("mg_gp_apply");
  return MG_OK;
}

extern "C" int mg_group_means(const float* x, int groups, int n, float* out, mg_stream_t stream) {
  MG_CHECK_ARG(x && out && groups >= 1 && groups <= 8 && n > 0, "mg_group_means: bad arguments");
  EW_LAUNCH(group_means_k, 1, 256, x, groups, n, out);
  MG_CHECK_LAUNCH("mg_group_means");
  return MG_OK;
}

extern "C" int mg_channel_sum(const float* x, float* out, int N, int C, int HW, int accumulate, mg_stream_t stream) {
  MG_CHECK_ARG(x && out && N > 0 && C > 0 && HW > 0, "mg_channel_sum: bad arguments");
  EW_LAUNCH(channel_sum_k, C, 1024, x, out, N, C, HW, accumulate);
  MG_CHECK_LAUNCH("mg_channel_sum");
  return MG_OK;
}

extern "C" int mg_adam_step(const mg_adam_tensor_t* desc, int n_tensors, float beta1, float beta2, float eps,
                            float grad_scale, mg_stream_t stream) {
  MG_CHECK_ARG(desc && n_tensors > 0, "mg_adam_step: bad arguments");
  for (int first = 0; first < n_tensors; first += ADAM_CHUNK) {
    const int n = n_tensors - first < ADAM_CHUNK ? n_tensors - first : ADAM_CHUNK;
    AdamChunk c;
    for (int i = 0; i < n; ++i) {
      c.t[i] = desc[first + i];
      MG_CHECK_ARG(c.t[i].param && c.t[i].grad && c.t[i].exp_avg && c.t[i].exp_avg_sq && c.t[i].numel >= 0,
                   "mg_adam_step: record %d has a null pointer", first + i);
    }
    hipLaunchKernelGGL(adam_k, dim3(64, n), dim3(256), 0, (hipStream_t)stream, c, beta1, beta2, eps, grad_scale);
    MG_CHECK_LAUNCH("mg_adam_step");
  }
  return MG_OK;
}
