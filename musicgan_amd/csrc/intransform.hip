// The per-batch input transform in front of the training step, fused on the device:
//   ChannelMinMaxNorm -> ChangeRange(-1, 1) -> Resize(S)      /root/reference/music_gan/audio/transforms.py:4-40,
//   /root/reference/music_gan/utils.py:70-86 (Grower.__get_transform), applied at train.py:138-140 on the CPU per batch.
// Resize is torchvision's tensor path = aten upsample_bilinear2d_aa (bilinear, antialias, align_corners = False), restated
// from its published definition: per output index i, centre = scale (i + 1/2), support = scale (for scale >= 1, else 1),
// taps j in [xmin, xmin + xsize), weight = max(0, 1 - |(j - centre + 1/2) / max(scale, 1)|) normalised by their sum; the
// horizontal pass runs first, then the vertical one, with a float32 intermediate (the CPU kernel's separable order).
// Input: the (N, 2, H, W) batch as the DataLoader delivers it -- float64 (the dataset's storage type, cast to float32 on read,
// exactly x.to(th.float)) or float32.  Two launches: partial min/max per plane (16-byte loads), then ONE resize kernel: a workgroup
// owns a band of IT_R output rows of one plane, runs the horizontal pass over the input rows that band needs (the normalisation
// applied to every tap, as the reference does before resizing) into LDS -- the reference's float32 intermediate never goes to
// memory -- and the vertical pass out of it; the tap weights of the band are tabulated once per workgroup.  HBM-bound: the batch is
// read twice (+ 2 / IT_R of it again as halo rows, mostly out of the L2), the output is 1/16 of it at 512 -> 128.
// (r01-r03: three launches, weights and the plane's min / max recomputed per output element, a 33 MB intermediate written and read
// back: 0.19 ms for a batch of 64 at 2x512x512 = 1.4 TB/s.)
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
  constexpr int V = 16 / (int)sizeof(T);  // elements per 16-byte load
  typedef T vec_t __attribute__((ext_vector_type(V)));
  if ((reinterpret_cast<size_t>(xp) & 15) == 0 && (per % V) == 0 && (plane % V) == 0) {
    const vec_t* xv = reinterpret_cast<const vec_t*>(xp);
    const size_t vlo = lo / V, vhi = hi / V;
    size_t i = vlo + threadIdx.x;
    for (; i + 3 * 256 < vhi; i += 4 * 256) {  // four loads in flight
      vec_t q[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) q[u] = xv[i + u * 256];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < V; ++k) {
          const float v = (float)q[u][k];
          mn = fminf(mn, v);
          mx = fmaxf(mx, v);
        }
    }
    for (; i < vhi; i += 256) {
      const vec_t q = xv[i];
#pragma unroll
      for (int k = 0; k < V; ++k) {
        const float v = (float)q[k];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
      }
    }
  } else {
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
      const float v = (float)xp[i];
      mn = fminf(mn, v);
      mx = fmaxf(mx, v);
    }
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

constexpr int IT_R = 8;  // output rows per workgroup

struct ItGeom {
  int H, W, S;
  int kh, kv;      // bounds on the taps per output column / row
  int rows_max;    // bound on the input rows a band needs
  int lgS;         // log2(S) if S is a power of two, else -1
  float eps;
};

// LDS: hw[S][kh] | hx[S][2] | vw[IT_R][kv] | vy[IT_R][2] | mm[2] | tl[rows_max][S]
__host__ __device__ inline size_t it_lds_floats(const ItGeom& g) {
  return (size_t)g.S * g.kh + 2 * (size_t)g.S + (size_t)IT_R * g.kv + 2 * IT_R + 2 + (size_t)g.rows_max * g.S;
}

template <typename T>
__global__ void __launch_bounds__(256) it_resize_fused(const T* __restrict__ x, const float* __restrict__ part, float* __restrict__ out,
                                                       const ItGeom g) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* hw = sm;
  int* hx = reinterpret_cast<int*>(hw + (size_t)g.S * g.kh);
  float* vw = reinterpret_cast<float*>(hx + 2 * g.S);
  int* vy = reinterpret_cast<int*>(vw + IT_R * g.kv);
  float* mm = reinterpret_cast<float*>(vy + 2 * IT_R);
  float* tl = mm + 2;
  const int nc = blockIdx.x, i0 = blockIdx.y * IT_R, tid = threadIdx.x;
  const int nrow_out = g.S - i0 < IT_R ? g.S - i0 : IT_R;
  const float sh = (float)g.W / (float)g.S, sv = (float)g.H / (float)g.S;

  // tables of the band: the plane's min / max, tap ranges and weights (w / total, as aten stores them)
  if (tid == 0) {
    float mn = INFINITY, mx = -INFINITY;
    for (int p = 0; p < MMP; ++p) {
      mn = fminf(mn, part[((size_t)nc * MMP + p) * 2]);
      mx = fmaxf(mx, part[((size_t)nc * MMP + p) * 2 + 1]);
    }
    mm[0] = mn;
    mm[1] = mx - mn + g.eps;
  }
  for (int j = tid; j < g.S + nrow_out; j += 256) {
    const bool horiz = j < g.S;
    const int idx = horiz ? j : j - g.S;
    int lo, n;
    float centre, invscale, tw;
    aa_span(horiz ? idx : i0 + idx, horiz ? g.W : g.H, horiz ? sh : sv, lo, n, centre, invscale, tw);
    float* wt = horiz ? hw + (size_t)idx * g.kh : vw + idx * g.kv;
    int* span = horiz ? hx + 2 * idx : vy + 2 * idx;
    span[0] = lo;
    span[1] = n;
    for (int b = 0; b < n; ++b) {
      float t = ((float)(b + lo) - centre + 0.5f) * invscale;
      t = t < 0.f ? -t : t;
      wt[b] = (t < 1.f ? 1.f - t : 0.f) / tw;
    }
  }
  __syncthreads();
  const float mn = mm[0], den = mm[1];
  const int ylo = vy[0], yhi = vy[2 * (nrow_out - 1)] + vy[2 * (nrow_out - 1) + 1];  // (tap ranges grow with the row index)
  const int nin = yhi - ylo;

  // horizontal pass: tl[r][j] = sum_b w[j][b] * norm(x[nc][ylo + r][xmin_j + b]),  norm(v) = (v - mn) / (mx - mn + eps) * 2 - 1
  const T* plane = x + (size_t)nc * g.H * g.W;
  for (int e = tid; e < nin * g.S; e += 256) {
    const int r = g.lgS >= 0 ? e >> g.lgS : e / g.S;
    const int j = e - r * g.S;
    const int xmin = hx[2 * j], xs = hx[2 * j + 1];
    const T* row = plane + (size_t)(ylo + r) * g.W + xmin;
    const float* w = hw + (size_t)j * g.kh;
    float acc = 0.f;
    int b = 0;
    for (; b + 3 < xs; b += 4) {  // loads first, arithmetic second (the sum keeps its order)
      const T q0 = row[b], q1 = row[b + 1], q2 = row[b + 2], q3 = row[b + 3];
      acc += (((float)q0 - mn) / den * 2.f + -1.f) * w[b];
      acc += (((float)q1 - mn) / den * 2.f + -1.f) * w[b + 1];
      acc += (((float)q2 - mn) / den * 2.f + -1.f) * w[b + 2];
      acc += (((float)q3 - mn) / den * 2.f + -1.f) * w[b + 3];
    }
    for (; b < xs; ++b) acc += (((float)row[b] - mn) / den * 2.f + -1.f) * w[b];
    tl[e] = acc;
  }
  __syncthreads();
  // vertical pass: out[nc][i0 + i][j] = sum_a w[i][a] * tl[ymin_i - ylo + a][j]
  for (int e = tid; e < nrow_out * g.S; e += 256) {
    const int i = g.lgS >= 0 ? e >> g.lgS : e / g.S;
    const int j = e - i * g.S;
    const int ymin = vy[2 * i], ys = vy[2 * i + 1];
    const float* col = tl + (size_t)(ymin - ylo) * g.S + j;
    const float* w = vw + i * g.kv;
    float acc = 0.f;
    for (int a = 0; a < ys; ++a) acc += col[(size_t)a * g.S] * w[a];
    out[((size_t)nc * g.S + i0 + i) * g.S + j] = acc;
  }
}

bool it_geometry(int H, int W, int S, float eps, ItGeom& g) {
  g.H = H; g.W = W; g.S = S; g.eps = eps;
  const float sh = (float)W / (float)S, sv = (float)H / (float)S;
  g.kh = (int)(2.f * (sh >= 1.f ? sh : 1.f)) + 3;
  g.kv = (int)(2.f * (sv >= 1.f ? sv : 1.f)) + 3;
  g.rows_max = (int)(sv * (IT_R - 1)) + g.kv + 2;
  if (g.rows_max > H) g.rows_max = H;
  g.lgS = (S & (S - 1)) == 0 ? mg_ilog2(S) : -1;
  return it_lds_floats(g) * sizeof(float) <= 160 * 1024;
}

template <typename T>
int run_it(const T* x, float* out, float* part, int N, const ItGeom& g, hipStream_t s) {
  const size_t plane = (size_t)g.H * g.W;
  hipLaunchKernelGGL(it_minmax_part<T>, dim3(N * 2, MMP), dim3(256), 0, s, x, part, plane);
  const size_t lds = it_lds_floats(g) * sizeof(float);
  static MgPerDevice once;
  if (mg_first_use_on_device(once))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&it_resize_fused<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(it_resize_fused<T>, dim3(N * 2, mg_cdiv(g.S, IT_R)), dim3(256), lds, s, x, part, out, g);
  MG_CHECK_LAUNCH("mg_input_transform");
  return MG_OK;
}

}  // namespace

extern "C" size_t mg_input_transform_ws_bytes(int N, int H, int W, int S) {
  (void)H; (void)W; (void)S;
  return (size_t)N * 2 * MMP * 2 * sizeof(float);
}

extern "C" int mg_input_transform(const void* x, int x_is_f64, float* out, void* ws, size_t ws_bytes, int N, int H, int W, int S,
                                  float eps, mg_stream_t stream) {
  MG_CHECK_ARG(x && out && ws && N > 0 && H > 0 && W > 0 && S > 0, "mg_input_transform: bad arguments");
  MG_CHECK_ARG(S <= H && S <= W, "mg_input_transform: only down-sampling (S=%d from %dx%d) is implemented", S, H, W);
  if (ws_bytes < mg_input_transform_ws_bytes(N, H, W, S)) {
    mg_set_error("mg_input_transform: workspace too small");
    return MG_EWORKSPACE;
  }
  ItGeom g;
  MG_CHECK_ARG(it_geometry(H, W, S, eps, g), "mg_input_transform: %dx%d -> %d needs more than 160 KB of LDS per row band", H, W, S);
  float* part = reinterpret_cast<float*>(ws);
  hipStream_t s = (hipStream_t)stream;
  if (x_is_f64) return run_it(reinterpret_cast<const double*>(x), out, part, N, g, s);
  return run_it(reinterpret_cast<const float*>(x), out, part, N, g, s);
}
