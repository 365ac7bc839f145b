// Framed, Hann-windowed 1024-point real FFT (the STFT of /root/reference/music_gan/audio/functions.py:38-62):
//   X[k,t] = 1/sqrt(sum w^2) * sum_{n<1024} w[n] xpad[256 t + n] e^{-2 pi i k n / 1024},  k = 0..511 (Nyquist dropped),
//   xpad = reflect-pad(x, 512), w = periodic Hann, T = 1 + L/256.
//
// One wave transforms one frame: the 1024 real samples are packed as 512 complex points (8 per lane), three radix-8
// Stockham passes (two exchanges through a per-wave 4 KiB LDS buffer) leave Z[j + 64 r] in lane j / register r; the
// real-FFT untangling needs Z[512-k], which lives in lane (64-j)&63 -> one wavefront shuffle per value, no LDS.
// A workgroup (4 waves) covers 16 consecutive frames and transposes its 512x16 output tile through LDS so that global
// stores are 128-byte runs along t (the output is frequency-major).  Twiddles and the window are built once per
// workgroup into LDS with sincospi (no global tables, no hidden state).
#include "mg_common.h"

namespace {

constexpr int NFFT = 1024, HOP = 256, NB = 512;  // NB = complex points = output bins
constexpr int FPW = 2;                            // frames per wave
constexpr int FPB = 8;                            // frames per workgroup
constexpr int OSTR = FPB + 1;                     // padded row of the output tile (float2 units)

struct cf {
  float x, y;
};
__device__ __forceinline__ cf cadd(cf a, cf b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cf csub(cf a, cf b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cf cmul(cf a, cf b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cf mul_mi(cf a) { return {a.y, -a.x}; }  // a * (-i)

__device__ __forceinline__ void dft4(cf y0, cf y1, cf y2, cf y3, cf& q0, cf& q1, cf& q2, cf& q3) {
  const cf s0 = cadd(y0, y2), s1 = csub(y0, y2), s2 = cadd(y1, y3), s3 = mul_mi(csub(y1, y3));
  q0 = cadd(s0, s2);
  q2 = csub(s0, s2);
  q1 = cadd(s1, s3);
  q3 = csub(s1, s3);
}

// in-place 8-point DFT, natural order in and out
__device__ __forceinline__ void dft8(cf (&v)[8]) {
  const float h = 0.70710678118654752440f;
  const cf a0 = cadd(v[0], v[4]), a1 = cadd(v[1], v[5]), a2 = cadd(v[2], v[6]), a3 = cadd(v[3], v[7]);
  cf d0 = csub(v[0], v[4]), d1 = csub(v[1], v[5]), d2 = csub(v[2], v[6]), d3 = csub(v[3], v[7]);
  d1 = cf{(d1.x + d1.y) * h, (d1.y - d1.x) * h};   // * W8^1 = (1 - i)/sqrt2
  d2 = mul_mi(d2);                                 // * W8^2 = -i
  d3 = cf{(d3.y - d3.x) * h, -(d3.x + d3.y) * h};  // * W8^3 = (-1 - i)/sqrt2
  dft4(a0, a1, a2, a3, v[0], v[2], v[4], v[6]);
  dft4(d0, d1, d2, d3, v[1], v[3], v[5], v[7]);
}

__global__ void __launch_bounds__(256) stft1024_kernel(const float* __restrict__ wav, float* __restrict__ out_re,
                                                       float* __restrict__ out_im, long long L, int T) {
  __shared__ __attribute__((aligned(16))) float2 tw[NFFT];       // tw[m] = exp(-2 pi i m / 1024)
  __shared__ __attribute__((aligned(16))) float win[NFFT];       // Hann / sqrt(sum w^2)
  __shared__ __attribute__((aligned(16))) float2 xbuf[4][NB];    // per-wave exchange buffer
  __shared__ __attribute__((aligned(16))) float2 otile[NB * OSTR];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int m = tid; m < NFFT; m += 256) {
    float s, c;
    sincospif((float)m * (1.0f / 512.0f), &s, &c);  // angle = 2 pi m / 1024 = pi * m / 512
    tw[m] = make_float2(c, -s);
    win[m] = (0.5f - 0.5f * c) * 0.05103103630798288f;  // 1/sqrt(384): sum of hann^2 over 1024 = 384
  }
  __syncthreads();

  const int t0 = blockIdx.x * FPB;
  float2* xb = xbuf[wave];
  for (int f = 0; f < FPW; ++f) {
    const int fl = wave * FPW + f;  // frame slot in the tile
    const int t = t0 + fl;
    cf v[8];
    if (t < T) {
      const long long base = (long long)t * HOP - NFFT / 2;  // sample index of padded position 0 of this frame
      const bool interior = (base >= 0) && (base + NFFT <= L);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int n = 2 * (lane + 64 * r);
        float x0, x1;
        if (interior) {
          const float2 p = *reinterpret_cast<const float2*>(wav + base + n);
          x0 = p.x;
          x1 = p.y;
        } else {
          long long s0 = base + n, s1 = base + n + 1;
          if (s0 < 0) s0 = -s0;
          if (s1 < 0) s1 = -s1;
          if (s0 >= L) s0 = 2 * (L - 1) - s0;
          if (s1 >= L) s1 = 2 * (L - 1) - s1;
          x0 = wav[s0];
          x1 = wav[s1];
        }
        v[r] = cf{x0 * win[n], x1 * win[n + 1]};
      }
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = cf{0.f, 0.f};
    }
    // pass 0 (Ns = 1): no twiddles; out[8 j + r]
    dft8(v);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) xb[8 * lane + r] = make_float2(v[r].x, v[r].y);
    __syncthreads();
    // pass 1 (Ns = 8): in[j + 64 r] * exp(-2 pi i r k / 64), k = j & 7; out[(j>>3)*64 + k + 8 r]
    {
      const int k = lane & 7;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float2 p = xb[lane + 64 * r];
        const float2 w = tw[(r * k * 16) & (NFFT - 1)];  // 1024/64 = 16
        v[r] = cmul(cf{p.x, p.y}, cf{w.x, w.y});
      }
      dft8(v);
      __syncthreads();
      const int j0 = (lane >> 3) * 64 + k;
#pragma unroll
      for (int r = 0; r < 8; ++r) xb[j0 + 8 * r] = make_float2(v[r].x, v[r].y);
      __syncthreads();
    }
    // pass 2 (Ns = 64): in[j + 64 r] * exp(-2 pi i r j / 512); result Z[j + 64 r] stays in lane j, register r
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float2 p = xb[lane + 64 * r];
      const float2 w = tw[(r * lane * 2) & (NFFT - 1)];  // 1024/512 = 2
      v[r] = cmul(cf{p.x, p.y}, cf{w.x, w.y});
    }
    dft8(v);
    // untangle: X[k] = (Z[k] + conj Z[512-k])/2 - i/2 * e^{-2 pi i k/1024} * (Z[k] - conj Z[512-k]),  k = lane + 64 r.
    // Z[512-k] sits in lane (64-lane)&63, register 7-r (lane != 0) or 8-r (lane == 0; r == 0 -> Z[0] itself).
    const int src = (64 - lane) & 63;
    cf part[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      // value this lane must SEND for the receiver's register r: receiver lane l' = (64-lane)&63 wants register
      // (l' == 0 ? 8-r : 7-r) of its source lane; every lane != 0 is read by a lane != 0 (7-r); lane 0 reads itself.
      const cf send_n0 = v[7 - r];
      const cf send_0 = v[(8 - r) & 7];
      const cf send = (lane == 0) ? send_0 : send_n0;
      part[r] = cf{__shfl(send.x, src), __shfl(send.y, src)};
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int k = lane + 64 * r;
      const cf z = v[r];
      const cf zc = cf{part[r].x, -part[r].y};
      const cf e = cf{0.5f * (z.x + zc.x), 0.5f * (z.y + zc.y)};
      const cf d = cf{0.5f * (z.x - zc.x), 0.5f * (z.y - zc.y)};
      const float2 w = tw[k];
      const cf wd = cmul(cf{w.x, w.y}, d);
      // -i * wd = (wd.y, -wd.x)
      otile[k * OSTR + fl] = make_float2(e.x + wd.y, e.y - wd.x);
    }
  }
  __syncthreads();
  // transposed write-out: FPB consecutive frames of one bin per FPB lanes
  const int f = tid & (FPB - 1);
  const int t = t0 + f;
  if (t < T) {
    for (int k = tid / FPB; k < NB; k += 256 / FPB) {
      const float2 o = otile[k * OSTR + f];
      const size_t idx = (size_t)k * T + t;
      if (out_im != nullptr) {
        out_re[idx] = o.x;
        out_im[idx] = o.y;
      } else {
        *reinterpret_cast<float2*>(out_re + 2 * idx) = o;
      }
    }
  }
}

}  // namespace

extern "C" int mg_stft_1024(const float* wav, float* out_re, float* out_im, int64_t L, mg_stream_t stream) {
  MG_CHECK_ARG(wav && out_re, "mg_stft_1024: bad arguments");
  MG_CHECK_ARG(L > NFFT / 2, "mg_stft_1024: reflect padding needs L > 512 (got %lld)", (long long)L);
  MG_CHECK_ARG(L / HOP + 1 < (1ll << 30), "mg_stft_1024: too many frames");
  const int T = (int)(L / HOP) + 1;
  const int blocks = (T + FPB - 1) / FPB;
  hipLaunchKernelGGL(stft1024_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, wav, out_re, out_im,
                     (long long)L, T);
  MG_CHECK_LAUNCH("mg_stft_1024");
  return MG_OK;
}
