// Magnitude/phase codec around the STFT and the inverse STFT, fp32 with the reference's operation order:
//   forward  /root/reference/music_gan/audio/functions.py:65-94   (abs/angle, bark scale, unwrap, first difference,
//            global min/max -> [-1,1], drop the leading remainder, split into nb_vec-frame images)
//   inverse  functions.py:97-139 (un-bark, /(max-min), phase -> [-pi,pi], cumulative sum, mod 2pi, polar -> complex,
//            zero Nyquist row, inverse_spectrogram == window * irfft, overlap-add / window envelope, centre trim)
// The forward unwrap's cumulative sum (functions.py:23, torch.cumsum: DOUBLE running sum, float32 outputs) is an exact blocked
// scan over the whole chip (codec_row_pass); the inverse's cumulative phase (functions.py:117-118) is a Python loop of float32
// adds in the reference, so it runs SEQUENTIALLY per frequency row in fp32 here too (inv_phase_cumsum: one lane per row walks
// time through LDS tiles).  Everything else is embarrassingly parallel and HBM-bound (8 B in + 8 B out per bin).
#include "mg_common.h"
#include "sleef_f32.h"

#include <cstdint>

namespace {

constexpr float PI_F = 3.14159274101257324f;      // float32(np.pi)
constexpr float TWO_PI_F = 6.28318548202514648f;  // float32(2*np.pi)
constexpr int NB = 512, NFFT = 1024, HOP = 256;

__device__ __forceinline__ float py_mod(float a, float b) {  // torch.remainder for b > 0
  float m = fmodf(a, b);
  if (m != 0.f && m < 0.f) m += b;
  return m;
}

__device__ __forceinline__ void block_minmax(float& mn, float& mx, float* red) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, d));
    mx = fmaxf(mx, __shfl_xor(mx, d));
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) {
    red[wave] = mn;
    red[16 + wave] = mx;
  }
  __syncthreads();
  mn = red[0];
  mx = red[16];
  for (int k = 1; k < nw; ++k) {
    mn = fminf(mn, red[k]);
    mx = fmaxf(mx, red[16 + k]);
  }
}

// ---- forward pass 1 (codec_row_pass): one workgroup per frequency row walks the whole track.
//   magn = |X| * s[k], phi = atan2(im, re); unwrap (np.unwrap semantics incl. the -pi -> +pi fix, functions.py:17-23); first
//   difference of the unwrapped phase; both written UN-normalised straight into the chunked output images; per-row min / max.
// The reference's `phi_adj.cumsum(1)` is torch.cumsum on a CPU float32 tensor: the running sum is kept in DOUBLE
// (at::acc_type<float, false>) and every output is rounded to float32.  The adjustments are float32 values of magnitude
// 2 pi (multiples of 2^-22, below 16), so a double holds any partial sum of up to 2^24 of them EXACTLY: the sum does not depend
// on its association, and a blocked scan -- per-lane sums, a DPP scan over the wave, an 8-entry scan over the waves, a carry
// per 2048-frame block -- returns bit for bit the doubles of the sequential loop.  (Rounds 1-2 kept a float32 running sum in one
// scanner lane per row: 2.0 ms per 10-minute track on 8 workgroups, and 5e-3 off the reference at that length.)
// A lane owns RQ = 4 consecutive OUTPUT columns (16-byte aligned stores into the images); a wave 256, the workgroup 2048 per
// iteration; samples are requested three iterations ahead in three register sets.  ONE barrier per iteration: each wave
// publishes (sum of its adjustments, its first and last phase) and then every wave redoes the 8-entry scan over the waves.
constexpr int RW = 8;           // waves per row workgroup
constexpr int RT = RW * 64;     // threads
constexpr int RQ = 4;           // consecutive output columns per lane
constexpr int RSPAN = RT * RQ;  // columns per workgroup iteration

// torch.remainder(a, 2 pi) for a = dphi + pi.  Both phases come from atan2f, so a lies in [-pi, 3 pi] and fmodf reduces to at most
// one exact subtraction (Sterbenz: a - b is exact for b <= a <= 2b) or, for negative a, the single rounded add of py_mod: the
// three-way select below returns the same bits as the library call at a few instructions instead of a few hundred.
// (phi is this file's own atan2f output; outside [-2 pi, 4 pi) the select would not be a remainder.)
__device__ __forceinline__ float py_mod_2pi_near(float a) {
  return a >= TWO_PI_F ? a - TWO_PI_F : (a < 0.f ? a + TWO_PI_F : a);
}

__device__ __forceinline__ float unwrap_adj(float cur, float prev) {
  const float dphi = cur - prev;
  float dm = py_mod_2pi_near(dphi + PI_F) - PI_F;
  if (dm == -PI_F && dphi > 0.f) dm = PI_F;
  float adj = dm - dphi;
  if (fabsf(dphi) < PI_F) adj = 0.f;
  return adj;
}

template <int CTRL, int RMASK>
__device__ __forceinline__ double dpp_f64(double v) {  // lanes without a source (row edge, masked row) read 0.0
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, RMASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, RMASK, 0xf, false);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}

__device__ __forceinline__ double wave_incl_scan_f64(double v) {
  v += dpp_f64<0x111, 0xf>(v);  // row_shr:1
  v += dpp_f64<0x112, 0xf>(v);  // row_shr:2
  v += dpp_f64<0x114, 0xf>(v);  // row_shr:4
  v += dpp_f64<0x118, 0xf>(v);  // row_shr:8   -> inclusive scan inside each row of 16
  v += dpp_f64<0x142, 0xa>(v);  // row_bcast15 -> rows 1, 3 add the total of the row below
  v += dpp_f64<0x143, 0xc>(v);  // row_bcast31 -> rows 2, 3 add the total of rows 0-1
  return v;
}

__device__ __forceinline__ double readlane_f64(double v, int l) {  // l wave-uniform
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)b, l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}

__global__ void __launch_bounds__(RT) codec_row_pass(const float2* __restrict__ X, const float* __restrict__ scale,
                                                     float* __restrict__ magn_out, float* __restrict__ phase_out,
                                                     float* __restrict__ part, int T, int nb_signed, int rem,
                                                     size_t img_stride) {
  __shared__ double tot_s[2][RW];
  __shared__ float first_s[2][RW], last_s[3][RW];  // last_s: three deep, see body()
  __shared__ float red[32];
  const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float sc = scale[k];
  const float2* __restrict__ Xr = X + (size_t)k * T;
  const int lead = 1 + rem;               // frames in front of output column 0 (frame 0 + the dropped remainder)
  const int J0 = -((lead + 3) & ~3);      // first column processed: 4-aligned, <= -lead (frames < 0 are masked)
  const int nout = T - lead;              // stored columns = S * nb
  const int niter = (nout - J0 + RSPAN - 1) / RSPAN;
  const bool vec = nb_signed > 0;         // host: a lane's 4 columns never straddle an image and are 16-byte aligned
  const int nb = vec ? nb_signed : -nb_signed;
  float mnm = INFINITY, mxm = -INFINITY, mnp = INFINITY, mxp = -INFINITY;
  double carry = 0.0;                     // running sum of the adjustments of all frames before this iteration's block

  auto load = [&](int it, float2 (&x)[RQ]) {  // unconditional, clamped: frames outside [0, T) re-read an edge frame and are masked
    const long long j = (long long)J0 + (long long)it * RSPAN + tid * RQ + lead;
#pragma unroll
    for (int e = 0; e < RQ; ++e) {
      long long t = j + e;
      t = t < 0 ? 0 : (t > T - 1 ? T - 1 : t);
      x[e] = Xr[t];
    }
  };
  auto body = [&](int it, int q, float2 (&x)[RQ]) {  // q = it % 3
    const int p = it & 1;
    const int jw = J0 + it * RSPAN;        // column of the workgroup's first element
    const int j0 = jw + tid * RQ;          // this lane's first column; frame t = column + lead
    float m[RQ], ph[RQ];
#pragma unroll
    for (int e = 0; e < RQ; ++e) {
      // th.abs / th.angle with torch's own bits (sleef_f32.h; rocm's hypotf / atan2f differ from them in the last bit on a third
      // of the bins and are 0.08 ms per 10-minute file cheaper)
      slf::abs_angle(x[e].x, x[e].y, m[e], ph[e]);
      m[e] *= sc;
    }
    load(it + 3, x);
    auto valid = [&](int col) { return col + lead >= 1 && col < nout; };  // 1 <= t <= T-1
    // adjustments; the wave's very first element (lane 0) needs the previous wave's last phase: after the barrier
    const float left = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ph[RQ - 1]), 0x138, 0xf, 0xf, false));
    float adj[RQ];
    adj[0] = (lane > 0 && valid(j0)) ? unwrap_adj(ph[0], left) : 0.f;
#pragma unroll
    for (int e = 1; e < RQ; ++e) adj[e] = valid(j0 + e) ? unwrap_adj(ph[e], ph[e - 1]) : 0.f;
    double sl[RQ];
    sl[0] = (double)adj[0];
#pragma unroll
    for (int e = 1; e < RQ; ++e) sl[e] = sl[e - 1] + (double)adj[e];
    const double incl = wave_incl_scan_f64(sl[RQ - 1]);
    if (lane == 63) {
      tot_s[p][wave] = incl;
      last_s[q][wave] = ph[RQ - 1];
    }
    if (lane == 0) first_s[p][wave] = ph[0];
    __syncthreads();
    // scan over the waves, redone by every wave in its lanes 0..RW-1 (entry l = wave l; lanes >= RW mirror the last entry)
    const int l = lane < RW ? lane : RW - 1;
    // The previous block's last phase is read AFTER this block's barrier while the last wave may already be writing the next
    // block's: with two slots that write could overtake a slow reader (nothing orders them), so this array has three.
    const float lp = l > 0 ? last_s[q][l - 1] : last_s[q == 0 ? 2 : q - 1][RW - 1];  // (first iteration: unwritten, masked by valid())
    const float af = valid(jw + l * 64 * RQ) ? unwrap_adj(first_s[p][l], lp) : 0.f;
    const double tl = tot_s[p][l];
    double v = (double)af + tl;
    v += dpp_f64<0x111, 0xf>(v);
    v += dpp_f64<0x112, 0xf>(v);
    v += dpp_f64<0x114, 0xf>(v);
    const double off = carry + (readlane_f64(v, wave) - readlane_f64(tl, wave));  // everything before this wave + its own first adj
    const float af_w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, af), wave));
    const float lp_w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lp), wave));
    carry += readlane_f64(v, RW - 1);
    const double base = off + (incl - sl[RQ - 1]);
    float d[RQ];
#pragma unroll
    for (int e = 0; e < RQ; ++e) {
      const double c = base + sl[e];                       // cumulative adjustment up to and including this frame (exact)
      const float a = e == 0 ? (lane > 0 ? adj[0] : af_w) : adj[e];
      const float pv = e == 0 ? (lane > 0 ? left : lp_w) : ph[e - 1];
      const float u = ph[e] + (float)c;                    // unwrapped[t]   = phi[t]   + float32(cumsum[t])
      const float up = pv + (float)(c - (double)a);        // unwrapped[t-1] = phi[t-1] + float32(cumsum[t-1])
      d[e] = u - up;
      if (valid(j0 + e)) {
        mnm = fminf(mnm, m[e]);
        mxm = fmaxf(mxm, m[e]);
        mnp = fminf(mnp, d[e]);
        mxp = fmaxf(mxp, d[e]);
      }
    }
    if (vec) {
      if (j0 >= 0 && j0 < nout) {
        const unsigned img = (unsigned)j0 / (unsigned)nb, col = (unsigned)j0 - img * (unsigned)nb;
        const size_t o = (size_t)img * img_stride + (size_t)k * nb + col;
        *reinterpret_cast<f32x4*>(magn_out + o) = f32x4{m[0], m[1], m[2], m[3]};
        *reinterpret_cast<f32x4*>(phase_out + o) = f32x4{d[0], d[1], d[2], d[3]};
      }
    } else {
#pragma unroll
      for (int e = 0; e < RQ; ++e) {
        const int j = j0 + e;
        if (j >= 0 && j < nout) {
          const unsigned img = (unsigned)j / (unsigned)nb, col = (unsigned)j - img * (unsigned)nb;
          const size_t o = (size_t)img * img_stride + (size_t)k * nb + col;
          magn_out[o] = m[e];
          phase_out[o] = d[e];
        }
      }
    }
  };

  float2 xa[RQ], xb[RQ], xc[RQ];
  load(0, xa);
  load(1, xb);
  load(2, xc);
  for (int it = 0; it < niter; it += 3) {  // (iterations past niter only touch masked columns)
    body(it, 0, xa);
    body(it + 1, 1, xb);
    body(it + 2, 2, xc);
  }
  block_minmax(mnm, mxm, red);
  float a0 = mnm, a1 = mxm;
  block_minmax(mnp, mxp, red);
  if (tid == 0) {
    part[4 * k] = a0;
    part[4 * k + 1] = a1;
    part[4 * k + 2] = mnp;
    part[4 * k + 3] = mxp;
  }
}

// ---- forward pass 2: global min / max from the 512 row records, then both images to [-1, 1] in place:
// (v - min) / (max - min) * 2 - 1 (functions.py:84-87).  blockIdx.y walks the images (plane = 512 * nb floats each, img_stride apart).
__global__ void __launch_bounds__(256) codec_normalize_inplace(float* __restrict__ magn, float* __restrict__ phase,
                                                               const float* __restrict__ part, int S, size_t plane,
                                                               size_t img_stride, int vec) {
  __shared__ float red[32];
  float mnm = INFINITY, mxm = -INFINITY, mnp = INFINITY, mxp = -INFINITY;
  for (int r = threadIdx.x; r < NB; r += 256) {
    const f32x4 q = *reinterpret_cast<const f32x4*>(part + 4 * r);
    mnm = fminf(mnm, q[0]);
    mxm = fmaxf(mxm, q[1]);
    mnp = fminf(mnp, q[2]);
    mxp = fmaxf(mxp, q[3]);
  }
  block_minmax(mnm, mxm, red);
  block_minmax(mnp, mxp, red);
  const float rm = mxm - mnm, rp = mxp - mnp;
  const size_t stride = (size_t)gridDim.x * blockDim.x, first = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int img = blockIdx.y; img < S; img += gridDim.y) {
    float* mi = magn + (size_t)img * img_stride;
    float* pi = phase + (size_t)img * img_stride;
    if (vec) {
      f32x4* m4 = reinterpret_cast<f32x4*>(mi);
      f32x4* p4 = reinterpret_cast<f32x4*>(pi);
      for (size_t i = first; i < plane / 4; i += stride) {
        f32x4 a = m4[i], b = p4[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a[e] = (a[e] - mnm) / rm * 2.f - 1.f;
          b[e] = (b[e] - mnp) / rp * 2.f - 1.f;
        }
        m4[i] = a;
        p4[i] = b;
      }
    } else {
      for (size_t i = first; i < plane; i += stride) {
        mi[i] = (mi[i] - mnm) / rm * 2.f - 1.f;
        pi[i] = (pi[i] - mnp) / rp * 2.f - 1.f;
      }
    }
  }
}

// ---- tiny: reduce (min,max) pairs
__global__ void __launch_bounds__(256) minmax_final(const float* __restrict__ part, int n, float* __restrict__ out) {
  __shared__ float red[32];
  float mn = INFINITY, mx = -INFINITY;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    mn = fminf(mn, part[2 * i]);
    mx = fmaxf(mx, part[2 * i + 1]);
  }
  block_minmax(mn, mx, red);
  if (threadIdx.x == 0) {
    out[0] = mn;
    out[1] = mx;
  }
}

// ---- inverse pass 1: m = ((magn + 1) / 2) / s[k]  and its global min/max   (input (N,2,512,W), time index = n*W + j)
__global__ void __launch_bounds__(256) inv_unbark(const float* __restrict__ mp, const float* __restrict__ scale,
                                                  float* __restrict__ m_out, float* __restrict__ part, int N, int W) {
  __shared__ float red[32];
  const int k = blockIdx.y;
  const int TT = N * W;
  const float s = scale[k];
  float mn = INFINITY, mx = -INFINITY;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < TT; t += gridDim.x * blockDim.x) {
    const int n = t / W, j = t - n * W;
    const float v = mp[(((size_t)n * 2 + 0) * NB + k) * W + j];
    const float m = (v + 1.f) / 2.f / s;
    m_out[(size_t)k * TT + t] = m;
    mn = fminf(mn, m);
    mx = fmaxf(mx, m);
  }
  block_minmax(mn, mx, red);
  if (threadIdx.x == 0) {
    part[2 * (blockIdx.y * gridDim.x + blockIdx.x)] = mn;
    part[2 * (blockIdx.y * gridDim.x + blockIdx.x) + 1] = mx;
  }
}

// ---- inverse pass 2: phase image -> [-pi, pi], cumulative sum over time (functions.py:115-118, a Python loop == sequential
// fp32 cumsum per frequency row), mod 2 pi, polar -> complex.  Only the running sum is serial: inv_phase_cumsum tiles it through
// LDS exactly like codec_unwrap_delta (4 loader waves, ONE scanner wave with lane = row, 4 storer waves; loads and stores in
// different waves) and writes the running phase; inv_polar then does the remainder / sincos / scaling on the whole chip.
constexpr int UT = 64;     // frames per tile
constexpr int USTR = 68;   // LDS row stride (floats): rows 16-byte aligned, 16-byte reads of 16 lanes hit 16 distinct slots
constexpr int UTILE = 64 * USTR;

__global__ void __launch_bounds__(576) inv_phase_cumsum(const float* __restrict__ mp, float* __restrict__ acc_out, int N,
                                                        int W) {
  extern __shared__ __attribute__((aligned(16))) float usm[];
  float* ph_s = usm;               // [2][64][USTR]
  float* ac_s = usm + 2 * UTILE;   // [2][64][USTR]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k0 = blockIdx.x * 64;
  const int TT = N * W;
  const int ntile = (TT + UT - 1) / UT;  // tile q covers frames t = 64 q + col
  const bool scanner = wave == 0, loader = wave >= 1 && wave <= 4, storer = wave >= 5;
  const int pcol = lane, prow0 = ((wave - 1) & 3) * 16;
  float r0[16], r1[16], r2[16];

  auto load_tile = [&](int q, float (&r)[16]) {  // unconditional: frames past the end re-read the last frame (never stored)
    const long long tq = (long long)q * UT + pcol;
    const int t = tq < TT ? (int)tq : TT - 1;
    const int n = t / W, j = t - n * W;
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = mp[(((size_t)n * 2 + 1) * NB + k0 + prow0 + i) * W + j];
  };
  auto stage_tile = [&](int q, const float (&r)[16]) {
    float* ph = ph_s + (q & 1) * UTILE;
#pragma unroll
    for (int i = 0; i < 16; ++i) ph[(prow0 + i) * USTR + pcol] = (r[i] + 1.f) / 2.f * 2.f * PI_F - PI_F;
  };
  auto store_tile = [&](int q) {
    const float* o = ac_s + (q & 1) * UTILE;
    const int t = q * UT + pcol;
    if (t < TT) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc_out[(size_t)(k0 + prow0 + i) * TT + t] = o[(prow0 + i) * USTR + pcol];
    }
  };
  float acc = 0.f;
  if (loader) {
    load_tile(0, r0);
    load_tile(1, r1);
    load_tile(2, r2);
    stage_tile(0, r0);
    load_tile(3, r0);
  }
  __syncthreads();
  auto scan_tile = [&](int q) {
    const float* ph = ph_s + (q & 1) * UTILE + lane * USTR;
    float* o = ac_s + (q & 1) * UTILE + lane * USTR;
#pragma unroll 4
    for (int j = 0; j < UT; j += 4) {
      const f32x4 pv = *reinterpret_cast<const f32x4*>(ph + j);
      f32x4 ov;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc = (q == 0 && j == 0 && e == 0) ? pv[e] : acc + pv[e];  // the first frame STARTS the sum (keeps a -0.0 phase as it is)
        ov[e] = acc;
      }
      *reinterpret_cast<f32x4*>(o + j) = ov;
    }
  };
  auto iteration = [&](int q, float (&r)[16]) {
    if (scanner) {
      if (q < ntile) scan_tile(q);
    } else if (loader) {
      stage_tile(q + 1, r);
      load_tile(q + 4, r);
    } else {
      if (q > 0 && q - 1 < ntile) store_tile(q - 1);
    }
    __syncthreads();
  };
  for (int q = 0; q < ntile; q += 3) {
    iteration(q, r1);
    iteration(q + 1, r2);
    iteration(q + 2, r0);
  }
  if (storer) {
    const int qe = ((ntile + 2) / 3) * 3;
    if (qe - 1 < ntile) store_tile(qe - 1);
  }
}

__global__ void __launch_bounds__(256) inv_polar(const float* __restrict__ acc, const float* __restrict__ m_in,
                                                 const float* __restrict__ mm, float2* __restrict__ Z, size_t total) {
  const float range = mm[1] - mm[0];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const float pm = py_mod(acc[i], TWO_PI_F);
    const float mag = m_in[i] / range;
    Z[i] = make_float2(mag * cosf(pm), mag * sinf(pm));
  }
}

__global__ void __launch_bounds__(256) inv_frames(const float2* __restrict__ Z, float* __restrict__ frames, int TT) {
  __shared__ float2 buf[NFFT];
  const int t = blockIdx.x;
  const int tid = threadIdx.x;
  // Hermitian-extend to a full 1024-point spectrum, bit-reversed order for an in-place DIT
  for (int i = tid; i < NFFT; i += 256) {
    float2 v;
    if (i < NB) v = Z[(size_t)i * TT + t];
    else if (i == NB) v = make_float2(0.f, 0.f);
    else {
      const float2 c = Z[(size_t)(NFFT - i) * TT + t];
      v = make_float2(c.x, -c.y);
    }
    if (i == 0) v.y = 0.f;  // irfft ignores the imaginary part of DC
    const int rev = __brev((unsigned)i) >> 22;
    buf[rev] = v;
  }
  __syncthreads();
  for (int len = 2; len <= NFFT; len <<= 1) {
    const int half = len >> 1;
    for (int i = tid; i < NFFT / 2; i += 256) {
      const int grp = i / half, pos = i - grp * half;
      const int a = grp * len + pos, b = a + half;
      float s, c;
      sincospif(2.0f * (float)pos / (float)len, &s, &c);  // inverse transform: e^{+2 pi i pos/len}
      const float2 x = buf[a], y = buf[b];
      const float2 wy = make_float2(y.x * c - y.y * s, y.x * s + y.y * c);
      buf[a] = make_float2(x.x + wy.x, x.y + wy.y);
      buf[b] = make_float2(x.x - wy.x, x.y - wy.y);
    }
    __syncthreads();
  }
  const float norm = 19.595917942265423f / (float)NFFT;  // sqrt(384) / N
  for (int i = tid; i < NFFT; i += 256) {
    float s, c;
    sincospif((float)i * (1.0f / 512.0f), &s, &c);
    const float w = 0.5f - 0.5f * c;
    frames[(size_t)t * NFFT + i] = buf[i].x * norm * w;
  }
}

// ---- inverse pass 4: overlap-add (gather: each output sample sums its <= 4 frames) / window envelope, centre trimmed
__global__ void __launch_bounds__(256) inv_overlap_add(const float* __restrict__ frames, float* __restrict__ wav, int TT,
                                                       long long out_len) {
  for (long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x; m < out_len;
       m += (long long)gridDim.x * blockDim.x) {
    const long long q = m + NFFT / 2;  // position in the un-trimmed signal
    int t_hi = (int)(q / HOP);
    if (t_hi > TT - 1) t_hi = TT - 1;
    float acc = 0.f, env = 0.f;
    for (int t = t_hi; t >= 0 && (long long)t * HOP + NFFT > q; --t) {
      const int i = (int)(q - (long long)t * HOP);
      float s, c;
      sincospif((float)i * (1.0f / 512.0f), &s, &c);
      const float w = 0.5f - 0.5f * c;
      acc += frames[(size_t)t * NFFT + i];
      env += w * w;
    }
    wav[m] = acc / env;
  }
}

}  // namespace

extern "C" size_t mg_codec_fwd_ws_bytes(int T) {
  (void)T;  // the per-row (min, max) records only: nothing of the track's size is staged any more
  return ((size_t)4 * NB + 16) * sizeof(float);
}

extern "C" int mg_codec_fwd_strided(const float* stft_c64, const float* bark_scale, float* magn_out, float* phase_out,
                                    size_t img_stride, void* ws, size_t ws_bytes, int T, int nb_vec, mg_stream_t stream) {
  MG_CHECK_ARG(stft_c64 && bark_scale && magn_out && phase_out && ws, "mg_codec_fwd: bad arguments");
  MG_CHECK_ARG(nb_vec > 0 && T - 1 >= nb_vec, "mg_codec_fwd: needs T-1 >= nb_vec (T=%d, nb_vec=%d)", T, nb_vec);
  MG_CHECK_ARG(T <= (1 << 24), "mg_codec_fwd: T=%d frames; the exact blocked scan is specified up to 2^24", T);
  MG_CHECK_ARG(img_stride >= (size_t)NB * nb_vec, "mg_codec_fwd: image stride smaller than an image");
  MG_CHECK_ARG(reinterpret_cast<uintptr_t>(ws) % 16 == 0 && reinterpret_cast<uintptr_t>(stft_c64) % 8 == 0,
               "mg_codec_fwd: the workspace must be 16-byte aligned, the input 8-byte aligned");
  if (ws_bytes < mg_codec_fwd_ws_bytes(T)) {
    mg_set_error("mg_codec_fwd: workspace too small");
    return MG_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  float* part = reinterpret_cast<float*>(ws);  // [512][mn_magn, mx_magn, mn_phase, mx_phase]
  const int S = (T - 1) / nb_vec;
  const int rem = (T - 1) % nb_vec;
  // 16-byte stores need 4-column groups that never straddle an image and aligned image bases; anything else: scalar stores
  const int vec = (nb_vec % 4 == 0 && img_stride % 4 == 0 &&
                   (reinterpret_cast<uintptr_t>(magn_out) | reinterpret_cast<uintptr_t>(phase_out)) % 16 == 0)
                      ? 1 : 0;
  hipLaunchKernelGGL(codec_row_pass, dim3(NB), dim3(RT), 0, s, reinterpret_cast<const float2*>(stft_c64), bark_scale,
                     magn_out, phase_out, part, T, vec ? nb_vec : -nb_vec, rem, img_stride);
  MG_CHECK_LAUNCH("mg_codec_fwd(row pass)");
  const size_t plane = (size_t)NB * nb_vec;
  int bx = (int)((plane / (vec ? 4 : 1) + 1023) / 1024);  // ~4 items per thread
  if (bx < 1) bx = 1;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(codec_normalize_inplace, dim3(bx, S > 1024 ? 1024 : S), dim3(256), 0, s, magn_out, phase_out, part, S,
                     plane, img_stride, vec);
  MG_CHECK_LAUNCH("mg_codec_fwd(normalize)");
  return MG_OK;
}

extern "C" int mg_codec_fwd(const float* stft_c64, const float* bark_scale, float* magn_out, float* phase_out, void* ws,
                            size_t ws_bytes, int T, int nb_vec, mg_stream_t stream) {
  return mg_codec_fwd_strided(stft_c64, bark_scale, magn_out, phase_out, (size_t)NB * (nb_vec > 0 ? nb_vec : 0), ws, ws_bytes,
                              T, nb_vec, stream);
}

extern "C" size_t mg_codec_inv_ws_bytes(int N, int W) {
  const size_t TT = (size_t)N * W;
  return ((size_t)NB * TT * 3 + TT * NFFT + 2 * (size_t)NB * 64 + 16) * sizeof(float);
}

extern "C" int mg_codec_inv(const float* magn_phase, const float* bark_scale, float* wav_out, void* ws, size_t ws_bytes,
                            int N, int W, mg_stream_t stream) {
  MG_CHECK_ARG(magn_phase && bark_scale && wav_out && ws && N > 0 && W > 0, "mg_codec_inv: bad arguments");
  MG_CHECK_ARG((long long)N * W >= 2, "mg_codec_inv: needs at least 2 frames");
  if (ws_bytes < mg_codec_inv_ws_bytes(N, W)) {
    mg_set_error("mg_codec_inv: workspace too small");
    return MG_EWORKSPACE;
  }
  const int TT = N * W;
  hipStream_t s = (hipStream_t)stream;
  float* m = reinterpret_cast<float*>(ws);
  float2* Z = reinterpret_cast<float2*>(m + (size_t)NB * TT);
  float* frames = m + (size_t)NB * TT * 3;
  float* part = frames + (size_t)TT * NFFT;
  float* mm = part + 2 * (size_t)NB * 64;
  int gx = (TT + 255) / 256;
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(inv_unbark, dim3(gx, NB), dim3(256), 0, s, magn_phase, bark_scale, m, part, N, W);
  hipLaunchKernelGGL(minmax_final, dim3(1), dim3(256), 0, s, part, gx * NB, mm);
  static MgPerDevice once;  // the LDS limit is a per-device function attribute
  if (mg_first_use_on_device(once)) {
    const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&inv_phase_cumsum),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    if (ea != hipSuccess) {
      mg_set_error("mg_codec_inv: hipFuncSetAttribute: %s", hipGetErrorString(ea));
      return MG_ELAUNCH;
    }
  }
  float* run = frames;  // running phase [NB][TT]: borrows the frame buffer (TT*1024 floats), which inv_frames fills afterwards
  hipLaunchKernelGGL(inv_phase_cumsum, dim3(NB / 64), dim3(576), (size_t)4 * UTILE * sizeof(float), s, magn_phase, run, N, W);
  {
    const size_t total = (size_t)NB * TT;
    int pb = (int)((total + 255) / 256);
    if (pb > 8192) pb = 8192;
    hipLaunchKernelGGL(inv_polar, dim3(pb), dim3(256), 0, s, run, m, mm, Z, total);
  }
  hipLaunchKernelGGL(inv_frames, dim3(TT), dim3(256), 0, s, Z, frames, TT);
  const long long out_len = (long long)HOP * (TT - 1);
  int blocks = (int)((out_len + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(inv_overlap_add, dim3(blocks), dim3(256), 0, s, frames, wav_out, TT, out_len);
  MG_CHECK_LAUNCH("mg_codec_inv");
  return MG_OK;
}
