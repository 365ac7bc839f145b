// Magnitude/phase codec around the STFT and the inverse STFT, fp32 with the reference's operation order:
//   forward  /root/reference/music_gan/audio/functions.py:65-94   (abs/angle, bark scale, unwrap, first difference,
//            global min/max -> [-1,1], drop the leading remainder, split into nb_vec-frame images)
//   inverse  functions.py:97-139 (un-bark, /(max-min), phase -> [-pi,pi], cumulative sum, mod 2pi, polar -> complex,
//            zero Nyquist row, inverse_spectrogram == window * irfft, overlap-add / window envelope, centre trim)
// The unwrap / cumulative sums are evaluated SEQUENTIALLY per frequency row in fp32, exactly as torch.cumsum does on the CPU:
// the running sum reaches hundreds of radians, so any re-association would change the low bits the reference produces.
// One lane per row walks time (the forward unwrap tiles it through LDS so that only the running sum itself is serial);
// everything else is embarrassingly parallel and HBM-bound (8 B in + 8 B out per bin).
#include "mg_common.h"

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

// ---- forward pass 1: magn = |X| * s[k], phi = atan2(im, re); per-block min/max of magn over t >= 1
__global__ void __launch_bounds__(256) codec_abs_angle(const float2* __restrict__ X, const float* __restrict__ scale,
                                                       float* __restrict__ magn, float* __restrict__ phi,
                                                       float* __restrict__ part, int T) {
  __shared__ float red[32];
  const int k = blockIdx.y;
  const float s = scale[k];
  float mn = INFINITY, mx = -INFINITY;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < T; t += gridDim.x * blockDim.x) {
    const float2 v = X[(size_t)k * T + t];
    const float m = hypotf(v.x, v.y) * s;
    magn[(size_t)k * T + t] = m;
    phi[(size_t)k * T + t] = atan2f(v.y, v.x);
    if (t >= 1) {
      mn = fminf(mn, m);
      mx = fmaxf(mx, m);
    }
  }
  block_minmax(mn, mx, red);
  if (threadIdx.x == 0) {
    part[2 * (blockIdx.y * gridDim.x + blockIdx.x)] = mn;
    part[2 * (blockIdx.y * gridDim.x + blockIdx.x) + 1] = mx;
  }
}

// ---- forward pass 2: per row, sequential unwrap (np.unwrap semantics incl. the -pi -> +pi fix), first difference of the
// unwrapped phase -> delta[k][t-1], t = 1..T-1; partial min/max.
// Only the running sum of the adjustments is sequential (c_t = c_{t-1} + adj_t, one fp32 add per frame, in frame order, exactly
// as torch.cumsum does it); everything else is parallel.  A workgroup owns 64 frequency rows and walks time in tiles of 64
// frames through LDS: 8 loader waves fetch a tile coalesced along t, compute the adjustments (wrap test, remainder, the
// -pi -> +pi fix) and write phi / adj tiles; ONE scanner wave (lane = row) runs the sequential part of the tile before --
// 3 adds per frame, operands by 16-byte LDS reads; 4 storer waves write the finished delta tile coalesced.  Loads and stores
// sit in different waves on purpose: gfx950 retires a wave's vector-memory operations in order, so a wave that interleaves
// them waits for a store round trip before every tile (measured: 5.8 us per tile instead of < 1).  One barrier per tile; all
// tiles double-buffered.
constexpr int UT = 64;     // frames per tile
constexpr int USTR = 68;   // LDS row stride (floats): rows 16-byte aligned, 16-byte reads of 16 lanes hit 16 distinct slots
constexpr int UTILE = 64 * USTR;
constexpr size_t UNWRAP_LDS = (size_t)6 * UTILE * sizeof(float);
constexpr int NLW = 8;             // loader waves
constexpr int RPL = 64 / NLW;      // rows per loader lane
constexpr int UNWRAP_THREADS = 64 * (1 + NLW + 4);

// torch.remainder(a, 2 pi) for a = dphi + pi.  Both phases come from atan2f, so a lies in [-pi, 3 pi] and fmodf reduces to at most
// one exact subtraction (Sterbenz: a - b is exact for b <= a <= 2b) or, for negative a, the single rounded add of py_mod: the
// three-way select below returns the same bits as the library call at a few instructions instead of a few hundred.
// (phi is this file's own atan2f output, codec_abs_angle above; outside [-2 pi, 4 pi) the select would not be a remainder.)
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

__global__ void __launch_bounds__(UNWRAP_THREADS) codec_unwrap_delta(const float* __restrict__ phi, float* __restrict__ delta,
                                                          float* __restrict__ part, int T) {
  extern __shared__ __attribute__((aligned(16))) float usm[];
  float* phi_s = usm;               // [2][64][USTR]
  float* adj_s = usm + 2 * UTILE;   // [2][64][USTR]
  float* out_s = usm + 4 * UTILE;   // [2][64][USTR]
  __shared__ float red[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k0 = blockIdx.x * 64;
  const int ntile = (T - 1 + UT - 1) / UT;  // tile q covers frames t = 1 + 64 q + col
  const bool scanner = wave == 0, loader = wave >= 1 && wave <= NLW, storer = wave > NLW;
  // producer geometry: column = lane (coalesced along t), 16 rows each.  Three register sets keep tiles q+1 .. q+3 in flight
  // (two full iterations of latency cover); phi[t-1] comes from the neighbour lane, for lane 0 from the previous tile's
  // last column (kept in scalars)
  const int pcol = lane, prow0 = loader ? (wave - 1) * RPL : (storer ? (wave - 1 - NLW) * 16 : 0);
  float r0[RPL], r1[RPL], r2[RPL];
  float lastcol[RPL];  // wave-uniform: column 63 of the tile staged last
  float mn = INFINITY, mx = -INFINITY;

  auto load_tile = [&](int q, float (&r)[RPL]) {
    // unconditional (frames past the end re-read frame T-1; their adjustment is forced to 0 and their delta never stored):
    // with divergent loads hipcc waits vmcnt(0) at every use and the three-tile prefetch collapses
    const long long tq = 1ll + (long long)q * UT + pcol;
    const int t = tq < T ? (int)tq : T - 1;
#pragma unroll
    for (int i = 0; i < RPL; ++i) r[i] = phi[(size_t)(k0 + prow0 + i) * T + t];
  };
  auto stage_tile = [&](int q, const float (&r)[RPL]) {  // registers -> phi / adj tiles of buffer q & 1 (past the end: adj 0)
    float* ph = phi_s + (q & 1) * UTILE;
    float* ad = adj_s + (q & 1) * UTILE;
    const bool ok = (1 + q * UT + pcol) < T;
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
      const float left = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, r[i]), 0x138, 0xf, 0xf, false));
      const float prev = pcol == 0 ? lastcol[i] : left;
      ph[(prow0 + i) * USTR + pcol] = r[i];
      ad[(prow0 + i) * USTR + pcol] = ok ? unwrap_adj(r[i], prev) : 0.f;
      lastcol[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r[i]), 63));
    }
  };
  auto store_tile = [&](int q) {  // finished delta tile -> global, coalesced along t; min/max on the way
    const float* o = out_s + (q & 1) * UTILE;
    const int t = 1 + q * UT + pcol;
    if (t < T) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float v = o[(prow0 + i) * USTR + pcol];
        delta[(size_t)(k0 + prow0 + i) * (T - 1) + (t - 1)] = v;
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
      }
    }
  };

  float c = 0.f, prev_u = 0.f;  // scanner state of row k0 + lane: running sum of adjustments, previous unwrapped value
  if (scanner) {
    prev_u = phi[(size_t)(k0 + lane) * T];  // unwrapped[0] = phi[0]
  } else if (loader) {
#pragma unroll
    for (int i = 0; i < RPL; ++i) lastcol[i] = phi[(size_t)(k0 + prow0 + i) * T];  // phi[0]: left neighbour of frame 1
    load_tile(0, r0);
    load_tile(1, r1);
    load_tile(2, r2);
    stage_tile(0, r0);
    load_tile(3, r0);
  }
  __syncthreads();
  auto scan_tile = [&](int q) {
    const float* ph = phi_s + (q & 1) * UTILE + lane * USTR;
    const float* ad = adj_s + (q & 1) * UTILE + lane * USTR;
    float* o = out_s + (q & 1) * UTILE + lane * USTR;
#pragma unroll 4
    for (int j = 0; j < UT; j += 4) {
      const f32x4 pv = *reinterpret_cast<const f32x4*>(ph + j);
      const f32x4 av = *reinterpret_cast<const f32x4*>(ad + j);
      f32x4 ov;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        c += av[e];
        const float u = pv[e] + c;
        ov[e] = u - prev_u;
        prev_u = u;
      }
      *reinterpret_cast<f32x4*>(o + j) = ov;
    }
  };
  // iteration q: the scanner runs tile q; the storers write tile q-1; the loaders stage tile q+1 (registers loaded two
  // iterations ago) and re-load that register set with tile q+4
  auto iteration = [&](int q, float (&r)[RPL]) {
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
    const int qe = ((ntile + 2) / 3) * 3;  // iterations run: tiles up to qe-2 are stored inside the loop
    if (qe - 1 < ntile) store_tile(qe - 1);
  }
  block_minmax(mn, mx, red);
  if (tid == 0) {
    part[2 * blockIdx.x] = mn;
    part[2 * blockIdx.x + 1] = mx;
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

// ---- forward pass 3: normalise to [-1,1], drop the leading remainder, chunk: out[s][k][j] = f(src[k][off + s*nb + j])
__global__ void __launch_bounds__(256) codec_normalize_chunk(const float* __restrict__ src, int row_stride, int off,
                                                             const float* __restrict__ mm, float* __restrict__ out, int S,
                                                             int nb) {
  const float mn = mm[0], mx = mm[1];
  const float range = mx - mn;
  const size_t total = (size_t)S * NB * nb;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % nb);
    const size_t r = i / nb;
    const int k = (int)(r % NB);
    const int s = (int)(r / NB);
    const float v = src[(size_t)k * row_stride + off + (size_t)s * nb + j];
    out[i] = (v - mn) / range * 2.f - 1.f;
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
  // magn[512*T] + phi[512*T] + delta[512*(T-1)] + partials
  return ((size_t)NB * T * 3 + 2 * (size_t)NB * 64 + 2 * NB + 16) * sizeof(float);
}

extern "C" int mg_codec_fwd(const float* stft_c64, const float* bark_scale, float* magn_out, float* phase_out, void* ws,
                            size_t ws_bytes, int T, int nb_vec, mg_stream_t stream) {
  MG_CHECK_ARG(stft_c64 && bark_scale && magn_out && phase_out && ws, "mg_codec_fwd: bad arguments");
  MG_CHECK_ARG(nb_vec > 0 && T - 1 >= nb_vec, "mg_codec_fwd: needs T-1 >= nb_vec (T=%d, nb_vec=%d)", T, nb_vec);
  if (ws_bytes < mg_codec_fwd_ws_bytes(T)) {
    mg_set_error("mg_codec_fwd: workspace too small");
    return MG_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  float* magn = reinterpret_cast<float*>(ws);
  float* phi = magn + (size_t)NB * T;
  float* delta = phi + (size_t)NB * T;
  float* part_m = delta + (size_t)NB * (T - 1);
  float* part_p = part_m + 2 * (size_t)NB * 64;
  float* mm = part_p + 2 * NB;  // [mn_m, mx_m, mn_p, mx_p]
  int gx = (T + 255) / 256;
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(codec_abs_angle, dim3(gx, NB), dim3(256), 0, s, reinterpret_cast<const float2*>(stft_c64),
                     bark_scale, magn, phi, part_m, T);
  static MgPerDevice once;  // the LDS limit is a per-device function attribute
  if (mg_first_use_on_device(once)) {
    const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&codec_unwrap_delta),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    if (ea != hipSuccess) {
      mg_set_error("mg_codec_fwd: hipFuncSetAttribute: %s", hipGetErrorString(ea));
      return MG_ELAUNCH;
    }
  }
  MG_CHECK_LAUNCH("mg_codec_fwd(abs_angle)");
  hipLaunchKernelGGL(codec_unwrap_delta, dim3(NB / 64), dim3(UNWRAP_THREADS), UNWRAP_LDS, s, phi, delta, part_p, T);
  MG_CHECK_LAUNCH("mg_codec_fwd(unwrap)");
  hipLaunchKernelGGL(minmax_final, dim3(1), dim3(256), 0, s, part_m, gx * NB, mm);
  hipLaunchKernelGGL(minmax_final, dim3(1), dim3(256), 0, s, part_p, NB / 64, mm + 2);
  const int S = (T - 1) / nb_vec;
  const int rem = (T - 1) % nb_vec;
  const size_t total = (size_t)S * NB * nb_vec;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  // magn[:, 1:] then drop `rem` leading frames: column offset 1 + rem in the T-wide rows; delta rows are (T-1) wide
  hipLaunchKernelGGL(codec_normalize_chunk, dim3(blocks), dim3(256), 0, s, magn, T, 1 + rem, mm, magn_out, S, nb_vec);
  hipLaunchKernelGGL(codec_normalize_chunk, dim3(blocks), dim3(256), 0, s, delta, T - 1, rem, mm + 2, phase_out, S,
                     nb_vec);
  MG_CHECK_LAUNCH("mg_codec_fwd");
  return MG_OK;
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
