// Magnitude/phase codec around the STFT and the inverse STFT, fp32 with the reference's operation order:
//   forward  /root/reference/music_gan/audio/functions.py:65-94   (abs/angle, bark scale, unwrap, first difference,
//            global min/max -> [-1,1], drop the leading remainder, split into nb_vec-frame images)
//   inverse  functions.py:97-139 (un-bark, /(max-min), phase -> [-pi,pi], cumulative sum, mod 2pi, polar -> complex,
//            zero Nyquist row, inverse_spectrogram == window * irfft, overlap-add / window envelope, centre trim)
// The unwrap / cumulative sums are evaluated SEQUENTIALLY per frequency row in fp32, exactly as torch.cumsum does on the CPU:
// the running sum reaches hundreds of radians, so any re-association would change the low bits the reference produces.
// One thread per row walks time; everything else is embarrassingly parallel and HBM-bound (8 B in + 8 B out per bin).
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
// unwrapped phase -> delta[k][t-1], t = 1..T-1; per-row min/max
__global__ void __launch_bounds__(64) codec_unwrap_delta(const float* __restrict__ phi, float* __restrict__ delta,
                                                         float* __restrict__ part, int T) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= NB) return;
  const float* p = phi + (size_t)k * T;
  float* d = delta + (size_t)k * (T - 1);
  float prev = p[0];
  float c = 0.f;       // running cumsum of the adjustments (fp32, sequential)
  float prev_u = prev;  // unwrapped[0] = phi[0] + 0
  float mn = INFINITY, mx = -INFINITY;
  // the recurrence is sequential, the loads are not: fetch 16 frames ahead so their latency overlaps the scan
  constexpr int BLK = 16;
  int t = 1;
  for (; t + BLK <= T; t += BLK) {
    float cur[BLK], out[BLK];
#pragma unroll
    for (int j = 0; j < BLK; ++j) cur[j] = p[t + j];
#pragma unroll
    for (int j = 0; j < BLK; ++j) {
      const float dphi = cur[j] - prev;
      float dm = py_mod(dphi + PI_F, TWO_PI_F) - PI_F;
      if (dm == -PI_F && dphi > 0.f) dm = PI_F;
      float adj = dm - dphi;
      if (fabsf(dphi) < PI_F) adj = 0.f;
      c += adj;
      const float u = cur[j] + c;
      const float dl = u - prev_u;
      out[j] = dl;
      mn = fminf(mn, dl);
      mx = fmaxf(mx, dl);
      prev = cur[j];
      prev_u = u;
    }
#pragma unroll
    for (int j = 0; j < BLK; ++j) d[t - 1 + j] = out[j];
  }
  for (; t < T; ++t) {
    const float cur = p[t];
    const float dphi = cur - prev;
    float dm = py_mod(dphi + PI_F, TWO_PI_F) - PI_F;
    if (dm == -PI_F && dphi > 0.f) dm = PI_F;
    float adj = dm - dphi;
    if (fabsf(dphi) < PI_F) adj = 0.f;
    c += adj;
    const float u = cur + c;
    const float dl = u - prev_u;
    d[t - 1] = dl;
    mn = fminf(mn, dl);
    mx = fmaxf(mx, dl);
    prev = cur;
    prev_u = u;
  }
  part[2 * k] = mn;
  part[2 * k + 1] = mx;
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

// ---- inverse pass 2: per row, phase -> [-pi,pi], sequential cumulative sum, mod 2pi, polar -> complex spectrum Z[k][t]
__global__ void __launch_bounds__(64) inv_phase_polar(const float* __restrict__ mp, const float* __restrict__ m_in,
                                                      const float* __restrict__ mm, float2* __restrict__ Z, int N,
                                                      int W) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= NB) return;
  const int TT = N * W;
  const float range = mm[1] - mm[0];
  float acc = 0.f;
  for (int t = 0; t < TT; ++t) {
    const int n = t / W, j = t - n * W;
    const float v = mp[(((size_t)n * 2 + 1) * NB + k) * W + j];
    const float ph = (v + 1.f) / 2.f * 2.f * PI_F - PI_F;
    acc = (t == 0) ? ph : acc + ph;
    const float pm = py_mod(acc, TWO_PI_F);
    const float mag = m_in[(size_t)k * TT + t] / range;
    Z[(size_t)k * TT + t] = make_float2(mag * cosf(pm), mag * sinf(pm));
  }
}

// ---- inverse pass 3: per frame, 1024-point inverse real FFT (Nyquist bin = 0) times window * sqrt(sum w^2).
// Plain O(N log N) radix-2 in LDS, one workgroup per frame: the inverse path is tiny (generate: a few thousand frames).
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
  hipLaunchKernelGGL(codec_unwrap_delta, dim3(NB / 64), dim3(64), 0, s, phi, delta, part_p, T);
  hipLaunchKernelGGL(minmax_final, dim3(1), dim3(256), 0, s, part_m, gx * NB, mm);
  hipLaunchKernelGGL(minmax_final, dim3(1), dim3(256), 0, s, part_p, NB, mm + 2);
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
  hipLaunchKernelGGL(inv_phase_polar, dim3(NB / 64), dim3(64), 0, s, magn_phase, m, mm, Z, N, W);
  hipLaunchKernelGGL(inv_frames, dim3(TT), dim3(256), 0, s, Z, frames, TT);
  const long long out_len = (long long)HOP * (TT - 1);
  int blocks = (int)((out_len + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(inv_overlap_add, dim3(blocks), dim3(256), 0, s, frames, wav_out, TT, out_len);
  MG_CHECK_LAUNCH("mg_codec_inv");
  return MG_OK;
}
