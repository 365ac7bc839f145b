// Element functions of the four weight re-packing layouts (one packed element / one filter per call), shared by the single-tensor
// pack kernels next to their convolution kernels and by the multi-tensor launch of pack.hip (mg_pack_multi: every stale layout of
// both networks in ONE launch per optimizer step instead of ~60).
#pragma once
#include "mg_common.h"

constexpr int MG_PACK_CC = 8;  // input channels per LDS chunk (CC / WCC of the convolution kernels)

// direct 3x3 form [Cin/8][9 taps][8][OPF];  dgrad=0: conv Ci->Co, W'[o][c][t] = w[o][c][t];  dgrad=1: conv Co->Ci, W'[o][c][t] = w[c][o][8-t]
__device__ __forceinline__ void pack_conv3x3_elem(size_t e, const float* __restrict__ w, float* __restrict__ wp, int Co, int Ci,
                                                  int dgrad) {
  const int cin_call = dgrad ? Co : Ci, cout_call = dgrad ? Ci : Co;
  const int OPF = 16 * mg_cdiv(cout_call, 16);
  const int o = (int)(e % OPF);
  size_t r = e / OPF;
  const int cl = (int)(r % MG_PACK_CC);
  r /= MG_PACK_CC;
  const int t = (int)(r % 9);
  const int ch = (int)(r / 9);
  const int c = ch * MG_PACK_CC + cl;
  float v = 0.f;
  if (c < cin_call && o < cout_call) v = dgrad ? w[((size_t)c * Ci + o) * 9 + (8 - t)] : w[((size_t)o * Ci + c) * 9 + t];
  wp[e] = v;
}
__host__ __device__ static inline size_t pack_conv3x3_total(int Co, int Ci, int dgrad) {
  const int cin_call = dgrad ? Co : Ci, cout_call = dgrad ? Ci : Co;
  return (size_t)mg_cdiv(cin_call, MG_PACK_CC) * 9 * MG_PACK_CC * (size_t)(16 * mg_cdiv(cout_call, 16));
}

// Winograd form U = G g G^T in MFMA operand order (wino3x3.hip); one thread per (chunk, out tile, lane, k-step) = one filter
__host__ __device__ static inline int pack_wino_nt_padded(int Cout) {
  const int nt = mg_cdiv(Cout, 16);
  const int p4 = mg_cdiv(nt, 4) * 4, p3 = mg_cdiv(nt, 3) * 3;
  return p4 > p3 ? p4 : p3;
}
__device__ __forceinline__ void pack_wino3x3_elem(size_t e, const float* __restrict__ w, float* __restrict__ up, int Co, int Ci,
                                                  int dgrad, int NT) {
  const int cin_call = dgrad ? Co : Ci, cout_call = dgrad ? Ci : Co;
  const int ks = (int)(e & 1);
  const int lane = (int)((e >> 1) & 63);
  const size_t r = e >> 7;
  const int ct = (int)(r % NT);
  const int ch = (int)(r / NT);
  const int c = ch * MG_PACK_CC + 2 * (lane >> 4) + ks;
  const int o = ct * 16 + (lane & 15);
  float g[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) g[t] = 0.f;
  if (c < cin_call && o < cout_call) {
    // dgrad=0: conv Ci->Co, g[t] = w[o][c][t];  dgrad=1: conv Co->Ci with the spatially flipped, transposed filter
#pragma unroll
    for (int t = 0; t < 9; ++t) g[t] = dgrad ? w[((size_t)c * Ci + o) * 9 + (8 - t)] : w[((size_t)o * Ci + c) * 9 + t];
  }
  float h[12];  // G g : 4x3
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float g0 = g[j], g1 = g[3 + j], g2 = g[6 + j];
    h[j] = g0;
    h[3 + j] = 0.5f * ((g0 + g1) + g2);
    h[6 + j] = 0.5f * ((g0 - g1) + g2);
    h[9 + j] = g2;
  }
  float* dst = up + (((size_t)ch * NT + ct) * 8) * 256 + lane * 4 + ks * 2;
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // (G g) G^T : row xi = i, columns nu0..nu3 -> pairs (nu0, nu3), (nu1, nu2)
    const float h0 = h[3 * i], h1 = h[3 * i + 1], h2 = h[3 * i + 2];
    *reinterpret_cast<float2*>(dst + (size_t)(2 * i) * 256) = make_float2(h0, h2);
    *reinterpret_cast<float2*>(dst + (size_t)(2 * i + 1) * 256) = make_float2(0.5f * ((h0 + h1) + h2), 0.5f * ((h0 - h1) + h2));
  }
}
__host__ __device__ static inline size_t pack_wino3x3_threads(int Co, int Ci, int dgrad) {
  const int cin_call = dgrad ? Co : Ci, cout_call = dgrad ? Ci : Co;
  return (size_t)mg_cdiv(cin_call, MG_PACK_CC) * pack_wino_nt_padded(cout_call) * 128;
}

// Winograd filters for an up-sampled input (wino_ups.hip): the 9 components (xi, nu), xi, nu in {0, 1, 3}, of U = G g G^T -- rows and
// columns 2 of the INPUT transform vanish on an up-sampled grid (forward), and drop out of the block-summed output transform (data
// gradient, dgrad = 1: flipped taps, channels transposed).  Factors folded in: forward 2 per index 1 (B^T d doubles the centre row /
// column); data gradient s = A 1 = (1, 2, ., -1).  Layout [chunk][out tile][component 3 i + j][64 lanes][2 k-steps]; one thread per
// (chunk, out tile, lane, k-step) = one (channel, out-channel) filter.
__device__ __forceinline__ void pack_winoups_elem(size_t e, const float* __restrict__ w, float* __restrict__ up, int Co, int Ci, int dgrad) {
  const int cin_call = dgrad ? Co : Ci, cout_call = dgrad ? Ci : Co;
  const int NT = mg_cdiv(cout_call, 16);
  const int ks = (int)(e & 1);
  const int lane = (int)((e >> 1) & 63);
  const size_t r = e >> 7;
  const int ct = (int)(r % NT);
  const int ch = (int)(r / NT);
  const int c = ch * MG_PACK_CC + 2 * (lane >> 4) + ks;
  const int o = ct * 16 + (lane & 15);
  float g[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) g[t] = 0.f;
  if (c < cin_call && o < cout_call) {
#pragma unroll
    for (int t = 0; t < 9; ++t) g[t] = dgrad ? w[((size_t)c * Ci + o) * 9 + (8 - t)] : w[((size_t)o * Ci + c) * 9 + t];
  }
  float h[12];  // G g : 4x3
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float g0 = g[j], g1 = g[3 + j], g2 = g[6 + j];
    h[j] = g0;
    h[3 + j] = 0.5f * ((g0 + g1) + g2);
    h[6 + j] = 0.5f * ((g0 - g1) + g2);
    h[9 + j] = g2;
  }
  const float f1 = 2.0f, f3 = dgrad ? -1.0f : 1.0f;  // factor of index 1 / index 3 (index 0: 1)
  float* dst = up + (((size_t)ch * NT + ct) * 9) * 128 + lane * 2 + ks;
  constexpr int XI[3] = {0, 1, 3};
#pragma unroll
  for (int ii = 0; ii < 3; ++ii) {
    const int i = XI[ii];
    const float fi = ii == 1 ? f1 : (ii == 2 ? f3 : 1.0f);
    const float h0 = h[3 * i], h1 = h[3 * i + 1], h2 = h[3 * i + 2];
    dst[(size_t)(3 * ii + 0) * 128] = fi * h0;
    dst[(size_t)(3 * ii + 1) * 128] = fi * f1 * (0.5f * ((h0 + h1) + h2));
    dst[(size_t)(3 * ii + 2) * 128] = fi * f3 * h2;
  }
}
__host__ __device__ static inline size_t pack_winoups_threads(int Co, int Ci, int dgrad) {
  const int cin_call = dgrad ? Co : Ci, cout_call = dgrad ? Ci : Co;
  return (size_t)mg_cdiv(cin_call, MG_PACK_CC) * mg_cdiv(cout_call, 16) * 128;
}
__host__ __device__ static inline size_t pack_winoups_total(int Co, int Ci, int dgrad) { return pack_winoups_threads(Co, Ci, dgrad) * 9; }

// taps k in {0,1,2} of the original filter landing on low-res offset t for sub-pixel phase p:  p=0: t=0 <- {0}, t=1 <- {1,2};  p=1: t=0 <- {0,1}, t=1 <- {2}
__device__ __forceinline__ float pack_subpixel_sum(const float* __restrict__ wk, int py, int ta, int px, int tb) {
  const int ky0 = py == 0 ? (ta == 0 ? 0 : 1) : (ta == 0 ? 0 : 2);
  const int ky1 = py == 0 ? (ta == 0 ? 0 : 2) : (ta == 0 ? 1 : 2);
  const int kx0 = px == 0 ? (tb == 0 ? 0 : 1) : (tb == 0 ? 0 : 2);
  const int kx1 = px == 0 ? (tb == 0 ? 0 : 2) : (tb == 0 ? 1 : 2);
  float v = 0.f;
  for (int ky = ky0; ky <= ky1; ++ky)
    for (int kx = kx0; kx <= kx1; ++kx) v += wk[ky * 3 + kx];
  return v;
}
// effective sub-pixel weights of Upsample(x2) -> Conv3x3, LDS image layout [Ci/8][16 = phase*4 + a*2 + b][8][OPF]
__device__ __forceinline__ void pack_upconv3x3_elem(size_t e, const float* __restrict__ w, float* __restrict__ wp, int Co, int Ci) {
  const int OPF = 16 * mg_cdiv(Co, 16);
  const int o = (int)(e % OPF);
  size_t r = e / OPF;
  const int cl = (int)(r % MG_PACK_CC);
  r /= MG_PACK_CC;
  const int q = (int)(r % 16);
  const int ch = (int)(r / 16);
  const int c = ch * MG_PACK_CC + cl;
  float v = 0.f;
  if (c < Ci && o < Co) v = pack_subpixel_sum(w + ((size_t)o * Ci + c) * 9, q >> 3, (q >> 1) & 1, (q >> 2) & 1, q & 1);
  wp[e] = v;
}
__host__ __device__ static inline size_t pack_upconv3x3_total(int Co, int Ci) {
  return (size_t)mg_cdiv(Ci, MG_PACK_CC) * 16 * MG_PACK_CC * (size_t)(16 * mg_cdiv(Co, 16));
}
// 4x4 stride-2 weights of the data gradient of that layer, layout [Co/8][16 = u*4 + v][8][OPF], OPF over Ci
__device__ __forceinline__ void pack_downconv_elem(size_t e, const float* __restrict__ w, float* __restrict__ wp, int Co, int Ci) {
  const int OPF = 16 * mg_cdiv(Ci, 16);
  const int c = (int)(e % OPF);  // forward input channel = output channel of this conv
  size_t r = e / OPF;
  const int cl = (int)(r % MG_PACK_CC);
  r /= MG_PACK_CC;
  const int t = (int)(r % 16);
  const int ch = (int)(r / 16);
  const int o = ch * MG_PACK_CC + cl;  // gradient (forward output) channel
  float v = 0.f;
  if (o < Co && c < Ci) {
    const int u = t >> 2, vv = t & 3;
    // offset u-1: -1 -> (p 1, t 1), 0 -> (0, 1), +1 -> (1, 0), +2 -> (0, 0)
    v = pack_subpixel_sum(w + ((size_t)o * Ci + c) * 9, (u == 0 || u == 2) ? 1 : 0, (u <= 1) ? 1 : 0, (vv == 0 || vv == 2) ? 1 : 0,
                          (vv <= 1) ? 1 : 0);
  }
  wp[e] = v;
}
__host__ __device__ static inline size_t pack_downconv_total(int Co, int Ci) {
  return (size_t)mg_cdiv(Co, MG_PACK_CC) * 16 * MG_PACK_CC * (size_t)(16 * mg_cdiv(Ci, 16));
}

// multi-layer small-map chains (smallnet.hip): the filter fragment IS the MFMA A operand, read from L2 straight into registers:
// [Cin/16 group g][tap][Cout/16 tile][lane = (oc % 16) + 16 * kq][s]  <->  filter (oc, ci = 16 g + 4 kq + s, tap); zero padded
__device__ __forceinline__ void pack_smallnet_elem(size_t e, const float* __restrict__ w, float* __restrict__ wp, int Co, int Ci,
                                                   int dgrad) {
  const int cin_call = dgrad ? Co : Ci, cout_call = dgrad ? Ci : Co;
  const int MT = mg_cdiv(cout_call, 16);
  const int s = (int)(e & 3), lane = (int)((e >> 2) & 63);
  size_t r = e >> 8;
  const int mt = (int)(r % MT);
  r /= MT;
  const int t = (int)(r % 9);
  const int g = (int)(r / 9);
  const int o = mt * 16 + (lane & 15), c = g * 16 + 4 * (lane >> 4) + s;
  float v = 0.f;
  if (c < cin_call && o < cout_call) v = dgrad ? w[((size_t)c * Ci + o) * 9 + (8 - t)] : w[((size_t)o * Ci + c) * 9 + t];
  wp[e] = v;
}
__host__ __device__ static inline size_t pack_smallnet_total(int Co, int Ci, int dgrad) {
  const int cin_call = dgrad ? Co : Ci, cout_call = dgrad ? Ci : Co;
  return (size_t)mg_cdiv(cin_call, 16) * 9 * mg_cdiv(cout_call, 16) * 256;
}
