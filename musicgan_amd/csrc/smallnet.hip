// Multi-layer chains on small maps: ONE launch carries an image through all <= 4x4 layers of a pass (include/musicgan_hip.h,
// mg_smallnet) instead of one launch per layer --
//   /root/reference/music_gan/networks/generator.py:15-40,67-76   (first generator blocks: conv, LeakyReLU, PixelNorm, Upsample)
//   /root/reference/music_gan/networks/discriminator.py:14-34,60-70,94-101   (last critic blocks, Flatten, Linear(160, 1))
// at the reference's batch 6 (train.py:43) and at BASELINE configs[0] / [1] those layers are 6-10 dependent launches per pass, each
// at its floor of a dispatch + a cold memory round trip + a 100-300-step dependent MFMA chain, for ~10 MFLOP per image.
//
// A workgroup (4 waves) owns `G` images.  Activations live in LDS as [pixel slot][channel]: (H+2) x (W+2) slots per image with a
// zero halo ring, channel stride CS = 16 * ceil(C / 16) + 4 floats, three buffers that the ops ping-pong between.  A 3x3
// convolution is the implicit GEMM  D[oc, pixel] = sum_{tap, ci} W[oc, ci, tap] * X[pixel + tap, ci]  on v_mfma_f32_16x16x4_f32
// (exact fp32) with the FILTER fragment as the A operand: it is read from L2 straight into operand registers -- the packed
// layout (pack_smallnet_elem) is the operand layout, lane (oc % 16, kq) holds input channels 16 g + 4 kq + {0..3} in one
// 16-byte load -- through a ring of 9 steps (one channel group's nine taps) x up to 3 out-channel tiles per wave, so ~100 KB of
// filter loads are in flight per CU and the stream never waits for the arithmetic; the activation fragment (B operand) is one
// ds_read_b128 per 16 pixels and step.  A lane ends up with 4 consecutive out-channels of one pixel = one 16-byte LDS store
// in the layout the next convolution reads.  Waves split the out-channel tiles (wave w: tiles w, w + 4, w + 8), every wave
// covers all pixels.  Point-wise / pooling / normalisation steps are small LDS passes between barriers; every tensor a later
// pass needs goes to global memory from the op that produces it.  1x1 maps take the centre tap with fp64 accumulation on the
// vector ALU (as conv3x3_tiny does: the classifier's gradient inherits the round-off of these sums one to one).
#include "mg_common.h"
#include "pack_kernels.h"

namespace {

constexpr int SN_THREADS = 256;
constexpr float SN_PN_EPS = 1e-8f;

struct SnProgram {
  mg_sn_op_t op[MG_SN_MAX_OPS];
  int nops, N, G, stride;  // stride: floats per LDS buffer
  float slope;
};

__host__ __device__ static inline int sn_cp(int C) { return 16 * mg_cdiv(C, 16); }
__host__ __device__ static inline int sn_cs(int C) { return sn_cp(C) + 4; }

struct Geo {  // one tensor's LDS geometry
  int H, W, W2, SPI, HW, CP, CS;
  __device__ Geo(int C, int h, int w) : H(h), W(w), W2(w + 2), SPI((h + 2) * (w + 2)), HW(h * w), CP(sn_cp(C)), CS(sn_cs(C)) {}
  __device__ int slot(int nl, int y, int x) const { return nl * SPI + (y + 1) * W2 + (x + 1); }
  __device__ int slot_px(int px) const {  // px = nl * HW + y * W + x
    const int nl = px / HW, r = px - nl * HW, y = r / W;
    return slot(nl, y, r - y * W);
  }
};

// element loop over [slot][channel < CP] of a G-image buffer: f(slot, nl, y, x, c, interior)
template <typename F>
__device__ __forceinline__ void for_slots(const Geo& g, int G, F f) {
  const int total = G * g.SPI * g.CP;
  for (int e = threadIdx.x; e < total; e += SN_THREADS) {
    const int c = e % g.CP, s = e / g.CP;
    const int nl = s / g.SPI, r = s - nl * g.SPI;
    const int yy = r / g.W2, xx = r - yy * g.W2;
    const bool inside = yy >= 1 && yy <= g.H && xx >= 1 && xx <= g.W;
    f(s, nl, yy - 1, xx - 1, c, inside);
  }
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ------------------------------------------------------------------------------------------------ the convolution core
// acc[j][nt] += sum over (channel group g, tap t) of A(g, t, tile mt0 + 4 j) x B(pixel tile nt shifted by tap t, group g)
template <int MTW, int NTW>
__device__ __forceinline__ void conv_core(const float* __restrict__ wpk, const float* src, int KG, int MT, int mt0, int CS, int W2,
                                          const int (&boff)[NTW], f32x4 (&acc)[MTW][NTW]) {
  const int lane = threadIdx.x & 63;
  const float* wl = wpk + (size_t)mt0 * 256 + lane * 4;
  const size_t gstride = (size_t)9 * MT * 256;
  f32x4 ring[9][MTW];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
#pragma unroll
    for (int j = 0; j < MTW; ++j) ring[t][j] = *reinterpret_cast<const f32x4*>(wl + ((size_t)t * MT + 4 * j) * 256);
    // issued in the order the loop consumes them: the wait in front of step t is then a counted one on both ways into the loop
    // (the scheduler had put slot 0 last: vmcnt(0) at the top of every iteration)
    __builtin_amdgcn_sched_barrier(0);
  }
  int toff[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) toff[t] = ((t / 3 - 1) * W2 + (t % 3 - 1)) * CS;
  f32x4 b[2][NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) b[0][nt] = *reinterpret_cast<const f32x4*>(src + boff[nt] + toff[0]);
  for (int g = 0; g < KG; ++g) {
    // Step (g, t): the activation fragment of the NEXT step is requested, the MFMAs of this one are issued, and its ring slot is
    // re-requested with the next group's filters (the last group re-requests itself: 9 x MTW KB nobody waits for) -- in that
    // order, pinned per step: left alone the scheduler sinks all re-requests to the end of the loop body (shorter live ranges)
    // and the next iteration opens with vmcnt(0), i.e. a memory round trip per channel group with the matrix pipe idle.
    const int gn = g + 1 < KG ? g + 1 : g;
    const float* wn = wl + (size_t)gn * gstride;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int cur = t & 1, nxt = cur ^ 1;
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
        b[nxt][nt] = *reinterpret_cast<const f32x4*>(src + boff[nt] + (t < 8 ? toff[t + 1] + g * 16 : toff[0] + gn * 16));
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < MTW; ++j)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt)
            acc[j][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[t][j][s], b[cur][nt][s], acc[j][nt], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < MTW; ++j) ring[t][j] = *reinterpret_cast<const f32x4*>(wn + ((size_t)t * MT + 4 * j) * 256);
      __builtin_amdgcn_sched_barrier(0);
    }
    // (nine steps per iteration is odd: the fragment buffers swap roles every iteration, so rotate them back)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) b[0][nt] = b[1][nt];
    // The filters are read-only memory, so without this the compiler replaces the ring by "load the current step, wait, use"
    // (fewer live registers, every step a full memory round trip).  Behind a memory clobber it cannot re-load what it holds.
    asm volatile("" ::: "memory");
  }
}

template <int MTW, int NTW>
__device__ __forceinline__ void conv_tiles(const mg_sn_op_t& o, const float* src, float* dst, int G, int img0, int N, float slope,
                                           int mt0) {
  const int lane = threadIdx.x & 63, col = lane & 15, q = lane >> 4;
  const Geo gi(o.C, o.H, o.W), go(o.C2, o.H, o.W);
  const int MT = go.CP / 16, KG = gi.CP / 16;
  const int npx = G * gi.HW;
  int boff[NTW], px[NTW];
  bool ok[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    px[nt] = nt * 16 + col;
    ok[nt] = px[nt] < npx && img0 + px[nt] / gi.HW < N;
    boff[nt] = gi.slot_px(px[nt] < npx ? px[nt] : 0) * gi.CS + 4 * q;  // lanes without a pixel read pixel 0 (results dropped)
  }
  // the mask source of the epilogue is requested before the K loop (its round trip hides under the filter stream)
  f32x4 mk[MTW][NTW];
  if (o.flags & MG_SN_MASK_AUX) {
#pragma unroll
    for (int j = 0; j < MTW; ++j)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const int oc0 = (mt0 + 4 * j) * 16 + 4 * q;
        const int nl = px[nt] / gi.HW, pr = px[nt] - nl * gi.HW;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          mk[j][nt][i] = (ok[nt] && oc0 + i < o.C2) ? o.aux[((size_t)(img0 + nl) * o.C2 + oc0 + i) * gi.HW + pr] : 1.f;
      }
  }
  f32x4 acc[MTW][NTW];
#pragma unroll
  for (int j = 0; j < MTW; ++j)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[j][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  conv_core<MTW, NTW>(o.in, src, KG, MT, mt0, gi.CS, gi.W2, boff, acc);
#pragma unroll
  for (int j = 0; j < MTW; ++j) {
    const int oc0 = (mt0 + 4 * j) * 16 + 4 * q;
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (o.bias != nullptr) {
#pragma unroll
      for (int i = 0; i < 4; ++i) bv[i] = oc0 + i < o.C2 ? o.bias[oc0 + i] : 0.f;
    }
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      if (!ok[nt]) continue;
      f32x4 v = acc[j][nt] + bv;
      if (o.flags & MG_SN_LRELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = mg_lrelu(v[i], slope);
      }
      if (o.flags & MG_SN_MASK_AUX) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] *= mg_lrelu_mask(mk[j][nt][i], slope);
      }
      *reinterpret_cast<f32x4*>(dst + go.slot_px(px[nt]) * go.CS + oc0) = v;
      if (o.out != nullptr) {
        const int nl = px[nt] / gi.HW, pr = px[nt] - nl * gi.HW;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (oc0 + i < o.C2) o.out[((size_t)(img0 + nl) * o.C2 + oc0 + i) * gi.HW + pr] = v[i];
      }
    }
  }
}

template <int NTW>
__device__ __forceinline__ void conv_op_nt(const mg_sn_op_t& o, const float* src, float* dst, int G, int img0, int N, float slope) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int MT = sn_cp(o.C2) / 16;
  const int mtw = wave < MT ? (MT - wave + 3) / 4 : 0;  // tiles wave, wave + 4, wave + 8
  if (mtw == 1) conv_tiles<1, NTW>(o, src, dst, G, img0, N, slope, wave);
  else if (mtw == 2) conv_tiles<2, NTW>(o, src, dst, G, img0, N, slope, wave);
  else if (mtw == 3) conv_tiles<3, NTW>(o, src, dst, G, img0, N, slope, wave);
}

// zero the halo ring of a destination tensor (its interior is written by someone else: disjoint addresses, no barrier between)
__device__ __forceinline__ void zero_halo(float* dst, const Geo& g, int G) {
  const int quads = g.CP / 4, total = G * g.SPI * quads;
  for (int e = threadIdx.x; e < total; e += SN_THREADS) {
    const int cq = e % quads, s = e / quads;
    const int r = s % g.SPI, yy = r / g.W2, xx = r - yy * g.W2;
    if (!(yy >= 1 && yy <= g.H && xx >= 1 && xx <= g.W)) *reinterpret_cast<f32x4*>(dst + s * g.CS + cq * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

// 3x3 convolution on a 1x1 map = the centre tap as a matrix-vector product; one thread per out-channel, fp64 accumulation
__device__ __forceinline__ void conv_1x1map(const mg_sn_op_t& o, const float* src, float* dst, int G, int img0, int N, float slope) {
  const Geo gi(o.C, 1, 1), go(o.C2, 1, 1);
  const int MT = go.CP / 16, KG = gi.CP / 16;
  const int oc = threadIdx.x;
  if (oc < go.CP) {
    const float* wl = o.in + ((size_t)4 * MT + (oc >> 4)) * 256 + (oc & 15) * 4;  // tap 4, tile oc / 16, lane (oc % 16) + 16 kq
    for (int nl = 0; nl < G; ++nl) {
      if (img0 + nl >= N) break;
      const float* x = src + gi.slot(nl, 0, 0) * gi.CS;
      double acc = 0.0;
      for (int g = 0; g < KG; ++g) {
        f32x4 wv[4];
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) wv[kq] = *reinterpret_cast<const f32x4*>(wl + (size_t)g * 9 * MT * 256 + kq * 64);
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
          const f32x4 xv = *reinterpret_cast<const f32x4*>(x + g * 16 + kq * 4);
#pragma unroll
          for (int s = 0; s < 4; ++s) acc = fma((double)wv[kq][s], (double)xv[s], acc);
        }
      }
      float v = 0.f;
      if (oc < o.C2) {
        v = (float)(o.bias != nullptr ? acc + (double)o.bias[oc] : acc);
        if (o.flags & MG_SN_LRELU) v = mg_lrelu(v, slope);
        if (o.flags & MG_SN_MASK_AUX) v *= mg_lrelu_mask(o.aux[(size_t)(img0 + nl) * o.C2 + oc], slope);
        if (o.out != nullptr) o.out[(size_t)(img0 + nl) * o.C2 + oc] = v;
      }
      dst[go.slot(nl, 0, 0) * go.CS + oc] = v;
    }
  }
  zero_halo(dst, go, G);
}

__global__ void __launch_bounds__(SN_THREADS) smallnet_k(const SnProgram P) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int G = P.G, N = P.N;
  const int img0 = blockIdx.x * G;
  const float slope = P.slope;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = 0; k < P.nops; ++k) {
    const mg_sn_op_t& o = P.op[k];
    float* src = smem + (size_t)o.src * P.stride;
    float* dst = smem + (size_t)o.dst * P.stride;
    switch (o.op) {
      case MG_SN_LOAD: {
        const Geo g(o.C, o.H, o.W);
        for_slots(g, G, [&](int s, int nl, int y, int x, int c, bool in) {
          float v = 0.f;
          if (in && c < o.C && img0 + nl < N) v = o.in[((size_t)(img0 + nl) * o.C + c) * g.HW + y * g.W + x];
          dst[s * g.CS + c] = v;
        });
        break;
      }
      case MG_SN_STORE: {
        const Geo g(o.C, o.H, o.W);
        for_slots(g, G, [&](int s, int nl, int y, int x, int c, bool in) {
          if (in && c < o.C && img0 + nl < N) o.out[((size_t)(img0 + nl) * o.C + c) * g.HW + y * g.W + x] = src[s * g.CS + c];
        });
        break;
      }
      case MG_SN_CONV: {
        if (o.H == 1 && o.W == 1) {
          conv_1x1map(o, src, dst, G, img0, N, slope);
          break;
        }
        const Geo go(o.C2, o.H, o.W);
        zero_halo(dst, go, G);
        const int ntw = (G * o.H * o.W + 15) / 16;
        if (ntw == 1) conv_op_nt<1>(o, src, dst, G, img0, N, slope);
        else if (ntw == 2) conv_op_nt<2>(o, src, dst, G, img0, N, slope);
        else conv_op_nt<4>(o, src, dst, G, img0, N, slope);
        break;
      }
      case MG_SN_MASK: {
        const Geo g(o.C, o.H, o.W);
        for_slots(g, G, [&](int s, int nl, int y, int x, int c, bool in) {
          if (in && c < o.C && img0 + nl < N) {
            const size_t gi = ((size_t)(img0 + nl) * o.C + c) * g.HW + y * g.W + x;
            const float v = src[s * g.CS + c] * mg_lrelu_mask(o.aux[gi], slope);
            src[s * g.CS + c] = v;
            if (o.out != nullptr) o.out[gi] = v;
          }
        });
        break;
      }
      case MG_SN_PIXNORM:
      case MG_SN_PNBWD: {
        // one wave per pixel (pixels wave, wave + 4, ...), lanes over channels, DPP sum over the wave
        const Geo g(o.C, o.H, o.W);
        const int npx = G * g.HW;
        for (int px = wave; px < npx; px += 4) {
          const int nl = px / g.HW, pr = px - nl * g.HW;
          if (img0 + nl >= N) break;
          float* row = src + g.slot_px(px) * g.CS;
          const size_t gbase = (size_t)(img0 + nl) * o.C * g.HW + pr;
          float v[3], pv[3];
          float part = 0.f;
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const int c = lane + 64 * i;
            v[i] = c < o.C ? row[c] : 0.f;
            if (o.op == MG_SN_PNBWD) {
              pv[i] = c < o.C ? o.in[gbase + (size_t)c * g.HW] : 0.f;
              part = fmaf(v[i], pv[i], part);
            } else {
              part = fmaf(v[i], v[i], part);
            }
          }
          const float tot = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mg_wave_sum_to_lane63(part)), 63));
          if (o.op == MG_SN_PIXNORM) {
            const float r = 1.0f / sqrtf(tot / (float)o.C + SN_PN_EPS);
            if (lane == 0 && o.out2 != nullptr) o.out2[(size_t)(img0 + nl) * g.HW + pr] = r;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
              const int c = lane + 64 * i;
              if (c < o.C) {
                const float pn = v[i] * r;
                row[c] = pn;
                if (o.out != nullptr) o.out[gbase + (size_t)c * g.HW] = pn;
              }
            }
          } else {
            const float r = o.aux[(size_t)(img0 + nl) * g.HW + pr];
            const float dot = tot / (float)o.C;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
              const int c = lane + 64 * i;
              if (c < o.C) {
                const float gp = mg_lrelu_mask(pv[i], slope) * r * (v[i] - pv[i] * dot);
                row[c] = gp;
                if (o.out != nullptr) o.out[gbase + (size_t)c * g.HW] = gp;
              }
            }
          }
        }
        break;
      }
      case MG_SN_POOL: {
        const Geo gi(o.C, o.H, o.W), go(o.C, o.H / 2, o.W / 2);
        for_slots(go, G, [&](int s, int nl, int y, int x, int c, bool in) {
          float v = 0.f;
          if (in && c < o.C && img0 + nl < N) {
            const float* p0 = src + gi.slot(nl, 2 * y, 2 * x) * gi.CS + c;
            v = ((p0[0] + p0[gi.CS]) + (p0[gi.W2 * gi.CS] + p0[(gi.W2 + 1) * gi.CS])) * 0.25f;
            if (o.out != nullptr) o.out[((size_t)(img0 + nl) * o.C + c) * go.HW + y * go.W + x] = v;
          }
          dst[s * go.CS + c] = v;
        });
        break;
      }
      case MG_SN_POOLBWD: {
        const Geo gi(o.C, o.H, o.W), go(o.C, 2 * o.H, 2 * o.W);
        const bool lds = !(o.flags & MG_SN_NOLDS);
        for_slots(go, G, [&](int s, int nl, int y, int x, int c, bool in) {
          float v = 0.f;
          if (in && c < o.C && img0 + nl < N) {
            const size_t gidx = ((size_t)(img0 + nl) * o.C + c) * go.HW + y * go.W + x;
            v = 0.25f * src[gi.slot(nl, y >> 1, x >> 1) * gi.CS + c] * mg_lrelu_mask(o.aux[gidx], slope);
            if (o.out != nullptr) o.out[gidx] = v;
          }
          if (lds) dst[s * go.CS + c] = v;
        });
        break;
      }
      case MG_SN_UP: {
        const Geo gi(o.C, o.H, o.W), go(o.C, 2 * o.H, 2 * o.W);
        for_slots(go, G, [&](int s, int nl, int y, int x, int c, bool in) {
          dst[s * go.CS + c] = in ? src[gi.slot(nl, y >> 1, x >> 1) * gi.CS + c] : 0.f;
        });
        break;
      }
      case MG_SN_UPBWD: {
        const Geo gi(o.C, o.H, o.W), go(o.C, o.H / 2, o.W / 2);
        for_slots(go, G, [&](int s, int nl, int y, int x, int c, bool in) {
          float v = 0.f;
          if (in) {
            const float* p0 = src + gi.slot(nl, 2 * y, 2 * x) * gi.CS + c;
            v = (p0[0] + p0[gi.CS]) + (p0[gi.W2 * gi.CS] + p0[(gi.W2 + 1) * gi.CS]);
          }
          dst[s * go.CS + c] = v;
        });
        break;
      }
      case MG_SN_LINEAR: {
        const Geo g(o.C, 1, 1);
        for (int nl = wave; nl < G; nl += 4) {
          if (img0 + nl >= N) break;
          const float* x = src + g.slot(nl, 0, 0) * g.CS;
          double acc = 0.0;
          for (int c = lane; c < o.C; c += 64) acc = fma((double)o.in[c], (double)x[c], acc);
          acc = wave_sum_f64(acc);
          if (lane == 0) o.out[img0 + nl] = (float)(acc + (double)o.aux[0]);
        }
        break;
      }
      case MG_SN_LINBWD: {
        const Geo g(o.C, 1, 1);
        for_slots(g, G, [&](int s, int nl, int y, int x, int c, bool in) {
          dst[s * g.CS + c] = (in && c < o.C && img0 + nl < N) ? o.in[img0 + nl] * o.aux[c] : 0.f;
        });
        break;
      }
      default: break;
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" size_t mg_smallnet_packed_floats(int Cin, int Cout) { return pack_smallnet_total(Cout, Cin, 0); }

extern "C" size_t mg_smallnet_buffer_floats(int imgs_per_wg, int C, int H, int W) {
  return (size_t)imgs_per_wg * (H + 2) * (W + 2) * sn_cs(C);
}

extern "C" int mg_smallnet(const mg_sn_op_t* ops, int nops, int N, int imgs_per_wg, size_t lds_floats_per_buffer, float slope,
                           mg_stream_t stream) {
  MG_CHECK_ARG(ops && nops > 0 && nops <= MG_SN_MAX_OPS && N > 0 && imgs_per_wg > 0, "mg_smallnet: bad arguments");
  const size_t lds_bytes = 3 * lds_floats_per_buffer * sizeof(float);
  MG_CHECK_ARG(lds_bytes <= 160 * 1024 && (lds_floats_per_buffer & 3) == 0, "mg_smallnet: %zu bytes of LDS", lds_bytes);
  SnProgram P;
  const int G = imgs_per_wg;
  for (int k = 0; k < nops; ++k) {
    const mg_sn_op_t& o = ops[k];
    MG_CHECK_ARG(o.op >= MG_SN_LOAD && o.op <= MG_SN_LINBWD && o.src >= 0 && o.src < 3 && o.dst >= 0 && o.dst < 3 && o.C > 0 &&
                     o.C <= 192 && o.H >= 1 && o.W >= 1 && o.H <= 16 && o.W <= 16,
                 "mg_smallnet: op %d malformed", k);
    // every tensor the op touches must fit its LDS buffer
    int Ho = o.H, Wo = o.W, Co = o.C;
    if (o.op == MG_SN_POOL || o.op == MG_SN_UPBWD) {
      MG_CHECK_ARG((o.H % 2) == 0 && (o.W % 2) == 0, "mg_smallnet: op %d needs an even map", k);
      Ho /= 2, Wo /= 2;
    }
    if ((o.op == MG_SN_POOLBWD && !(o.flags & MG_SN_NOLDS)) || o.op == MG_SN_UP) Ho *= 2, Wo *= 2;
    if (o.op == MG_SN_CONV) {
      MG_CHECK_ARG(o.C2 > 0 && o.C2 <= 192 && o.in != nullptr && o.src != o.dst, "mg_smallnet: op %d: bad convolution", k);
      MG_CHECK_ARG(G * o.H * o.W <= 64 && G * o.H * o.W != 48, "mg_smallnet: op %d: %d pixels per workgroup", k, G * o.H * o.W);
      MG_CHECK_ARG(!(o.flags & MG_SN_MASK_AUX) || o.aux != nullptr, "mg_smallnet: op %d: mask without a source", k);
      Co = o.C2;
    }
    MG_CHECK_ARG(mg_smallnet_buffer_floats(G, o.C, o.H, o.W) <= lds_floats_per_buffer &&
                     mg_smallnet_buffer_floats(G, Co, Ho, Wo) <= lds_floats_per_buffer,
                 "mg_smallnet: op %d does not fit %zu floats of LDS per buffer", k, lds_floats_per_buffer);
    const bool need_in = o.op == MG_SN_LOAD || o.op == MG_SN_PNBWD || o.op == MG_SN_LINEAR || o.op == MG_SN_LINBWD;
    const bool need_aux = o.op == MG_SN_MASK || o.op == MG_SN_PNBWD || o.op == MG_SN_POOLBWD || o.op == MG_SN_LINEAR || o.op == MG_SN_LINBWD;
    const bool need_out = o.op == MG_SN_STORE || o.op == MG_SN_LINEAR || (o.op == MG_SN_POOLBWD && (o.flags & MG_SN_NOLDS));
    MG_CHECK_ARG((!need_in || o.in) && (!need_aux || o.aux) && (!need_out || o.out), "mg_smallnet: op %d lacks a pointer", k);
    MG_CHECK_ARG(!((o.op == MG_SN_POOL || o.op == MG_SN_POOLBWD || o.op == MG_SN_UP || o.op == MG_SN_UPBWD) && o.src == o.dst &&
                   !(o.flags & MG_SN_NOLDS)),
                 "mg_smallnet: op %d resamples in place", k);
    P.op[k] = o;
  }
  P.nops = nops;
  P.N = N;
  P.G = G;
  P.stride = (int)lds_floats_per_buffer;
  P.slope = slope;
  static MgPerDevice once;
  if (mg_first_use_on_device(once))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&smallnet_k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(smallnet_k, dim3((unsigned)mg_cdiv(N, G)), dim3(SN_THREADS), lds_bytes, (hipStream_t)stream, P);
  MG_CHECK_LAUNCH("mg_smallnet");
  return MG_OK;
}
