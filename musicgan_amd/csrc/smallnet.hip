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
#include <cstdlib>

#include "mg_common.h"
#include "pack_kernels.h"

extern "C" size_t mg_smallnet_buffer_floats(int imgs_per_wg, int C, int H, int W);

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

// G images (N, C, Hin, Win) -> LDS [slot][channel] with zero halo / zero pad channels; `ups`: nearest-upsampled x2 on the way.
// Loads in batches of 8 per thread, all issued before the first LDS store (left to itself hipcc waits for every load before it
// issues the next: one memory round trip per element), in global order (coalesced); the zero fill touches other addresses.
__device__ __forceinline__ void stage_images(const float* __restrict__ x, float* xs, const Geo& g, int G, int C, int img0, int N,
                                             bool ups, int Hfull = 0, int row0 = 0) {
  // `g` describes what is staged: g.H rows starting at image row `row0` of maps that are Hfull rows high (Hfull = 0: whole
  // images); the slot rows above / below them hold the neighbouring image rows where those exist (a band's halo is real data)
  if (Hfull == 0) Hfull = g.H;
  const int Hin = ups ? Hfull >> 1 : Hfull, Win = ups ? g.W >> 1 : g.W;
  const int nimg = N - img0 < G ? N - img0 : G;
  // image rows that exist among row0 - 1 .. row0 + g.H (slot rows 0 .. g.H + 1)
  const int ylo = row0 > 0 ? row0 - 1 : 0, yhi = row0 + g.H + 1 < Hfull ? row0 + g.H + 1 : Hfull;
  const int rows = yhi - ylo, PW = rows * g.W, npos = nimg * PW;
  const float* xb = x + (size_t)img0 * C * (Hin * Win);
  // A thread keeps ONE pixel position (image, row, column) and walks the channels in steps of `cpp` (threads / positions): the
  // only divisions are these, once; 16 loads are issued before the first LDS store -- every batch is a memory round trip (~2.5 us
  // on a tensor the previous launch wrote), and a 4x4 image of up to 256 channels, four 2x2 images or a two-row band of an 8x8
  // one take ONE.  Loads of a wave run along a row (and over consecutive channels): contiguous 32..128-byte pieces.
  for (int p0 = 0; p0 < npos; p0 += SN_THREADS) {
    const int cpp = npos - p0 >= SN_THREADS ? 1 : SN_THREADS / (npos - p0);
    const int pos = p0 + (int)threadIdx.x % (npos - p0 < SN_THREADS ? npos - p0 : SN_THREADS);
    const int c0 = npos - p0 >= SN_THREADS ? 0 : (int)threadIdx.x / (npos - p0);
    const bool live = c0 < cpp;
    const int nl = pos / PW, pr = pos - nl * PW, yr = pr / g.W, xx = pr - yr * g.W, y = ylo + yr;
    const float* src = xb + (size_t)nl * C * (Hin * Win) + (ups ? (y >> 1) * Win + (xx >> 1) : y * Win + xx);
    float* dst = xs + (nl * g.SPI + (y - row0 + 1) * g.W2 + xx + 1) * g.CS;
    constexpr int U = 16;
    for (int cb = c0; cb < C; cb += U * cpp) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = cb + u * cpp;
        v[u] = src[(size_t)(c < C ? c : C - 1) * (Hin * Win)];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = cb + u * cpp;
        if (live && c < C) dst[c] = v[u];
      }
    }
  }
  // what the loop above does not write -- left / right halo columns, rows outside the image, pad channels, images past the
  // batch -- is zero
  const int quads = g.CP / 4, slots = G * g.SPI;
  for (int e = threadIdx.x; e < slots * quads; e += SN_THREADS) {
    const int cq = e % quads, sl = e / quads;
    const int nl = sl / g.SPI, r = sl - nl * g.SPI, yy = r / g.W2, xx = r - yy * g.W2;
    const int y = row0 + yy - 1;
    const bool inside = xx >= 1 && xx <= g.W && y >= 0 && y < Hfull && nl < nimg;
    if (!inside) {
      *reinterpret_cast<f32x4*>(xs + sl * g.CS + cq * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    } else if (cq * 4 + 4 > C) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (cq * 4 + i >= C) xs[sl * g.CS + cq * 4 + i] = 0.f;
    }
  }
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ------------------------------------------------------------------------------------------------ the convolution core
// acc[j][nt] += sum over (channel group g, tap t) of A(g, t, tile mt0 + 4 j) x B(pixel tile nt shifted by tap t, group g)
// The filter ring of one wave: 9 taps x MTW out-channel tiles of ONE input-channel group, as buffer loads: address = base +
// per-lane offset (one VGPR, set once) + a scalar offset per (group, tap, tile).  fp32 MFMA and vector-ALU time add up on
// gfx950, so the 64-bit per-lane pointer arithmetic of plain global loads (~8 vector instructions per step) came straight out
// of the matrix rate.
template <int MTW>
struct FilterRing {
  f32x4 r[9][MTW];
  __amdgpu_buffer_rsrc_t rs;
  int voff, MT;
  __device__ __forceinline__ void init(const float* wpk, int KG, int MT_, int mt0) {
    rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wpk), 0, KG * 9 * MT_ * 1024, 0x00020000);
    voff = (mt0 * 256 + (threadIdx.x & 63) * 4) * 4;
    MT = MT_;
  }
  __device__ __forceinline__ f32x4 load(int g, int t, int j) const {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, ((g * 9 + t) * MT + 4 * j) * 1024, 0));
  }
  // group g0's filters, issued in the order the loop consumes them: the wait in front of step t is then a counted one on both
  // ways into the loop (the scheduler had put slot 0 last: vmcnt(0) at the top of every iteration)
  __device__ __forceinline__ void fill(int g0) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int j = 0; j < MTW; ++j) r[t][j] = load(g0, t, j);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
};

// acc[j][nt] += sum over (channel group g0 <= g < g1, tap t) of A(g, t, tile mt0 + 4 j) x B(pixel tile nt shifted by tap t, group
// g); all groups when the waves split the out-channel tiles, a quarter when they split K.  `F` holds group g0 (FilterRing::fill).
template <int MTW, int NTW>
__device__ __forceinline__ void conv_core(FilterRing<MTW>& F, const float* src, int g0, int g1, int CS, int W2, const int (&boff)[NTW],
                                          f32x4 (&acc)[MTW][NTW]) {
  if (g0 >= g1) return;
  auto& ring = F.r;
  auto wload = [&](int g, int t, int j) __attribute__((always_inline)) { return F.load(g, t, j); };
  int toff[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) toff[t] = ((t / 3 - 1) * W2 + (t % 3 - 1)) * CS;
  f32x4 b[3][NTW];  // activation fragments, three deep: nine steps per group, so slot (t % 3) lines up across groups
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) b[0][nt] = *reinterpret_cast<const f32x4*>(src + boff[nt] + toff[0] + g0 * 16);
  for (int g = g0; g < g1; ++g) {
    // Step (g, t): the activation fragment of the NEXT step is requested, the MFMAs of this one are issued, and its ring slot is
    // re-requested with the next group's filters (the last group re-requests itself: 9 x MTW KB nobody waits for) -- in that
    // order, pinned per step: left alone the scheduler sinks all re-requests to the end of the loop body (shorter live ranges)
    // and the next iteration opens with vmcnt(0), i.e. a memory round trip per channel group with the matrix pipe idle.
    const int gn = g + 1 < g1 ? g + 1 : g;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int cur = t % 3, nxt = (t + 1) % 3;
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
        b[nxt][nt] = *reinterpret_cast<const f32x4*>(src + boff[nt] + (t < 8 ? toff[t + 1] + g * 16 : toff[0] + gn * 16));
      __builtin_amdgcn_sched_barrier(0);  // (else the reads sink below the MFMAs and the next step opens with their latency)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < MTW; ++j)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt)
#ifdef SN_NO_MFMA  // ablation build (tools/bench_smallnet.py): the filter stream alone
            acc[j][nt][s] += ring[t][j][s] * b[cur][nt][s];
#else
            acc[j][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[t][j][s], b[cur][nt][s], acc[j][nt], 0, 0, 0);
#endif
#ifndef SN_NO_STREAM  // ablation build: the matrix instructions alone (the ring keeps the first group's filters)
#pragma unroll
      for (int j = 0; j < MTW; ++j) ring[t][j] = wload(gn, t, j);
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
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
  FilterRing<MTW> F;
  F.init(o.in, KG, MT, mt0);
  F.fill(0);
  conv_core<MTW, NTW>(F, src, 0, KG, gi.CS, gi.W2, boff, acc);
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
        stage_images(o.in, dst, g, G, o.C, img0, N, false);
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

// ------------------------------------------------------------------------------------------------ one layer, whole chip
// The same convolution as ONE launch per layer for maps of at most 8x8 with few images (mg_conv3x3_small): a workgroup takes G
// images x ONE 16-out-channel tile, stages the images' input channels in LDS ([slot][channel], as above), and its four waves
// split the input-channel groups (split-K), so a wave's whole filter share -- 2-3 groups x 9 taps x 1 KB -- is in flight from
// the first instruction: one memory round trip, a few dozen MFMAs, a partial-sum exchange through LDS and the epilogue.  The
// direct kernel of conv3x3.hip walks the same K as 16 dependent chunk round trips (13-16 us on these layers, whatever their
// size).  Epilogues: bias + LeakyReLU, LeakyReLU-derivative mask, AvgPool2d / 2x2 block sums of the result as a second output,
// AvgPool2d backward x mask as the output, nearest-upsampled input.
struct ScArgs {
  const float* x;
  const float* wpk;
  const float* bias;
  const float* aux;
  float* y;
  float* p;
  float* pn_p;   // SC_PN_IN: PixelNorm(x) (N, Cin, Hin, Win) and
  float* pn_rn;  //           its per-pixel 1 / norm (N, 1, Hin, Win), written by the workgroups of out-channel tile 0 (may be NULL)
  int N, Cin, Cout, H, W, flags, G;
  int R;  // image rows per workgroup (R < H: one image per workgroup, H / R bands of it; their halo rows are staged too)
  float slope, pool_scale;
};
constexpr int SC_UNPOOL = 1 << 20;  // internal: y is (N, Cout, 2H, 2W) = 0.25 * result * lrelu'(aux), aux of that shape
constexpr int SC_PN_IN = 1 << 21;   // internal: x is the un-normalised activation; PixelNorm is applied to the staged copy

template <int NTW>
__global__ void __launch_bounds__(SN_THREADS) smallconv_k(const ScArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int G = a.G, mt = blockIdx.y;
  const int nbands = a.H / a.R;
  const int img0 = (blockIdx.x / nbands) * G, row0 = (blockIdx.x % nbands) * a.R;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const Geo gi(a.Cin, a.R, a.W);
  const int HWf = a.H * a.W;  // the full map (global indexing); gi.HW is this workgroup's band
  const int MT = sn_cp(a.Cout) / 16, KG = gi.CP / 16;
  float* xs = smem;
  f32x4* part = reinterpret_cast<f32x4*>(smem + (size_t)G * gi.SPI * gi.CS);  // [4 waves][NTW][64]
  float* tile = reinterpret_cast<float*>(part + 4 * NTW * 64);                  // [16 out-channels][NTW * 16 + 1] (pooling)
  // the wave's first filter group is requested before the images are staged: both round trips run at once
  const int g0 = KG * wave / 4, g1 = KG * (wave + 1) / 4;
  FilterRing<1> F;
  F.init(a.wpk, KG, MT, mt);
  if (g0 < g1) F.fill(g0);
  stage_images(a.x, xs, gi, G, a.Cin, img0, a.N, a.flags & MG_CONV_UPS_IN, a.H, row0);
  __syncthreads();
  if (a.flags & SC_PN_IN) {
    // PixelNorm (layers.py:11-17) of the layer in front, applied to the staged pixels instead of by a launch of its own between two
    // small convolutions: every workgroup holds ALL input channels of its pixels (it needs them for K), so the channel reduction
    // is local -- one wave per pixel, lanes over channels, DPP sum; every out-channel tile's workgroup repeats it (a few hundred
    // cycles), tile 0's also writes p and 1/norm for the backward pass.  Halo rows of a band are neighbours' pixels: normalised
    // too, stored by their owners.  Up-sampled input: a staged pixel is a copy of its low-resolution one (same norm); the
    // even-coordinate copy stores.
    const bool ups = a.flags & MG_CONV_UPS_IN;
    const int Hin = ups ? a.H >> 1 : a.H, Win = ups ? a.W >> 1 : a.W;
    const int rows = gi.H + 2, npix = G * rows * gi.W;
    for (int pi = wave; pi < npix; pi += 4) {
      const int nl = pi / (rows * gi.W), r = pi - nl * (rows * gi.W), yy = r / gi.W, xx = r - yy * gi.W;
      const int y = row0 + yy - 1;
      if (y < 0 || y >= a.H || img0 + nl >= a.N) continue;  // (wave-uniform)
      float* row = xs + (nl * gi.SPI + yy * gi.W2 + xx + 1) * gi.CS;
      float v[3];
      float part = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < a.Cin ? row[c] : 0.f;
        part = fmaf(v[i], v[i], part);
      }
      const float tot = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mg_wave_sum_to_lane63(part)), 63));
      const float rn = 1.0f / sqrtf(tot / (float)a.Cin + SN_PN_EPS);
      const bool owner = mt == 0 && yy >= 1 && yy <= gi.H && (!ups || (((y | xx) & 1) == 0));
      const size_t pix = (size_t)(ups ? (y >> 1) * Win + (xx >> 1) : y * Win + xx);
      if (owner && lane == 0 && a.pn_rn != nullptr) a.pn_rn[(size_t)(img0 + nl) * (Hin * Win) + pix] = rn;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int c = lane + 64 * i;
        if (c < a.Cin) {
          const float pn = v[i] * rn;
          row[c] = pn;
          if (owner && a.pn_p != nullptr) a.pn_p[((size_t)(img0 + nl) * a.Cin + c) * (Hin * Win) + pix] = pn;
        }
      }
    }
    __syncthreads();
  }
  const int npx = G * gi.HW;
  const int col = lane & 15, q = lane >> 4;
  int boff[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const int px = nt * 16 + col;
    boff[nt] = gi.slot_px(px < npx ? px : 0) * gi.CS + 4 * q;
  }
  f32x4 acc[1][NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) acc[0][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  conv_core<1, NTW>(F, xs, g0, g1, gi.CS, gi.W2, boff, acc);
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) part[(wave * NTW + nt) * 64 + lane] = acc[0][nt];
  __syncthreads();
  const bool pool = a.flags & MG_CONV_POOL_OUT;
  constexpr int TS = NTW * 16 + 1;
  for (int idx = threadIdx.x; idx < NTW * 64; idx += SN_THREADS) {
    const int nt = idx >> 6, l = idx & 63;
    f32x4 v = (part[nt * 64 + l] + part[(NTW + nt) * 64 + l]) + (part[(2 * NTW + nt) * 64 + l] + part[(3 * NTW + nt) * 64 + l]);
    const int px = nt * 16 + (l & 15), oc0 = mt * 16 + 4 * (l >> 4);
    const int nl = px / gi.HW, prl = px - nl * gi.HW;
    const int pr = row0 * a.W + prl;  // position in the full map
    const bool ok = px < npx && img0 + nl < a.N;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int oc = oc0 + i;
      float r = v[i];
      if (a.bias != nullptr && oc < a.Cout) r += a.bias[oc];
      if (a.flags & MG_CONV_LRELU) r = mg_lrelu(r, a.slope);
      if (ok && oc < a.Cout) {
        const size_t o = ((size_t)(img0 + nl) * a.Cout + oc) * HWf + pr;
        if (a.flags & SC_UNPOOL) {
          const int y = pr / a.W, x = pr - y * a.W, W2o = 2 * a.W;
          const size_t o2 = ((size_t)(img0 + nl) * a.Cout + oc) * (4 * HWf) + (size_t)(2 * y) * W2o + 2 * x;
          const float2 m0 = *reinterpret_cast<const float2*>(a.aux + o2), m1 = *reinterpret_cast<const float2*>(a.aux + o2 + W2o);
          const float h = 0.25f * r;
          *reinterpret_cast<float2*>(a.y + o2) = make_float2(h * mg_lrelu_mask(m0.x, a.slope), h * mg_lrelu_mask(m0.y, a.slope));
          *reinterpret_cast<float2*>(a.y + o2 + W2o) = make_float2(h * mg_lrelu_mask(m1.x, a.slope), h * mg_lrelu_mask(m1.y, a.slope));
        } else {
          if (a.flags & MG_CONV_MASK_AUX) r *= mg_lrelu_mask(a.aux[o], a.slope);
          if (a.y != nullptr) a.y[o] = r;
        }
      }
      if (pool) tile[(4 * (l >> 4) + i) * TS + px] = r;
    }
  }
  if (pool) {  // second output: 2x2 block means (AvgPool2d) or sums (nearest-upsample backward) of the result
    __syncthreads();
    const int Rp = a.R >> 1, Wp = a.W >> 1, bp = Rp * Wp, HWp = (a.H >> 1) * Wp;
    for (int e = threadIdx.x; e < 16 * G * bp; e += SN_THREADS) {
      const int ocl = e / (G * bp), r = e - ocl * (G * bp);
      const int nl = r / bp, pp = r - nl * bp, yy = pp / Wp, xx = pp - yy * Wp;
      const int oc = mt * 16 + ocl;
      if (oc < a.Cout && img0 + nl < a.N) {
        const float* t0 = tile + ocl * TS + nl * gi.HW + (2 * yy) * a.W + 2 * xx;
        a.p[((size_t)(img0 + nl) * a.Cout + oc) * HWp + (row0 >> 1) * Wp + pp] = ((t0[0] + t0[1]) + (t0[a.W] + t0[a.W + 1])) * a.pool_scale;
      }
    }
  }
}

}  // namespace

extern "C" int mg_conv3x3_small_supported(int N, int Cin, int Cout, int H, int W) {
  // whole images of at most 16 pixels, or bands of an even number of rows with 16, 32 or 64 pixels (W = 4, 8 or 16)
  return H >= 2 && W >= 2 && H <= 16 && W <= 16 && (H * W <= 16 || ((H % 2) == 0 && (W == 4 || W == 8 || W == 16))) && Cin <= 192 &&
         Cout <= 192 && N > 0;
}

static int small_conv_launch(const float* x, const float* wpk, const float* bias, const float* aux, float* y, float* p, float* pn_p,
                             float* pn_rn, bool pn_in, int N, int Cin, int Cout, int H, int W, int flags, float slope,
                             mg_stream_t stream);

extern "C" int mg_conv3x3_small(const float* x, const float* wpk, const float* bias, const float* aux, float* y, float* p, int N,
                                int Cin, int Cout, int H, int W, int flags, float slope, mg_stream_t stream) {
  return small_conv_launch(x, wpk, bias, aux, y, p, nullptr, nullptr, false, N, Cin, Cout, H, W, flags, slope, stream);
}

extern "C" int mg_conv3x3_small_pn(const float* x_raw, const float* wpk, const float* bias, float* y, float* pn_p, float* pn_rn, int N,
                                   int Cin, int Cout, int H, int W, int flags, float slope, mg_stream_t stream) {
  MG_CHECK_ARG(!(flags & ~(MG_CONV_UPS_IN | MG_CONV_LRELU)), "mg_conv3x3_small_pn: flags %d (UPS_IN and LRELU only)", flags);
  return small_conv_launch(x_raw, wpk, bias, nullptr, y, nullptr, pn_p, pn_rn, true, N, Cin, Cout, H, W, flags, slope, stream);
}

static int small_conv_launch(const float* x, const float* wpk, const float* bias, const float* aux, float* y, float* p, float* pn_p,
                             float* pn_rn, bool pn_in, int N, int Cin, int Cout, int H, int W, int flags, float slope,
                             mg_stream_t stream) {
  MG_CHECK_ARG(x && wpk && N > 0 && mg_conv3x3_small_supported(N, Cin, Cout, H, W), "mg_conv3x3_small: unsupported shape");
  const bool pool = flags & (MG_CONV_POOL_OUT | MG_CONV_UPSUM_OUT), unpool = flags & MG_CONV_UNPOOL;
  MG_CHECK_ARG(!(flags & ~(MG_CONV_UPS_IN | MG_CONV_LRELU | MG_CONV_MASK_AUX | MG_CONV_POOL_OUT | MG_CONV_UPSUM_OUT | MG_CONV_UNPOOL)),
               "mg_conv3x3_small: unsupported flags %d", flags);
  MG_CHECK_ARG(!pool || (p && (H % 2) == 0 && (W % 2) == 0), "mg_conv3x3_small: pooled output needs p and an even map");
  MG_CHECK_ARG(!((flags & MG_CONV_UPS_IN) && ((H % 2) || (W % 2))), "mg_conv3x3_small: upsampled input needs an even map");
  MG_CHECK_ARG(!(unpool || (flags & MG_CONV_MASK_AUX)) || aux, "mg_conv3x3_small: mask without a source");
  MG_CHECK_ARG(!unpool || (y && !pool && !(flags & (MG_CONV_MASK_AUX | MG_CONV_LRELU))), "mg_conv3x3_small: UNPOOL stands alone");
  MG_CHECK_ARG(y || pool, "mg_conv3x3_small: no output");
  ScArgs a;
  a.x = x; a.wpk = wpk; a.bias = bias; a.aux = aux; a.y = y; a.p = p; a.pn_p = pn_p; a.pn_rn = pn_rn;
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.slope = slope;
  a.flags = (flags & (MG_CONV_UPS_IN | MG_CONV_LRELU | MG_CONV_MASK_AUX)) | (pool ? MG_CONV_POOL_OUT : 0) | (unpool ? SC_UNPOOL : 0) | (pn_in ? SC_PN_IN : 0);
  a.pool_scale = (flags & MG_CONV_UPSUM_OUT) ? 1.0f : 0.25f;
  const int MT = mg_cdiv(Cout, 16), HW = H * W;
  // Images per workgroup: one, doubled while the grid is beyond ~1024 workgroups (every workgroup re-reads its out-channel tile's
  // filters: past a few workgroups per CU that traffic, not latency, is the run time), at most 64 pixels.  (Filling the
  // 16-pixel tile of a 2x2 map with four images first was measured slower at 24 images: 9.0 against 7.1 us.)
  static const int wg_target = getenv("MG_SMALLCONV_WGS") ? atoi(getenv("MG_SMALLCONV_WGS")) : 1024;
  int G = 1, R = H;
  if (HW > 16) {
    // maps of more than 16 pixels: ONE image per workgroup, split into bands of R rows (16, 32 or 64 pixels) while the grid
    // stays within the target -- a band stages its rows + one halo row either side
    for (R = 2; R < H && (R * W < 16 || (H % R) || (R & 1)); ++R) {}
    while (N * (H / R) * MT > wg_target && 2 * R <= H && 2 * R * W <= 64) R *= 2;
    while (R * W == 48 && 2 * R <= H) R *= 2;
    G = 1;
  } else {
    while (mg_cdiv(N, G) * MT > wg_target && 2 * G * HW <= 64) G *= 2;
    while (G > 1 && mg_cdiv(G * HW, 16) == 3) --G;  // 48 pixels would need three pixel tiles: instantiated for 1, 2, 4
  }
  a.G = G;
  a.R = R;
  const int tiles = mg_cdiv(G * R * W, 16);
  MG_CHECK_ARG(tiles <= 4 && tiles != 3 && (H % R) == 0, "mg_conv3x3_small: %d x %d map does not tile", H, W);
  const int NTW = tiles;
  const size_t lds = (mg_smallnet_buffer_floats(G, Cin, R, W) + (size_t)4 * NTW * 64 * 4 + (size_t)16 * (NTW * 16 + 1)) * sizeof(float);
  MG_CHECK_ARG(lds <= 160 * 1024, "mg_conv3x3_small: %zu bytes of LDS", lds);
  static MgPerDevice once;
  if (mg_first_use_on_device(once)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&smallconv_k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&smallconv_k<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&smallconv_k<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const dim3 grid((unsigned)(mg_cdiv(N, G) * (H / R)), (unsigned)MT);
  if (NTW == 1) hipLaunchKernelGGL(smallconv_k<1>, grid, dim3(SN_THREADS), lds, (hipStream_t)stream, a);
  else if (NTW == 2) hipLaunchKernelGGL(smallconv_k<2>, grid, dim3(SN_THREADS), lds, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(smallconv_k<4>, grid, dim3(SN_THREADS), lds, (hipStream_t)stream, a);
  MG_CHECK_LAUNCH("mg_conv3x3_small");
  return MG_OK;
}

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
