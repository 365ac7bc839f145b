// nn.Upsample(x2, nearest) -> nn.Conv2d(3x3, pad 1) (+ LeakyReLU + PixelNorm) of the generator blocks
// (/root/reference/music_gan/networks/generator.py:26-39) evaluated in SUB-PIXEL form on the low-resolution input:
//
//   out[2y+py, 2x+px] = b + sum_c sum_{a,b in {0,1}} Weff[py][px][a][b][o][c] * in[y + py - 1 + a, x + px - 1 + b]
//   Weff[py][.][a][.] sums the original taps ky that land on low-res row offset a:  py=0: a=0 <- {0}, a=1 <- {1,2};
//                                                                                  py=1: a=0 <- {0,1}, a=1 <- {2}   (same in x)
//
// i.e. four 2x2 convolutions (one per output phase) instead of one 3x3 over the 4x larger up-sampled tensor: 16 instead of
// 36 multiply-adds per low-res pixel and channel pair, 2.25x fewer MFMAs for the layers that hold 75 % of the generator's
// FLOPs.  Zero padding is identical (a tap set never mixes in-range and out-of-range rows).  Only the summation order differs
// from the direct form (effective weights are sums of 1, 2 or 4 fp32 taps), inside the stated 1e-5 forward tolerance.
//
// Same skeleton as conv3x3.hip: implicit GEMM on v_mfma_f32_16x16x4_f32, a wave = 16 low-res pixels x 4 phases x all output
// channels (64 output pixels, PixelNorm reduction inside the wave), 8-channel chunks through LDS with issue-early/write-late
// register prefetch, weights pre-packed as the LDS image [chunk][16 = phase*4 + a*2 + b][8][16*ceil(Cout/16)].
#include <type_traits>
#include <utility>

#include "mg_common.h"
#include "pack_kernels.h"

namespace {

constexpr int CC = 8;
template <class F, int... Is>
__device__ __forceinline__ void up_static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void up_static_for(F&& f) {
  up_static_for_impl(f, std::make_integer_sequence<int, N>{});
}
constexpr float PN_EPS = 1e-8f;
constexpr int UP_NIN = 8;  // halo positions per thread in flight (plane <= 256)

struct UpArgs {
  const float* x;
  const float* wp;
  const float* bias;
  float* y;
  float* p;
  float* rn;
  int N, Cin, Cout, Hin, Win;
  int flags;
  float slope;
  int TH, TW, TN, lgTH, lgTW, THp, TWp;  // low-res tile geometry (64 low-res pixels per workgroup)
  int tiles_x, tiles_y, tiles_n;
  int plane, ch_stride, tab_floats;
  int OPF, nchunk;
};

template <int NI>
__global__ void __launch_bounds__(256) upconv3x3_mfma(const UpArgs a) {
  constexpr int OPL = (NI & 1) ? NI * 16 : NI * 16 + 16;
  constexpr int NW4 = (16 * CC * NI * 4 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int* tab = reinterpret_cast<int*>(smem);
  float* in_t = smem + a.tab_floats;
  float* w_t = in_t + CC * a.ch_stride;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, rq = lane >> 4;
  const int bid = mg_xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % a.tiles_x;
  const int t2 = bid / a.tiles_x;
  const int ty = t2 % a.tiles_y;
  const int tn = t2 / a.tiles_y;
  const int HWin = a.Hin * a.Win;
  const int H = 2 * a.Hin, W = 2 * a.Win;
  const float* xn = a.x + (size_t)tn * a.TN * a.Cin * HWin;

  {
    const int THpTWp = a.THp * a.TWp;
    for (int pos = tid; pos < a.plane; pos += 256) {
      const int n_l = pos / THpTWp;
      const int rem = pos - n_l * THpTWp;
      const int rr = rem / a.TWp;
      const int cc = rem - rr * a.TWp;
      const int n = tn * a.TN + n_l, Y = ty * a.TH + rr - 1, X = tx * a.TW + cc - 1;
      const bool ok = (n < a.N) && (Y >= 0) && (Y < a.Hin) && (X >= 0) && (X < a.Win);
      tab[pos] = ok ? n_l * a.Cin * HWin + Y * a.Win + X : -1;
    }
  }

  // this lane's low-res pixel as A-operand row (col) of the wave's m-tile
  int pix_off;
  {
    const int p = wave * 16 + col;
    const int c = p & (a.TW - 1);
    const int r = (p >> a.lgTW) & (a.TH - 1);
    const int n_l = p >> (a.lgTW + a.lgTH);
    pix_off = (n_l * a.THp + r) * a.TWp + c + rq * a.ch_stride;
  }

  f32x4 acc[4][NI];  // [phase = py*2 + px][channel tile]
#pragma unroll
  for (int ph = 0; ph < 4; ++ph)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[ph][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int cl_ = tid >> 5, l32_ = tid & 31;
  float rin[UP_NIN];
  f32x4 rw[NW4];

  auto load_chunk = [&](int ch) {
    const int c = ch * CC + cl_;
    const bool cok = c < a.Cin;
    const float* xc = xn + (size_t)c * HWin;
#pragma unroll
    for (int j = 0; j < UP_NIN; ++j) {
      const int pos = l32_ + 32 * j;
      float v = 0.f;
      if (pos < a.plane) {
        const int off = tab[pos];
        if (cok && off >= 0) v = xc[off];
      }
      rin[j] = v;
    }
    const float* src = a.wp + (size_t)ch * (16 * CC) * a.OPF;
#pragma unroll
    for (int j = 0; j < NW4; ++j) {
      const int e = tid + 256 * j;
      if (e < 16 * CC * NI * 4) {
        const int row = e / (NI * 4);
        const int jj = e - row * (NI * 4);
        rw[j] = *reinterpret_cast<const f32x4*>(src + (size_t)row * a.OPF + 4 * jj);
      }
    }
  };
  auto store_chunk = [&]() {
    float* dst = in_t + cl_ * a.ch_stride;
#pragma unroll
    for (int j = 0; j < UP_NIN; ++j) {
      const int pos = l32_ + 32 * j;
      if (pos < a.plane) dst[pos] = rin[j];
    }
#pragma unroll
    for (int j = 0; j < NW4; ++j) {
      const int e = tid + 256 * j;
      if (e < 16 * CC * NI * 4) {
        const int row = e / (NI * 4);
        const int jj = e - row * (NI * 4);
        *reinterpret_cast<f32x4*>(w_t + row * OPL + 4 * jj) = rw[j];
      }
    }
  };
  // One group = one low-res neighbour (dy, dx) and k-step: its pixel value and the weights of the 1 / 2 / 4 phases that read it.  The
  // operands of group g + 1 are requested BEFORE the MFMAs of group g are issued (two register sets, scheduling fences): left to the
  // scheduler every pair of MFMAs sat behind its own ds_read + s_waitcnt lgkmcnt(0) (an LDS round trip per two MFMAs).
  constexpr int NG = 9 * (CC / 4);
  float gav[2];
  float gwb[2][4][NI];
  auto load_group = [&](auto g_, auto s_) __attribute__((always_inline)) {
    constexpr int g = decltype(g_)::value, S = decltype(s_)::value;
    constexpr int tp = g / (CC / 4), ks = g % (CC / 4), dy = tp / 3, dx = tp % 3;
    gav[S] = in_t[pix_off + ks * 4 * a.ch_stride + dy * a.TWp + dx];
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
      for (int px = 0; px < 2; ++px) {
        const int ta = dy - py, tb = dx - px;  // tap of this phase that reads the neighbour, if any
        if (ta >= 0 && ta <= 1 && tb >= 0 && tb <= 1) {
          const int wrow = ((py * 2 + px) * 4 + ta * 2 + tb) * CC + ks * 4;
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) gwb[S][py * 2 + px][ni] = w_t[(wrow + rq) * OPL + ni * 16 + col];
        }
      }
  };
  auto mfma_group = [&](auto g_, auto s_) __attribute__((always_inline)) {
    constexpr int g = decltype(g_)::value, S = decltype(s_)::value;
    constexpr int tp = g / (CC / 4), dy = tp / 3, dx = tp % 3;
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
      for (int px = 0; px < 2; ++px) {
        const int ta = dy - py, tb = dx - px;
        if (ta >= 0 && ta <= 1 && tb >= 0 && tb <= 1) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[py * 2 + px][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(gav[S], gwb[S][py * 2 + px][ni], acc[py * 2 + px][ni], 0, 0, 0);
        }
      }
  };
  auto compute_chunk = [&]() __attribute__((always_inline)) {
    load_group(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    up_static_for<NG>([&](auto g_) __attribute__((always_inline)) {
      constexpr int g = decltype(g_)::value;
      if constexpr (g + 1 < NG) load_group(std::integral_constant<int, g + 1>{}, std::integral_constant<int, (g + 1) & 1>{});
      __builtin_amdgcn_sched_barrier(0);
      mfma_group(g_, std::integral_constant<int, g & 1>{});
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  __syncthreads();
  load_chunk(0);
  for (int ch = 0; ch < a.nchunk; ++ch) {
    __syncthreads();
    store_chunk();
    __syncthreads();
    if (ch + 1 < a.nchunk) load_chunk(ch + 1);
    compute_chunk();
  }

  // ---------------------------------------------------------------- epilogue
  const bool lrelu = (a.flags & MG_CONV_LRELU) != 0;
  const bool pixnorm = (a.flags & MG_CONV_PIXNORM) != 0;
  const bool vec = (a.TW >= 4) && ((a.Win & 3) == 0);
  float bvs[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int o = ni * 16 + col;
    bvs[ni] = (a.bias != nullptr && o < a.Cout) ? a.bias[o] : 0.f;
  }
  f32x4 rnv[4];
#pragma unroll
  for (int ph = 0; ph < 4; ++ph) {
    rnv[ph] = f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v = acc[ph][ni][g] + bvs[ni];
        if (lrelu) v = mg_lrelu(v, a.slope);
        acc[ph][ni][g] = v;
      }
    }
    if (pixnorm) {
      f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) s += acc[ph][ni] * acc[ph][ni];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float t = s[g];
        t += __shfl_xor(t, 1);
        t += __shfl_xor(t, 2);
        t += __shfl_xor(t, 4);
        t += __shfl_xor(t, 8);
        rnv[ph][g] = 1.0f / sqrtf(t / (float)a.Cout + PN_EPS);
      }
    }
  }
  const int pb = wave * 16 + rq * 4;  // this lane's 4 consecutive low-res pixels
  if (vec) {
    const int c = pb & (a.TW - 1);
    const int r = (pb >> a.lgTW) & (a.TH - 1);
    const int n_l = pb >> (a.lgTW + a.lgTH);
    const int n = tn * a.TN + n_l, Yl = ty * a.TH + r, Xl = tx * a.TW + c;
    if ((n < a.N) && (Yl < a.Hin) && (Xl < a.Win)) {
#pragma unroll
      for (int py = 0; py < 2; ++py) {
        const size_t rowoff = (size_t)(2 * Yl + py) * W + 2 * Xl;  // 8 consecutive outputs = 4 low-res pixels x px {0,1}
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int o = ni * 16 + col;
          if (o < a.Cout) {
            const size_t idx = ((size_t)n * a.Cout + o) * H * W + rowoff;
            const f32x4 e = acc[py * 2 + 0][ni], d = acc[py * 2 + 1][ni];
            const f32x4 lo = f32x4{e[0], d[0], e[1], d[1]}, hi = f32x4{e[2], d[2], e[3], d[3]};
            if (a.y != nullptr) {
              *reinterpret_cast<f32x4*>(a.y + idx) = lo;
              *reinterpret_cast<f32x4*>(a.y + idx + 4) = hi;
            }
            if (pixnorm) {
              const f32x4 re = rnv[py * 2 + 0], rd = rnv[py * 2 + 1];
              *reinterpret_cast<f32x4*>(a.p + idx) = lo * f32x4{re[0], rd[0], re[1], rd[1]};
              *reinterpret_cast<f32x4*>(a.p + idx + 4) = hi * f32x4{re[2], rd[2], re[3], rd[3]};
            }
          }
        }
        if (pixnorm && col == 0 && a.rn != nullptr) {
          const f32x4 re = rnv[py * 2 + 0], rd = rnv[py * 2 + 1];
          float* rp = a.rn + (size_t)n * H * W + rowoff;
          *reinterpret_cast<f32x4*>(rp) = f32x4{re[0], rd[0], re[1], rd[1]};
          *reinterpret_cast<f32x4*>(rp + 4) = f32x4{re[2], rd[2], re[3], rd[3]};
        }
      }
    }
  } else {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int pl = pb + g;
      const int c = pl & (a.TW - 1);
      const int r = (pl >> a.lgTW) & (a.TH - 1);
      const int n_l = pl >> (a.lgTW + a.lgTH);
      const int n = tn * a.TN + n_l, Yl = ty * a.TH + r, Xl = tx * a.TW + c;
      if ((n < a.N) && (Yl < a.Hin) && (Xl < a.Win)) {
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
          const size_t sp = (size_t)(2 * Yl + (ph >> 1)) * W + 2 * Xl + (ph & 1);
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            const int o = ni * 16 + col;
            if (o < a.Cout) {
              const size_t idx = ((size_t)n * a.Cout + o) * H * W + sp;
              const float v = acc[ph][ni][g];
              if (a.y != nullptr) a.y[idx] = v;
              if (pixnorm) a.p[idx] = v * rnv[ph][g];
            }
          }
          if (pixnorm && col == 0 && a.rn != nullptr) a.rn[(size_t)n * H * W + sp] = rnv[ph][g];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Data gradient of the same layer in sub-pixel form.  The transpose of "four 2x2 convs + pixel interleave" is ONE stride-2
// convolution with a 4x4 effective kernel over the high-resolution output gradient:
//   gx[c, y, x] = sum_o sum_{u,v in 0..3} K4[u][v][o][c] * gy[o, 2y + u - 1, 2x + v - 1],
//   K4[u][.] = Weff[py(u)][.][a(u)][.]   with row offsets u-1: -1 -> (py 1, a 1), 0 -> (0, 1), +1 -> (1, 0), +2 -> (0, 0)
// (it replaces conv3x3-dgrad at 2H x 2W followed by the 2x2 block sum of Upsample's backward: 16 instead of 36 multiply-adds
// per low-res pixel and channel pair, and the 4x larger intermediate gradient is never written).
// Implicit GEMM like conv3x3.hip: M = 16 low-res pixels per MFMA tile (MI = 2 tiles per wave), N = input channels of the
// forward conv, K = 4 gradient channels of one of the 16 taps.  The A operand walks the staged high-res halo tile with stride 2
// (16 lanes x 8 bytes = 32 distinct even banks); the channel stride is odd so the second k-lane lands on the odd banks.
struct DownArgs {
  const float* gy;  // (N, Cg, 2H, 2W)
  const float* wp;  // [Cg/8][16][8][OPF]
  float* gx;        // (N, Cx, H, W)
  int N, Cg, Cx, H, W;
  int TH, TW, TN, lgTH, lgTW, THp, TWp;
  int tiles_x, tiles_y, tiles_n;
  int plane, ch_stride, tab_floats;
  int OPF, nchunk;
};

constexpr int DN_MI = 2;
constexpr int DN_NIN = 24;  // halo positions per thread in flight (plane <= 768)

template <int NI>
__global__ void __launch_bounds__(256) downconv4x4s2_mfma(const DownArgs a) {
  constexpr int MI = DN_MI;
  constexpr int OPL = (NI & 1) ? NI * 16 : NI * 16 + 16;
  constexpr int NW4 = (16 * CC * NI * 4 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int* tab = reinterpret_cast<int*>(smem);
  float* in_t = smem + a.tab_floats;
  float* w_t = in_t + CC * a.ch_stride;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, rq = lane >> 4;
  const int bid = mg_xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % a.tiles_x;
  const int t2 = bid / a.tiles_x;
  const int ty = t2 % a.tiles_y;
  const int tn = t2 / a.tiles_y;
  const int Hh = 2 * a.H, Wh = 2 * a.W;
  const int HWh = Hh * Wh;
  const float* gn = a.gy + (size_t)tn * a.TN * a.Cg * HWh;

  {
    const int THpTWp = a.THp * a.TWp;
    for (int pos = tid; pos < a.plane; pos += 256) {
      const int n_l = pos / THpTWp;
      const int rem = pos - n_l * THpTWp;
      const int rr = rem / a.TWp;
      const int cc = rem - rr * a.TWp;
      const int n = tn * a.TN + n_l, Y = 2 * ty * a.TH + rr - 1, X = 2 * tx * a.TW + cc - 1;
      const bool ok = (n < a.N) && (Y >= 0) && (Y < Hh) && (X >= 0) && (X < Wh);
      tab[pos] = ok ? n_l * a.Cg * HWh + Y * Wh + X : -1;
    }
  }

  int pix_off[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int p = (wave * MI + mi) * 16 + col;
    const int c = p & (a.TW - 1);
    const int r = (p >> a.lgTW) & (a.TH - 1);
    const int n_l = p >> (a.lgTW + a.lgTH);
    pix_off[mi] = (n_l * a.THp + 2 * r) * a.TWp + 2 * c + rq * a.ch_stride;
  }

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int cl_ = tid >> 5, l32_ = tid & 31;
  float rin[DN_NIN];
  f32x4 rw[NW4];
  auto load_chunk = [&](int ch) {
    const int c = ch * CC + cl_;
    const bool cok = c < a.Cg;
    const float* gc = gn + (size_t)c * HWh;
#pragma unroll
    for (int j = 0; j < DN_NIN; ++j) {
      const int pos = l32_ + 32 * j;
      float v = 0.f;
      if (pos < a.plane) {
        const int off = tab[pos];
        if (cok && off >= 0) v = gc[off];
      }
      rin[j] = v;
    }
    const float* src = a.wp + (size_t)ch * (16 * CC) * a.OPF;
#pragma unroll
    for (int j = 0; j < NW4; ++j) {
      const int e = tid + 256 * j;
      if (e < 16 * CC * NI * 4) {
        const int row = e / (NI * 4);
        const int jj = e - row * (NI * 4);
        rw[j] = *reinterpret_cast<const f32x4*>(src + (size_t)row * a.OPF + 4 * jj);
      }
    }
  };
  auto store_chunk = [&]() {
    float* dst = in_t + cl_ * a.ch_stride;
#pragma unroll
    for (int j = 0; j < DN_NIN; ++j) {
      const int pos = l32_ + 32 * j;
      if (pos < a.plane) dst[pos] = rin[j];
    }
#pragma unroll
    for (int j = 0; j < NW4; ++j) {
      const int e = tid + 256 * j;
      if (e < 16 * CC * NI * 4) {
        const int row = e / (NI * 4);
        const int jj = e - row * (NI * 4);
        *reinterpret_cast<f32x4*>(w_t + row * OPL + 4 * jj) = rw[j];
      }
    }
  };
  // (measured: requesting the operands of the next tap / k-step group ahead of the MFMAs, as upconv3x3_mfma does, is 3 % SLOWER here --
  // ten MFMAs per group already cover the reads: 107 -> 111 us at 80<-64 @32 x64)
  auto compute_chunk = [&]() {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int tap = (t >> 2) * a.TWp + (t & 3);
#pragma unroll
      for (int ks = 0; ks < CC / 4; ++ks) {
        float av[MI], bv[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) av[mi] = in_t[pix_off[mi] + ks * 4 * a.ch_stride + tap];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bv[ni] = w_t[(t * CC + ks * 4 + rq) * OPL + ni * 16 + col];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mi], bv[ni], acc[mi][ni], 0, 0, 0);
      }
    }
  };

  __syncthreads();
  load_chunk(0);
  for (int ch = 0; ch < a.nchunk; ++ch) {
    __syncthreads();
    store_chunk();
    __syncthreads();
    if (ch + 1 < a.nchunk) load_chunk(ch + 1);
    compute_chunk();
  }

  const bool vec = (a.TW >= 4) && ((a.W & 3) == 0);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int pb = (wave * MI + mi) * 16 + rq * 4;
    if (vec) {
      const int c = pb & (a.TW - 1);
      const int r = (pb >> a.lgTW) & (a.TH - 1);
      const int n_l = pb >> (a.lgTW + a.lgTH);
      const int n = tn * a.TN + n_l, Y = ty * a.TH + r, X = tx * a.TW + c;
      if ((n < a.N) && (Y < a.H) && (X < a.W)) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int o = ni * 16 + col;
          if (o < a.Cx) *reinterpret_cast<f32x4*>(a.gx + (((size_t)n * a.Cx + o) * a.H + Y) * a.W + X) = acc[mi][ni];
        }
      }
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int pl = pb + g;
        const int c = pl & (a.TW - 1);
        const int r = (pl >> a.lgTW) & (a.TH - 1);
        const int n_l = pl >> (a.lgTW + a.lgTH);
        const int n = tn * a.TN + n_l, Y = ty * a.TH + r, X = tx * a.TW + c;
        if ((n < a.N) && (Y < a.H) && (X < a.W)) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            const int o = ni * 16 + col;
            if (o < a.Cx) a.gx[(((size_t)n * a.Cx + o) * a.H + Y) * a.W + X] = acc[mi][ni][g];
          }
        }
      }
    }
  }
}

// K4 weights of the data-gradient form, LDS image layout [Cg/8][16][8][OPF]; w is the module weight [Co][Ci][3][3], Cg = Co, Cx = Ci.
__global__ void downconv_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int Co, int Ci, size_t total) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < total) pack_downconv_elem(e, w, wp, Co, Ci);
}

template <int NI>
int launch_down(const DownArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  static MgPerDevice once;  // the LDS limit is a per-device function attribute
  if (mg_first_use_on_device(once)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&downconv4x4s2_mfma<NI>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  hipLaunchKernelGGL((downconv4x4s2_mfma<NI>), grid, dim3(256), lds, s, a);
  MG_CHECK_LAUNCH("mg_upconv3x3_dgrad");
  return MG_OK;
}

// Effective sub-pixel weights in the LDS image layout.  w is the module weight [Co][Ci][3][3].
__global__ void upconv3x3_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int Co, int Ci, size_t total) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < total) pack_upconv3x3_elem(e, w, wp, Co, Ci);
}

template <int NI>
int launch_up(const UpArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  static MgPerDevice once;  // the LDS limit is a per-device function attribute
  if (mg_first_use_on_device(once)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&upconv3x3_mfma<NI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
  }
  hipLaunchKernelGGL((upconv3x3_mfma<NI>), grid, dim3(256), lds, s, a);
  MG_CHECK_LAUNCH("mg_upconv3x3");
  return MG_OK;
}

}  // namespace

extern "C" size_t mg_upconv3x3_packed_floats(int Cin, int Cout) {
  return (size_t)mg_cdiv(Cin, CC) * 16 * CC * (size_t)(16 * mg_cdiv(Cout, 16));
}

extern "C" int mg_upconv3x3_pack(const float* w, float* wp, int Co, int Ci, mg_stream_t stream) {
  MG_CHECK_ARG(w && wp && Co > 0 && Ci > 0, "mg_upconv3x3_pack: bad arguments");
  const size_t total = mg_upconv3x3_packed_floats(Ci, Co);
  hipLaunchKernelGGL(upconv3x3_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, wp,
                     Co, Ci, total);
  MG_CHECK_LAUNCH("mg_upconv3x3_pack");
  return MG_OK;
}

extern "C" int mg_upconv3x3(const float* x, const float* wp, const float* bias, float* y, float* p, float* rn, int N,
                            int Cin, int Cout, int Hin, int Win, int flags, float slope, mg_stream_t stream) {
  MG_CHECK_ARG(x && wp && N > 0 && Cin > 0 && Cout > 0 && Hin > 0 && Win > 0, "mg_upconv3x3: bad arguments");
  MG_CHECK_ARG(Cout <= 160, "mg_upconv3x3: Cout=%d > 160 unsupported", Cout);
  const bool pn = flags & MG_CONV_PIXNORM;
  MG_CHECK_ARG(!pn || ((flags & MG_CONV_LRELU) && p), "mg_upconv3x3: PIXNORM needs LRELU and p");
  MG_CHECK_ARG(pn || y, "mg_upconv3x3: y is NULL");
  MG_CHECK_ARG(!(flags & ~(MG_CONV_LRELU | MG_CONV_PIXNORM)), "mg_upconv3x3: unsupported flag");
  MG_CHECK_ARG((long long)N * Cin * Hin * Win < (1ll << 31), "mg_upconv3x3: tensor too large");
  UpArgs a;
  a.x = x; a.wp = wp; a.bias = bias; a.y = y; a.p = p; a.rn = rn;
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.Hin = Hin; a.Win = Win;
  a.flags = flags; a.slope = slope;
  const int NI = mg_cdiv(Cout, 16);
  a.OPF = NI * 16;
  a.nchunk = mg_cdiv(Cin, CC);
  const int P = 64;  // low-res pixels per workgroup (256 output pixels)
  a.TW = mg_pow2_ceil(Win) < 16 ? mg_pow2_ceil(Win) : 16;
  a.TH = mg_pow2_ceil(Hin) < P / a.TW ? mg_pow2_ceil(Hin) : P / a.TW;
  a.TN = P / (a.TW * a.TH);
  a.lgTW = mg_ilog2(a.TW); a.lgTH = mg_ilog2(a.TH);
  a.THp = a.TH + 2; a.TWp = a.TW + 2;
  a.tiles_x = mg_cdiv(Win, a.TW); a.tiles_y = mg_cdiv(Hin, a.TH); a.tiles_n = mg_cdiv(N, a.TN);
  a.plane = a.TN * a.THp * a.TWp;
  MG_CHECK_ARG(a.plane <= 32 * UP_NIN, "mg_upconv3x3: halo tile too large");
  a.ch_stride = ((a.plane + 15) / 32) * 32 + 16;
  if (a.ch_stride < a.plane) a.ch_stride += 32;
  a.tab_floats = (a.plane + 3) & ~3;
  const int OPL = (NI & 1) ? NI * 16 : NI * 16 + 16;
  const size_t lds = (size_t)(a.tab_floats + CC * a.ch_stride + 16 * CC * OPL) * sizeof(float);
  MG_CHECK_ARG(lds <= 160 * 1024, "mg_upconv3x3: LDS tile too large");
  dim3 grid(a.tiles_x * a.tiles_y * a.tiles_n);
  hipStream_t s = (hipStream_t)stream;
  switch (NI) {
    case 1: return launch_up<1>(a, grid, lds, s);
    case 2: return launch_up<2>(a, grid, lds, s);
    case 3: return launch_up<3>(a, grid, lds, s);
    case 4: return launch_up<4>(a, grid, lds, s);
    case 5: return launch_up<5>(a, grid, lds, s);
    case 6: return launch_up<6>(a, grid, lds, s);
    case 7: return launch_up<7>(a, grid, lds, s);
    case 8: return launch_up<8>(a, grid, lds, s);
    case 9: return launch_up<9>(a, grid, lds, s);
    default: return launch_up<10>(a, grid, lds, s);
  }
}

extern "C" size_t mg_upconv3x3_dgrad_packed_floats(int Cin, int Cout) {
  return (size_t)mg_cdiv(Cout, CC) * 16 * CC * (size_t)(16 * mg_cdiv(Cin, 16));
}

extern "C" int mg_upconv3x3_dgrad_pack(const float* w, float* wp, int Co, int Ci, mg_stream_t stream) {
  MG_CHECK_ARG(w && wp && Co > 0 && Ci > 0, "mg_upconv3x3_dgrad_pack: bad arguments");
  const size_t total = mg_upconv3x3_dgrad_packed_floats(Ci, Co);
  hipLaunchKernelGGL(downconv_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, wp,
                     Co, Ci, total);
  MG_CHECK_LAUNCH("mg_upconv3x3_dgrad_pack");
  return MG_OK;
}

extern "C" int mg_upconv3x3_dgrad(const float* gy, const float* wp, float* gx, int N, int Cin, int Cout, int Hin, int Win,
                                  mg_stream_t stream) {
  MG_CHECK_ARG(gy && wp && gx && N > 0 && Cin > 0 && Cout > 0 && Hin > 0 && Win > 0, "mg_upconv3x3_dgrad: bad arguments");
  MG_CHECK_ARG(Cin <= 160, "mg_upconv3x3_dgrad: Cin=%d > 160 unsupported", Cin);
  MG_CHECK_ARG((long long)N * Cout * 4 * Hin * Win < (1ll << 31), "mg_upconv3x3_dgrad: tensor too large");
  DownArgs a;
  a.gy = gy; a.wp = wp; a.gx = gx;
  a.N = N; a.Cg = Cout; a.Cx = Cin; a.H = Hin; a.W = Win;
  const int NI = mg_cdiv(Cin, 16);
  a.OPF = NI * 16;
  a.nchunk = mg_cdiv(Cout, CC);
  const int P = 64 * DN_MI;
  a.TW = mg_pow2_ceil(Win) < 32 ? mg_pow2_ceil(Win) : 32;
  a.TH = mg_pow2_ceil(Hin) < P / a.TW ? mg_pow2_ceil(Hin) : P / a.TW;
  a.TN = P / (a.TW * a.TH);
  a.lgTW = mg_ilog2(a.TW); a.lgTH = mg_ilog2(a.TH);
  a.THp = 2 * a.TH + 2; a.TWp = 2 * a.TW + 2;
  a.tiles_x = mg_cdiv(Win, a.TW); a.tiles_y = mg_cdiv(Hin, a.TH); a.tiles_n = mg_cdiv(N, a.TN);
  a.plane = a.TN * a.THp * a.TWp;
  MG_CHECK_ARG(a.plane <= 32 * DN_NIN, "mg_upconv3x3_dgrad: halo tile too large (plane %d)", a.plane);
  a.ch_stride = a.plane | 1;  // odd: the stride-2 operand reads of the two k-lanes of a half-wave use disjoint bank parities
  a.tab_floats = (a.plane + 3) & ~3;
  const int OPL = (NI & 1) ? NI * 16 : NI * 16 + 16;
  const size_t lds = (size_t)(a.tab_floats + ((CC * a.ch_stride + 3) & ~3) + 16 * CC * OPL) * sizeof(float);
  MG_CHECK_ARG(lds <= 160 * 1024, "mg_upconv3x3_dgrad: LDS tile too large");
  dim3 grid(a.tiles_x * a.tiles_y * a.tiles_n);
  hipStream_t s = (hipStream_t)stream;
  switch (NI) {
    case 1: return launch_down<1>(a, grid, lds, s);
    case 2: return launch_down<2>(a, grid, lds, s);
    case 3: return launch_down<3>(a, grid, lds, s);
    case 4: return launch_down<4>(a, grid, lds, s);
    case 5: return launch_down<5>(a, grid, lds, s);
    case 6: return launch_down<6>(a, grid, lds, s);
    case 7: return launch_down<7>(a, grid, lds, s);
    case 8: return launch_down<8>(a, grid, lds, s);
    case 9: return launch_down<9>(a, grid, lds, s);
    default: return launch_down<10>(a, grid, lds, s);
  }
}
