// 3x3 / stride 1 / pad 1 convolution as an implicit GEMM on the exact-fp32 matrix cores of gfx950
// (v_mfma_f32_16x16x4_f32), forward and data-gradient (the latter through re-packed weights).
//
// Replaces nn.Conv2d(3x3) + LeakyReLU (+ PixelNorm) (+ preceding nearest Upsample) of
//   /root/reference/music_gan/networks/generator.py:9-40 and discriminator.py:8-34.
//
// GEMM view:  D[pixel, out-channel] = sum_{tap, c} X[pixel + tap, c] * W[tap, c, out-channel]
//   M (A rows)  = 16 output pixels per MFMA tile (lane&15), MI tiles per wave, 4 waves per workgroup
//   N (B cols)  = 16 output channels per MFMA tile, NI tiles per wave (a wave sees ALL channels of its pixels, so the
//                 PixelNorm channel reduction stays inside the wave)
//   K           = 4 input channels of one tap per MFMA (lane>>4), 8-channel chunks staged through LDS
// LDS image per chunk: input halo tile [8][ch_stride] (ch_stride % 32 == 16 => the two k-lanes of a half-wave hit disjoint
// banks) and weights [9 taps][8 ch][OPL] copied verbatim from the pre-packed global layout (OPL % 32 == 16, same reason).
// D layout (16x16x4): lane holds out-channel (lane&15) of pixels 4*(lane>>4)+{0..3} => one 16-byte store per tile.
#include <cstdlib>
#include <type_traits>

#include "mg_common.h"
#include "pack_kernels.h"

namespace {

constexpr int CC = 8;  // input channels per LDS chunk
constexpr float PN_EPS = 1e-8f;

struct ConvArgs {
  const float* x;
  const float* wp;
  const float* bias;
  const float* aux;
  float* y;
  float* p;
  float* rn;
  int N, Cin, Cout, H, W, Hin, Win;
  int flags;
  float slope;
  int TH, TW, TN, lgTH, lgTW, THp, TWp;
  int tiles_x, tiles_y, tiles_n;
  int plane, ch_stride, tab_floats;
  int OPF;  // row stride of the packed weights (floats) = 16*ceil(Cout/16)
  int nchunk;
  int sub;  // 8-channel chunks per barrier interval (1 or 4)
  int ksplit;  // 4: the workgroup's waves split those chunks over one shared 16-pixel tile (small maps)
};

// PF: software-pipelined variant -- the next chunk's global loads are issued into registers right after the barrier that
// starts the current chunk's MFMA phase and written to LDS after it (T14 "issue early / write late"), so staging latency hides
// under the matrix work of the same workgroup instead of relying on other workgroups being in a different phase.
constexpr int PF_NIN = 12;  // halo positions per thread held in flight (covers plane <= 384: every tile of a >= 16-wide image)

// SUB: 8-channel chunks staged and consumed per barrier interval.  The small maps at the ends of both networks (<= 8x8, 112-160
// channels) are latency-bound -- 14-20 chunks of 18 MFMAs, each behind two barriers and a global-load round trip -- so their
// variants take 4 chunks at a time.
// KS = 4 (split-K, with SUB = 4): the four waves share ONE 16*MI-pixel tile and each consumes one of the four chunks of a barrier
// interval; their partial accumulators meet in LDS and wave 0 runs the epilogue.  On <= 8x8 maps a launch is a chain of Cin/8
// chunk phases that no width shortens (4 us + 0.113 us per input channel whatever Cout and batch, tools/bench_smallconv.py: every
// MFMA of the 16x16 tile waits for its own two LDS operands); split four ways the chain is a quarter as long.
template <int NI, int MI, bool PF, int SUB = 1, int KS = 1>
__global__ void __launch_bounds__(256) conv3x3_mfma(const ConvArgs a) {
  static_assert(KS == 1 || (KS == 4 && SUB == 4 && MI == 1), "split-K variant: 4 waves x 4 chunks per interval");
  constexpr int OPL = (NI & 1) ? NI * 16 : NI * 16 + 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int* tab = reinterpret_cast<int*>(smem);
  float* in_t = smem + a.tab_floats;
  float* w_t = in_t + SUB * CC * a.ch_stride;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, rq = lane >> 4;
  const int bid = mg_xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % a.tiles_x;
  const int t2 = bid / a.tiles_x;
  const int ty = t2 % a.tiles_y;
  const int tn = t2 / a.tiles_y;
  const int o0 = blockIdx.y * NI * 16;
  const int HWin = a.Hin * a.Win;
  const float* xn = a.x + (size_t)tn * a.TN * a.Cin * HWin;
  const bool ups = (a.flags & MG_CONV_UPS_IN) != 0;

  // source-offset table of the halo tile (chunk-invariant): -1 => zero padding / out of range
  {
    const int THpTWp = a.THp * a.TWp;
    for (int pos = tid; pos < a.plane; pos += 256) {
      const int n_l = pos / THpTWp;
      const int rem = pos - n_l * THpTWp;
      const int rr = rem / a.TWp;
      const int cc = rem - rr * a.TWp;
      const int n = tn * a.TN + n_l, Y = ty * a.TH + rr - 1, X = tx * a.TW + cc - 1;
      const bool ok = (n < a.N) && (Y >= 0) && (Y < a.H) && (X >= 0) && (X < a.W);
      const int sp = ups ? (Y >> 1) * a.Win + (X >> 1) : Y * a.Win + X;
      tab[pos] = ok ? n_l * a.Cin * HWin + sp : -1;
    }
  }

  int pix_off[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int p = ((KS > 1 ? 0 : wave) * MI + mi) * 16 + col;
    const int c = p & (a.TW - 1);
    const int r = (p >> a.lgTW) & (a.TH - 1);
    const int n_l = p >> (a.lgTW + a.lgTH);
    pix_off[mi] = (n_l * a.THp + r) * a.TWp + c + rq * a.ch_stride;
  }

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int NW4 = (9 * CC * NI * 4 + 255) / 256;
  const int cl_ = tid >> 5, l32_ = tid & 31;
  constexpr int RIN = SUB > 1 ? 8 : PF_NIN;  // SUB variants serve small maps only: halo tile <= 256 positions
  float rin[PF ? RIN * SUB : 1];
  f32x4 rw[PF ? NW4 * SUB : 1];

  auto compute_chunk = [&](int ch) {
#pragma unroll
    for (int sub = 0; sub < SUB; ++sub) {
      if (SUB > 1 && ch * SUB + sub >= a.nchunk) break;
      if (KS > 1 && sub != wave) continue;
      const float* in_s = in_t + sub * CC * a.ch_stride;
      const float* w_s = w_t + sub * 9 * CC * OPL;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int tap = (t / 3) * a.TWp + (t % 3);
#pragma unroll
        for (int ks = 0; ks < CC / 4; ++ks) {
          float av[MI], bv[NI];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) av[mi] = in_s[pix_off[mi] + ks * 4 * a.ch_stride + tap];
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) bv[ni] = w_s[(t * CC + ks * 4 + rq) * OPL + ni * 16 + col];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mi], bv[ni], acc[mi][ni], 0, 0, 0);
        }
      }
    }
  };

  if constexpr (PF) {
    const int nouter = (a.nchunk + SUB - 1) / SUB;
    int toff[RIN];  // this thread's entries of the (chunk-invariant) source-offset table, in registers: read from LDS in front of
                    // every load they put an LDS round trip between consecutive loads
    auto load_chunk = [&](int ch) {
#pragma unroll
      for (int sub = 0; sub < SUB; ++sub) {
        const int chs = ch * SUB + sub;
        if (SUB > 1 && chs >= a.nchunk) break;
        const int c = chs * CC + cl_;
        const bool cok = c < a.Cin;
        const float* xc = xn + (size_t)c * HWin;
#pragma unroll
        for (int j = 0; j < RIN; ++j) {
          float v = 0.f;
          if (cok && toff[j] >= 0) v = xc[toff[j]];
          rin[sub * RIN + j] = v;
        }
        const float* src = a.wp + (size_t)chs * (9 * CC) * a.OPF + o0;
#pragma unroll
        for (int j = 0; j < NW4; ++j) {
          const int e = tid + 256 * j;
          if (e < 9 * CC * NI * 4) {
            const int row = e / (NI * 4);
            const int jj = e - row * (NI * 4);
            rw[sub * NW4 + j] = *reinterpret_cast<const f32x4*>(src + (size_t)row * a.OPF + 4 * jj);
          }
        }
      }
    };
    auto store_chunk = [&](int ch) {
#pragma unroll
      for (int sub = 0; sub < SUB; ++sub) {
        if (SUB > 1 && ch * SUB + sub >= a.nchunk) break;
        float* dst = in_t + (sub * CC + cl_) * a.ch_stride;
#pragma unroll
        for (int j = 0; j < RIN; ++j) {
          const int pos = l32_ + 32 * j;
          if (pos < a.plane) dst[pos] = rin[sub * RIN + j];
        }
        float* wd = w_t + sub * 9 * CC * OPL;
#pragma unroll
        for (int j = 0; j < NW4; ++j) {
          const int e = tid + 256 * j;
          if (e < 9 * CC * NI * 4) {
            const int row = e / (NI * 4);
            const int jj = e - row * (NI * 4);
            *reinterpret_cast<f32x4*>(wd + row * OPL + 4 * jj) = rw[sub * NW4 + j];
          }
        }
      }
    };
    __syncthreads();  // tab visible
#pragma unroll
    for (int j = 0; j < RIN; ++j) toff[j] = l32_ + 32 * j < a.plane ? tab[l32_ + 32 * j] : -1;
    load_chunk(0);
    for (int ch = 0; ch < nouter; ++ch) {
      __syncthreads();  // previous chunk's LDS reads done
      store_chunk(ch);
      __syncthreads();
      if (ch + 1 < nouter) load_chunk(ch + 1);  // in flight during the MFMA phase below
      compute_chunk(ch);
    }
  } else {
    for (int ch = 0; ch < a.nchunk; ++ch) {
      __syncthreads();
      {  // input halo tile: 8 half-waves, one channel each, 32 consecutive positions per pass
        const int c = ch * CC + cl_;
        const bool cok = c < a.Cin;
        const float* xc = xn + (size_t)c * HWin;
        float* dst = in_t + cl_ * a.ch_stride;
#pragma unroll 4
        for (int pos = l32_; pos < a.plane; pos += 32) {
          const int off = tab[pos];
          float v = 0.f;
          if (cok && off >= 0) v = xc[off];
          dst[pos] = v;
        }
      }
      {  // weights: verbatim 16-byte copy of this chunk's packed rows
        const float* src = a.wp + (size_t)ch * (9 * CC) * a.OPF + o0;
        for (int e = tid; e < 9 * CC * NI * 4; e += 256) {
          const int row = e / (NI * 4);
          const int j = e - row * (NI * 4);
          const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)row * a.OPF + 4 * j);
          *reinterpret_cast<f32x4*>(w_t + row * OPL + 4 * j) = v;
        }
      }
      __syncthreads();
      compute_chunk(ch);
    }
  }

  if constexpr (KS > 1) {  // partial sums of waves 1..3 -> LDS (over the staging area), summed by wave 0 in a fixed order
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(in_t);
    if (wave > 0) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) red[(((wave - 1) * MI + mi) * NI + ni) * 64 + lane] = acc[mi][ni];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        acc[mi][ni] = ((acc[mi][ni] + red[((0 * MI + mi) * NI + ni) * 64 + lane]) + red[((1 * MI + mi) * NI + ni) * 64 + lane]) +
                      red[((2 * MI + mi) * NI + ni) * 64 + lane];
  }

  // ---------------------------------------------------------------- epilogue
  const bool lrelu = (a.flags & MG_CONV_LRELU) != 0;
  const bool mask_aux = (a.flags & MG_CONV_MASK_AUX) != 0;
  const bool pixnorm = (a.flags & MG_CONV_PIXNORM) != 0;
  const bool vec = (a.TW >= 4) && ((a.W & 3) == 0);
  float bv[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int o = o0 + ni * 16 + col;
    bv[ni] = (a.bias != nullptr && o < a.Cout) ? a.bias[o] : 0.f;
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int pb = (wave * MI + mi) * 16 + rq * 4;
    f32x4 rnv = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v = acc[mi][ni][g] + bv[ni];
        if (lrelu) v = mg_lrelu(v, a.slope);
        acc[mi][ni][g] = v;
      }
    }
    if (pixnorm) {
      f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) s += acc[mi][ni] * acc[mi][ni];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float t = s[g];
        t += __shfl_xor(t, 1);
        t += __shfl_xor(t, 2);
        t += __shfl_xor(t, 4);
        t += __shfl_xor(t, 8);
        rnv[g] = 1.0f / sqrtf(t / (float)a.Cout + PN_EPS);
      }
    }
    if (vec) {
      const int c = pb & (a.TW - 1);
      const int r = (pb >> a.lgTW) & (a.TH - 1);
      const int n_l = pb >> (a.lgTW + a.lgTH);
      const int n = tn * a.TN + n_l, Y = ty * a.TH + r, X = tx * a.TW + c;
      const bool valid = (n < a.N) && (Y < a.H) && (X < a.W);
      if (valid) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int o = o0 + ni * 16 + col;
          if (o < a.Cout) {
            const size_t idx = (((size_t)n * a.Cout + o) * a.H + Y) * a.W + X;
            f32x4 v = acc[mi][ni];
            if (mask_aux) {
              const f32x4 ax = *reinterpret_cast<const f32x4*>(a.aux + idx);
#pragma unroll
              for (int g = 0; g < 4; ++g) v[g] *= mg_lrelu_mask(ax[g], a.slope);
              acc[mi][ni] = v;  // the pooled output below averages the masked values
            }
            if (a.y != nullptr) *reinterpret_cast<f32x4*>(a.y + idx) = v;
            if (pixnorm) *reinterpret_cast<f32x4*>(a.p + idx) = v * rnv;
          }
        }
        if (pixnorm && col == 0 && a.rn != nullptr)
          *reinterpret_cast<f32x4*>(a.rn + ((size_t)n * a.H + Y) * a.W + X) = rnv;
      }
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int p = pb + g;
        const int c = p & (a.TW - 1);
        const int r = (p >> a.lgTW) & (a.TH - 1);
        const int n_l = p >> (a.lgTW + a.lgTH);
        const int n = tn * a.TN + n_l, Y = ty * a.TH + r, X = tx * a.TW + c;
        const bool valid = (n < a.N) && (Y < a.H) && (X < a.W);
        if (valid) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            const int o = o0 + ni * 16 + col;
            if (o < a.Cout) {
              const size_t idx = (((size_t)n * a.Cout + o) * a.H + Y) * a.W + X;
              float v = acc[mi][ni][g];
              if (mask_aux) v *= mg_lrelu_mask(a.aux[idx], a.slope);
              if (a.y != nullptr) a.y[idx] = v;
              if (pixnorm) a.p[idx] = v * rnv[g];
            }
          }
          if (pixnorm && col == 0 && a.rn != nullptr) a.rn[((size_t)n * a.H + Y) * a.W + X] = rnv[g];
        }
      }
    }
  }

  // Fused AvgPool2d(2,2) of the output just written (discriminator.py:24): a wave holds complete row pairs -- m-tile mi and
  // mi + TPR cover the same 16 columns of two consecutive rows, and a lane's 4 pixels are 2 horizontal pairs -- so the
  // pooled tensor costs no extra read of the full-resolution activation.  Host guarantees: vector path, TW >= 16, an even
  // number of tile rows per wave, even H and W.
  if (a.flags & MG_CONV_POOL_OUT) {
    auto pool = [&](auto tpr_) {
      constexpr int TPR = decltype(tpr_)::value;  // m-tiles per tile row
      const int Hp = a.H >> 1, Wp = a.W >> 1;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        if (((mi / TPR) & 1) == 0 && mi + TPR < MI) {
          const int pb = (wave * MI + mi) * 16 + rq * 4;
          const int c = pb & (a.TW - 1);
          const int r = (pb >> a.lgTW) & (a.TH - 1);
          const int n_l = pb >> (a.lgTW + a.lgTH);
          const int n = tn * a.TN + n_l, Y = ty * a.TH + r, X = tx * a.TW + c;
          if ((n < a.N) && (Y < a.H) && (X < a.W)) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              const int o = o0 + ni * 16 + col;
              if (o < a.Cout) {
                const f32x4 u = acc[mi][ni], d = acc[(mi + TPR) % MI][ni];
                const float2 pv = make_float2(((u[0] + u[1]) + (d[0] + d[1])) * 0.25f, ((u[2] + u[3]) + (d[2] + d[3])) * 0.25f);
                *reinterpret_cast<float2*>(a.p + (((size_t)n * a.Cout + o) * Hp + (Y >> 1)) * Wp + (X >> 1)) = pv;
              }
            }
          }
        }
      }
    };
    if (a.TW == 32) pool(std::integral_constant<int, 2>{});
    else pool(std::integral_constant<int, 1>{});
  }
}

// Direct (non-MFMA) variant for 1x1 images (the last critic conv; 2x2 measured slower than the MFMA path).  A 256-pixel MFMA
// tile would hold 64..256 samples with a 4x..9x halo and 5 of 9 taps entirely in the zero padding; here one thread owns one
// output value, lanes run along the output channel (coalesced packed weights), and only in-range taps are visited.
__global__ void __launch_bounds__(256) conv3x3_tiny(const ConvArgs a) {
  // workgroup = 64 consecutive output values (lane = out-channel: coalesced packed weights); its 4 waves split the input channels
  // and combine their partial sums through LDS in a fixed order -- a single thread walking all 160 channels is a chain of memory
  // round trips (19-25 us per launch with 4 loads in flight, 10 with 16; this form keeps 8-10 batches of them in parallel)
  __shared__ double part[3][64];
  const int HW = a.H * a.W;
  const size_t total = (size_t)a.N * HW * a.OPF;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t e = (size_t)blockIdx.x * 64 + lane;
  const bool live = e < total;
  const int o = live ? (int)(e % a.OPF) : 0;
  const size_t r = live ? e / a.OPF : 0;
  const int pix = (int)(r % HW);
  const int n = (int)(r / HW);
  const bool act = live && o < a.Cout;
  const int y = pix / a.W, x = pix - y * a.W;
  const bool ups = (a.flags & MG_CONV_UPS_IN) != 0;
  const int HWin = a.Hin * a.Win;
  const float* xn = a.x + (size_t)n * a.Cin * HWin;
  // fp64 accumulation (each fp32 product is exact in fp64; the work is a few hundred fmas per thread on a full-rate fp64 VALU): this
  // layer's output is the classifier's input, and the classifier weight gradient -- a difference of mean features -- inherits its
  // round-off one to one (tools/diag_act_noise.py: 4.2x the CPU library's rms error with a sequential fp32 sum, 1x with this)
  double acc64 = 0.0;
  if (act) {
    for (int t = 0; t < 9; ++t) {
      const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
      if (yy < 0 || yy >= a.H || xx < 0 || xx >= a.W) continue;
      const int sp = ups ? (yy >> 1) * a.Win + (xx >> 1) : yy * a.Win + xx;
      const float* wt = a.wp + (size_t)t * CC * a.OPF + o;
      // loads first, arithmetic second, 8 channels at a time (as a plain loop hipcc waits for each pair of loads before it issues
      // the next one); channels past Cin re-read the wave's first channel and are multiplied by 0
      for (int c0 = wave; c0 < a.Cin; c0 += 32) {
        float xv[8], wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int c = c0 + 4 * u < a.Cin ? c0 + 4 * u : wave;
          xv[u] = xn[(size_t)c * HWin + sp];
          wv[u] = wt[((size_t)(c >> 3) * 9 * CC + (c & 7)) * a.OPF];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc64 += (double)(c0 + 4 * u < a.Cin ? xv[u] : 0.f) * (double)wv[u];
      }
    }
  }
  if (wave > 0) part[wave - 1][lane] = acc64;
  __syncthreads();
  if (wave != 0 || !act) return;
  acc64 = ((acc64 + part[0][lane]) + part[1][lane]) + part[2][lane];
  if (a.bias != nullptr) acc64 += (double)a.bias[o];
  float acc = (float)acc64;
  const size_t idx = ((size_t)n * a.Cout + o) * HW + pix;
  if (a.flags & MG_CONV_LRELU) acc = mg_lrelu(acc, a.slope);
  if (a.flags & MG_CONV_MASK_AUX) acc *= mg_lrelu_mask(a.aux[idx], a.slope);
  a.y[idx] = acc;
}

__global__ void conv3x3_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int Co, int Ci, int dgrad, size_t total) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < total) pack_conv3x3_elem(e, w, wp, Co, Ci, dgrad);
}

template <int NI, int MI, bool PF, int SUB = 1, int KS = 1>
int launch_conv_pf(const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  static MgPerDevice once;  // the LDS limit is a per-device function attribute
  if (mg_first_use_on_device(once)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_mfma<NI, MI, PF, SUB, KS>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  hipLaunchKernelGGL((conv3x3_mfma<NI, MI, PF, SUB, KS>), grid, dim3(256), lds, s, a);
  MG_CHECK_LAUNCH("mg_conv3x3");
  return MG_OK;
}

bool pf_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("MG_CONV_PF");
    v = (e == nullptr) ? 1 : (atoi(e) != 0);
  }
  return v != 0;
}

template <int NI, int MI>
int launch_conv(const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  // small maps with many channels: 4 chunks per barrier interval (the host sized `lds` for it)
  if constexpr (MI == 1 && NI <= 2) {
    if (a.sub == 4 && a.ksplit == 4) return launch_conv_pf<NI, MI, true, 4, 4>(a, grid, lds, s);
    if (a.sub == 4) return launch_conv_pf<NI, MI, true, 4>(a, grid, lds, s);
  }
  // the pipelined variant needs the halo tile to fit its in-flight registers and more than one chunk to overlap
  if (pf_enabled() && a.plane <= 32 * PF_NIN && a.nchunk > 1) return launch_conv_pf<NI, MI, true>(a, grid, lds, s);
  return launch_conv_pf<NI, MI, false>(a, grid, lds, s);
}

template <int MI>
int dispatch_ni(int NI, const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  switch (NI) {
    case 1: return launch_conv<1, MI>(a, grid, lds, s);
    case 2: return launch_conv<2, MI>(a, grid, lds, s);
    case 3: return launch_conv<3, MI>(a, grid, lds, s);
    case 4: return launch_conv<4, MI>(a, grid, lds, s);
    case 5: return launch_conv<5, MI>(a, grid, lds, s);
    case 6: return launch_conv<6, MI>(a, grid, lds, s);
    default: break;
  }
  if constexpr (MI <= 2) {
    switch (NI) {
      case 7: return launch_conv<7, MI>(a, grid, lds, s);
      case 8: return launch_conv<8, MI>(a, grid, lds, s);
      case 9: return launch_conv<9, MI>(a, grid, lds, s);
      case 10: return launch_conv<10, MI>(a, grid, lds, s);
      default: break;
    }
  }
  mg_set_error("mg_conv3x3: unsupported tile NI=%d MI=%d", NI, MI);
  return MG_EINVAL;
}

}  // namespace

extern "C" size_t mg_conv3x3_packed_floats(int Cin, int Cout) {
  return (size_t)mg_cdiv(Cin, CC) * 9 * CC * (size_t)(16 * mg_cdiv(Cout, 16));
}

extern "C" int mg_conv3x3_pack(const float* w, float* wp, int Co, int Ci, int dgrad, mg_stream_t stream) {
  MG_CHECK_ARG(w && wp && Co > 0 && Ci > 0, "mg_conv3x3_pack: bad arguments");
  const size_t total = pack_conv3x3_total(Co, Ci, dgrad);
  const int blocks = (int)((total + 255) / 256);
  hipLaunchKernelGGL(conv3x3_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wp, Co, Ci, dgrad, total);
  MG_CHECK_LAUNCH("mg_conv3x3_pack");
  return MG_OK;
}

extern "C" int mg_conv3x3(const float* x, const float* wp, const float* bias, const float* aux, float* y, float* p,
                          float* rn, int N, int Cin, int Cout, int H, int W, int flags, float slope,
                          mg_stream_t stream) {
  MG_CHECK_ARG(x && wp && N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "mg_conv3x3: bad arguments");
  MG_CHECK_ARG(Cout <= 160, "mg_conv3x3: Cout=%d > 160 unsupported", Cout);
  const bool ups = flags & MG_CONV_UPS_IN, pn = flags & MG_CONV_PIXNORM;
  MG_CHECK_ARG(!ups || ((H % 2 == 0) && (W % 2 == 0)), "mg_conv3x3: upsampled input needs even H,W");
  MG_CHECK_ARG(!(flags & MG_CONV_MASK_AUX) || aux, "mg_conv3x3: MASK_AUX without aux");
  MG_CHECK_ARG(!pn || ((flags & MG_CONV_LRELU) && p), "mg_conv3x3: PIXNORM needs LRELU and p");
  MG_CHECK_ARG(pn || y, "mg_conv3x3: y is NULL");
  const bool want_pool = (flags & MG_CONV_POOL_OUT) != 0;
  MG_CHECK_ARG(!want_pool || (!pn && p && (H % 2 == 0) && (W % 2 == 0)), "mg_conv3x3: POOL_OUT needs p, even H,W, no PIXNORM");
  MG_CHECK_ARG(!((flags & MG_CONV_MASK_AUX) && (flags & (MG_CONV_LRELU | MG_CONV_PIXNORM))),
               "mg_conv3x3: MASK_AUX excludes LRELU/PIXNORM");
  const long long in_elems = (long long)N * Cin * (ups ? (H / 2) * (W / 2) : H * W);
  MG_CHECK_ARG(in_elems < (1ll << 31) && (long long)N * Cout * H * W < (1ll << 40), "mg_conv3x3: tensor too large");

  ConvArgs a;
  a.x = x; a.wp = wp; a.bias = bias; a.aux = aux; a.y = y; a.p = p; a.rn = rn;
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  a.Hin = ups ? H / 2 : H; a.Win = ups ? W / 2 : W;
  a.flags = flags; a.slope = slope;
  const int NIfull = mg_cdiv(Cout, 16);
  a.OPF = NIfull * 16;
  a.nchunk = mg_cdiv(Cin, CC);

  if (H * W == 1 && !pn) {
    const size_t total = (size_t)N * H * W * a.OPF;
    hipLaunchKernelGGL(conv3x3_tiny, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, (hipStream_t)stream, a);
    MG_CHECK_LAUNCH("mg_conv3x3(tiny)");
    return MG_OK;
  }
  // tile selection: fill the chip first (>= ~2 workgroups per CU), then grow the per-wave pixel tile
  const long long px = (long long)N * H * W;
  int MI = NIfull <= 6 ? 4 : 2;
  while (MI > 1 && px / (64 * MI) < 512) MI >>= 1;
  int NI = NIfull;
  if (!pn && MI == 1) {
    const long long nwg = (px + 63) / 64;
    if (nwg * 2 <= 256) NI = 1;
    else if (nwg <= 256 && (NIfull % 2 == 0)) NI = 2;
    if (NI > NIfull) NI = NIfull;
  }
  // the smallest maps, long K: split-K variant -- 16 pixels per workgroup, its 4 waves share them and split each interval's 4
  // chunks.  Measured (tools/bench_smallconv.py, Cin 128): 96 x 2x2 21.9 -> 13.4 us, 32 x 4x4 18.4 -> 12.9, 96 x 4x4 18.5 -> 16.4; at
  // 96 x 8x8 the 4x larger grid re-stages the weights 4x as often and loses (27.7 -> 58 us), hence the pixel bound.
  a.ksplit = (MI == 1 && NI <= 2 && !pn && H * W <= 64 && px <= 1536 && a.nchunk >= 8 && pf_enabled() &&
              getenv("MG_CONV_NOSUB") == nullptr && getenv("MG_CONV_NOKSPLIT") == nullptr) ? 4 : 1;
  const int P = a.ksplit == 4 ? 16 : 64 * MI;
  a.TW = mg_pow2_ceil(W) < 32 ? mg_pow2_ceil(W) : 32;
  if (a.TW > P) a.TW = P;
  a.TH = mg_pow2_ceil(H) < P / a.TW ? mg_pow2_ceil(H) : P / a.TW;
  a.TN = P / (a.TW * a.TH);
  a.lgTW = mg_ilog2(a.TW); a.lgTH = mg_ilog2(a.TH);
  a.THp = a.TH + 2; a.TWp = a.TW + 2;
  a.tiles_x = mg_cdiv(W, a.TW); a.tiles_y = mg_cdiv(H, a.TH); a.tiles_n = mg_cdiv(N, a.TN);
  a.plane = a.TN * a.THp * a.TWp;
  a.ch_stride = ((a.plane + 15) / 32) * 32 + 16;  // smallest s >= plane with s % 32 == 16
  if (a.ch_stride < a.plane) a.ch_stride += 32;
  a.tab_floats = (a.plane + 3) & ~3;
  const int OPL = (NI & 1) ? NI * 16 : NI * 16 + 16;
  a.sub = 1;
  if (MI == 1 && NI <= 2 && pf_enabled() && a.plane <= 256 && a.nchunk >= 8 && getenv("MG_CONV_NOSUB") == nullptr) a.sub = 4;
  MG_CHECK_ARG(a.ksplit == 1 || a.sub == 4, "mg_conv3x3: internal error (split-K without 4-chunk intervals)");
  const size_t lds = (size_t)(a.tab_floats + a.sub * (CC * a.ch_stride + 9 * CC * OPL)) * sizeof(float);
  MG_CHECK_ARG(lds <= 160 * 1024, "mg_conv3x3: LDS tile %zu B too large", lds);
  dim3 grid(a.tiles_x * a.tiles_y * a.tiles_n, NIfull / NI);
  MG_CHECK_ARG(NIfull % NI == 0, "mg_conv3x3: internal tile error");
  hipStream_t s = (hipStream_t)stream;
  // the pooled output is produced in the epilogue when a wave owns whole row pairs; otherwise by the stand-alone kernel
  bool pool_fused = false;
  if (want_pool) {
    const int rows_per_wave = (MI * 16) / a.TW;
    pool_fused = ((W & 3) == 0) && a.TW >= 16 && rows_per_wave >= 2 && (rows_per_wave % 2 == 0) && (a.TH % 2 == 0);
    if (!pool_fused) a.flags &= ~MG_CONV_POOL_OUT;
  }
  int rc;
  switch (MI) {
    case 4: rc = dispatch_ni<4>(NI, a, grid, lds, s); break;
    case 2: rc = dispatch_ni<2>(NI, a, grid, lds, s); break;
    default: rc = dispatch_ni<1>(NI, a, grid, lds, s); break;
  }
  if (rc == MG_OK && want_pool && !pool_fused) rc = mg_avgpool2_fwd(y, p, N * Cout, H, W, stream);
  return rc;
}
