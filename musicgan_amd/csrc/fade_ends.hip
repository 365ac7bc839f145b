// The two-channel ENDS of both networks while a block fades in, each as one launch instead of three or four:
//   critic input   [/root/reference/music_gan/networks/discriminator.py:107-113]: MagPhaseLayer(x) for the new block, AvgPool2d(x) and the
//                  old block's MagPhaseLayer on the pooled input  -> mg_stem_pair;  their data gradient back to x (two transposed 1x1
//                  convs, AvgPool2d backward, the sum)              -> mg_stem_pair_gx
//   generator head [generator.py:118-126]: ToMagnPhaseLayer of the last block and of the one before it, nearest up-sampling of the
//                  latter and the alpha blend                        -> mg_head_pair;  the blend's backward (scale, 2x2 block sums)
//                                                                       -> mg_head_pair_bwd
// Every one of these is a stream over tensors of 2 .. 160 channels with a handful of FLOPs per byte; below 64x64 maps a launch costs
// ~5 us whatever it does, and an update at the reference's batch sizes makes ~25 of them at these two ends (profiles/r05_trace_table_l3*).
// The arithmetic of each output is the one of the kernels these replace (conv1x1.hip, elementwise.hip), in a different summation
// grouping where channels are split over waves.
#include "mg_common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct StemArgs {
  const float* x;
  const float *ws, *bs, *wo, *bo;
  float *h0, *xp, *o;
  unsigned char* hm;  // optional: tile mask of h0, one byte per 2x2 tile and channel (bit 2i+j <-> h0[2Y+i][2X+j] > 0; wino3x3.hip's format)
  int N, C0, C1, H, W;
  int flags;
  float slope;
  int co_per;
};

// One thread: a patch of 2 rows x 4 columns of x (both channels) = 2 pooled pixels.  blockIdx.y slices the out-channels of both
// convolutions (small maps are latency-bound on the serial channel loop); slice 0 also writes the pooled input.
// MG_C1_MASK_AUX: h0 / o hold activations; the (bias-free) results are multiplied by their LeakyReLU derivative and written over
// them (the penalty's tangent pass, engine.disc_step_fused).
__global__ void __launch_bounds__(256) stem_pair_k(const StemArgs a) {
  const int Wq = a.W >> 2, Hh = a.H >> 1, Wh = a.W >> 1;
  const size_t HW = (size_t)a.H * a.W, PP = (size_t)Hh * Wh;
  const size_t total = (size_t)a.N * Hh * Wq;
  const bool masked = (a.flags & MG_C1_MASK_AUX) != 0, lrelu = (a.flags & MG_C1_LRELU) != 0;
  const int lo = blockIdx.y * a.co_per;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int qx = (int)(i % Wq);
    const size_t r = i / Wq;
    const int py = (int)(r % Hh);
    const int n = (int)(r / Hh);
    const size_t off = (size_t)(2 * py) * a.W + 4 * qx;    // inside a full-resolution plane
    const size_t offp = (size_t)py * Wh + 2 * qx;          // inside a pooled plane
    f32x4 xv[2][2];
    f32x2 pv[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float* p = a.x + ((size_t)n * 2 + c) * HW + off;
      xv[c][0] = *reinterpret_cast<const f32x4*>(p);
      xv[c][1] = *reinterpret_cast<const f32x4*>(p + a.W);
      pv[c] = f32x2{((xv[c][0][0] + xv[c][0][1]) + (xv[c][1][0] + xv[c][1][1])) * 0.25f,
                    ((xv[c][0][2] + xv[c][0][3]) + (xv[c][1][2] + xv[c][1][3])) * 0.25f};
      if (blockIdx.y == 0 && a.xp != nullptr) *reinterpret_cast<f32x2*>(a.xp + ((size_t)n * 2 + c) * PP + offp) = pv[c];
    }
    constexpr int OB = 4;  // out-channels whose weights (and masks) are requested before any is used
    {
      const int hi = lo + a.co_per < a.C0 ? lo + a.co_per : a.C0;
      for (int ob = lo; ob < hi; ob += OB) {
        float w0[OB], w1[OB], bv[OB];
        f32x4 m[OB][2];
#pragma unroll
        for (int u = 0; u < OB; ++u) {
          const int oc = ob + u < hi ? ob + u : hi - 1;
          w0[u] = a.ws[oc * 2]; w1[u] = a.ws[oc * 2 + 1];
          bv[u] = a.bs ? a.bs[oc] : 0.f;
          if (masked) {
            const float* mp = a.h0 + ((size_t)n * a.C0 + oc) * HW + off;
            m[u][0] = *reinterpret_cast<const f32x4*>(mp);
            m[u][1] = *reinterpret_cast<const f32x4*>(mp + a.W);
          }
        }
#pragma unroll
        for (int u = 0; u < OB; ++u) {
          if (ob + u >= hi) break;
          float* dst = a.h0 + ((size_t)n * a.C0 + ob + u) * HW + off;
          unsigned bits = 0;  // byte 0: the patch's left tile, byte 1: its right tile
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float t = fmaf(w1[u], xv[1][rr][e], fmaf(w0[u], xv[0][rr][e], bv[u]));
              if (lrelu) t = mg_lrelu(t, a.slope);
              if (masked) t *= mg_lrelu_mask(m[u][rr][e], a.slope);
              v[e] = t;
              bits |= mg_pos_bit(t) << (8 * (e >> 1) + 2 * rr + (e & 1));
            }
            *reinterpret_cast<f32x4*>(dst + rr * a.W) = v;
          }
          if (a.hm != nullptr) *reinterpret_cast<unsigned short*>(a.hm + ((size_t)n * a.C0 + ob + u) * PP + offp) = (unsigned short)bits;
        }
      }
    }
    {
      const int hi = lo + a.co_per < a.C1 ? lo + a.co_per : a.C1;
      for (int ob = lo; ob < hi; ob += OB) {
        float w0[OB], w1[OB], bv[OB];
        f32x2 m[OB];
#pragma unroll
        for (int u = 0; u < OB; ++u) {
          const int oc = ob + u < hi ? ob + u : hi - 1;
          w0[u] = a.wo[oc * 2]; w1[u] = a.wo[oc * 2 + 1];
          bv[u] = a.bo ? a.bo[oc] : 0.f;
          if (masked) m[u] = *reinterpret_cast<const f32x2*>(a.o + ((size_t)n * a.C1 + oc) * PP + offp);
        }
#pragma unroll
        for (int u = 0; u < OB; ++u) {
          if (ob + u >= hi) break;
          f32x2 v;
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            float t = fmaf(w1[u], pv[1][e], fmaf(w0[u], pv[0][e], bv[u]));
            if (lrelu) t = mg_lrelu(t, a.slope);
            if (masked) t *= mg_lrelu_mask(m[u][e], a.slope);
            v[e] = t;
          }
          *reinterpret_cast<f32x2*>(a.o + ((size_t)n * a.C1 + ob + u) * PP + offp) = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Many channels in, two out, at two resolutions.  A lane owns one 2x2 quad of the full-resolution map (= one low-resolution
// pixel); the four waves of a workgroup split the channels of BOTH inputs (wave w: channels w, w + 4, ...; loads stay coalesced:
// 8 bytes per lane and row) and combine their ten partial sums through LDS in a fixed order -- conv1x1_few_out_split's scheme.
struct PairArgs {
  const float* xf;   // (N, Cf, H, W)   full resolution
  const float* xl;   // (N, Cl, H/2, W/2)
  const float *wf, *wl;  // element (o, c) at w[o * so + c * sc]
  int sof, scf, sol, scl;
  const float *bf, *bl;
  const float* coef;  // device (a, b) or NULL
  float ca, cb;
  float *yf, *yl, *out;
  int N, Cf, Cl, H, W;
  int mode;  // 0: gx = yf_sum + 0.25 * yl_sum -> out;   1: yf = tanh(.), yl = tanh(.), out = a * yf + b * up(yl)
};

template <int MODE>
__global__ void __launch_bounds__(256) pair_few_out_k(const PairArgs a) {
  __shared__ float red[3][10][64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int Hh = a.H >> 1, Wh = a.W >> 1;
  const size_t HW = (size_t)a.H * a.W, PP = (size_t)Hh * Wh;
  const size_t total = (size_t)a.N * PP;
  float ca = a.ca, cb = a.cb;
  if (MODE != 0 && a.coef != nullptr) { ca = a.coef[0]; cb = a.coef[1]; }
  for (size_t base = (size_t)blockIdx.x * 64; base < total; base += (size_t)gridDim.x * 64) {
    const size_t i = base + lane;
    const bool ok = i < total;
    const int n = ok ? (int)(i / PP) : 0;
    const size_t q = ok ? i - (size_t)n * PP : 0;
    const int py = (int)(q / Wh), px = (int)(q - (size_t)py * Wh);
    const size_t off = (size_t)(2 * py) * a.W + 2 * px;
    float acc[2][4], accl[2];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      accl[o] = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[o][e] = 0.f;
    }
    constexpr int U = 8;
    if (ok) {
      const float* xp = a.xf + (size_t)n * a.Cf * HW + off;
      for (int c0 = wave; MODE != 2 && c0 < a.Cf; c0 += 4 * U) {  // (MODE 2: the front head's values are given in yf)
        f32x2 r0[U], r1[U];
        float w[U][2];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int c = c0 + 4 * u < a.Cf ? c0 + 4 * u : wave;
          r0[u] = *reinterpret_cast<const f32x2*>(xp + (size_t)c * HW);
          r1[u] = *reinterpret_cast<const f32x2*>(xp + (size_t)c * HW + a.W);
          w[u][0] = a.wf[c * a.scf];
          w[u][1] = a.wf[a.sof + c * a.scf];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool live = c0 + 4 * u < a.Cf;  // (channels past the end re-read the wave's first one and count as 0)
          const f32x2 v0 = live ? r0[u] : f32x2{0.f, 0.f}, v1 = live ? r1[u] : f32x2{0.f, 0.f};
#pragma unroll
          for (int o = 0; o < 2; ++o) {
            acc[o][0] = fmaf(w[u][o], v0[0], acc[o][0]);
            acc[o][1] = fmaf(w[u][o], v0[1], acc[o][1]);
            acc[o][2] = fmaf(w[u][o], v1[0], acc[o][2]);
            acc[o][3] = fmaf(w[u][o], v1[1], acc[o][3]);
          }
        }
      }
      const float* lp = a.xl + (size_t)n * a.Cl * PP + q;
      for (int c0 = wave; c0 < a.Cl; c0 += 4 * U) {
        float v[U], w[U][2];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int c = c0 + 4 * u < a.Cl ? c0 + 4 * u : wave;
          v[u] = lp[(size_t)c * PP];
          w[u][0] = a.wl[c * a.scl];
          w[u][1] = a.wl[a.sol + c * a.scl];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const float vv = c0 + 4 * u < a.Cl ? v[u] : 0.f;
          accl[0] = fmaf(w[u][0], vv, accl[0]);
          accl[1] = fmaf(w[u][1], vv, accl[1]);
        }
      }
    }
    if (wave > 0) {
#pragma unroll
      for (int o = 0; o < 2; ++o) {
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave - 1][o * 4 + e][lane] = acc[o][e];
        red[wave - 1][8 + o][lane] = accl[o];
      }
    }
    __syncthreads();
    if (wave == 0 && ok) {
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        float s[4], sl;
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] = ((acc[o][e] + red[0][o * 4 + e][lane]) + red[1][o * 4 + e][lane]) + red[2][o * 4 + e][lane];
        sl = ((accl[o] + red[0][8 + o][lane]) + red[1][8 + o][lane]) + red[2][8 + o][lane];
        float* po = a.out + ((size_t)n * 2 + o) * HW + off;
        if (MODE == 0) {
          const float l = 0.25f * sl;
          *reinterpret_cast<f32x2*>(po) = f32x2{s[0] + l, s[1] + l};
          *reinterpret_cast<f32x2*>(po + a.W) = f32x2{s[2] + l, s[3] + l};
        } else {
          const float bfv = a.bf ? a.bf[o] : 0.f, blv = a.bl ? a.bl[o] : 0.f;
          const float t = tanhf(sl + blv);
          float m[4];
          if (a.yl != nullptr) a.yl[((size_t)n * 2 + o) * PP + q] = t;
          if (MODE == 2) {
            const float* pm = a.yf + ((size_t)n * 2 + o) * HW + off;
            const f32x2 r0 = *reinterpret_cast<const f32x2*>(pm), r1 = *reinterpret_cast<const f32x2*>(pm + a.W);
            m[0] = r0[0]; m[1] = r0[1]; m[2] = r1[0]; m[3] = r1[1];
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) m[e] = tanhf(s[e] + bfv);
            if (a.yf != nullptr) {
              float* pm = a.yf + ((size_t)n * 2 + o) * HW + off;
              *reinterpret_cast<f32x2*>(pm) = f32x2{m[0], m[1]};
              *reinterpret_cast<f32x2*>(pm + a.W) = f32x2{m[2], m[3]};
            }
          }
          const float yl = cb * t;
          *reinterpret_cast<f32x2*>(po) = f32x2{fmaf(ca, m[0], yl), fmaf(ca, m[1], yl)};
          *reinterpret_cast<f32x2*>(po + a.W) = f32x2{fmaf(ca, m[2], yl), fmaf(ca, m[3], yl)};
        }
      }
    }
    __syncthreads();
  }
}

// backward of out = a * x + b * up2(y):  gx = a * g,  gy = b * (2x2 block sums of g).  One thread: 2 rows x 4 columns.
__global__ void __launch_bounds__(256) blend_up_bwd_k(const float* __restrict__ g, const float* __restrict__ coef, float ca, float cb,
                                                      float* __restrict__ gx, float* __restrict__ gy, size_t total, int Hh, int Wq) {
  if (coef) { ca = coef[0]; cb = coef[1]; }
  const int W = Wq * 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int qx = (int)(i % Wq);
    const size_t r = i / Wq;
    const int py = (int)(r % Hh);
    const size_t nc = r / Hh;
    const size_t off = (nc * (2 * Hh) + 2 * py) * (size_t)W + 4 * qx;
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(g + off), g1 = *reinterpret_cast<const f32x4*>(g + off + W);
    *reinterpret_cast<f32x4*>(gx + off) = g0 * ca;
    *reinterpret_cast<f32x4*>(gx + off + W) = g1 * ca;
    *reinterpret_cast<f32x2*>(gy + (nc * Hh + py) * (size_t)(2 * Wq) + 2 * qx) =
        f32x2{cb * ((g0[0] + g0[1]) + (g1[0] + g1[1])), cb * ((g0[2] + g0[3]) + (g1[2] + g1[3]))};
  }
}

int grid_for(size_t items, int per_block, int cap) {
  size_t b = (items + per_block - 1) / per_block;
  if (b > (size_t)cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}


// ---- generator head, backward, with the PixelNorm + LeakyReLU backward of the last block's second conv in the same pass
// [generator.py:118-126 ToMagnPhaseLayer = tanh(conv1x1), layers.py:11-17 PixelNorm, generator.py:31-39]: replaces, per pixel,
//   t_f     = g_mp[f] (1 - mp[f]^2)                              tanh backward (conv1x1.hip: MG_C1_TANH_BWD_IN)
//   gw[f][c] += t_f p[c],  gb[f] += t_f                         conv1x1_wgrad (head weights (2, C), x = p: the block's output)
//   g[c]    = w[0][c] t_0 + w[1][c] t_1                         conv1x1 transposed (the head's data gradient)
//   gpre[c] = lrelu'(p[c]) rn (g[c] - p[c] mean_c(g p))         pixelnorm_lrelu_bwd, from_p form (elementwise.hip)
// -- three launches that read p twice and wrote / re-read the C-channel g (5 tensor passes) -- as one read of p and one write of gpre.
// One pixel per thread and turn; the pixel's C values of p wait in LDS between the two sweeps (a thread reads back its own column only:
// no barrier); the 2 C weight-gradient sums stay in registers over the thread's pixels and are reduced once per workgroup.
// GIN: a second gradient arriving at p (the data gradient of the conv that reads p; the OLD head of a fading-in level sits on the input
// of the last block, which that block's first conv reads too) is added to the head's: g[c] = gin[c] + w^T t; the second sweep reads gin
// again (the workgroup's own lines, minutes of L2 residency ago) rather than parking it beside p (twice the LDS, half the waves).
// Channels go in groups of 16 (a scheduling fence between groups): all C loads of a pixel in flight at once cost C registers on top of
// the 2 C sums and left one wave per SIMD at C = 64.
template <int C, int NT, bool GIN>
__global__ void __launch_bounds__(NT) gen_head_bwd_k(const float* __restrict__ gm, const float* __restrict__ mp, const float* __restrict__ w,
                                                     const float* __restrict__ p, const float* __restrict__ rn, const float* __restrict__ gin,
                                                     float* __restrict__ gpre, float* __restrict__ part, int N, int HW, float slope) {
  extern __shared__ __attribute__((aligned(16))) float park[];  // [C][NT], then (reduction) [NT / 64][2 C + 2]
  constexpr int GS = 16;  // channels per group
  const int tid = threadIdx.x;
  float s0[C], s1[C], sb0 = 0.f, sb1 = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) s0[c] = s1[c] = 0.f;
  const size_t total = (size_t)N * HW, stride = (size_t)gridDim.x * NT;
  for (size_t i = (size_t)blockIdx.x * NT + tid; i < total; i += stride) {
    const int n = (int)(i / HW);
    const int px = (int)(i - (size_t)n * HW);
    const size_t fb = (size_t)n * 2 * HW + px, base = (size_t)n * C * HW + px;
    const float m0 = mp[fb], m1 = mp[fb + HW];
    const float t0 = gm[fb] * (1.f - m0 * m0), t1 = gm[fb + HW] * (1.f - m1 * m1);
    const float r = rn[i];
    sb0 += t0;
    sb1 += t1;
    float dot = 0.f;
#pragma unroll
    for (int c0 = 0; c0 < C; c0 += GS) {
      float tv[GS], gv[GIN ? GS : 1];
#pragma unroll
      for (int u = 0; u < GS; ++u) {
        tv[u] = p[base + (size_t)(c0 + u) * HW];
        if constexpr (GIN) gv[u] = gin[base + (size_t)(c0 + u) * HW];
      }
#pragma unroll
      for (int u = 0; u < GS; ++u) {
        const int c = c0 + u;
        const float t = tv[u];
        park[c * NT + tid] = t;
        float g = fmaf(w[C + c], t1, w[c] * t0);
        if constexpr (GIN) g += gv[u];
        dot = fmaf(g, t, dot);
        s0[c] = fmaf(t0, t, s0[c]);
        s1[c] = fmaf(t1, t, s1[c]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    dot /= (float)C;
    // (an opaque copy of the pointer: hipcc would otherwise keep the first sweep's C values of gin in registers for this one)
    const float* gin2 = gin;
    if constexpr (GIN) asm volatile("" : "+s"(gin2));
#pragma unroll
    for (int c0 = 0; c0 < C; c0 += GS) {
      float gv[GIN ? GS : 1];
      if constexpr (GIN) {
#pragma unroll
        for (int u = 0; u < GS; ++u) gv[u] = gin2[base + (size_t)(c0 + u) * HW];
      }
#pragma unroll
      for (int u = 0; u < GS; ++u) {
        const int c = c0 + u;
        const float t = park[c * NT + tid];
        float g = fmaf(w[C + c], t1, w[c] * t0);
        if constexpr (GIN) g += gv[u];
        gpre[base + (size_t)c * HW] = mg_lrelu_mask(t, slope) * r * (g - t * dot);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // the workgroup's sums: wave totals by DPP, the four waves through LDS, one row of 2 C + 2 partials per workgroup
  __syncthreads();  // (every thread is done with its parked column)
  const int lane = tid & 63, wave = tid >> 6;
  constexpr int PER = 2 * C + 2;
  float* red = park;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float a0 = mg_wave_sum_to_lane63(s0[c]), a1 = mg_wave_sum_to_lane63(s1[c]);
    if (lane == 63) {
      red[wave * PER + c] = a0;
      red[wave * PER + C + c] = a1;
    }
  }
  {
    const float a0 = mg_wave_sum_to_lane63(sb0), a1 = mg_wave_sum_to_lane63(sb1);
    if (lane == 63) {
      red[wave * PER + 2 * C] = a0;
      red[wave * PER + 2 * C + 1] = a1;
    }
  }
  __syncthreads();
  for (int e = tid; e < PER; e += NT) {
    float v = red[e];
#pragma unroll
    for (int wv = 1; wv < NT / 64; ++wv) v += red[wv * PER + e];
    part[(size_t)blockIdx.x * PER + e] = v;
  }
}

// The same for ANY channel count up to 128 on small maps (the heads of levels 3 / 4: 80 .. 128 channels on 32x32 / 16x16): a workgroup takes
// 64 pixels at a time (lane = pixel), its 4 waves split the channels (wave w: channels w, w + 4, ...: at most 32 each, p and gin of a
// pixel in registers), the four partial dot products meet in LDS.  One pass, no parking; 2 C / 4 sums per lane.
constexpr int GHS_MAXQ = 32;  // channels per wave
template <bool GIN>
__global__ void __launch_bounds__(256) gen_head_bwd_small_k(const float* __restrict__ gm, const float* __restrict__ mp, const float* __restrict__ w,
                                                            const float* __restrict__ p, const float* __restrict__ rn,
                                                            const float* __restrict__ gin, float* __restrict__ gpre, float* __restrict__ part,
                                                            int N, int C, int HW, float slope) {
  __shared__ float dots[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float s0[GHS_MAXQ], s1[GHS_MAXQ], sb0 = 0.f, sb1 = 0.f;
#pragma unroll
  for (int k = 0; k < GHS_MAXQ; ++k) s0[k] = s1[k] = 0.f;
  const size_t total = (size_t)N * HW;
  for (size_t b0 = (size_t)blockIdx.x * 64; b0 < total; b0 += (size_t)gridDim.x * 64) {
    const size_t i = b0 + lane;
    const bool ok = i < total;
    const int n = ok ? (int)(i / HW) : 0;
    const int px = ok ? (int)(i - (size_t)n * HW) : 0;
    const size_t fb = (size_t)n * 2 * HW + px, base = (size_t)n * C * HW + px;
    const float m0 = ok ? mp[fb] : 0.f, m1 = ok ? mp[fb + HW] : 0.f;
    const float t0 = ok ? gm[fb] * (1.f - m0 * m0) : 0.f, t1 = ok ? gm[fb + HW] * (1.f - m1 * m1) : 0.f;
    const float r = ok ? rn[i] : 0.f;
    if (wave == 0) {
      sb0 += t0;
      sb1 += t1;
    }
    float tv[GHS_MAXQ], gv[GHS_MAXQ];
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < GHS_MAXQ; ++k) {
      const int c = wave + 4 * k;
      const bool live = ok && c < C;
      tv[k] = live ? p[base + (size_t)c * HW] : 0.f;
      gv[k] = (GIN && live) ? gin[base + (size_t)c * HW] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < GHS_MAXQ; ++k) {
      const int c = wave + 4 * k < C ? wave + 4 * k : 0;
      gv[k] += fmaf(w[C + c], t1, w[c] * t0);
      dot = fmaf(gv[k], tv[k], dot);
      s0[k] = fmaf(t0, tv[k], s0[k]);
      s1[k] = fmaf(t1, tv[k], s1[k]);
    }
    dots[wave][lane] = dot;
    __syncthreads();
    dot = (((dots[0][lane] + dots[1][lane]) + dots[2][lane]) + dots[3][lane]) / (float)C;
#pragma unroll
    for (int k = 0; k < GHS_MAXQ; ++k) {
      const int c = wave + 4 * k;
      if (ok && c < C) gpre[base + (size_t)c * HW] = mg_lrelu_mask(tv[k], slope) * r * (gv[k] - tv[k] * dot);
    }
    __syncthreads();  // (dots is rewritten in the next turn)
  }
  // per workgroup: row of 2 C + 2 partial sums in the layout of gen_head_bwd_k ([f][c], then the two bias sums)
  const int per = 2 * C + 2;
#pragma unroll
  for (int k = 0; k < GHS_MAXQ; ++k) {
    const float a0 = mg_wave_sum_to_lane63(s0[k]), a1 = mg_wave_sum_to_lane63(s1[k]);
    const int c = wave + 4 * k;
    if (lane == 63 && c < C) {
      part[(size_t)blockIdx.x * per + c] = a0;
      part[(size_t)blockIdx.x * per + C + c] = a1;
    }
  }
  if (wave == 0) {
    const float a0 = mg_wave_sum_to_lane63(sb0), a1 = mg_wave_sum_to_lane63(sb1);
    if (lane == 63) {
      part[(size_t)blockIdx.x * per + 2 * C] = a0;
      part[(size_t)blockIdx.x * per + 2 * C + 1] = a1;
    }
  }
}

// gw (2, C) and gb (2) from the per-workgroup partials: one wave per output, lanes stride over the workgroups, a fixed shuffle tree
__global__ void __launch_bounds__(256) gen_head_bwd_final_k(const float* __restrict__ part, int G, int C, float* __restrict__ gw,
                                                            float* __restrict__ gb, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6), per = 2 * C + 2;
  if (e >= per) return;
  float s = 0.f;
  for (int b = lane; b < G; b += 64) s += part[(size_t)b * per + e];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
  if (lane == 0) {
    float* dst = e < 2 * C ? gw + e : gb + (e - 2 * C);
    *dst = accumulate ? *dst + s : s;
  }
}

template <int C, int NT, bool GIN>
int gen_head_bwd_launch_v(const float* gm, const float* mp, const float* w, const float* p, const float* rn, const float* gin, float* gpre,
                          float* part, int G, int N, int HW, float slope, hipStream_t s) {
  static MgPerDevice once;
  if (mg_first_use_on_device(once))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gen_head_bwd_k<C, NT, GIN>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
  hipLaunchKernelGGL((gen_head_bwd_k<C, NT, GIN>), dim3(G), dim3(NT), (size_t)C * NT * sizeof(float), s, gm, mp, w, p, rn, gin,
                     gpre, part, N, HW, slope);
  MG_CHECK_LAUNCH("mg_gen_head_bwd");
  return MG_OK;
}
template <int C>
int gen_head_bwd_launch(const float* gm, const float* mp, const float* w, const float* p, const float* rn, const float* gin, float* gpre,
                        float* part, int G, int N, int HW, float slope, hipStream_t s) {
  if (gin != nullptr) return gen_head_bwd_launch_v<C, 256, true>(gm, mp, w, p, rn, gin, gpre, part, G, N, HW, slope, s);
  return gen_head_bwd_launch_v<C, 256, false>(gm, mp, w, p, rn, nullptr, gpre, part, G, N, HW, slope, s);
}

// workgroups: as many as fit the chip at once (LDS: C KB each), never more than there are pixel groups
int gen_head_bwd_grid(int C, size_t pixels, bool) {
  int per_cu = (160 * 1024) / (C * 1024);
  if (per_cu > 8) per_cu = 8;
  if (per_cu < 1) per_cu = 1;
  size_t g = (size_t)per_cu * mg_cu_count();
  const size_t nt = 256, need = (pixels + nt - 1) / nt;
  if (g > need) g = need;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" int mg_stem_pair(const float* x, const float* ws, const float* bs, const float* wo, const float* bo, float* h0, float* xp,
                            float* o, unsigned char* h0_mask, int N, int C0, int C1, int H, int W, int flags, float slope,
                            mg_stream_t stream) {
  MG_CHECK_ARG(x && ws && wo && h0 && o && N > 0 && C0 > 0 && C1 > 0 && H > 0 && W > 0, "mg_stem_pair: bad arguments");
  MG_CHECK_ARG((H % 2) == 0 && (W % 4) == 0, "mg_stem_pair: needs H %% 2 == 0 and W %% 4 == 0 (got %dx%d)", H, W);
  MG_CHECK_ARG(!(flags & ~(MG_C1_LRELU | MG_C1_MASK_AUX)), "mg_stem_pair: flags other than MG_C1_LRELU | MG_C1_MASK_AUX");
  MG_CHECK_ARG(!((flags & MG_C1_MASK_AUX) && (bs || bo)), "mg_stem_pair: the masked (tangent) form is bias-free");
  StemArgs a;
  MG_CHECK_ARG(!(h0_mask && (flags & MG_C1_MASK_AUX)), "mg_stem_pair: the tile mask is an output of the forward form");
  a.x = x; a.ws = ws; a.bs = bs; a.wo = wo; a.bo = bo; a.h0 = h0; a.xp = xp; a.o = o; a.hm = h0_mask;
  a.N = N; a.C0 = C0; a.C1 = C1; a.H = H; a.W = W; a.flags = flags; a.slope = slope;
  const size_t px = (size_t)N * H * W;
  const int cmax = C0 > C1 ? C0 : C1;
  a.co_per = px >= ((size_t)1 << 21) ? cmax : (px >= ((size_t)1 << 19) ? 16 : 8);
  const dim3 grid(grid_for(px / 8, 256, 2048), mg_cdiv(cmax, a.co_per));
  hipLaunchKernelGGL(stem_pair_k, grid, dim3(256), 0, (hipStream_t)stream, a);
  MG_CHECK_LAUNCH("mg_stem_pair");
  return MG_OK;
}

extern "C" int mg_stem_pair_gx(const float* gs, const float* ws, const float* go, const float* wo, float* gx, int N, int C0, int C1,
                               int H, int W, mg_stream_t stream) {
  MG_CHECK_ARG(gs && ws && go && wo && gx && N > 0 && C0 >= 4 && C1 >= 4, "mg_stem_pair_gx: bad arguments (at least 4 channels each)");
  MG_CHECK_ARG(H > 0 && W > 0 && (H % 2) == 0 && (W % 2) == 0, "mg_stem_pair_gx: needs even H and W (got %dx%d)", H, W);
  PairArgs a = {};
  a.xf = gs; a.xl = go; a.wf = ws; a.wl = wo;
  a.sof = 1; a.scf = 2; a.sol = 1; a.scl = 2;  // w[c][o] of the (C, 2) stem weights: the transposed use
  a.out = gx; a.N = N; a.Cf = C0; a.Cl = C1; a.H = H; a.W = W; a.mode = 0;
  const size_t quads = (size_t)N * (H / 2) * (W / 2);
  hipLaunchKernelGGL(pair_few_out_k<0>, dim3(grid_for(quads, 64, 4096)), dim3(256), 0, (hipStream_t)stream, a);
  MG_CHECK_LAUNCH("mg_stem_pair_gx");
  return MG_OK;
}

extern "C" int mg_head_pair(const float* x, const float* wh, const float* bh, const float* xl, const float* wo, const float* bo,
                            const float* coef, float ca, float cb, float* mp, float* old, float* out, int N, int C, int Cl, int H,
                            int W, mg_stream_t stream) {
  MG_CHECK_ARG(x && wh && xl && wo && out && N > 0 && C >= 4 && Cl >= 4, "mg_head_pair: bad arguments (at least 4 channels each)");
  MG_CHECK_ARG(H > 0 && W > 0 && (H % 2) == 0 && (W % 2) == 0, "mg_head_pair: needs even H and W (got %dx%d)", H, W);
  PairArgs a = {};
  a.xf = x; a.xl = xl; a.wf = wh; a.wl = wo; a.bf = bh; a.bl = bo;
  a.sof = C; a.scf = 1; a.sol = Cl; a.scl = 1;  // (2, C) head weights
  a.coef = coef; a.ca = ca; a.cb = cb;
  a.yf = mp; a.yl = old; a.out = out; a.N = N; a.Cf = C; a.Cl = Cl; a.H = H; a.W = W; a.mode = 1;
  const size_t quads = (size_t)N * (H / 2) * (W / 2);
  hipLaunchKernelGGL(pair_few_out_k<1>, dim3(grid_for(quads, 64, 4096)), dim3(256), 0, (hipStream_t)stream, a);
  MG_CHECK_LAUNCH("mg_head_pair");
  return MG_OK;
}

// mg_head_pair with the new head's values given (mg_winoups3x3_head wrote them in the last conv's epilogue): old = tanh(wo xl + bo),
// out = a mp + b up2(old)
extern "C" int mg_head_pair_from_mp(const float* mp, const float* xl, const float* wo, const float* bo, const float* coef, float ca, float cb,
                                    float* old, float* out, int N, int Cl, int H, int W, mg_stream_t stream) {
  MG_CHECK_ARG(mp && xl && wo && out && N > 0 && Cl >= 4, "mg_head_pair_from_mp: bad arguments (at least 4 channels)");
  MG_CHECK_ARG(H > 0 && W > 0 && (H % 2) == 0 && (W % 2) == 0, "mg_head_pair_from_mp: needs even H and W (got %dx%d)", H, W);
  PairArgs a = {};
  a.xf = xl; a.xl = xl; a.wf = wo; a.wl = wo; a.bf = nullptr; a.bl = bo;  // (the front side's operands are not read in this mode)
  a.sof = Cl; a.scf = 1; a.sol = Cl; a.scl = 1;
  a.coef = coef; a.ca = ca; a.cb = cb;
  a.yf = const_cast<float*>(mp); a.yl = old; a.out = out; a.N = N; a.Cf = Cl; a.Cl = Cl; a.H = H; a.W = W; a.mode = 2;
  const size_t quads = (size_t)N * (H / 2) * (W / 2);
  hipLaunchKernelGGL(pair_few_out_k<2>, dim3(grid_for(quads, 64, 4096)), dim3(256), 0, (hipStream_t)stream, a);
  MG_CHECK_LAUNCH("mg_head_pair_from_mp");
  return MG_OK;
}

extern "C" int mg_blend_up_bwd(const float* g, const float* coef, float ca, float cb, float* gx, float* gy, int NC, int H, int W,
                               mg_stream_t stream) {
  MG_CHECK_ARG(g && gx && gy && NC > 0 && H > 0 && W > 0, "mg_blend_up_bwd: bad arguments");
  MG_CHECK_ARG((H % 2) == 0 && (W % 4) == 0, "mg_blend_up_bwd: needs H %% 2 == 0 and W %% 4 == 0 (got %dx%d)", H, W);
  const size_t total = (size_t)NC * (H / 2) * (W / 4);
  hipLaunchKernelGGL(blend_up_bwd_k, dim3(grid_for(total, 256, 2048)), dim3(256), 0, (hipStream_t)stream, g, coef, ca, cb, gx, gy, total,
                     H / 2, W / 4);
  MG_CHECK_LAUNCH("mg_blend_up_bwd");
  return MG_OK;
}

// the streaming form: C in {16, 32, 48, 64}; the small-map form: any C <= 128 that is a multiple of 4, up to 2^15 pixels
static bool ghb_stream_ok(int C) { return C == 16 || C == 32 || C == 48 || C == 64; }
static bool ghb_small_ok(int C, size_t pixels) { return C >= 4 && C <= 4 * GHS_MAXQ && (C % 4) == 0 && pixels <= (1u << 15); }
static int ghb_small_grid(size_t pixels) {
  size_t g = (pixels + 63) / 64;
  const size_t cap = 2 * (size_t)mg_cu_count();
  return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

extern "C" int mg_gen_head_bwd_supported(int C, int Cout) { return (Cout == 2 && ghb_stream_ok(C)) ? 1 : 0; }
extern "C" int mg_gen_head_bwd_supported_at(int C, int Cout, int N, int HW) {
  return (Cout == 2 && (ghb_stream_ok(C) || ghb_small_ok(C, (size_t)N * HW))) ? 1 : 0;
}

extern "C" size_t mg_gen_head_bwd_ws_floats(int N, int C, int HW) {
  const size_t pixels = (size_t)N * HW;
  const size_t g = ghb_stream_ok(C) ? (size_t)gen_head_bwd_grid(C, pixels, true) : (size_t)ghb_small_grid(pixels);
  return g * (2 * C + 2);
}

extern "C" int mg_gen_head_bwd(const float* g_mp, const float* mp, const float* w, const float* p, const float* rn, const float* g_in,
                               float* gpre, float* gw, float* gb, float* ws, size_t ws_floats, int N, int C, int HW, float slope,
                               int accumulate, mg_stream_t stream) {
  MG_CHECK_ARG(g_mp && mp && w && p && rn && gpre && gw && gb && ws && N > 0 && HW > 0, "mg_gen_head_bwd: bad arguments");
  MG_CHECK_ARG(mg_gen_head_bwd_supported_at(C, 2, N, HW), "mg_gen_head_bwd: C = %d (16, 32, 48, 64; any multiple of 4 up to 128 on <= 32 768 pixels)", C);
  if (!ghb_stream_ok(C)) {
    const int G = ghb_small_grid((size_t)N * HW);
    MG_CHECK_ARG(ws_floats >= (size_t)G * (2 * C + 2), "mg_gen_head_bwd: workspace of %zu floats, needs %zu", ws_floats, (size_t)G * (2 * C + 2));
    hipStream_t s = (hipStream_t)stream;
    if (g_in != nullptr)
      hipLaunchKernelGGL(gen_head_bwd_small_k<true>, dim3(G), dim3(256), 0, s, g_mp, mp, w, p, rn, g_in, gpre, ws, N, C, HW, slope);
    else
      hipLaunchKernelGGL(gen_head_bwd_small_k<false>, dim3(G), dim3(256), 0, s, g_mp, mp, w, p, rn, g_in, gpre, ws, N, C, HW, slope);
    MG_CHECK_LAUNCH("mg_gen_head_bwd(small)");
    hipLaunchKernelGGL(gen_head_bwd_final_k, dim3((2 * C + 2 + 3) / 4), dim3(256), 0, s, ws, G, C, gw, gb, accumulate);
    MG_CHECK_LAUNCH("mg_gen_head_bwd(final)");
    return MG_OK;
  }
  const int G = gen_head_bwd_grid(C, (size_t)N * HW, g_in != nullptr);
  MG_CHECK_ARG(ws_floats >= (size_t)G * (2 * C + 2), "mg_gen_head_bwd: workspace of %zu floats, needs %zu", ws_floats, (size_t)G * (2 * C + 2));
  hipStream_t s = (hipStream_t)stream;
  int rc;
  switch (C) {
    case 16: rc = gen_head_bwd_launch<16>(g_mp, mp, w, p, rn, g_in, gpre, ws, G, N, HW, slope, s); break;
    case 32: rc = gen_head_bwd_launch<32>(g_mp, mp, w, p, rn, g_in, gpre, ws, G, N, HW, slope, s); break;
    case 48: rc = gen_head_bwd_launch<48>(g_mp, mp, w, p, rn, g_in, gpre, ws, G, N, HW, slope, s); break;
    default: rc = gen_head_bwd_launch<64>(g_mp, mp, w, p, rn, g_in, gpre, ws, G, N, HW, slope, s); break;
  }
  if (rc != MG_OK) return rc;
  hipLaunchKernelGGL(gen_head_bwd_final_k, dim3((2 * C + 2 + 3) / 4), dim3(256), 0, s, ws, G, C, gw, gb, accumulate);
  MG_CHECK_LAUNCH("mg_gen_head_bwd(final)");
  return MG_OK;
}
