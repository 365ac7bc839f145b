// 3x3 / stride 1 / pad 1 convolution in Winograd F(2x2,3x3) form on the exact-fp32 matrix cores of gfx950: 16 multiplies per
// 2x2 output tile and channel pair instead of 36, i.e. 2.25x fewer MFMA cycles than the implicit GEMM of conv3x3.hip for the
// same nn.Conv2d(3x3) + LeakyReLU (+ PixelNorm) (+ AvgPool2d) of
//   /root/reference/music_gan/networks/generator.py:9-40 and discriminator.py:8-34
// (forward, data gradient and the tangent pass of the gradient penalty all go through it, the latter two with re-packed weights).
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A      d: 4x4 input patch (stride 2), g: 3x3 filter, Y: 2x2 outputs
//
// Per Winograd component xi (16 of them) this is a GEMM  M_xi[tile, out-ch] = V_xi[tile, c] * U_xi[c, out-ch]:
//   M (A rows)  = 16 tiles per wave (one 16x16x4 MFMA tile), 4 tile groups per workgroup = 64 tiles = 256 output pixels
//   N (B cols)  = NIW x 16 output channels per wave, WC channel groups per workgroup
//   K           = input channels, 8 per LDS chunk (two MFMA k-steps: lane k-index rq holds channels 2rq and 2rq+1)
// A wave keeps all 16 components of its (tile, channel) block in registers (16 x NIW accumulator tiles), so the output
// transform, bias, LeakyReLU, PixelNorm, mask and 2x2 average pool are in-register epilogues; the input transform is done once
// per (tile, channel) by the staging threads on the way from HBM to LDS; U = G g G^T is pre-computed by the pack kernel.
// LDS images are laid out in operand order ([component][lane][2 k-steps]) so every operand read is one conflict-free
// ds_read_b64 and the weight image is a verbatim copy of the packed global layout.
// fp32 throughout; rounding differs from the direct form by ~1.2x rms (measured against fp64, tests/test_ops_gpu.py).
#include <cstdlib>
#include <type_traits>

#include "mg_common.h"

namespace {

constexpr int WCC = 8;  // input channels per LDS chunk
constexpr int WT = 4;   // tile groups (16 tiles each) per workgroup
constexpr float PN_EPS = 1e-8f;

struct WinoArgs {
  const float* x;
  const float* up;
  const float* bias;
  const float* aux;
  float* y;
  float* p;
  float* rn;
  int N, Cin, Cout, H, W;
  int flags;
  float slope;
  int TBW, TBH, TBN, lgTBW, lgTBH;  // tile-block geometry in TILES: TBW * TBH * TBN == 64
  int blocks_x, blocks_y, blocks_n;
  int nchunk;
  int NT;  // out-channel tiles in the packed weights (padded)
};

template <int NIW, int WC>
__global__ void __launch_bounds__(256 * WC) wino3x3_mfma(const WinoArgs a) {
  constexpr int NTHR = 256 * WC;
  constexpr int NWAVE = 4 * WC;
  constexpr int NITEM = WCC / NWAVE;  // (tile, channel) items transformed per thread and chunk
  constexpr int NU4 = NIW * 2;        // 16-byte pieces of the weight image per thread and chunk
  constexpr int V_FLOATS = 16 * WT * 128;
  constexpr int STAGE = V_FLOATS + WC * NIW * 2048;  // one pipeline stage: V image + U image
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // stage s: Vs = smem + s*STAGE  [16 comps][WT][64 lanes][2],  Us = Vs + V_FLOATS  [WC*NIW tiles][16 comps][64 lanes][2]
  float* red = smem + 2 * STAGE;  // PixelNorm cross-wave partial sums [WC][64 tiles][4]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, rq = lane >> 4;
  const int wt = wave & 3, wc = wave >> 2;
  const int ct0 = blockIdx.y * (WC * NIW);
  const int HW = a.H * a.W;
  const int Ht = a.H >> 1, Wt = a.W >> 1;
  const int nblk = a.blocks_x * a.blocks_y * a.blocks_n;
  const int first = mg_xcd_remap(blockIdx.x, gridDim.x);
  const int nmine = (nblk - first + (int)gridDim.x - 1) / (int)gridDim.x;  // spatial blocks first, first+grid, ...
  const int Q = nmine * a.nchunk;                                          // pipeline steps of this workgroup

  // staging geometry of one block: this thread transforms tile `lane`, for input channels wave + k*NWAVE of every chunk
  unsigned voff[16];   // byte offsets of the 4x4 patch from the block's first image (0 where masked)
  unsigned okmask = 0;
  size_t img0 = 0;     // element offset of the block's first image
  auto geometry = [&](int blk) {
    const int bx = blk % a.blocks_x;
    const int t2 = blk / a.blocks_x;
    const int by = t2 % a.blocks_y;
    const int bn = t2 / a.blocks_y;
    const int txl = lane & (a.TBW - 1);
    const int tyl = (lane >> a.lgTBW) & (a.TBH - 1);
    const int nl = lane >> (a.lgTBW + a.lgTBH);
    const int n = bn * a.TBN + nl, TY = by * a.TBH + tyl, TX = bx * a.TBW + txl;
    const bool ok = (n < a.N) && (TY < Ht) && (TX < Wt);
    const int y0 = 2 * TY - 1, x0 = 2 * TX - 1;
    const int base = nl * a.Cin * HW + y0 * a.W + x0;
    okmask = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool v = ok && (y0 + r >= 0) && (y0 + r < a.H) && (x0 + j >= 0) && (x0 + j < a.W);
        voff[r * 4 + j] = v ? (unsigned)(base + r * a.W + j) * 4u : 0u;
        okmask |= v ? (1u << (r * 4 + j)) : 0u;
      }
    img0 = (size_t)bn * a.TBN * a.Cin * HW;
  };

  f32x4 acc[16][NIW];
#pragma unroll
  for (int c = 0; c < 16; ++c)
#pragma unroll
    for (int ni = 0; ni < NIW; ++ni) acc[c][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  float rin[NITEM][16];
  f32x4 rw[NU4];

  // branch-free: masked positions read element 0 of the channel plane and are zeroed on the way to LDS
  auto load_in = [&](int ch) {
#pragma unroll
    for (int k = 0; k < NITEM; ++k) {
      int c = ch * WCC + wave + k * NWAVE;
      c = c < a.Cin ? c : a.Cin - 1;
      const char* xc = reinterpret_cast<const char*>(a.x + img0 + (size_t)c * HW);
#pragma unroll
      for (int e = 0; e < 16; ++e) rin[k][e] = *reinterpret_cast<const float*>(xc + voff[e]);
    }
  };
  auto load_u = [&](int ch) {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.up + ((size_t)ch * a.NT + ct0) * 2048);
#pragma unroll
    for (int j = 0; j < NU4; ++j) rw[j] = src[tid + NTHR * j];
  };
  auto store_in = [&](int ch, float* Vs) {
#pragma unroll
    for (int k = 0; k < NITEM; ++k) {
      const int cl = wave + k * NWAVE;
      const bool cok = ch * WCC + cl < a.Cin;
      float d[16], t[16], v[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) d[e] = (cok && ((okmask >> e) & 1u)) ? rin[k][e] : 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {  // B^T d  (rows)
        t[j] = d[j] - d[8 + j];
        t[4 + j] = d[4 + j] + d[8 + j];
        t[8 + j] = d[8 + j] - d[4 + j];
        t[12 + j] = d[4 + j] - d[12 + j];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {  // (B^T d) B  (columns)
        v[4 * i] = t[4 * i] - t[4 * i + 2];
        v[4 * i + 1] = t[4 * i + 1] + t[4 * i + 2];
        v[4 * i + 2] = t[4 * i + 2] - t[4 * i + 1];
        v[4 * i + 3] = t[4 * i + 1] - t[4 * i + 3];
      }
      float* dst = Vs + (((lane >> 4) * 64 + (cl >> 1) * 16 + (lane & 15)) * 2 + (cl & 1));
#pragma unroll
      for (int c = 0; c < 16; ++c) dst[c * (WT * 128)] = v[c];
    }
  };
  auto store_u = [&](float* Us) {
    f32x4* dstw = reinterpret_cast<f32x4*>(Us);
#pragma unroll
    for (int j = 0; j < NU4; ++j) dstw[tid + NTHR * j] = rw[j];
  };
  auto compute_part = [&](const float* Vs, const float* Us, auto part_) {
    constexpr int PART = decltype(part_)::value;
    const float* va = Vs + wt * 128 + lane * 2;
    const float* ub = Us + (wc * NIW) * 2048 + lane * 2;
#pragma unroll
    for (int c = PART * 4; c < PART * 4 + 4; ++c) {
      const float2 av = *reinterpret_cast<const float2*>(va + c * (WT * 128));
      float2 bv[NIW];
#pragma unroll
      for (int ni = 0; ni < NIW; ++ni) bv[ni] = *reinterpret_cast<const float2*>(ub + (ni * 16 + c) * 128);
#pragma unroll
      for (int ni = 0; ni < NIW; ++ni) acc[c][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv[ni].x, acc[c][ni], 0, 0, 0);
#pragma unroll
      for (int ni = 0; ni < NIW; ++ni) acc[c][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv[ni].y, acc[c][ni], 0, 0, 0);
    }
  };

  // ---------------------------------------------------------------- epilogue: A^T M A, then the fused point-wise tail
  auto epilogue = [&](int eblk) {
  const int bx = eblk % a.blocks_x;
  const int et2 = eblk / a.blocks_x;
  const int by = et2 % a.blocks_y;
  const int bn = et2 / a.blocks_y;
  const bool lrelu = (a.flags & MG_CONV_LRELU) != 0;
  const bool mask_aux = (a.flags & MG_CONV_MASK_AUX) != 0;
  const bool pixnorm = (a.flags & MG_CONV_PIXNORM) != 0;
  const bool pool = (a.flags & MG_CONV_POOL_OUT) != 0;

  float o[NIW][4][4];  // [ni][g = tile 4*rq+g of the wave][2*i + j = pixel (i, j) of the tile]
#pragma unroll
  for (int ni = 0; ni < NIW; ++ni) {
    const int oc = (ct0 + wc * NIW + ni) * 16 + col;
    const float bvv = (a.bias != nullptr && oc < a.Cout) ? a.bias[oc] : 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float s0[4], s1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float m0 = acc[j][ni][g], m1 = acc[4 + j][ni][g], m2 = acc[8 + j][ni][g], m3 = acc[12 + j][ni][g];
        s0[j] = (m0 + m1) + m2;
        s1[j] = (m1 - m2) - m3;
      }
      float r4[4];
      r4[0] = (s0[0] + s0[1]) + s0[2];
      r4[1] = (s0[1] - s0[2]) - s0[3];
      r4[2] = (s1[0] + s1[1]) + s1[2];
      r4[3] = (s1[1] - s1[2]) - s1[3];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v = r4[q] + bvv;
        if (lrelu) v = mg_lrelu(v, a.slope);
        o[ni][g][q] = v;
      }
    }
  }

  float rnv[4][4];
  if (pixnorm) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float t = 0.f;
#pragma unroll
        for (int ni = 0; ni < NIW; ++ni) t += o[ni][g][q] * o[ni][g][q];
        t += __shfl_xor(t, 1);
        t += __shfl_xor(t, 2);
        t += __shfl_xor(t, 4);
        t += __shfl_xor(t, 8);
        rnv[g][q] = t;
        if (WC > 1 && col == 0) red[((wc * 64 + wt * 16 + rq * 4 + g) * 4) + q] = t;
      }
    if (WC > 1) {
      __syncthreads();
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float t = 0.f;
#pragma unroll
          for (int w2 = 0; w2 < WC; ++w2) t += red[((w2 * 64 + wt * 16 + rq * 4 + g) * 4) + q];
          rnv[g][q] = t;
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int q = 0; q < 4; ++q) rnv[g][q] = 1.0f / sqrtf(rnv[g][q] / (float)a.Cout + PN_EPS);
  }

  const int Hp = Ht, Wp = Wt;
  const bool vec = (a.TBW >= 4) && ((a.W & 7) == 0);
  if (vec) {
    // the lane's 4 tiles are consecutive in x: 8 output pixels per row = two 16-byte stores
    const int tl = wt * 16 + rq * 4;
    const int txl = tl & (a.TBW - 1);
    const int tyl = (tl >> a.lgTBW) & (a.TBH - 1);
    const int nl = tl >> (a.lgTBW + a.lgTBH);
    const int n = bn * a.TBN + nl, TY = by * a.TBH + tyl, TX = bx * a.TBW + txl;
    if ((n < a.N) && (TY < Ht) && (TX < Wt)) {
#pragma unroll
      for (int ni = 0; ni < NIW; ++ni) {
        const int oc = (ct0 + wc * NIW + ni) * 16 + col;
        if (oc < a.Cout) {
          const size_t plane = ((size_t)n * a.Cout + oc);
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const size_t idx = (plane * a.H + 2 * TY + i) * a.W + 2 * TX;
            f32x4 v0 = f32x4{o[ni][0][2 * i], o[ni][0][2 * i + 1], o[ni][1][2 * i], o[ni][1][2 * i + 1]};
            f32x4 v1 = f32x4{o[ni][2][2 * i], o[ni][2][2 * i + 1], o[ni][3][2 * i], o[ni][3][2 * i + 1]};
            if (mask_aux) {
              const f32x4 a0 = *reinterpret_cast<const f32x4*>(a.aux + idx);
              const f32x4 a1 = *reinterpret_cast<const f32x4*>(a.aux + idx + 4);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v0[e] *= mg_lrelu_mask(a0[e], a.slope);
                v1[e] *= mg_lrelu_mask(a1[e], a.slope);
              }
              o[ni][0][2 * i] = v0[0]; o[ni][0][2 * i + 1] = v0[1]; o[ni][1][2 * i] = v0[2]; o[ni][1][2 * i + 1] = v0[3];
              o[ni][2][2 * i] = v1[0]; o[ni][2][2 * i + 1] = v1[1]; o[ni][3][2 * i] = v1[2]; o[ni][3][2 * i + 1] = v1[3];
            }
            if (a.y != nullptr) {
              *reinterpret_cast<f32x4*>(a.y + idx) = v0;
              *reinterpret_cast<f32x4*>(a.y + idx + 4) = v1;
            }
            if (pixnorm) {
              const f32x4 r0 = f32x4{rnv[0][2 * i], rnv[0][2 * i + 1], rnv[1][2 * i], rnv[1][2 * i + 1]};
              const f32x4 r1 = f32x4{rnv[2][2 * i], rnv[2][2 * i + 1], rnv[3][2 * i], rnv[3][2 * i + 1]};
              *reinterpret_cast<f32x4*>(a.p + idx) = v0 * r0;
              *reinterpret_cast<f32x4*>(a.p + idx + 4) = v1 * r1;
            }
          }
          if (pool) {
            f32x4 pv;
#pragma unroll
            for (int g = 0; g < 4; ++g) pv[g] = ((o[ni][g][0] + o[ni][g][1]) + (o[ni][g][2] + o[ni][g][3])) * 0.25f;
            *reinterpret_cast<f32x4*>(a.p + (plane * Hp + TY) * Wp + TX) = pv;
          }
        }
      }
      if (pixnorm && col == 0 && wc == 0 && a.rn != nullptr && blockIdx.y == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const size_t idx = ((size_t)n * a.H + 2 * TY + i) * a.W + 2 * TX;
          *reinterpret_cast<f32x4*>(a.rn + idx) = f32x4{rnv[0][2 * i], rnv[0][2 * i + 1], rnv[1][2 * i], rnv[1][2 * i + 1]};
          *reinterpret_cast<f32x4*>(a.rn + idx + 4) = f32x4{rnv[2][2 * i], rnv[2][2 * i + 1], rnv[3][2 * i], rnv[3][2 * i + 1]};
        }
      }
    }
  } else {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int tl = wt * 16 + rq * 4 + g;
      const int txl = tl & (a.TBW - 1);
      const int tyl = (tl >> a.lgTBW) & (a.TBH - 1);
      const int nl = tl >> (a.lgTBW + a.lgTBH);
      const int n = bn * a.TBN + nl, TY = by * a.TBH + tyl, TX = bx * a.TBW + txl;
      if ((n < a.N) && (TY < Ht) && (TX < Wt)) {
#pragma unroll
        for (int ni = 0; ni < NIW; ++ni) {
          const int oc = (ct0 + wc * NIW + ni) * 16 + col;
          if (oc < a.Cout) {
            const size_t plane = ((size_t)n * a.Cout + oc);
            float ps = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const size_t idx = (plane * a.H + 2 * TY + (q >> 1)) * a.W + 2 * TX + (q & 1);
              float v = o[ni][g][q];
              if (mask_aux) v *= mg_lrelu_mask(a.aux[idx], a.slope);
              ps += v;
              if (a.y != nullptr) a.y[idx] = v;
              if (pixnorm) a.p[idx] = v * rnv[g][q];
            }
            if (pool) a.p[(plane * Hp + TY) * Wp + TX] = ps * 0.25f;
          }
        }
        if (pixnorm && col == 0 && wc == 0 && a.rn != nullptr && blockIdx.y == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) a.rn[((size_t)n * a.H + 2 * TY + (q >> 1)) * a.W + 2 * TX + (q & 1)] = rnv[g][q];
        }
      }
    }
  }
  };

  if (Q <= 0) return;
  // Pipeline over steps q = (block, chunk): in iteration q the MFMAs of step q run from LDS stage q&1 while step q+1 goes
  // registers -> LDS stage (q+1)&1 and the global loads of step q+2 are issued.  Everything inside the iteration is
  // branch-free (steps past the end re-load valid data and write a stage nobody reads) so that the staging instructions
  // interleave with the matrix instructions; the geometry of a new block is computed in the tail of the iteration before.
  const int gstride = (int)gridDim.x;
  auto advance = [&](int& b, int& c) {
    if (++c == a.nchunk) { c = 0; b += gstride; }
  };
  int blk = first, ch = 0;  // step q
  int blk1 = blk, ch1 = ch;
  advance(blk1, ch1);       // step q+1
  int blk2 = blk1, ch2 = ch1;
  advance(blk2, ch2);       // step q+2
  unsigned okmask_st;
  {
    geometry(first);
    load_in(0);
    load_u(0);
    store_in(0, smem);
    store_u(smem + V_FLOATS);
    const bool has1 = Q > 1;
    if (has1 && ch1 == 0) geometry(blk1);
    load_in(has1 ? ch1 : 0);
    load_u(has1 ? ch1 : 0);
    okmask_st = okmask;
    if (Q > 2 && ch2 == 0) geometry(blk2);
  }
  __syncthreads();

  for (int q = 0; q < Q; ++q) {
    float* cur = smem + (q & 1) * STAGE;
    float* nxt = smem + ((q + 1) & 1) * STAGE;
    const int lch2 = (q + 2 < Q) ? ch2 : 0;  // past the end: any valid chunk of the current geometry

#ifndef WINO_EXP_NOMFMA
    compute_part(cur, cur + V_FLOATS, std::integral_constant<int, 0>{});
#endif
#ifndef WINO_EXP_NOSTAGE
    {
      const unsigned keep = okmask;
      okmask = okmask_st;
      store_in(ch1, nxt);
      okmask = keep;
    }
    load_in(lch2);  // re-uses the registers just drained: almost a full iteration of latency cover
    okmask_st = okmask;
#endif
#ifndef WINO_EXP_NOMFMA
    compute_part(cur, cur + V_FLOATS, std::integral_constant<int, 1>{});
#endif
#ifndef WINO_EXP_NOSTAGE
    store_u(nxt + V_FLOATS);
    load_u(lch2);
#endif
#ifndef WINO_EXP_NOMFMA
    compute_part(cur, cur + V_FLOATS, std::integral_constant<int, 2>{});
    compute_part(cur, cur + V_FLOATS, std::integral_constant<int, 3>{});
#endif

    if (ch + 1 == a.nchunk) {
#ifndef WINO_EXP_NOEPI
      epilogue(blk);
#else
      if (a.N < 0) {  // keeps the accumulators alive
        f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 16; ++c)
#pragma unroll
          for (int ni = 0; ni < NIW; ++ni) t += acc[c][ni];
        a.y[tid] = (t[0] + t[1]) + (t[2] + t[3]);
      }
#endif
#pragma unroll
      for (int c = 0; c < 16; ++c)
#pragma unroll
        for (int ni = 0; ni < NIW; ++ni) acc[c][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    blk = blk1; ch = ch1;
    blk1 = blk2; ch1 = ch2;
    advance(blk2, ch2);
    if (ch2 == 0 && q + 3 < Q) geometry(blk2);  // step q+3 opens a new block: its loads are issued in the next iteration
#ifndef WINO_EXP_NOBARRIER
    __syncthreads();
#endif
  }
}

// U = G g G^T for every (out, in) channel pair, written in MFMA operand order:
//   up[((ch*NT + ct)*16 + comp)*128 + lane*2 + ks]  with  in-channel = ch*8 + 2*(lane>>4) + ks,  out-channel = ct*16 + (lane&15)
__global__ void wino3x3_pack_kernel(const float* __restrict__ w, float* __restrict__ up, int Co, int Ci, int dgrad,
                                    int cin_call, int cout_call, int NT, size_t total) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int ks = (int)(e & 1);
  const int lane = (int)((e >> 1) & 63);
  const size_t r = e >> 7;
  const int ct = (int)(r % NT);
  const int ch = (int)(r / NT);
  const int c = ch * WCC + 2 * (lane >> 4) + ks;
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
  float* dst = up + (((size_t)ch * NT + ct) * 16) * 128 + lane * 2 + ks;
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // (G g) G^T : 4x4
    const float h0 = h[3 * i], h1 = h[3 * i + 1], h2 = h[3 * i + 2];
    dst[(size_t)(4 * i + 0) * 128] = h0;
    dst[(size_t)(4 * i + 1) * 128] = 0.5f * ((h0 + h1) + h2);
    dst[(size_t)(4 * i + 2) * 128] = 0.5f * ((h0 - h1) + h2);
    dst[(size_t)(4 * i + 3) * 128] = h2;
  }
}

template <int NIW, int WC>
int launch_wino(const WinoArgs& a, dim3 grid, hipStream_t s) {
  constexpr size_t lds = (size_t)(2 * (16 * WT * 128 + WC * NIW * 2048) + WC * 256) * sizeof(float);
  static bool attr_set = false;  // benign race: idempotent
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino3x3_mfma<NIW, WC>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((wino3x3_mfma<NIW, WC>), grid, dim3(256 * WC), lds, s, a);
  MG_CHECK_LAUNCH("mg_wino3x3");
  return MG_OK;
}

int wino_nt_padded(int Cout) {
  const int nt = mg_cdiv(Cout, 16);
  const int p4 = mg_cdiv(nt, 4) * 4, p3 = mg_cdiv(nt, 3) * 3;
  return p4 > p3 ? p4 : p3;
}

}  // namespace

extern "C" size_t mg_wino3x3_packed_floats(int Cin, int Cout) {
  return (size_t)mg_cdiv(Cin, WCC) * wino_nt_padded(Cout) * 2048;
}

extern "C" int mg_wino3x3_pack(const float* w, float* up, int Co, int Ci, int dgrad, mg_stream_t stream) {
  MG_CHECK_ARG(w && up && Co > 0 && Ci > 0, "mg_wino3x3_pack: bad arguments");
  const int cin_call = dgrad ? Co : Ci, cout_call = dgrad ? Ci : Co;
  const int NT = wino_nt_padded(cout_call);
  const size_t total = (size_t)mg_cdiv(cin_call, WCC) * NT * 128;  // one thread per (chunk, tile, lane, k-step)
  hipLaunchKernelGGL(wino3x3_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, up, Co,
                     Ci, dgrad, cin_call, cout_call, NT, total);
  MG_CHECK_LAUNCH("mg_wino3x3_pack");
  return MG_OK;
}

extern "C" int mg_wino3x3(const float* x, const float* up, const float* bias, const float* aux, float* y, float* p, float* rn,
                          int N, int Cin, int Cout, int H, int W, int flags, float slope, mg_stream_t stream) {
  MG_CHECK_ARG(x && up && N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "mg_wino3x3: bad arguments");
  MG_CHECK_ARG((H % 2 == 0) && (W % 2 == 0), "mg_wino3x3: H=%d W=%d must be even", H, W);
  MG_CHECK_ARG(Cout <= 160, "mg_wino3x3: Cout=%d > 160 unsupported", Cout);
  const bool pn = flags & MG_CONV_PIXNORM;
  MG_CHECK_ARG(!(flags & MG_CONV_UPS_IN), "mg_wino3x3: UPS_IN unsupported (use mg_upconv3x3)");
  MG_CHECK_ARG(!(flags & MG_CONV_MASK_AUX) || aux, "mg_wino3x3: MASK_AUX without aux");
  MG_CHECK_ARG(!pn || ((flags & MG_CONV_LRELU) && p), "mg_wino3x3: PIXNORM needs LRELU and p");
  MG_CHECK_ARG(pn || y, "mg_wino3x3: y is NULL");
  MG_CHECK_ARG(!(flags & MG_CONV_POOL_OUT) || (!pn && p), "mg_wino3x3: POOL_OUT needs p and no PIXNORM");
  MG_CHECK_ARG(!((flags & MG_CONV_MASK_AUX) && (flags & (MG_CONV_LRELU | MG_CONV_PIXNORM))),
               "mg_wino3x3: MASK_AUX excludes LRELU/PIXNORM");
  MG_CHECK_ARG(!(flags & ~(MG_CONV_LRELU | MG_CONV_PIXNORM | MG_CONV_MASK_AUX | MG_CONV_POOL_OUT)), "mg_wino3x3: unknown flag");
  MG_CHECK_ARG((long long)N * Cin * H * W < (1ll << 40) && (long long)N * Cout * H * W < (1ll << 40), "mg_wino3x3: tensor too large");
  const int nt = mg_cdiv(Cout, 16);
  MG_CHECK_ARG(!pn || nt <= 4, "mg_wino3x3: PIXNORM needs Cout <= 64 (all channels of a pixel in one workgroup)");

  WinoArgs a;
  a.x = x; a.up = up; a.bias = bias; a.aux = aux; a.y = y; a.p = p; a.rn = rn;
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  a.flags = flags; a.slope = slope;
  a.nchunk = mg_cdiv(Cin, WCC);
  a.NT = wino_nt_padded(Cout);
  const int Ht = H / 2, Wt = W / 2;
  a.TBW = mg_pow2_ceil(Wt) < 8 ? mg_pow2_ceil(Wt) : 8;
  a.TBH = mg_pow2_ceil(Ht) < 64 / a.TBW ? mg_pow2_ceil(Ht) : 64 / a.TBW;
  a.TBN = 64 / (a.TBW * a.TBH);
  a.lgTBW = mg_ilog2(a.TBW); a.lgTBH = mg_ilog2(a.TBH);
  a.blocks_x = mg_cdiv(Wt, a.TBW); a.blocks_y = mg_cdiv(Ht, a.TBH); a.blocks_n = mg_cdiv(N, a.TBN);

  // out-channel tiling: 4 tiles (2 per wave x 2 wave groups), 3 (3 per wave) or 2 per workgroup -- least padding wins
  int cfg = 4, best = mg_cdiv(nt, 4) * 4;
  if (mg_cdiv(nt, 3) * 3 < best) { cfg = 3; best = mg_cdiv(nt, 3) * 3; }
  if (mg_cdiv(nt, 2) * 2 < best) { cfg = 2; best = mg_cdiv(nt, 2) * 2; }
  if (pn) cfg = nt <= 2 ? 2 : (nt == 3 ? 3 : 4);
  {
    const char* e = getenv("MG_WINO_CFG");  // measurement override: 2, 3 or 4 out-channel tiles per workgroup
    if (e != nullptr && !pn) {
      const int v = atoi(e);
      if (v == 2 || v == 4 || (v == 3 && mg_cdiv(nt, 3) * 3 <= a.NT)) cfg = v;
    }
  }
  MG_CHECK_ARG(mg_cdiv(nt, cfg) * cfg <= a.NT, "mg_wino3x3: internal tile error");
  MG_CHECK_ARG((long long)a.TBN * Cin * H * W < (1ll << 29), "mg_wino3x3: image block too large for 32-bit offsets");
  // persistent workgroups (one per CU: 128 KB of LDS each), each walking the spatial blocks b, b + grid.x, ... so that the
  // loads of the next block and the stores of the previous one overlap the matrix work of the current one
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    n_cu = v;
  }
  const int nblk = a.blocks_x * a.blocks_y * a.blocks_n, gy = mg_cdiv(nt, cfg);
  int gx = n_cu / gy > 0 ? n_cu / gy : 1;
  if (gx > nblk) gx = nblk;
  gx = mg_cdiv(nblk, mg_cdiv(nblk, gx));  // same number of rounds, evenly spread
  dim3 grid(gx, gy);
  hipStream_t s = (hipStream_t)stream;
  switch (cfg) {
    case 4: return launch_wino<2, 2>(a, grid, s);
    case 3: return launch_wino<3, 1>(a, grid, s);
    default: return launch_wino<2, 1>(a, grid, s);
  }
}
