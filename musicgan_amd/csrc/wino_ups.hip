// nn.Upsample(x2, nearest) -> nn.Conv2d(3x3, pad 1) (+ LeakyReLU + PixelNorm) of the generator blocks
// (/root/reference/music_gan/networks/generator.py:24-39) and its data gradient, in Winograd F(2x2,3x3) form ON THE UP-SAMPLED GRID
// WITHOUT THE COMPONENTS THAT VANISH THERE (round 6).
//
// The 4x4 input patch of output tile (TY, TX) of an up-sampled image is the 3x3 low-res neighbourhood with its centre row and column
// doubled, d = [l(TY-1); l(TY); l(TY); l(TY+1)] (same in x).  Its Winograd transform V = B^T d B has rows
//     0: l(TY-1) - l(TY)      1: 2 l(TY)      2: l(TY) - l(TY) = 0      3: l(TY) - l(TY+1)
// and the same columns: component row 2 and column 2 are identically zero -- 9 of the 16 element-wise products remain, 0.25 of the
// direct convolution's multiplies (the sub-pixel form of upconv3x3.hip: 16 of 36 = 0.44; plain Winograd on a materialised up-sampled
// tensor: 16 of 36 = 0.44 too).  The data gradient mirrors it: the gradient w.r.t. the low-res input is the 2x2 block sum of the
// hi-res data gradient, sum_ij (A^T M A)_ij = s^T M s with s = A 1 = (1, 2, 0, -1): again row / column 2 drop out.  The factors
// (2 for index 1, -1 for index 3 of s) are folded into the packed filters (MG_PACK_WINOUPS).
//
// Structure (wino_strip.hip's): persistent workgroups, the transformed 9-component filter bank of ALL out-channels resident in LDS,
// one wave per block of 16 horizontally adjacent tiles, lane (rq, col) = tile col x channels 8 ch + 2 rq + {0, 1}; operands built in
// registers (halo pixels from the neighbouring lanes by DPP, the block's edge pixel fetched by lanes 0 / 15 only), the two channels of
// a lane packed into one register pair so that every transform step is one packed instruction; no barrier after the start.
//   forward: 3 row loads of 4 bytes per channel, 12 packed adds + 12 DPP moves per 8-channel chunk, 18 NT MFMAs (NT out tiles);
//   all out-channels of a pixel sit in one wave: bias + LeakyReLU + PixelNorm in the epilogue.
//   data gradient: the hi-res patch rows as in wino_strip.hip (8-byte own pair + 4-byte edge pixel), 9 of the 16 components of the
//   input transform, 18 NT MFMAs per chunk, epilogue = sum of the nine accumulators (one low-res pixel per tile and channel).
// Filter reads are inline-assembly ds_read_b64, one component ahead of the MFMAs (wu_mma).
#include <cstdlib>
#include <utility>

#include "mg_common.h"
#include "pack_kernels.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr float WU_PN_EPS = 1e-8f;
constexpr int WU_NWAVE = 8;

struct WuArgs {
  const float* x;     // forward: (N, Cin, Hl, Wl) low-res;  dgrad: gy (N, K, 2 Hl, 2 Wl)
  const float* up;    // MG_PACK_WINOUPS bank [chunk][tile][9][64][2]
  const float* bias;  // forward
  float* y;           // forward: (N, Cout, 2 Hl, 2 Wl) activation (may be NULL with PixelNorm);  dgrad: gx (N, Cout, Hl, Wl)
  float* p;           // forward + PixelNorm: normalised activation
  float* rn;          // forward + PixelNorm: (N, 1, 2 Hl, 2 Wl) 1 / norm
  const float *hw, *hb;  // forward + PixelNorm + HEAD: the 1x1 head on the normalised activation, weights (2, Cout), bias (2) or NULL
  float* mp;             //   its output tanh(hw p + hb), (N, 2, 2 Hl, 2 Wl)
  int N, K, Cout, Hl, Wl;  // K = channels of the kernel's input tensor
  int flags;
  float slope;
  int nchunk, blocks_x, G;  // K / 8; Wl / 16; workgroups
  int tile0, nt_total;      // data gradient: this launch computes out-channel tiles tile0 .. tile0 + NT - 1 of nt_total (Cout = 16 nt_total)
};

__device__ __forceinline__ f32x2 wu_sub(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ int wu_f2i(float v) { return __builtin_bit_cast(int, v); }
__device__ __forceinline__ float wu_i2f(int v) { return __builtin_bit_cast(float, v); }
// lane - 1's / lane + 1's value inside a row of 16 lanes; at the row's ends the destination keeps `old`
__device__ __forceinline__ float wu_from_left(float old, float v) { return wu_i2f(__builtin_amdgcn_update_dpp(wu_f2i(old), wu_f2i(v), 0x111, 0xf, 0xf, false)); }
__device__ __forceinline__ float wu_from_right(float old, float v) { return wu_i2f(__builtin_amdgcn_update_dpp(wu_f2i(old), wu_f2i(v), 0x101, 0xf, 0xf, false)); }

// The nine products of one chunk: acc[c][t] += U[c](filters of tile t) x V[c], both k-steps (the lane's two channels)
// (FIRST: a block's first chunk starts its sums from the zero constant -- no 36 NT register clears per block)
template <class F, int... Is>
__device__ __forceinline__ void wu_static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void wu_static_for(F&& f) {
  wu_static_for_impl(f, std::make_integer_sequence<int, N>{});
}
// 8 bytes of LDS at byte address addr + OFF as ONE ds_read_b64, outside hipcc's s_waitcnt bookkeeping (wu_wait before the first use):
// left to the compiler, the filter reads either merge into half-rate ds_read2st64_b64 or -- as volatile loads -- sink back to their
// uses, where every pair of MFMAs waits out an LDS round trip (first version: 165 us for 64 -> 48 @128x128 x 64)
template <int OFF>
__device__ __forceinline__ f32x2 wu_lds64(unsigned addr) {
  f32x2 v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ void wu_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wu_tie(f32x2& v) { asm volatile("" : "+v"(v)); }

// The nine products of one chunk: acc[c][t] += U[c](filters of tile t) x V[c], both k-steps (the lane's two channels).  The filters of
// component c + 1 are requested before the MFMAs of component c are issued (two register sets).
template <int NT, bool FIRST>
__device__ __forceinline__ void wu_mma(unsigned bank_addr, const f32x2 (&V)[9], f32x4 (&acc)[9][NT]) {
  f32x2 fa[2][NT];
  wu_static_for<NT>([&](auto tc) __attribute__((always_inline)) {
    constexpr int t = decltype(tc)::value;
    fa[0][t] = wu_lds64<(t * 9) * 512>(bank_addr);
  });
  wu_static_for<9>([&](auto cc) __attribute__((always_inline)) {
    constexpr int c = decltype(cc)::value;
    wu_wait();
#pragma unroll
    for (int t = 0; t < NT; ++t) wu_tie(fa[c & 1][t]);
    if constexpr (c + 1 < 9) {
      wu_static_for<NT>([&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value;
        fa[(c + 1) & 1][t] = wu_lds64<(t * 9 + c + 1) * 512>(bank_addr);
      });
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const f32x2 a = fa[c & 1][t];
      const f32x4 z = FIRST ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[c][t];
      acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], V[c][0], z, 0, 0, 0);
      acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], V[c][1], acc[c][t], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  });
}

// Work items: groups of 8 vertically adjacent tile blocks, item = (image * groups_y + gy) * blocks_x + bx, walked with stride G.
struct WuWalk {
  int item, bx, gy, n, dbx, dgy, dn, groups_y, blocks_x, G, nitems;
  __device__ __forceinline__ void init(int wg, int G_, int N, int groups_y_, int blocks_x_) {
    G = G_; groups_y = groups_y_; blocks_x = blocks_x_;
    nitems = N * groups_y * blocks_x;
    item = mg_xcd_remap(wg, G);
    bx = item % blocks_x; gy = (item / blocks_x) % groups_y; n = item / (blocks_x * groups_y);
    dbx = G % blocks_x; dgy = (G / blocks_x) % groups_y; dn = G / (blocks_x * groups_y);
  }
  __device__ __forceinline__ void step() {
    item += G;
    bx += dbx;
    const int c1 = bx >= blocks_x ? 1 : 0;
    bx -= c1 * blocks_x;
    gy += dgy + c1;
    const int c2 = gy >= groups_y ? 1 : 0;
    gy -= c2 * groups_y;
    n += dn + c2;
  }
};

// the same for a launch that owns NT of nt_total tiles: per chunk a run of NT x 9 x 128 floats out of nt_total x 9 x 128
__device__ __forceinline__ void wu_load_bank_tiles(float* Us, const float* up, int nchunk, int nt, int nt_total, int tile0, int tid) {
  const int per = nt * 9 * 32;  // 16-byte pieces per chunk
  f32x4* dst = reinterpret_cast<f32x4*>(Us);
  for (int i = tid; i < nchunk * per; i += 64 * WU_NWAVE) {
    const int ch = i / per, r = i - ch * per;
    dst[i] = reinterpret_cast<const f32x4*>(up)[(size_t)(ch * nt_total + tile0) * 9 * 32 + r];
  }
  __syncthreads();
}

__device__ __forceinline__ void wu_load_bank(float* Us, const float* up, int n16, int tid) {
  const f32x4* src = reinterpret_cast<const f32x4*>(up);
  f32x4* dst = reinterpret_cast<f32x4*>(Us);
  for (int i = tid; i < n16; i += 64 * WU_NWAVE) dst[i] = src[i];
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------ forward
template <int NT, bool PN, bool HEAD>
__global__ void __launch_bounds__(64 * WU_NWAVE, 1) winoups_fwd(const WuArgs a) {
  extern __shared__ __attribute__((aligned(16))) float Us[];
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 15, rq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int HWl = a.Hl * a.Wl, H = 2 * a.Hl, W = 2 * a.Wl, HW = H * W;
  wu_load_bank(Us, a.up, a.nchunk * NT * 9 * 32, tid);
  const unsigned us_addr = (unsigned)reinterpret_cast<size_t>(Us + lane * 2);  // (low 32 bits of a shared-aperture address = LDS offset)
  WuWalk w;
  w.init(blockIdx.x, a.G, a.N, a.Hl / WU_NWAVE, a.blocks_x);
  if (w.item >= w.nitems) return;

  // input: lane part = channel 2 rq of the chunk, own low-res pixel; scalar part = image, chunk, k-step, row, block
  const int lp = ((2 * rq) * HWl + col) * 4;
  const bool lrelu = (a.flags & MG_CONV_LRELU) != 0;
  const float slope_eff = lrelu ? a.slope : 1.0f;
  f32x2 rC[3];   // [row]: own pixel of the lane's two channels
  f32x2 rX[3];   // the block-edge pixel (left for lane 0, right for lane 15 of a row; 0 elsewhere and at the image edge)
  const float* img = nullptr;
  unsigned vC = 0, vX = 0;
  int rowoff[3], rowrec[3];
  auto geometry = [&](int ty) __attribute__((always_inline)) {
    img = a.x + (size_t)w.n * a.K * HWl;
    vC = (unsigned)(lp + w.bx * 64);
    vX = col == 0 ? (w.bx == 0 ? 0x80000000u : vC - 4u) : ((col == 15 && w.bx != a.blocks_x - 1) ? vC + 4u : 0x80000000u);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int Y = ty - 1 + r;
      rowoff[r] = Y * a.Wl * 4;
      rowrec[r] = (w.item < w.nitems && Y >= 0 && Y < a.Hl) ? a.K * HWl * 4 : 0;
    }
  };
  auto load_rows = [&](int ch) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(img), 0, rowrec[r], 0x00020000);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int so = (ch * 8 + ks) * HWl * 4 + rowoff[r];
        rC[r][ks] = wu_i2f(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)vC, so, 0));
        rX[r][ks] = wu_i2f(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)vX, so, 0));
      }
    }
  };
  f32x2 V[9];  // component (i, j), i, j in {0, 1, 3} -> index 3 i' + j'
  auto transform = [&]() __attribute__((always_inline)) {
    f32x2 L[3], R[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        L[r][ks] = wu_from_left(rX[r][ks], rC[r][ks]);
        R[r][ks] = wu_from_right(rX[r][ks], rC[r][ks]);
      }
    // rows: u0 = l0 - l1, u1 = l1 (x 2 in the filters), u3 = l1 - l2;  columns: v0 = uL - uC, v1 = uC (x 2), v3 = uC - uR
    const f32x2 uL[3] = {wu_sub(L[0], L[1]), L[1], wu_sub(L[1], L[2])};
    const f32x2 uC[3] = {wu_sub(rC[0], rC[1]), rC[1], wu_sub(rC[1], rC[2])};
    const f32x2 uR[3] = {wu_sub(R[0], R[1]), R[1], wu_sub(R[1], R[2])};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      V[3 * i + 0] = wu_sub(uL[i], uC[i]);
      V[3 * i + 1] = uC[i];
      V[3 * i + 2] = wu_sub(uC[i], uR[i]);
    }
    // (registers written by inline-assembly vector instructions and read as MFMA sources next: wait states, wino_strip.hip)
    asm volatile("s_nop 1" : "+v"(V[0]), "+v"(V[1]), "+v"(V[2]), "+v"(V[3]), "+v"(V[4]), "+v"(V[5]), "+v"(V[6]), "+v"(V[7]), "+v"(V[8]));
  };

  f32x4 acc[9][NT];
  const int ly = ((rq * 4) * HW + 2 * col) * 4;
  // HEAD: the 1x1 head's weights of this lane's channels (tile t, accumulator row g <-> channel 16 t + 4 rq + g)
  float hwv[HEAD ? 2 : 1][NT][4];
  float hbv[2] = {0.f, 0.f};
  if constexpr (HEAD) {
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      if (a.hb != nullptr) hbv[f] = a.hb[f];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) hwv[f][t][g] = a.hw[f * a.Cout + t * 16 + rq * 4 + g];
    }
  }
  int ty = w.gy * WU_NWAVE + wave;
  geometry(ty);
  load_rows(0);
#pragma nounroll
  for (;;) {
    const int ebx = w.bx, ety = ty, en = w.n;
    // (at least two chunks: wu_shape_ok)
    transform();
    load_rows(1);
    wu_mma<NT, true>(us_addr, V, acc);
    for (int ch = 1; ch + 1 < a.nchunk; ++ch) {
      transform();
      load_rows(ch + 1);
      wu_mma<NT, false>(us_addr + (unsigned)ch * (NT * 9 * 512), V, acc);
    }
    transform();
    // the first chunk of this wave's next block rides under the last chunk's MFMAs and the epilogue
    w.step();
    ty = w.gy * WU_NWAVE + wave;
    geometry(ty);
    load_rows(0);
    wu_mma<NT, false>(us_addr + (unsigned)(a.nchunk - 1) * (NT * 9 * 512), V, acc);
    // ---- epilogue: 2x2 outputs of the lane's tile from the nine components (A^T M A without row / column 2):
    //   y00 = M00 + M01 + M10 + M11   y01 = M01 - M03 + M11 - M13   y10 = M10 + M11 - M30 - M31   y11 = M11 - M13 - M31 + M33
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y ? a.y + (size_t)en * a.Cout * HW : nullptr, 0, a.y ? a.Cout * HW * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(PN ? a.p + (size_t)en * a.Cout * HW : nullptr, 0, PN ? a.Cout * HW * 4 : 0, 0x00020000);
    const int sy = ((2 * ety) * W + 32 * ebx) * 4;
    f32x4 o[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const f32x4 m00 = acc[0][t], m01 = acc[1][t], m03 = acc[2][t], m10 = acc[3][t], m11 = acc[4][t], m13 = acc[5][t], m30 = acc[6][t],
                  m31 = acc[7][t], m33 = acc[8][t];
      f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
      if (a.bias != nullptr) b4 = *reinterpret_cast<const f32x4*>(a.bias + t * 16 + rq * 4);
      const f32x4 s01 = m01 + m11, s13 = m03 + m13, s10 = m10 + m11, s31 = m30 + m31;
      o[t][0] = ((m00 + m10) + s01) + b4;
      o[t][1] = (s01 - s13) + b4;
      o[t][2] = (s10 - s31) + b4;
      o[t][3] = (((m11 - m13) - m31) + m33) + b4;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 ws = o[t][q] * slope_eff;
#pragma unroll
        for (int g = 0; g < 4; ++g) o[t][q][g] = fmaxf(o[t][q][g], ws[g]);
      }
    }
    if (a.y != nullptr) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int so = (t * 16 + g) * HW * 4 + sy;
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{o[t][0][g], o[t][1][g]}), ry, ly, so, 0);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{o[t][2][g], o[t][3][g]}), ry, ly, so + W * 4, 0);
        }
    }
    if constexpr (PN) {
      float rnv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          float tn = 0.f;
#pragma unroll
          for (int g = 0; g < 4; ++g) tn += o[t][q][g] * o[t][q][g];
          tn += __shfl_xor(tn, 16);
          tn += __shfl_xor(tn, 32);
          s += tn;
        }
        rnv[q] = 1.0f / sqrtf(s / (float)a.Cout + WU_PN_EPS);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int so = (t * 16 + g) * HW * 4 + sy;
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{o[t][0][g] * rnv[0], o[t][1][g] * rnv[1]}), rp, ly, so, 0);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{o[t][2][g] * rnv[2], o[t][3][g] * rnv[3]}), rp, ly, so + W * 4, 0);
        }
      if (rq == 0 && a.rn != nullptr) {
        float* rn = a.rn + ((size_t)en * H + 2 * ety) * W + 32 * ebx + 2 * col;
        *reinterpret_cast<float2*>(rn) = make_float2(rnv[0], rnv[1]);
        *reinterpret_cast<float2*>(rn + W) = make_float2(rnv[2], rnv[3]);
      }
      if constexpr (HEAD) {
        // tanh(conv1x1(p)): the lane's channels, then the four channel groups of a pixel (lanes 16 apart) -- the sum PixelNorm took
        float hs[2][4];
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float v = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
              for (int g = 0; g < 4; ++g) v = fmaf(hwv[f][t][g], o[t][q][g] * rnv[q], v);
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            hs[f][q] = v;
          }
        if (rq == 0) {
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            float* m = a.mp + (((size_t)en * 2 + f) * H + 2 * ety) * W + 32 * ebx + 2 * col;
            *reinterpret_cast<float2*>(m) = make_float2(tanhf(hs[f][0] + hbv[f]), tanhf(hs[f][1] + hbv[f]));
            *reinterpret_cast<float2*>(m + W) = make_float2(tanhf(hs[f][2] + hbv[f]), tanhf(hs[f][3] + hbv[f]));
          }
        }
      }
    }
    if (w.item >= w.nitems) break;
  }
}

// ------------------------------------------------------------------------------------------------ data gradient
template <int NT, bool PNB>
__global__ void __launch_bounds__(64 * WU_NWAVE, 1) winoups_dgrad(const WuArgs a) {
  extern __shared__ __attribute__((aligned(16))) float Us[];
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 15, rq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int HWl = a.Hl * a.Wl, H = 2 * a.Hl, W = 2 * a.Wl, HW = H * W;
  if (a.nt_total == NT) wu_load_bank(Us, a.up, a.nchunk * NT * 9 * 32, tid);
  else wu_load_bank_tiles(Us, a.up, a.nchunk, NT, a.nt_total, a.tile0, tid);
  const unsigned us_addr = (unsigned)reinterpret_cast<size_t>(Us + lane * 2);  // (low 32 bits of a shared-aperture address = LDS offset)
  WuWalk w;
  w.init(blockIdx.x, a.G, a.N, a.Hl / WU_NWAVE, a.blocks_x);
  if (w.item >= w.nitems) return;

  const int lp = ((2 * rq) * HW + 2 * col) * 4;
  f32x2 rP[2][4];  // [k-step][patch row]: own pixel pair
  float rX[2][4];  // the block-edge halo pixel
  const float* img = nullptr;
  unsigned vP = 0, vX = 0;
  int rowoff[4], rowrec[4];
  auto geometry = [&](int ty) __attribute__((always_inline)) {
    img = a.x + (size_t)w.n * a.K * HW;
    vP = (unsigned)(lp + w.bx * 128);
    vX = col == 0 ? (w.bx == 0 ? 0x80000000u : vP - 4u) : ((col == 15 && w.bx != a.blocks_x - 1) ? vP + 8u : 0x80000000u);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int Y = 2 * ty - 1 + r;
      rowoff[r] = Y * W * 4;
      rowrec[r] = (w.item < w.nitems && Y >= 0 && Y < H) ? a.K * HW * 4 : 0;
    }
  };
  auto load_rows = [&](int ch) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(img), 0, rowrec[r], 0x00020000);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int so = (ch * 8 + ks) * HW * 4 + rowoff[r];
        rP[ks][r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)vP, so, 0));
        rX[ks][r] = wu_i2f(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)vX, so, 0));
      }
    }
  };
  f32x2 V[9];
  auto transform = [&]() __attribute__((always_inline)) {
    // per k-step: patch row = [e0, p0, p1, e1]; rows 0, 1, 3 of B^T d: d0 - d2, d1 + d2, d1 - d3; columns 0, 1, 3: u0 - u2, u1 + u2, u1 - u3
    float c[2][9];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f32x2 E[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) E[r] = f32x2{wu_from_left(rX[ks][r], rP[ks][r][1]), wu_from_right(rX[ks][r], rP[ks][r][0])};
      const f32x2 uE[3] = {wu_sub(E[0], E[2]), E[1] + E[2], wu_sub(E[1], E[3])};
      const f32x2 uP[3] = {wu_sub(rP[ks][0], rP[ks][2]), rP[ks][1] + rP[ks][2], wu_sub(rP[ks][1], rP[ks][3])};
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        c[ks][3 * i + 0] = uE[i][0] - uP[i][1];  // e0 - p1
        c[ks][3 * i + 1] = uP[i][0] + uP[i][1];  // p0 + p1
        c[ks][3 * i + 2] = uP[i][0] - uE[i][1];  // p0 - e1
      }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) V[k] = f32x2{c[0][k], c[1][k]};
  };

  f32x4 acc[9][NT];
  int ty = w.gy * WU_NWAVE + wave;
  geometry(ty);
  load_rows(0);
#pragma nounroll
  for (;;) {
    const int ebx = w.bx, ety = ty, en = w.n;
    // (at least two chunks: wu_shape_ok)
    transform();
    load_rows(1);
    wu_mma<NT, true>(us_addr, V, acc);
    for (int ch = 1; ch + 1 < a.nchunk; ++ch) {
      transform();
      load_rows(ch + 1);
      wu_mma<NT, false>(us_addr + (unsigned)ch * (NT * 9 * 512), V, acc);
    }
    transform();
    // the first chunk of this wave's next block rides under the last chunk's MFMAs and the epilogue
    w.step();
    ty = w.gy * WU_NWAVE + wave;
    geometry(ty);
    load_rows(0);
    wu_mma<NT, false>(us_addr + (unsigned)(a.nchunk - 1) * (NT * 9 * 512), V, acc);
    // gx(tile) = s^T M s: the factors of s are in the filters, so the low-res pixel is the plain sum of the nine accumulators
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)en * a.Cout * HWl, 0, a.Cout * HWl * 4, 0x00020000);
    const int lo = ((rq * 4) * HWl + col) * 4, so0 = (ety * a.Wl + 16 * ebx) * 4;
    if constexpr (!PNB) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const f32x4 s = (((acc[0][t] + acc[1][t]) + (acc[2][t] + acc[3][t])) + ((acc[4][t] + acc[5][t]) + (acc[6][t] + acc[7][t]))) + acc[8][t];
#pragma unroll
        for (int g = 0; g < 4; ++g)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)s[g]), ry, lo, ((a.tile0 + t) * 16 + g) * HWl * 4 + so0, 0);
      }
    } else {
      // PNB: the PixelNorm + LeakyReLU backward of the layer below (its normalised output a.p, 1/norm a.rn; elementwise.hip's from_p
      // form) on the way out: all channels of the low-res pixel are in this wave -- the lane's NT x 4, the other three row groups
      // 16 lanes apart -- so the gradient at the lower conv's pre-activation leaves instead of the gradient at its output
      const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.p) + (size_t)en * a.Cout * HWl, 0,
                                                                          a.Cout * HWl * 4, 0x00020000);
      f32x4 s[NT], pv[NT];
      float dot = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) pv[t][g] = wu_i2f(__builtin_amdgcn_raw_buffer_load_b32(rp, lo, (t * 16 + g) * HWl * 4 + so0, 0));
      const float r = a.rn[(size_t)en * HWl + (size_t)ety * a.Wl + 16 * ebx + col];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        s[t] = (((acc[0][t] + acc[1][t]) + (acc[2][t] + acc[3][t])) + ((acc[4][t] + acc[5][t]) + (acc[6][t] + acc[7][t]))) + acc[8][t];
#pragma unroll
        for (int g = 0; g < 4; ++g) dot = fmaf(s[t][g], pv[t][g], dot);
      }
      dot += __shfl_xor(dot, 16);
      dot += __shfl_xor(dot, 32);
      dot /= (float)a.Cout;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float o = mg_lrelu_mask(pv[t][g], a.slope) * r * (s[t][g] - pv[t][g] * dot);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), ry, lo, (t * 16 + g) * HWl * 4 + so0, 0);
        }
    }
    if (w.item >= w.nitems) break;
  }
}

template <int NT, bool PN, bool HEAD = false>
int wu_launch_fwd(const WuArgs& a, size_t lds, hipStream_t s) {
  static MgPerDevice once;
  if (mg_first_use_on_device(once))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&winoups_fwd<NT, PN, HEAD>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((winoups_fwd<NT, PN, HEAD>), dim3(a.G), dim3(64 * WU_NWAVE), lds, s, a);
  MG_CHECK_LAUNCH("mg_winoups3x3");
  return MG_OK;
}
template <int NT, bool PNB = false>
int wu_launch_dgrad(const WuArgs& a, size_t lds, hipStream_t s) {
  static MgPerDevice once;
  if (mg_first_use_on_device(once))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&winoups_dgrad<NT, PNB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((winoups_dgrad<NT, PNB>), dim3(a.G), dim3(64 * WU_NWAVE), lds, s, a);
  MG_CHECK_LAUNCH("mg_winoups3x3_dgrad");
  return MG_OK;
}

// K input channels, Cout out-channels (of THIS kernel: the data gradient calls it with the layer's channels swapped)
// tile_groups (data gradient without an epilogue over all channels): up to 6 out-channel tiles, five or six as two launches of <= 3 tiles
// each (the input is read and transformed once per launch); otherwise all tiles -- at most four -- in one wave
bool wu_shape_ok(int N, int K, int Cout, int Hl, int Wl, bool tile_groups = false) {
  if (N <= 0 || K < 16 || Cout <= 0 || (K % 8) != 0 || (Cout % 16) != 0 || Cout > (tile_groups ? 96 : 64)) return false;
  if ((Wl % 16) != 0 || (Hl % WU_NWAVE) != 0) return false;
  const int nt_launch = Cout > 64 ? 3 : Cout / 16;
  if ((size_t)(K / 8) * nt_launch * 9 * 512 > 160 * 1024) return false;                         // the filter bank of a launch in LDS
  if ((long long)(K > Cout ? K : Cout) * 4 * Hl * Wl * 16 >= (1ll << 31)) return false;         // 32-bit byte offsets inside an image
  return true;
}

void wu_fill(WuArgs& a, int N, int K, int Cout, int Hl, int Wl) {
  a.N = N; a.K = K; a.Cout = Cout; a.Hl = Hl; a.Wl = Wl;
  a.nchunk = K / 8;
  a.tile0 = 0; a.nt_total = Cout / 16;
  a.blocks_x = Wl / 16;
  const int nitems = N * (Hl / WU_NWAVE) * a.blocks_x;
  int g = mg_cu_count() & ~7;
  if (g < 8) g = 8;
  a.G = g < nitems ? g : nitems;
}

}  // namespace

extern "C" int mg_winoups3x3_supported(int N, int Cin, int Cout, int Hin, int Win, int dgrad) {
  // forward: Cin -> Cout on the up-sampled grid; data gradient: Cout (gy) -> Cin
  return dgrad ? wu_shape_ok(N, Cout, Cin, Hin, Win, true) : wu_shape_ok(N, Cin, Cout, Hin, Win);
}

extern "C" size_t mg_winoups3x3_packed_floats(int Cin, int Cout, int dgrad) { return pack_winoups_total(Cout, Cin, dgrad); }

extern "C" int mg_winoups3x3(const float* x, const float* up, const float* bias, float* y, float* p, float* rn, int N, int Cin, int Cout,
                             int Hin, int Win, int flags, float slope, mg_stream_t stream) {
  MG_CHECK_ARG(x && up && wu_shape_ok(N, Cin, Cout, Hin, Win), "mg_winoups3x3: unsupported shape (N=%d %d->%d %dx%d)", N, Cin, Cout, Hin, Win);
  MG_CHECK_ARG(!(flags & ~(MG_CONV_LRELU | MG_CONV_PIXNORM)), "mg_winoups3x3: flags %d (LRELU, PIXNORM)", flags);
  const bool pn = (flags & MG_CONV_PIXNORM) != 0;
  MG_CHECK_ARG(pn ? (p != nullptr) : (y != nullptr), "mg_winoups3x3: no output");
  WuArgs a;
  a.x = x; a.up = up; a.bias = bias; a.y = y; a.p = p; a.rn = rn; a.flags = flags; a.slope = slope;
  a.hw = nullptr; a.hb = nullptr; a.mp = nullptr;
  wu_fill(a, N, Cin, Cout, Hin, Win);
  const size_t lds = (size_t)a.nchunk * (Cout / 16) * 9 * 512;
  hipStream_t s = (hipStream_t)stream;
  switch ((Cout / 16) * 2 + (pn ? 1 : 0)) {
    case 2: return wu_launch_fwd<1, false>(a, lds, s);
    case 3: return wu_launch_fwd<1, true>(a, lds, s);
    case 4: return wu_launch_fwd<2, false>(a, lds, s);
    case 5: return wu_launch_fwd<2, true>(a, lds, s);
    case 6: return wu_launch_fwd<3, false>(a, lds, s);
    case 7: return wu_launch_fwd<3, true>(a, lds, s);
    case 8: return wu_launch_fwd<4, false>(a, lds, s);
    default: return wu_launch_fwd<4, true>(a, lds, s);
  }
}

// (at most three out-channel tiles: with four the head's weights and sums no longer fit the register file beside 144 accumulators)
extern "C" int mg_winoups3x3_head_supported(int N, int Cin, int Cout, int Hin, int Win) {
  return (wu_shape_ok(N, Cin, Cout, Hin, Win) && Cout <= 48) ? 1 : 0;
}

// The same with the generator's 1x1 head on the normalised activation in the epilogue (generator.py:118-126 ToMagnPhaseLayer on the last
// block's output): mp = tanh(hw p + hb), (N, 2, 2 Hin, 2 Win) -- the head kernel's pass over p is gone.  LeakyReLU + PixelNorm implied.
extern "C" int mg_winoups3x3_head(const float* x, const float* up, const float* bias, float* y, float* p, float* rn, const float* hw,
                                  const float* hb, float* mp, int N, int Cin, int Cout, int Hin, int Win, float slope, mg_stream_t stream) {
  MG_CHECK_ARG(x && up && mg_winoups3x3_head_supported(N, Cin, Cout, Hin, Win), "mg_winoups3x3_head: unsupported shape (N=%d %d->%d %dx%d)", N, Cin, Cout, Hin, Win);
  MG_CHECK_ARG(p && hw && mp, "mg_winoups3x3_head: p, head weights and mp are required");
  WuArgs a;
  a.x = x; a.up = up; a.bias = bias; a.y = y; a.p = p; a.rn = rn; a.flags = MG_CONV_LRELU | MG_CONV_PIXNORM; a.slope = slope;
  a.hw = hw; a.hb = hb; a.mp = mp;
  wu_fill(a, N, Cin, Cout, Hin, Win);
  const size_t lds = (size_t)a.nchunk * (Cout / 16) * 9 * 512;
  hipStream_t s = (hipStream_t)stream;
  switch (Cout / 16) {
    case 1: return wu_launch_fwd<1, true, true>(a, lds, s);
    case 2: return wu_launch_fwd<2, true, true>(a, lds, s);
    default: return wu_launch_fwd<3, true, true>(a, lds, s);
  }
}

// The data gradient with the PixelNorm + LeakyReLU backward of the layer below in the epilogue (generator.py:31-39 backward): p (N,Cin,
// Hin,Win) that layer's normalised output, rn (N,1,Hin,Win) its 1/norm; gpre = lrelu'(p) rn (gx - p mean_c(gx p)).
extern "C" int mg_winoups3x3_dgrad_pn(const float* gy, const float* up, const float* p, const float* rn, float* gpre, int N, int Cin, int Cout,
                                      int Hin, int Win, float slope, mg_stream_t stream) {
  MG_CHECK_ARG(gy && up && p && rn && gpre && wu_shape_ok(N, Cout, Cin, Hin, Win), "mg_winoups3x3_dgrad_pn: unsupported shape (N=%d %d<-%d %dx%d)",
               N, Cin, Cout, Hin, Win);
  WuArgs a;
  a.x = gy; a.up = up; a.bias = nullptr; a.y = gpre; a.p = const_cast<float*>(p); a.rn = const_cast<float*>(rn); a.flags = 0; a.slope = slope;
  a.hw = nullptr; a.hb = nullptr; a.mp = nullptr;
  wu_fill(a, N, Cout, Cin, Hin, Win);
  const size_t lds = (size_t)a.nchunk * (Cin / 16) * 9 * 512;
  hipStream_t s = (hipStream_t)stream;
  switch (Cin / 16) {
    case 1: return wu_launch_dgrad<1, true>(a, lds, s);
    case 2: return wu_launch_dgrad<2, true>(a, lds, s);
    case 3: return wu_launch_dgrad<3, true>(a, lds, s);
    default: return wu_launch_dgrad<4, true>(a, lds, s);
  }
}

extern "C" int mg_winoups3x3_dgrad(const float* gy, const float* up, float* gx, int N, int Cin, int Cout, int Hin, int Win,
                                   mg_stream_t stream) {
  MG_CHECK_ARG(gy && up && gx && wu_shape_ok(N, Cout, Cin, Hin, Win, true), "mg_winoups3x3_dgrad: unsupported shape (N=%d %d<-%d %dx%d)", N, Cin,
               Cout, Hin, Win);
  WuArgs a;
  a.x = gy; a.up = up; a.bias = nullptr; a.y = gx; a.p = nullptr; a.rn = nullptr; a.flags = 0; a.slope = 1.0f;
  a.hw = nullptr; a.hb = nullptr; a.mp = nullptr;
  wu_fill(a, N, Cout, Cin, Hin, Win);
  hipStream_t s = (hipStream_t)stream;
  const int nt = Cin / 16;
  auto lds = [&](int tiles) { return (size_t)a.nchunk * tiles * 9 * 512; };
  if (nt > 4) {  // 80 / 96 input channels of the layer: tiles 0-2, then the other two or three
    a.tile0 = 0;
    int rc = wu_launch_dgrad<3>(a, lds(3), s);
    if (rc != MG_OK) return rc;
    a.tile0 = 3;
    return nt == 5 ? wu_launch_dgrad<2>(a, lds(2), s) : wu_launch_dgrad<3>(a, lds(3), s);
  }
  switch (nt) {
    case 1: return wu_launch_dgrad<1>(a, lds(1), s);
    case 2: return wu_launch_dgrad<2>(a, lds(2), s);
    case 3: return wu_launch_dgrad<3>(a, lds(3), s);
    default: return wu_launch_dgrad<4>(a, lds(4), s);
  }
}
