// Weight (+bias) gradient of the 3x3 convolution on the fp32 matrix cores (v_mfma_f32_16x16x4_f32).
//
// Replaces aten::convolution_backward (weight/bias part) as reached from the reference's
//   generator.py:16-22,31-37 / discriminator.py:15-21,26-32 under loss.backward()  (train.py:174,213).
//
// GEMM view per tap t=(ky,kx):  GW_t[o, c] = sum_{pixels} GY[o, pixel] * X[c, pixel + tap]
//   K = pixels: 4 consecutive pixels per MFMA (lane>>4), A rows = 16 out-channels, B cols = 16 in-channels.
//   One wave owns ONE 16-wide in-channel tile and WO (<=4) out-channel tiles for all 9 taps: 9*WO accumulators (f32x4),
//   13 LDS operand reads per 36 MFMAs at WO=4.  A workgroup = one wave per in-channel tile of its (o-block, c-block).
//   Workgroups are persistent over pixel tiles (split-K); partials go to a slab [split][tap][o][c] and a second kernel sums
//   the splits in a fixed order (deterministic, no float atomics) while transposing to the module layout [o][c][tap].
// LDS image per pixel tile: GY [WO*16][P+2] and X halo [nwaves*16][plane(+pad)], both with channel stride % 4 == 2 so
// that the 16 channel-lanes x 2 k-lanes of a half-wave fall on 32 distinct banks.
#include <cstdlib>

#include "mg_common.h"

namespace {

struct WgradArgs {
  const float* x;
  const float* gy;
  float* slab;    // [nsplit][9][Cout][Cin]
  float* slab_b;  // [nsplit][Cout]
  int N, Cin, Cout, H, W, Hin, Win, ups;
  int TH, TW, TN, lgTH, lgTW, THp, TWp, P;
  int tiles_x, tiles_y, tiles_n, ntiles;
  int plane, x_stride, gy_stride, tab_floats;
  int oblocks;  // grid.y = oblocks * cblocks
  int ogroups;  // waves along the out-channel tiles of the block (each owns WO tiles)
  int nct;      // in-channel tiles per workgroup; waves = ogroups * nct
  int bias_n;   // only samples n < bias_n contribute to the bias gradient
};

// NIX > 0: software-pipelined variant.  Every global load of the NEXT pixel tile is issued into registers right after the
// barrier that starts the current tile's MFMA phase (NIX = 8 / ogroups channel passes per half-wave x 7 halo positions +
// up to 8 float4 of gy) and written to LDS after it, so the ~70 KB of staging per tile hides under ~18k cycles of matrix work.
// NIX == 0: simple load->store->compute variant (tiny images, odd shapes).
constexpr int PF_NJ = 7;   // halo positions per lane (plane <= 224)
constexpr int PF_NG4 = 6;  // gy float4 per thread

// T32: the pixel tile is the 4 x 32 tile of every image at least 32 wide (TW 32, TH 4, one sample): all LDS strides become
// compile-time, operand reads use immediate offsets and the k-loop carries no index arithmetic (4.5 -> ~1.5 VALU per MFMA).
template <int WO, int NIX, bool T32>
__global__ void __launch_bounds__(WO == 1 ? 768 : 512) wgrad3x3_mfma(const WgradArgs a) {
  constexpr bool PF = NIX > 0;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int* tab = reinterpret_cast<int*>(smem);
  float* gy_t = smem + a.tab_floats;
  const int TB = WO * a.ogroups;  // out-channel tiles staged per workgroup
  float* x_t = gy_t + TB * 16 * a.gy_stride;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int og = wave % a.ogroups, ct = wave / a.ogroups;
  const int nthr = blockDim.x;
  const int col = lane & 15, rq = lane >> 4;
  const int ob = blockIdx.y % a.oblocks, cb = blockIdx.y / a.oblocks;
  const int o0 = ob * TB * 16;
  const int c0 = cb * a.nct * 16;
  const int HWin = a.Hin * a.Win;
  const int HW = a.H * a.W;
  const int THpTWp = a.THp * a.TWp;
  const bool vec = (a.TW >= 4) && ((a.W & 3) == 0);
  const int q4 = a.P >> 2;  // float4 groups per gy channel row
  const int nhalf = nthr >> 5, hw_ = tid >> 5, l32 = tid & 31;

  f32x4 acc[WO][9];
  float gb[WO];
#pragma unroll
  for (int wo = 0; wo < WO; ++wo) {
    gb[wo] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[wo][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  auto build_tab = [&](int tile) {
    const int tx = tile % a.tiles_x;
    const int t2 = tile / a.tiles_x;
    const int ty = t2 % a.tiles_y;
    const int tn = t2 / a.tiles_y;
    for (int pos = tid; pos < a.plane; pos += nthr) {
      const int n_l = pos / THpTWp;
      const int rem = pos - n_l * THpTWp;
      const int rr = rem / a.TWp;
      const int cc = rem - rr * a.TWp;
      const int n = tn * a.TN + n_l, Y = ty * a.TH + rr - 1, X = tx * a.TW + cc - 1;
      const bool ok = (n < a.N) && (Y >= 0) && (Y < a.H) && (X >= 0) && (X < a.W);
      const int sp = a.ups ? (Y >> 1) * a.Win + (X >> 1) : Y * a.Win + X;
      tab[pos] = ok ? n_l * a.Cin * HWin + sp : -1;
    }
  };

  int cur_tn = 0;  // sample-tile index of the tile being consumed (bias masking)
  auto compute_tile = [&]() {
    if constexpr (T32) {
      constexpr int TWP = 34, GS = 130, XS = 206;
      const float* xw = x_t + (ct * 16 + col) * XS + rq;
      const float* gw_ = gy_t + (og * WO * 16 + col) * GS + rq;
      const bool bias_on = cur_tn < a.bias_n;
#pragma unroll 1
      for (int r = 0; r < 4; ++r) {
        const float* xr = xw + r * TWP;
        const float* gr = gw_ + r * 32;
#pragma unroll
        for (int xg = 0; xg < 8; ++xg) {
          float av[WO], bv[9];
#pragma unroll
          for (int wo = 0; wo < WO; ++wo) av[wo] = gr[wo * 16 * GS + 4 * xg];
#pragma unroll
          for (int t = 0; t < 9; ++t) bv[t] = xr[(t / 3) * TWP + (t % 3) + 4 * xg];
#pragma unroll
          for (int wo = 0; wo < WO; ++wo) {
            gb[wo] += bias_on ? av[wo] : 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t)
              acc[wo][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[wo], bv[t], acc[wo][t], 0, 0, 0);
          }
        }
      }
      return;
    }
    const float* xw = x_t + (ct * 16 + col) * a.x_stride;
    const float* gw_ = gy_t + (og * WO * 16 + col) * a.gy_stride;
    const int nk = a.P >> 2;
#pragma unroll 2
    for (int kk = 0; kk < nk; ++kk) {
      const int p = kk * 4 + rq;
      const int c = p & (a.TW - 1);
      const int r = (p >> a.lgTW) & (a.TH - 1);
      const int n_l = p >> (a.lgTW + a.lgTH);
      const int xpos = (n_l * a.THp + r) * a.TWp + c;
      float av[WO], bv[9];
#pragma unroll
      for (int wo = 0; wo < WO; ++wo) av[wo] = gw_[wo * 16 * a.gy_stride + p];
#pragma unroll
      for (int t = 0; t < 9; ++t) bv[t] = xw[xpos + (t / 3) * a.TWp + (t % 3)];
      const bool bias_on = (cur_tn * a.TN + n_l) < a.bias_n;
#pragma unroll
      for (int wo = 0; wo < WO; ++wo) {
        gb[wo] += bias_on ? av[wo] : 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t)
          acc[wo][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[wo], bv[t], acc[wo][t], 0, 0, 0);
      }
    }
  };

  if constexpr (PF) {
    float rx[NIX][PF_NJ];
    f32x4 rg[PF_NG4];
    // `z` is an opaque zero refreshed every iteration: staging indices derived from it cannot be hoisted out of the tile
    // loop (hipcc otherwise keeps ~40 loop-invariant LDS/global offsets live across the MFMA phase and halves occupancy).
    int z = 0;
    auto load_tile = [&](int tile) {
      const int tid = threadIdx.x + z, hw_ = (threadIdx.x >> 5) + z, l32 = (threadIdx.x & 31) + z;
      const int tx = tile % a.tiles_x;
      const int t2 = tile / a.tiles_x;
      const int ty = t2 % a.tiles_y;
      const int tn = t2 / a.tiles_y;
#pragma unroll
      for (int j = 0; j < PF_NG4; ++j) {
        const int e = tid + j * nthr;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (e < TB * 16 * q4) {
          const int ol = e / q4;
          const int p = (e - ol * q4) << 2;
          const int c = p & (a.TW - 1);
          const int r = (p >> a.lgTW) & (a.TH - 1);
          const int n_l = p >> (a.lgTW + a.lgTH);
          const int n = tn * a.TN + n_l, Y = ty * a.TH + r, X = tx * a.TW + c, o = o0 + ol;
          if (o < a.Cout && n < a.N && Y < a.H && X < a.W)
            v = *reinterpret_cast<const f32x4*>(a.gy + ((size_t)n * a.Cout + o) * HW + (size_t)Y * a.W + X);
        }
        rg[j] = v;
      }
      const float* xn = a.x + (size_t)tn * a.TN * a.Cin * HWin;
      int offs[PF_NJ];
#pragma unroll
      for (int j = 0; j < PF_NJ; ++j) {
        const int pos = l32 + 32 * j;
        offs[j] = pos < a.plane ? tab[pos] : -1;
      }
#pragma unroll
      for (int i = 0; i < NIX; ++i) {
        const int cl = hw_ + i * nhalf;
        const int c = c0 + cl;
        const bool cok = (cl < a.nct * 16) && (c < a.Cin);
        const float* xc = xn + (size_t)c * HWin;
#pragma unroll
        for (int j = 0; j < PF_NJ; ++j) {
          float v = 0.f;
          if (cok && offs[j] >= 0) v = xc[offs[j]];
          rx[i][j] = v;
        }
      }
    };
    auto store_tile = [&]() {
      const int tid = threadIdx.x + z, hw_ = (threadIdx.x >> 5) + z, l32 = (threadIdx.x & 31) + z;
#pragma unroll
      for (int j = 0; j < PF_NG4; ++j) {
        const int e = tid + j * nthr;
        if (e < TB * 16 * q4) {
          const int ol = e / q4;
          const int p = (e - ol * q4) << 2;
          float* d = gy_t + ol * a.gy_stride + p;  // 8-byte aligned (gy_stride even, p % 4 == 0)
          *reinterpret_cast<float2*>(d) = make_float2(rg[j][0], rg[j][1]);
          *reinterpret_cast<float2*>(d + 2) = make_float2(rg[j][2], rg[j][3]);
        }
      }
#pragma unroll
      for (int i = 0; i < NIX; ++i) {
        const int cl = hw_ + i * nhalf;
        if (cl < a.nct * 16) {
          float* dst = x_t + cl * a.x_stride;
#pragma unroll
          for (int j = 0; j < PF_NJ; ++j) {
            const int pos = l32 + 32 * j;
            if (pos < a.plane) dst[pos] = rx[i][j];
          }
        }
      }
    };
    int tile = blockIdx.x;
    build_tab(tile);
    __syncthreads();
    load_tile(tile);
    for (; tile < a.ntiles; tile += gridDim.x) {
      const int next = tile + gridDim.x;
      asm volatile("" : "+v"(z));
      __syncthreads();  // previous tile's MFMA reads done; every thread has issued the loads that used tab
      store_tile();
      if (next < a.ntiles) build_tab(next);
      __syncthreads();
      if (next < a.ntiles) load_tile(next);  // in flight during the MFMA phase below
      cur_tn = (tile / a.tiles_x) / a.tiles_y;
      compute_tile();
    }
  } else {
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
      const int tx = tile % a.tiles_x;
      const int t2 = tile / a.tiles_x;
      const int ty = t2 % a.tiles_y;
      const int tn = t2 / a.tiles_y;
      __syncthreads();  // previous tile's MFMA reads done
      build_tab(tile);
      if (vec) {
        for (int e = tid; e < TB * 16 * q4; e += nthr) {
          const int ol = e / q4;
          const int p = (e - ol * q4) << 2;
          const int c = p & (a.TW - 1);
          const int r = (p >> a.lgTW) & (a.TH - 1);
          const int n_l = p >> (a.lgTW + a.lgTH);
          const int n = tn * a.TN + n_l, Y = ty * a.TH + r, X = tx * a.TW + c, o = o0 + ol;
          f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
          if (o < a.Cout && n < a.N && Y < a.H && X < a.W)
            v = *reinterpret_cast<const f32x4*>(a.gy + ((size_t)n * a.Cout + o) * HW + (size_t)Y * a.W + X);
          float* d = gy_t + ol * a.gy_stride + p;  // 8-byte aligned (gy_stride even, p % 4 == 0)
          *reinterpret_cast<float2*>(d) = make_float2(v[0], v[1]);
          *reinterpret_cast<float2*>(d + 2) = make_float2(v[2], v[3]);
        }
      } else {
        for (int e = tid; e < TB * 16 * a.P; e += nthr) {
          const int ol = e / a.P;
          const int p = e - ol * a.P;
          const int c = p & (a.TW - 1);
          const int r = (p >> a.lgTW) & (a.TH - 1);
          const int n_l = p >> (a.lgTW + a.lgTH);
          const int n = tn * a.TN + n_l, Y = ty * a.TH + r, X = tx * a.TW + c, o = o0 + ol;
          float v = 0.f;
          if (o < a.Cout && n < a.N && Y < a.H && X < a.W) v = a.gy[((size_t)n * a.Cout + o) * HW + (size_t)Y * a.W + X];
          gy_t[ol * a.gy_stride + p] = v;
        }
      }
      __syncthreads();  // tab ready
      {                 // X halo tile [nct*16][plane]: half-wave per channel, 32 consecutive positions per pass
        const float* xn = a.x + (size_t)tn * a.TN * a.Cin * HWin;
        for (int cl = hw_; cl < a.nct * 16; cl += nhalf) {
          const int c = c0 + cl;
          const bool cok = c < a.Cin;
          const float* xc = xn + (size_t)c * HWin;
          float* dst = x_t + cl * a.x_stride;
#pragma unroll 4
          for (int pos = l32; pos < a.plane; pos += 32) {
            const int off = tab[pos];
            float v = 0.f;
            if (cok && off >= 0) v = xc[off];
            dst[pos] = v;
          }
        }
      }
      __syncthreads();
      cur_tn = tn;
      compute_tile();
    }
  }

  // partial slab: D rows = out-channels 4*rq+g, cols = in-channel col
  const int c = c0 + ct * 16 + col;
  float* slab = a.slab + (size_t)blockIdx.x * 9 * a.Cout * a.Cin;
#pragma unroll
  for (int wo = 0; wo < WO; ++wo) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int o = o0 + (og * WO + wo) * 16 + rq * 4 + g;
        if (o < a.Cout && c < a.Cin) slab[((size_t)t * a.Cout + o) * a.Cin + c] = acc[wo][t][g];
      }
    }
  }
  if (cb == 0 && ct == 0 && a.slab_b != nullptr) {
#pragma unroll
    for (int wo = 0; wo < WO; ++wo) {
      float v = gb[wo];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      const int o = o0 + (og * WO + wo) * 16 + col;
      if (rq == 0 && o < a.Cout) a.slab_b[(size_t)blockIdx.x * a.Cout + o] = v;
    }
  }
}

// Sum the split-K slabs in a fixed order.  Block = 64 consecutive slab elements x 4 split-lanes (each lane sums every 4th
// split, 4 loads in flight), LDS-combined as ((l0 + l1) + (l2 + l3)) => deterministic.
__device__ __forceinline__ void wgrad3x3_reduce_body(const float* __restrict__ slab, const float* __restrict__ slab_b,
                                                     int nsplit, float* __restrict__ gw, float* __restrict__ gb,
                                                     int Cout, int Cin, int accumulate, int block) {
  __shared__ float red[4][64];
  const int total = 9 * Cout * Cin;
  const int all = total + (gb != nullptr ? Cout : 0);
  const int el = threadIdx.x & 63, kl = threadIdx.x >> 6;
  const int e = block * 64 + el;
  float s = 0.f;
  if (e < all) {
    const float* src = e < total ? slab + e : slab_b + (e - total);
    const size_t stride = e < total ? (size_t)total : (size_t)Cout;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = kl;
    for (; k + 12 < nsplit; k += 16) {
      s0 += src[(size_t)k * stride];
      s1 += src[(size_t)(k + 4) * stride];
      s2 += src[(size_t)(k + 8) * stride];
      s3 += src[(size_t)(k + 12) * stride];
    }
    for (; k < nsplit; k += 4) s0 += src[(size_t)k * stride];
    s = (s0 + s1) + (s2 + s3);
  }
  red[kl][el] = s;
  __syncthreads();
  if (kl == 0 && e < all) {
    s = (red[0][el] + red[1][el]) + (red[2][el] + red[3][el]);
    if (e < total) {
      const int c = e % Cin;
      const int r = e / Cin;
      const int o = r % Cout;
      const int t = r / Cout;
      const size_t idx = ((size_t)o * Cin + c) * 9 + t;
      gw[idx] = accumulate ? gw[idx] + s : s;
    } else {
      const int o = e - total;
      gb[o] = accumulate ? gb[o] + s : s;
    }
  }
}

// The reductions of several layers in one launch (jobs by value, blockIdx.y = job; blocks past a job's extent return).
constexpr int WG_JOBS = 40;
struct WgJobs {
  mg_wgrad_job_t j[WG_JOBS];
};
__global__ void __launch_bounds__(256) wgrad3x3_reduce_multi(const WgJobs jobs) {
  const mg_wgrad_job_t j = jobs.j[blockIdx.y];
  if ((int)blockIdx.x * 64 >= 9 * j.Cout * j.Cin + j.Cout) return;
  wgrad3x3_reduce_body(j.slab, j.slab_b, j.nsplit, j.gw, j.gb, j.Cout, j.Cin, j.accumulate, blockIdx.x);
}

struct WgradPlan {
  WgradArgs a;
  int WO, nsplit, cblocks;
  size_t lds, ws_floats;
};

bool plan_wgrad(int N, int Cin, int Cout, int H, int W, WgradPlan& pl) {
  WgradArgs& a = pl.a;
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  const int otiles = mg_cdiv(Cout, 16), ctiles = mg_cdiv(Cin, 16);
  a.oblocks = mg_cdiv(otiles, 4);
  const int TB = mg_cdiv(otiles, a.oblocks);  // out-channel tiles per workgroup (<= 4)
  // Wave grid = ogroups (out-channel tiles, WO each) x nct (in-channel tiles).  Preference order: a shape the pipelined
  // kernel is instantiated for (WO == 1 with up to 12 waves, or WO == 2 x 2 groups), then the fewest in-channel blocks (each
  // re-reads the gy tile), then the most waves.
  int best_d = 1, best_cb = 1 << 30, best_w = 0, best_pf = -1;
  for (int d = 1; d <= TB; ++d) {
    if (TB % d) continue;
    const int wo = TB / d;
    const int pf = (wo == 1 || (wo == 2 && d == 2)) ? 1 : 0;
    const int wmax = wo == 1 ? 12 : 8;
    const int nct_max = wmax / d;
    if (nct_max < 1) continue;
    const int cbk = mg_cdiv(ctiles, nct_max);
    const int nct = mg_cdiv(ctiles, cbk);
    const int w = d * nct;
    const bool better = pf > best_pf || (pf == best_pf && (cbk < best_cb || (cbk == best_cb && w > best_w)));
    if (better) { best_d = d; best_cb = cbk; best_w = w; best_pf = pf; }
  }
  a.ogroups = best_d;
  pl.WO = TB / best_d;
  pl.cblocks = best_cb;
  a.nct = mg_cdiv(ctiles, pl.cblocks);
  // pixel tile: 128 pixels, shrunk for tiny images (halo overhead 4x..9x) until the LDS image fits comfortably
  for (a.P = 128;; a.P >>= 1) {
    a.TW = mg_pow2_ceil(W) < 32 ? mg_pow2_ceil(W) : 32;
    if (a.TW > a.P) a.TW = a.P;
    a.TH = mg_pow2_ceil(H) < a.P / a.TW ? mg_pow2_ceil(H) : a.P / a.TW;
    a.TN = a.P / (a.TW * a.TH);
    a.lgTW = mg_ilog2(a.TW); a.lgTH = mg_ilog2(a.TH);
    a.THp = a.TH + 2; a.TWp = a.TW + 2;
    a.plane = a.TN * a.THp * a.TWp;
    a.x_stride = a.plane + ((6 - (a.plane & 3)) & 3);  // smallest s >= plane with s % 4 == 2
    a.gy_stride = a.P + 2;
    a.tab_floats = (a.plane + 3) & ~3;
    pl.lds = (size_t)(a.tab_floats + TB * 16 * a.gy_stride + a.nct * 16 * a.x_stride) * sizeof(float);
    if (pl.lds <= 96 * 1024 || a.P <= 16) break;
  }
  a.tiles_x = mg_cdiv(W, a.TW); a.tiles_y = mg_cdiv(H, a.TH); a.tiles_n = mg_cdiv(N, a.TN);
  a.ntiles = a.tiles_x * a.tiles_y * a.tiles_n;
  const int gy_blocks = a.oblocks * pl.cblocks;
  int nsplit = 512 / gy_blocks;  // persistent workgroups: ~2 per CU in total
  if (nsplit < 1) nsplit = 1;
  if (nsplit > a.ntiles) nsplit = a.ntiles;
  pl.nsplit = nsplit;
  pl.ws_floats = (size_t)nsplit * ((size_t)9 * Cout * Cin + Cout);
  return pl.lds <= 160 * 1024;
}

template <int WO, int NIX, bool T32>
int launch_wgrad_t(const WgradPlan& pl, hipStream_t s) {
  static MgPerDevice once;  // the LDS limit is a per-device function attribute
  if (mg_first_use_on_device(once)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3_mfma<WO, NIX, T32>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  dim3 grid(pl.nsplit, pl.a.oblocks * pl.cblocks);
  hipLaunchKernelGGL((wgrad3x3_mfma<WO, NIX, T32>), grid, dim3(64 * pl.a.ogroups * pl.a.nct), pl.lds, s, pl.a);
  MG_CHECK_LAUNCH("mg_conv3x3_wgrad");
  return MG_OK;
}

template <int WO, int NIX>
int launch_wgrad(const WgradPlan& pl, hipStream_t s) {
  const WgradArgs& a = pl.a;
  const bool t32 = a.TW == 32 && a.TH == 4 && a.TN == 1 && a.P == 128 && a.gy_stride == 130 && a.x_stride == 206;
  if constexpr (NIX > 0) {
    if (t32) return launch_wgrad_t<WO, NIX, true>(pl, s);
  }
  return launch_wgrad_t<WO, NIX, false>(pl, s);
}

bool wgrad_pf_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("MG_WGRAD_PF");
    v = (e == nullptr) ? 1 : (atoi(e) != 0);
  }
  return v != 0;
}

int dispatch_wgrad(const WgradPlan& pl, hipStream_t s) {
  const WgradArgs& a = pl.a;
  const int nthr = 64 * a.ogroups * a.nct;
  const bool vec = (a.TW >= 4) && ((a.W & 3) == 0);
  const bool pf = wgrad_pf_enabled() && vec && a.plane <= 32 * PF_NJ &&
                  pl.WO * a.ogroups * 16 * (a.P / 4) <= PF_NG4 * nthr && a.ntiles > pl.nsplit;
  const int key = pl.WO * 10 + (pf ? a.ogroups : 0);
  switch (key) {
    case 10: return launch_wgrad<1, 0>(pl, s);
    case 20: return launch_wgrad<2, 0>(pl, s);
    case 30: return launch_wgrad<3, 0>(pl, s);
    case 40: return launch_wgrad<4, 0>(pl, s);
    case 11: return launch_wgrad<1, 8>(pl, s);
    case 12: return launch_wgrad<1, 4>(pl, s);
    case 22: return launch_wgrad<2, 4>(pl, s);
    case 13: return launch_wgrad<1, 3>(pl, s);
    case 14: return launch_wgrad<1, 2>(pl, s);
    default: break;
  }
  // ogroups == 3 (three out-channel tiles split one per wave): no pipelined instantiation, use the simple variant
  switch (pl.WO) {
    case 1: return launch_wgrad<1, 0>(pl, s);
    case 2: return launch_wgrad<2, 0>(pl, s);
    case 3: return launch_wgrad<3, 0>(pl, s);
    default: return launch_wgrad<4, 0>(pl, s);
  }
}

// 1x1 maps (the last critic conv, discriminator.py:14-34 after the final pool): only the centre tap ever meets an input value, so
// gw[o][c][1][1] = sum_n gy[n, o] * x[n, c] and the other eight taps are zero -- one thread per (o, c) pair, float64 sum in the
// order of n (as conv3x3_tiny accumulates the forward), no workspace, no reduce launch.  The MFMA kernel spends 18 us + a 5 us
// reduce on this layer (9 workgroups walking a 256-pixel tile geometry that holds one pixel per image).
__global__ void __launch_bounds__(256) wgrad3x3_1x1map_k(const float* __restrict__ x, const float* __restrict__ gy,
                                                        float* __restrict__ gw, float* __restrict__ gb, int N, int Cin, int Cout,
                                                        int accumulate, int bias_n) {
  // workgroup = 32 (o, c) pairs x 8 slices of the samples (one thread walking 192 samples is 24 memory round trips in a row);
  // the slices' float64 partial sums meet in LDS and are added in slice order
  __shared__ double red[2][8][32];
  const int pl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + pl;
  const bool live = e < Cin * Cout;
  const int o = live ? e / Cin : 0, c = live ? e - o * Cin : 0;
  const int per = (N + 7) / 8;
  const int n_lo = sl * per, n_hi = n_lo + per < N ? n_lo + per : N;
  double acc = 0.0, accb = 0.0;
  for (int n0 = n_lo; n0 < n_hi; n0 += 8) {
    float xv[8], gv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {  // loads first, arithmetic second (samples past the slice re-read its first one and count for nothing)
      const int n = n0 + u < n_hi ? n0 + u : n_lo;
      xv[u] = x[(size_t)n * Cin + c];
      gv[u] = gy[(size_t)n * Cout + o];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (n0 + u < n_hi) {
        acc += (double)gv[u] * (double)xv[u];
        if (n0 + u < bias_n) accb += (double)gv[u];
      }
    }
  }
  red[0][sl][pl] = acc;
  red[1][sl][pl] = accb;
  __syncthreads();
  if (sl != 0 || !live) return;
  double tw = 0.0, tb = 0.0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    tw += red[0][k][pl];
    tb += red[1][k][pl];
  }
  float* w9 = gw + (size_t)e * 9;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float v = t == 4 ? (float)tw : 0.f;
    w9[t] = accumulate ? w9[t] + v : v;
  }
  if (gb && c == 0) gb[o] = accumulate ? gb[o] + (float)tb : (float)tb;
}

}  // namespace

extern "C" int mg_conv3x3_wgrad_1x1map(const float* x, const float* gy, float* gw, float* gb, int N, int Cin, int Cout, int accumulate,
                                       int bias_n, mg_stream_t stream) {
  MG_CHECK_ARG(x && gy && gw && N > 0 && Cin > 0 && Cout > 0, "mg_conv3x3_wgrad_1x1map: bad arguments");
  MG_CHECK_ARG((long long)Cin * Cout < (1ll << 27), "mg_conv3x3_wgrad_1x1map: too many filters");
  const int bn = (bias_n <= 0 || bias_n > N) ? N : bias_n;
  hipLaunchKernelGGL(wgrad3x3_1x1map_k, dim3(mg_cdiv(Cin * Cout, 32)), dim3(256), 0, (hipStream_t)stream, x, gy, gw, gb, N, Cin, Cout,
                     accumulate, bn);
  MG_CHECK_LAUNCH("mg_conv3x3_wgrad_1x1map");
  return MG_OK;
}

extern "C" size_t mg_conv3x3_wgrad_ws_bytes(int N, int Cin, int Cout, int H, int W) {
  WgradPlan pl;
  plan_wgrad(N, Cin, Cout, H, W, pl);
  return pl.ws_floats * sizeof(float);
}

extern "C" int mg_conv3x3_wgrad_partial(const float* x, const float* gy, float* gw, float* gb, void* ws, size_t ws_bytes, int N,
                                        int Cin, int Cout, int H, int W, int flags, int accumulate, int bias_n,
                                        mg_wgrad_job_t* job, mg_stream_t stream) {
  MG_CHECK_ARG(x && gy && gw && ws && job && N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "mg_conv3x3_wgrad: bad arguments");
  const bool ups = flags & MG_CONV_UPS_IN;
  MG_CHECK_ARG(!ups || ((H % 2 == 0) && (W % 2 == 0)), "mg_conv3x3_wgrad: upsampled input needs even H,W");
  const long long in_elems = (long long)N * Cin * (ups ? (H / 2) * (W / 2) : H * W);
  MG_CHECK_ARG(in_elems < (1ll << 31), "mg_conv3x3_wgrad: tensor too large");
  hipStream_t s = (hipStream_t)stream;
  WgradPlan pl;
  MG_CHECK_ARG(plan_wgrad(N, Cin, Cout, H, W, pl), "mg_conv3x3_wgrad: LDS tile too large");
  if (ws_bytes < pl.ws_floats * sizeof(float)) {
    mg_set_error("mg_conv3x3_wgrad: workspace %zu < %zu bytes", ws_bytes, pl.ws_floats * sizeof(float));
    return MG_EWORKSPACE;
  }
  WgradArgs& a = pl.a;
  a.x = x; a.gy = gy;
  a.slab = reinterpret_cast<float*>(ws);
  a.slab_b = a.slab + (size_t)pl.nsplit * 9 * Cout * Cin;
  a.ups = ups ? 1 : 0;
  a.Hin = ups ? H / 2 : H; a.Win = ups ? W / 2 : W;
  a.bias_n = (bias_n <= 0 || bias_n > N) ? N : bias_n;
  const int rc = dispatch_wgrad(pl, s);
  if (rc != MG_OK) return rc;
  job->slab = a.slab; job->slab_b = a.slab_b; job->gw = gw; job->gb = gb;
  job->nsplit = pl.nsplit; job->Cout = Cout; job->Cin = Cin; job->CoutP = 0; job->CinP = 0; job->accumulate = accumulate;
  return MG_OK;
}

extern "C" int mg_conv3x3_wgrad_reduce(const mg_wgrad_job_t* jobs, int n, mg_stream_t stream) {
  MG_CHECK_ARG(jobs && n > 0, "mg_conv3x3_wgrad_reduce: bad arguments");
  for (int first = 0; first < n; first += WG_JOBS) {
    const int m = n - first < WG_JOBS ? n - first : WG_JOBS;
    WgJobs c;
    int most = 0;
    for (int i = 0; i < m; ++i) {
      c.j[i] = jobs[first + i];
      MG_CHECK_ARG(c.j[i].slab && c.j[i].gw && c.j[i].nsplit > 0 && c.j[i].Cout > 0 && c.j[i].Cin > 0,
                   "mg_conv3x3_wgrad_reduce: bad job %d", first + i);
      const int blocks = mg_cdiv(9 * c.j[i].Cout * c.j[i].Cin + c.j[i].Cout, 64);
      if (blocks > most) most = blocks;
    }
    hipLaunchKernelGGL(wgrad3x3_reduce_multi, dim3(most, m), dim3(256), 0, (hipStream_t)stream, c);
    MG_CHECK_LAUNCH("mg_conv3x3_wgrad_reduce");
  }
  return MG_OK;
}

extern "C" int mg_conv3x3_wgrad(const float* x, const float* gy, float* gw, float* gb, void* ws, size_t ws_bytes, int N,
                                int Cin, int Cout, int H, int W, int flags, int accumulate, int bias_n,
                                mg_stream_t stream) {
  mg_wgrad_job_t job;
  const int rc = mg_conv3x3_wgrad_partial(x, gy, gw, gb, ws, ws_bytes, N, Cin, Cout, H, W, flags, accumulate, bias_n, &job, stream);
  return rc != MG_OK ? rc : mg_conv3x3_wgrad_reduce(&job, 1, stream);
}
