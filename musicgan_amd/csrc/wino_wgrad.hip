// Weight gradient of the 3x3 / stride 1 / pad 1 convolution in Winograd F(3x3, 2x2) form on the fp32 matrix cores of gfx950
// (aten::convolution_backward, weight + bias grads, of nn.Conv2d(3x3) in /root/reference/music_gan/networks/generator.py:9-40 and
// discriminator.py:8-34).  For every 2x2 tile of the output gradient t and the 4x4 input patch d around it
//
//   dW (3x3)  =  G^T [ sum_tiles (A t A^T) .* (B^T d B) ] G        B^T, G: the matrices of wino3x3.hip;  A = (A^T)^T  (4x2)
//
// (the transposed F(2x2,3x3) algorithm: same input transform as the forward pass), i.e. 16 multiplies per tile and channel pair
// instead of 36.  Per Winograd component xi the sum over tiles is a GEMM  M_xi[c, o] = sum_tile V_xi[tile, c] * Y_xi[tile, o]:
//   M (A rows) = in-channels  (CT tiles of 16),  N (B cols) = out-channels (OT tiles of 16),  K = tiles, 8 per LDS chunk
//   (two MFMA k-steps: lane k-index rq holds tiles 2rq and 2rq+1 of the chunk)
// One workgroup of 8 waves owns a (CT*16) x (OT*16) block of ALL 16 components -- wave w accumulates component pair w, 2*CT*OT
// accumulator tiles -- over a slab of tiles (split-K over workgroups); both operands are transformed on the way from HBM to LDS
// (x: coalesced 8-byte row loads + DPP halo exchange + packed adds exactly as in wino3x3.hip; gy: two 8-byte loads), the LDS
// stages are double-buffered with one barrier per chunk, G^T . G is applied per slab (the components of a channel pair meet in LDS
// once the tile loop is done), and a second kernel sums the 9-tap slabs in a fixed order
// (bitwise deterministic, no float atomics).  The bias gradient rides along in the gy staging threads.
#include <cstdlib>
#include <type_traits>

#include "mg_common.h"

namespace {

typedef float wg_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ wg_f32x2 pk_sub(wg_f32x2 x, wg_f32x2 y) {  // x - y as one packed instruction
  wg_f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(x), "v"(y));
  return d;
}

constexpr int KT = 8;                  // tiles per chunk
constexpr int CH = 64;                 // channel slots per operand image (CT, OT <= 4)
constexpr int IMG = 8 * 4 * CH * 4;    // floats per operand image: [8 comp pairs][4 tile pairs][64 channels][k-step 2][parity 2]
constexpr int STAGE = 2 * IMG;         // V image + Y image

struct WwArgs {
  const float* x;
  const float* gy;
  float* slab;    // [nsplit][9 taps][CinP][CoutP]
  float* slab_b;  // [nsplit][CoutP]
  int N, Cin, Cout, H, W;
  int TBW, TBH, TBN, lgTBW, lgTBH;  // chunk geometry in TILES: TBW * TBH * TBN == 8
  int blocks_x, blocks_y, blocks_n, nblk, per;
  int CinP, CoutP;
  int nob;  // out-channel blocks (blockIdx.y = cb * nob + ob)
  int bias_n;
  unsigned x_bytes, gy_bytes;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

// UPS: x is (N, Cin, H/2, W/2) and the convolution input is its nearest x2 up-sampling (generator.py:24-25): the 4x4 patch of tile
// (TY, TX) is then the 3x3 low-res neighbourhood with the centre row / column doubled -- one dword per row and lane.
// The body takes its block coordinates as arguments: `split` (slab index, blockIdx.x of a single-layer launch) and `yblk`
// (channel block pair, blockIdx.y) -- a grouped launch (wino_wgrad_group_mfma below) derives them from a table instead.
// FAST: chunks of 8 x 1 x 1 tiles on maps whose width is a multiple of 16 (every layer from 16x16 maps up): which rows and halo pixels
// of a chunk exist is then the same for all its lanes, so a lane's byte offset is fixed for the kernel (+ one add of the chunk's
// column), rows travel in the scalar offset, and a row / chunk outside the tensor is read through a descriptor of zero records --
// the general form spends ~45 of its ~115 vector instructions per chunk and wave on per-lane predicates and offsets.
template <int CT, int OT, bool UPS, bool FAST = false>
__device__ __forceinline__ void ww_body(const WwArgs& a, const int split, const int yblk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // = component pair
  const int col = lane & 15, rq = lane >> 4;
  const int cb = yblk / a.nob, ob = yblk % a.nob;
  const int c0 = cb * CT * 16, o0 = ob * OT * 16;
  const int HW = a.H * a.W;
  const int Ht = a.H >> 1, Wt = a.W >> 1;

  // staging item of this thread: tile t of the chunk, channel slot chs: x channel c0 + chs and gy channel o0 + chs
  const int t = tid & 7, chs = tid >> 3;
  const int txl = t & (a.TBW - 1);
  const int tyl = (t >> a.lgTBW) & (a.TBH - 1);
  const int nl = t >> (a.lgTBW + a.lgTBH);
  const bool xch = (chs < CT * 16) && (c0 + chs < a.Cin);
  const bool ych = (chs < OT * 16) && (o0 + chs < a.Cout);
  const bool ledge = txl == 0, redge = txl == a.TBW - 1;
  const int HWx = UPS ? Ht * Wt : HW;
  const int xlane = UPS ? (nl * a.Cin + c0 + chs) * HWx + tyl * Wt + txl             // low-res pixel (TY, TX)
                        : (nl * a.Cin + c0 + chs) * HWx + (2 * tyl - 1) * a.W + 2 * txl;  // patch row 0, own pair
  const int ylane = (nl * a.Cout + o0 + chs) * HW + (2 * tyl) * a.W + 2 * txl;
  // LDS float offset of the item's first component pair: [cp][tile pair t>>1][swizzled channel][k-step t&1][parity]
  const int ldst = ((t >> 1) * CH + (chs ^ ((t >> 1) << 1))) * 4 + (t & 1) * 2;

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.gy), 0, (int)a.gy_bytes, 0x00020000);
  // FAST: lane parts of the byte offsets (channel plane + own pixel pair inside the chunk's 16 pixels); out of range = no such channel
  const unsigned fxP = xch ? (unsigned)(((c0 + chs) * HWx + (UPS ? t : 2 * t)) * 4) : 0x80000000u;
  const unsigned fyP = ych ? (unsigned)(((o0 + chs) * HW + 2 * t) * 4) : 0x80000000u;
  const int fdelta = t == 0 ? -4 : (UPS ? 4 : 8);  // halo pixel of an edge lane relative to its own pair

  f32x4 acc[2][CT][OT];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
      for (int j = 0; j < OT; ++j) acc[p][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x2 rP[4], rG[2];
  float rE[4];  // halo column of an edge lane (left OR right: a lane is at most one; a 1-tile-wide chunk has both outside)
  float bsum = 0.f;
  bool bnext = false;  // whether the gy tile in flight counts for the bias gradient

  // chunk `blk` (8 tiles): global loads into registers; tiles / rows / columns outside the image get an out-of-range offset and
  // read back as 0.0
  // the chunks of a slab are consecutive tile blocks: (bx, by, bn) of the next chunk to request is carried along (scalar selects)
  // instead of being divided out of the chunk index for every chunk (~80 scalar instructions per chunk and wave)
  int nq = 0, bx, by, bn;
  {
    const int b0 = split * a.per;
    bx = b0 % a.blocks_x;
    const int t2 = b0 / a.blocks_x;
    by = t2 % a.blocks_y;
    bn = t2 / a.blocks_y;
  }
  auto load_chunk = [&]() {  // the slab's next chunk (all-zero once past its end)
    const int blk = nq < a.per ? split * a.per + nq : a.nblk;
    if constexpr (FAST) {
      // scalar: the chunk is tile row `by` of image `bn`, tiles 8 bx .. 8 bx + 7
      const bool ok = blk < a.nblk;
      const unsigned vP = fxP + (unsigned)((UPS ? 8 : 16) * bx * 4);
      const bool ev = t == 0 ? bx > 0 : bx < a.blocks_x - 1;
      const unsigned vE = (xch && (t == 0 || t == 7) && ev) ? vP + (unsigned)fdelta : 0x80000000u;
      constexpr int NR = UPS ? 3 : 4;
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const bool rv = ok && (r == 0 ? by > 0 : (r == NR - 1 ? by < Ht - 1 : true));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, rv ? (int)a.x_bytes : 0, 0x00020000);
        const int so = UPS ? ((bn * a.Cin) * HWx + (by - 1 + r) * Wt) * 4 : ((bn * a.Cin) * HWx + (2 * by - 1 + r) * a.W) * 4;
        if constexpr (UPS) {
          const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)vP, so, 0));
          const float ve = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)vE, so, 0));
          const int rr = r == 0 ? 0 : (r == 1 ? 1 : 3);
          rP[rr] = f32x2{v, v};
          rE[rr] = ve;
          if (r == 1) { rP[2] = f32x2{v, v}; rE[2] = ve; }
        } else {
          rP[r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)vP, so, 0));
          rE[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)vE, so, 0));
        }
      }
      const __amdgpu_buffer_rsrc_t ys = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.gy), 0, ok ? (int)a.gy_bytes : 0, 0x00020000);
      const unsigned vY = fyP + (unsigned)(16 * bx * 4);
      const int sy = ((bn * a.Cout) * HW + (2 * by) * a.W) * 4;
      rG[0] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(ys, (int)vY, sy, 0));
      rG[1] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(ys, (int)vY, sy + a.W * 4, 0));
      bnext = bn < a.bias_n;
      ++nq;
      ++bx;
      const int wx = bx == a.blocks_x ? 1 : 0;
      bx = wx ? 0 : bx;
      by += wx;
      const int wy = by == a.blocks_y ? 1 : 0;
      by = wy ? 0 : by;
      bn += wy;
      return;
    }
    const int n = bn * a.TBN + nl, TY = by * a.TBH + tyl, TX = bx * a.TBW + txl;
    const bool ok = (blk < a.nblk) && (n < a.N) && (TY < Ht) && (TX < Wt);
    const int ux = UPS ? (bn * a.TBN * a.Cin) * HWx + (by * a.TBH) * Wt + bx * a.TBW
                       : (bn * a.TBN * a.Cin) * HWx + (2 * by * a.TBH) * a.W + 2 * bx * a.TBW;
    const int uy = (bn * a.TBN * a.Cout) * HW + (2 * by * a.TBH) * a.W + 2 * bx * a.TBW;
    const unsigned xo = (unsigned)(xlane + ux) * 4u;
    const bool xok = ok && xch;
    if constexpr (UPS) {
#pragma unroll
      for (int r3 = 0; r3 < 3; ++r3) {  // low-res rows TY-1, TY, TY+1 -> patch rows 0, (1, 2), 3
        const bool rv = xok && (r3 == 1 || (r3 == 0 ? TY > 0 : TY < Ht - 1));
        const unsigned o = xo + (unsigned)((r3 - 1) * Wt) * 4u;
        const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (int)(rv ? o : 0x80000000u), 0, 0));
        const unsigned oe = (rv && ledge && TX > 0) ? o - 4u : ((rv && redge && TX < Wt - 1) ? o + 4u : 0x80000000u);
        const float ve = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (int)oe, 0, 0));
        const int r = r3 == 0 ? 0 : (r3 == 1 ? 1 : 3);
        rP[r] = f32x2{v, v};
        rE[r] = ve;
        if (r3 == 1) { rP[2] = f32x2{v, v}; rE[2] = ve; }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool rv = xok && (r == 1 || r == 2 || (r == 0 ? TY > 0 : TY < Ht - 1));
        const unsigned o = xo + (unsigned)(r * a.W) * 4u;
        rP[r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xrs, (int)(rv ? o : 0x80000000u), 0, 0));
        const unsigned oe = (rv && ledge && TX > 0) ? o - 4u : ((rv && redge && TX < Wt - 1) ? o + 8u : 0x80000000u);
        rE[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (int)oe, 0, 0));
      }
    }
    const unsigned yo = (unsigned)(ylane + uy) * 4u;
    const bool yok = ok && ych;
    rG[0] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(yrs, (int)(yok ? yo : 0x80000000u), 0, 0));
    rG[1] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(yrs, (int)(yok ? yo + (unsigned)a.W * 4u : 0x80000000u), 0, 0));
    bnext = n < a.bias_n;
    ++nq;
    ++bx;
    const int wx = bx == a.blocks_x ? 1 : 0;
    bx = wx ? 0 : bx;
    by += wx;
    const int wy = by == a.blocks_y ? 1 : 0;
    by = wy ? 0 : by;
    bn += wy;
  };

  // registers -> transformed operand images of one stage
  auto store_chunk = [&](float* st) {
    {  // V = B^T d B, component slots of row i: [v0, v3 | v1, v2]  (see wino3x3.hip)
      f32x2 E[4], P[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        P[r] = rP[r];
        const float own_x = rP[r][0], own_y = rP[r][1];
        const float fl = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, own_y), 0x138, 0xf, 0xf, false));  // lane-1
        const float fr = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, own_x), 0x130, 0xf, 0xf, false));  // lane+1
        E[r] = f32x2{ledge ? (a.TBW > 1 ? rE[r] : 0.f) : fl, redge ? (a.TBW > 1 ? rE[r] : 0.f) : fr};
      }
      // one v_pk_add_f32 per result pair, swaps and negations in the operand modifiers (hipcc builds them with v_mov / v_xor)
      f32x2 UE[4], UP[4];
      UE[0] = pk_sub(E[0], E[2]);  UP[0] = pk_sub(P[0], P[2]);
      UE[1] = E[1] + E[2];         UP[1] = P[1] + P[2];
      UE[2] = pk_sub(E[2], E[1]);  UP[2] = pk_sub(P[2], P[1]);
      UE[3] = pk_sub(E[1], E[3]);  UP[3] = pk_sub(P[1], P[3]);
      float* dst = st + ldst;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x2 v03, v12;
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(v03) : "v"(UE[i]), "v"(UP[i]));  // (e0 - p1, p0 - e1)
        asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(v12) : "v"(UP[i]));             // (p0 + p1, p1 - p0)
        *reinterpret_cast<f32x2*>(dst + (2 * i) * (4 * CH * 4)) = v03;
        *reinterpret_cast<f32x2*>(dst + (2 * i + 1) * (4 * CH * 4)) = v12;
      }
    }
    {  // Y = A t A^T with A = [[1,0],[1,1],[1,-1],[0,-1]], same slot order: row i -> [y0, y3 | y1, y2]
      const f32x2 t0 = rG[0], t1 = rG[1];
      if (bnext) bsum += (t0[0] + t0[1]) + (t1[0] + t1[1]);
      f32x2 R[4];
      R[0] = t0;
      R[1] = t0 + t1;
      R[2] = pk_sub(t0, t1);
      R[3] = t1;  // stands for -t1: the sign is folded into the modifiers below
      float* dst = st + IMG + ldst;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x2 y03, y12;
        if (i < 3) {
          asm("v_pk_mul_f32 %0, %1, %2 neg_hi:[1,0]" : "=v"(y03) : "v"(R[i]), "v"(f32x2{1.f, 1.f}));                               // (r0, -r1)
          asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(y12) : "v"(R[i]));           // (r0 + r1, r0 - r1)
        } else {
          asm("v_pk_mul_f32 %0, %1, %2 neg_lo:[1,0]" : "=v"(y03) : "v"(R[i]), "v"(f32x2{1.f, 1.f}));                               // (-t0, t1)
          asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[1,1] neg_hi:[1,0]" : "=v"(y12) : "v"(R[i]));           // (-t0 - t1, -t0 + t1)
        }
        *reinterpret_cast<f32x2*>(dst + (2 * i) * (4 * CH * 4)) = y03;
        *reinterpret_cast<f32x2*>(dst + (2 * i + 1) * (4 * CH * 4)) = y12;
      }
    }
  };

  // The operand reads of a chunk are issued first (pinned by a scheduling fence), the next chunk's transform + LDS writes and
  // the loads of the one after run while they are in flight, and the MFMAs come last: +1..3 % over reads placed directly in
  // front of the MFMAs, where the matrix pipe waits out an LDS round trip per chunk.
  f32x4 av[CT], bv[OT];  // {par0 k0, par1 k0, par0 k1, par1 k1}
  auto read_operands = [&](const float* st) {
    const float* vb = st + (wave * 4 + rq) * (CH * 4);
    const float* yb = vb + IMG;
#pragma unroll
    for (int i = 0; i < CT; ++i) av[i] = *reinterpret_cast<const f32x4*>(vb + ((i * 16 + col) ^ (rq << 1)) * 4);
#pragma unroll
    for (int j = 0; j < OT; ++j) bv[j] = *reinterpret_cast<const f32x4*>(yb + ((j * 16 + col) ^ (rq << 1)) * 4);
  };
  auto mma_chunk = [&]() {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
          for (int j = 0; j < OT; ++j)
            acc[p][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][ks * 2 + p], bv[j][ks * 2 + p], acc[p][i][j], 0, 0, 0);
  };

  // pipeline: iteration q computes chunk q from stage q&1, writes chunk q+1 (registers) into the other stage and issues the
  // loads of chunk q+2; chunks past the slab (or past the tensor) are all-zero and add nothing
  load_chunk();
  store_chunk(smem);
  load_chunk();
  __syncthreads();
  for (int q = 0; q < a.per; ++q) {
    float* cur = smem + (q & 1) * STAGE;
    float* nxt = smem + ((q + 1) & 1) * STAGE;
    read_operands(cur);
    __builtin_amdgcn_sched_barrier(0);
    store_chunk(nxt);
    load_chunk();
    mma_chunk();
    __syncthreads();
  }

  // Slab of this split: dW_split = G^T M G per (c, o), 9 planes [split][k][c][o] -- the transform is linear, so it is applied per
  // split and the reduce kernel only sums (9/16 of the bytes, which is what the small-map layers' weight gradients cost: their
  // slabs are larger than their inputs).  A (c, o) pair's 16 components sit in 8 different waves: they meet in LDS (the two
  // stages are free now), 32 in-channels x 64 out-channels x 16 slots = exactly its 128 KB, in two passes over the in-channel
  // tiles; the out-channel tile index is XOR-ed with the row group so that the four row groups of a wave hit disjoint banks.
  float* G = smem;  // [slot 16][cc 32][o 64]
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    __syncthreads();  // MFMA loop / previous pass done with the buffer
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int i = 2 * h + ii;
        if (i < CT) {
#pragma unroll
          for (int j = 0; j < OT; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
              G[((2 * wave + p) * 32 + ii * 16 + rq * 4 + g) * 64 + ((j ^ rq) * 16 + col)] = acc[p][i][j][g];
        }
      }
    __syncthreads();
    constexpr int SL[4] = {0, 2, 3, 1};  // slot of column nu within a row of components
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      const int cc = (tid >> 6) + 8 * k4;  // wave-uniform row of the half block
      const int ol = tid & 63;
      const int i = 2 * h + (cc >> 4);
      const int c = c0 + i * 16 + (cc & 15), o = o0 + ol;
      if (i < CT && ol < OT * 16 && c < a.CinP && o < a.CoutP) {
        const int osw = ((ol >> 4) ^ ((cc >> 2) & 3)) * 16 + (ol & 15);
        float M[4][4];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi)
#pragma unroll
          for (int nu = 0; nu < 4; ++nu) M[xi][nu] = G[((4 * xi + SL[nu]) * 32 + cc) * 64 + osw];
        float hh[3][4];  // G^T M
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
          hh[0][nu] = M[0][nu] + 0.5f * (M[1][nu] + M[2][nu]);
          hh[1][nu] = 0.5f * (M[1][nu] - M[2][nu]);
          hh[2][nu] = 0.5f * (M[1][nu] + M[2][nu]) + M[3][nu];
        }
        float* sl = a.slab + (size_t)split * 9 * a.CinP * a.CoutP + (size_t)c * a.CoutP + o;
        const size_t plane = (size_t)a.CinP * a.CoutP;
#pragma unroll
        for (int aa = 0; aa < 3; ++aa) {
          sl[(size_t)(aa * 3 + 0) * plane] = hh[aa][0] + 0.5f * (hh[aa][1] + hh[aa][2]);
          sl[(size_t)(aa * 3 + 1) * plane] = 0.5f * (hh[aa][1] - hh[aa][2]);
          sl[(size_t)(aa * 3 + 2) * plane] = 0.5f * (hh[aa][1] + hh[aa][2]) + hh[aa][3];
        }
      }
    }
  }
  // bias gradient: the 8 tile lanes of a channel slot, then one value per (split, out-channel); in-channel block 0 only
  bsum += __shfl_xor(bsum, 1);
  bsum += __shfl_xor(bsum, 2);
  bsum += __shfl_xor(bsum, 4);
  if (cb == 0 && t == 0 && chs < OT * 16 && o0 + chs < a.CoutP) a.slab_b[(size_t)split * a.CoutP + o0 + chs] = bsum;
}

template <int CT, int OT, bool UPS, bool FAST>
__global__ void __launch_bounds__(512) wino_wgrad_mfma(const WwArgs a) {
  ww_body<CT, OT, UPS, FAST>(a, blockIdx.x, blockIdx.y);
}

// ---- several layers in ONE launch.  At small batch / on small maps a layer's weight gradient is a 10-30 us launch of a few dozen
// to 256 workgroups, a step has ten of them back to back, and every one is split over ~256 workgroups just to fill the chip, each
// split writing (and the reduce re-reading) a full 9-tap slab of the filter.  The small layers of a sweep share a launch: workgroup
// b looks its layer up in a table of at most WW_GROUP entries (first[] = prefix sums of workgroups) and runs that layer's body --
// the block shapes <CT, OT, UPS> of layers that are ever small (>= 96 channels: 3 or 4 channel tiles per block) are all inlined
// here; the host sizes the splits so that the GROUP fills the chip once (mg_wino3x3_wgrad_partial_multi).
constexpr int WW_GROUP = 16;
struct WwGroup {
  int n;
  int first[WW_GROUP + 1];
  int nsplit[WW_GROUP];
  int var[WW_GROUP];  // (CT * 10 + OT) * 2 + UPS
  WwArgs a[WW_GROUP];
};

__global__ void __launch_bounds__(512) wino_wgrad_group_mfma(const WwGroup g) {
  int j = 0;
#pragma unroll 1
  while (j + 1 < g.n && (int)blockIdx.x >= g.first[j + 1]) ++j;
  const int local = (int)blockIdx.x - g.first[j];
  const int ns = g.nsplit[j];
  const int split = local % ns, yblk = local / ns;
  switch (g.var[j]) {
    case 66: ww_body<3, 3, false>(g.a[j], split, yblk); break;
    case 67: ww_body<3, 3, true>(g.a[j], split, yblk); break;
    case 68: ww_body<3, 4, false>(g.a[j], split, yblk); break;
    case 69: ww_body<3, 4, true>(g.a[j], split, yblk); break;
    case 86: ww_body<4, 3, false>(g.a[j], split, yblk); break;
    case 87: ww_body<4, 3, true>(g.a[j], split, yblk); break;
    case 88: ww_body<4, 4, false>(g.a[j], split, yblk); break;
    case 89: ww_body<4, 4, true>(g.a[j], split, yblk); break;
    default: break;
  }
}

// ---- narrow blocks (CT + OT <= 4 channel tiles): same algorithm and LDS layout as wino_wgrad_mfma above, re-balanced for blocks
// whose per-chunk work (8 tiles x a few dozen channels) is far below an HBM round trip -- see the comments inside.
// UPS: x is (N, Cin, H/2, W/2) and the convolution input is its nearest x2 up-sampling (generator.py:24-25): the 4x4 patch of tile
// (TY, TX) is then the 3x3 low-res neighbourhood with the centre row / column doubled -- one dword per row and lane.
template <int CT, int OT, bool UPS, bool FAST>
__global__ void __launch_bounds__(512, 4) wino_wgrad_narrow_mfma(const WwArgs a) {
  static_assert(CT + OT <= 4, "the narrow form: at most four channel tiles in all");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // = component pair
  const int col = lane & 15, rq = lane >> 4;
  const int cb = blockIdx.y / a.nob, ob = blockIdx.y % a.nob;
  const int c0 = cb * CT * 16, o0 = ob * OT * 16;
  const int split = blockIdx.x;
  const int HW = a.H * a.W;
  const int Ht = a.H >> 1, Wt = a.W >> 1;

  // staging item of this thread: tile t of the chunk and a channel slot -- x channel c0 + xs and gy channel o0 + ys.  With up to
  // four channel tiles in all (CT + OT <= 4: the 16/32/48-channel layers at 256x256 and 512x512, where a run of the reference
  // spends its time) the x slots and the gy slots sit on DIFFERENT waves (x: waves 0 .. 2CT-1, gy: the next 2OT), so the two
  // transforms run side by side instead of one after the other on the first waves while the rest wait at the barrier; wider
  // blocks use every wave for both (slot = tid / 8).  Waves without slots skip loads, transform and LDS writes (wave-uniform).
  constexpr int YOFF = CT * 16;
  const int t = tid & 7, xs = tid >> 3, ys = (tid >> 3) - YOFF;
  const int txl = t & (a.TBW - 1);
  const int tyl = (t >> a.lgTBW) & (a.TBH - 1);
  const int nl = t >> (a.lgTBW + a.lgTBH);
  const bool xch = (xs < CT * 16) && (c0 + xs < a.Cin);
  const bool ych = (ys >= 0) && (ys < OT * 16) && (o0 + ys < a.Cout);
  const bool ledge = txl == 0, redge = txl == a.TBW - 1;
  // (wider blocks keep wino_wgrad_mfma: every wave stages both operands, slots past the block read zeros through the bounds check;
  // this kernel's structure costs the 48..64-channel layers of level 5 3-7 %, tools/ab_wgrad.py)
  const bool xw = wave * 8 < CT * 16, yw = (wave * 8 >= YOFF) && (wave * 8 < YOFF + OT * 16);
  const int HWx = UPS ? Ht * Wt : HW;
  const int xlane = UPS ? (nl * a.Cin + c0 + xs) * HWx + tyl * Wt + txl             // low-res pixel (TY, TX)
                        : (nl * a.Cin + c0 + xs) * HWx + (2 * tyl - 1) * a.W + 2 * txl;  // patch row 0, own pair
  const int ylane = (nl * a.Cout + o0 + ys) * HW + (2 * tyl) * a.W + 2 * txl;
  // LDS float offset of the item's first component pair: [cp][tile pair t>>1][swizzled channel][k-step t&1][parity]
  const int ldst = ((t >> 1) * CH + (xs ^ ((t >> 1) << 1))) * 4 + (t & 1) * 2;
  // ONE 64-slot operand image per stage holds both operands -- x in slots 0 .. 16CT-1, gy behind them (CT + OT <= 4) -- so a
  // stage is 32 KB, the workgroup 64 KB, and TWO workgroups share a CU: with blocks this thin a workgroup spends most of a chunk
  // waiting (barrier, LDS and memory round trips), and the second one fills those gaps

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.gy), 0, (int)a.gy_bytes, 0x00020000);
  // FAST (see ww_body): lane parts of the byte offsets, fixed for the kernel
  const unsigned fxP = xch ? (unsigned)(((c0 + xs) * HWx + (UPS ? t : 2 * t)) * 4) : 0x80000000u;
  const unsigned fyP = ych ? (unsigned)(((o0 + ys) * HW + 2 * t) * 4) : 0x80000000u;
  const int fdelta = t == 0 ? -4 : (UPS ? 4 : 8);

  f32x4 acc[2][CT][OT];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
      for (int j = 0; j < OT; ++j) acc[p][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Chunks in flight: a chunk is 8 tiles x (CT + OT) x 16 channels -- with one or two channel tiles that is a few hundred cycles
  // of work, an order of magnitude less than an HBM round trip, so the loads of such a block run PD chunks ahead in PD register
  // sets (16 registers each; measured on 16 x 32 channels at 512x512: 3 500 cycles per chunk with one set).
  constexpr int PD = CT * OT <= 1 ? 6 : (CT * OT <= 2 ? 5 : (CT * OT <= 3 ? 4 : 3));  // (128 registers: two workgroups per CU)
  struct Regs {
    f32x2 rP[4], rG[2];
    float rE[4];  // halo column of an edge lane (left OR right: a lane is at most one; a 1-tile-wide chunk has both outside)
    bool bnext;   // whether the gy tile in flight counts for the bias gradient
  };
  Regs R[PD];
  float bsum = 0.f;

  // chunk `blk` (8 tiles): global loads into registers; tiles / rows / columns outside the image get an out-of-range offset and
  // read back as 0.0
  // chunks of a slab are consecutive tile blocks: the (bx, by, bn) of the next chunk to load is carried along instead of being
  // divided out of the chunk index for every chunk (three integer divisions = ~80 scalar instructions per chunk and wave)
  int nq = 0;  // chunks of this slab requested so far
  int bx, by, bn;
  {
    const int blk = split * a.per;
    bx = blk % a.blocks_x;
    const int t2 = blk / a.blocks_x;
    by = t2 % a.blocks_y;
    bn = t2 / a.blocks_y;
  }
  auto load_chunk = [&](Regs& rr, auto role_) __attribute__((always_inline)) {  // the slab's next chunk (all-zero once past its end)
    constexpr int ROLE = decltype(role_)::value;  // bit 0: this wave stages x slots, bit 1: gy slots
    auto& rP = rr.rP; auto& rG = rr.rG; auto& rE = rr.rE; bool& bnext = rr.bnext;
    const int blk = nq < a.per ? split * a.per + nq : a.nblk;
    if constexpr (FAST) {
      const bool ok = blk < a.nblk;
      if constexpr ((ROLE & 1) != 0) {
        const unsigned vP = fxP + (unsigned)((UPS ? 8 : 16) * bx * 4);
        const bool ev = t == 0 ? bx > 0 : bx < a.blocks_x - 1;
        const unsigned vE = (xch && (t == 0 || t == 7) && ev) ? vP + (unsigned)fdelta : 0x80000000u;
        constexpr int NR = UPS ? 3 : 4;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const bool rv = ok && (r == 0 ? by > 0 : (r == NR - 1 ? by < Ht - 1 : true));
          const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, rv ? (int)a.x_bytes : 0, 0x00020000);
          const int so = UPS ? ((bn * a.Cin) * HWx + (by - 1 + r) * Wt) * 4 : ((bn * a.Cin) * HWx + (2 * by - 1 + r) * a.W) * 4;
          if constexpr (UPS) {
            const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)vP, so, 0));
            const float ve = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)vE, so, 0));
            const int r4 = r == 0 ? 0 : (r == 1 ? 1 : 3);
            rP[r4] = f32x2{v, v};
            rE[r4] = ve;
            if (r == 1) { rP[2] = f32x2{v, v}; rE[2] = ve; }
          } else {
            rP[r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)vP, so, 0));
            rE[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)vE, so, 0));
          }
        }
      }
      if constexpr ((ROLE & 2) != 0) {
        const __amdgpu_buffer_rsrc_t ysr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.gy), 0, ok ? (int)a.gy_bytes : 0, 0x00020000);
        const unsigned vY = fyP + (unsigned)(16 * bx * 4);
        const int sy = ((bn * a.Cout) * HW + (2 * by) * a.W) * 4;
        rG[0] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(ysr, (int)vY, sy, 0));
        rG[1] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(ysr, (int)vY, sy + a.W * 4, 0));
        bnext = bn < a.bias_n;
      }
      ++nq;
      ++bx;
      const int wx = bx == a.blocks_x ? 1 : 0;
      bx = wx ? 0 : bx;
      by += wx;
      const int wy = by == a.blocks_y ? 1 : 0;
      by = wy ? 0 : by;
      bn += wy;
      return;
    }
    const int n = bn * a.TBN + nl, TY = by * a.TBH + tyl, TX = bx * a.TBW + txl;
    const bool ok = (blk < a.nblk) && (n < a.N) && (TY < Ht) && (TX < Wt);
    const int ux = UPS ? (bn * a.TBN * a.Cin) * HWx + (by * a.TBH) * Wt + bx * a.TBW
                       : (bn * a.TBN * a.Cin) * HWx + (2 * by * a.TBH) * a.W + 2 * bx * a.TBW;
    const int uy = (bn * a.TBN * a.Cout) * HW + (2 * by * a.TBH) * a.W + 2 * bx * a.TBW;
    const unsigned xo = (unsigned)(xlane + ux) * 4u;
    const bool xok = ok && xch;
    if constexpr (!(ROLE & 1)) {
    } else if constexpr (UPS) {
#pragma unroll
      for (int r3 = 0; r3 < 3; ++r3) {  // low-res rows TY-1, TY, TY+1 -> patch rows 0, (1, 2), 3
        const bool rv = xok && (r3 == 1 || (r3 == 0 ? TY > 0 : TY < Ht - 1));
        const unsigned o = xo + (unsigned)((r3 - 1) * Wt) * 4u;
        const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (int)(rv ? o : 0x80000000u), 0, 0));
        const unsigned oe = (rv && ledge && TX > 0) ? o - 4u : ((rv && redge && TX < Wt - 1) ? o + 4u : 0x80000000u);
        const float ve = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (int)oe, 0, 0));
        const int r = r3 == 0 ? 0 : (r3 == 1 ? 1 : 3);
        rP[r] = f32x2{v, v};
        rE[r] = ve;
        if (r3 == 1) { rP[2] = f32x2{v, v}; rE[2] = ve; }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool rv = xok && (r == 1 || r == 2 || (r == 0 ? TY > 0 : TY < Ht - 1));
        const unsigned o = xo + (unsigned)(r * a.W) * 4u;
        rP[r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xrs, (int)(rv ? o : 0x80000000u), 0, 0));
        const unsigned oe = (rv && ledge && TX > 0) ? o - 4u : ((rv && redge && TX < Wt - 1) ? o + 8u : 0x80000000u);
        rE[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (int)oe, 0, 0));
      }
    }
    if constexpr ((ROLE & 2) != 0) {
      const unsigned yo = (unsigned)(ylane + uy) * 4u;
      const bool yok = ok && ych;
      rG[0] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(yrs, (int)(yok ? yo : 0x80000000u), 0, 0));
      rG[1] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(yrs, (int)(yok ? yo + (unsigned)a.W * 4u : 0x80000000u), 0, 0));
      bnext = n < a.bias_n;
    }
    ++nq;  // next tile block, branch-free (scalar selects)
    ++bx;
    const int wx = bx == a.blocks_x ? 1 : 0;
    bx = wx ? 0 : bx;
    by += wx;
    const int wy = by == a.blocks_y ? 1 : 0;
    by = wy ? 0 : by;
    bn += wy;
  };

  // registers -> transformed operand images of one stage
  auto store_chunk = [&](float* st, const Regs& rr, auto role_) __attribute__((always_inline)) {
    constexpr int ROLE = decltype(role_)::value;
    const auto& rP = rr.rP; const auto& rG = rr.rG; const auto& rE = rr.rE; const bool bnext = rr.bnext;
    if constexpr ((ROLE & 1) != 0) {  // V = B^T d B, component slots of row i: [v0, v3 | v1, v2]  (see wino3x3.hip)
      f32x2 E[4], P[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        P[r] = rP[r];
        const float own_x = rP[r][0], own_y = rP[r][1];
        const float fl = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, own_y), 0x138, 0xf, 0xf, false));  // lane-1
        const float fr = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, own_x), 0x130, 0xf, 0xf, false));  // lane+1
        E[r] = f32x2{ledge ? (a.TBW > 1 ? rE[r] : 0.f) : fl, redge ? (a.TBW > 1 ? rE[r] : 0.f) : fr};
      }
      // one v_pk_add_f32 per result pair, swaps and negations in the operand modifiers (hipcc builds them with v_mov / v_xor)
      f32x2 UE[4], UP[4];
      UE[0] = pk_sub(E[0], E[2]);  UP[0] = pk_sub(P[0], P[2]);
      UE[1] = E[1] + E[2];         UP[1] = P[1] + P[2];
      UE[2] = pk_sub(E[2], E[1]);  UP[2] = pk_sub(P[2], P[1]);
      UE[3] = pk_sub(E[1], E[3]);  UP[3] = pk_sub(P[1], P[3]);
      float* dst = st + ldst;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x2 v03, v12;
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(v03) : "v"(UE[i]), "v"(UP[i]));  // (e0 - p1, p0 - e1)
        asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(v12) : "v"(UP[i]));             // (p0 + p1, p1 - p0)
        *reinterpret_cast<f32x2*>(dst + (2 * i) * (4 * CH * 4)) = v03;
        *reinterpret_cast<f32x2*>(dst + (2 * i + 1) * (4 * CH * 4)) = v12;
      }
    }
    if constexpr ((ROLE & 2) != 0) {  // Y = A t A^T with A = [[1,0],[1,1],[1,-1],[0,-1]], same slot order: row i -> [y0, y3 | y1, y2]
      const f32x2 t0 = rG[0], t1 = rG[1];
      if (bnext) bsum += (t0[0] + t0[1]) + (t1[0] + t1[1]);
      f32x2 R[4];
      R[0] = t0;
      R[1] = t0 + t1;
      R[2] = pk_sub(t0, t1);
      R[3] = t1;  // stands for -t1: the sign is folded into the modifiers below
      float* dst = st + ldst;  // slot tid / 8 = 16 CT + ys
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x2 y03, y12;
        if (i < 3) {
          asm("v_pk_mul_f32 %0, %1, %2 neg_hi:[1,0]" : "=v"(y03) : "v"(R[i]), "v"(f32x2{1.f, 1.f}));                               // (r0, -r1)
          asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(y12) : "v"(R[i]));           // (r0 + r1, r0 - r1)
        } else {
          asm("v_pk_mul_f32 %0, %1, %2 neg_lo:[1,0]" : "=v"(y03) : "v"(R[i]), "v"(f32x2{1.f, 1.f}));                               // (-t0, t1)
          asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[1,1] neg_hi:[1,0]" : "=v"(y12) : "v"(R[i]));           // (-t0 - t1, -t0 + t1)
        }
        *reinterpret_cast<f32x2*>(dst + (2 * i) * (4 * CH * 4)) = y03;
        *reinterpret_cast<f32x2*>(dst + (2 * i + 1) * (4 * CH * 4)) = y12;
      }
    }
  };

  // The operand reads of a chunk are issued first (pinned by a scheduling fence), the next chunk's transform + LDS writes and
  // the loads of the one after run while they are in flight, and the MFMAs come last: +1..3 % over reads placed directly in
  // front of the MFMAs, where the matrix pipe waits out an LDS round trip per chunk.
  f32x4 av[CT], bv[OT];  // {par0 k0, par1 k0, par0 k1, par1 k1}
  auto read_operands = [&](const float* st) {
    const float* vb = st + (wave * 4 + rq) * (CH * 4);
#pragma unroll
    for (int i = 0; i < CT; ++i) av[i] = *reinterpret_cast<const f32x4*>(vb + ((i * 16 + col) ^ (rq << 1)) * 4);
#pragma unroll
    for (int j = 0; j < OT; ++j) bv[j] = *reinterpret_cast<const f32x4*>(vb + ((YOFF + j * 16 + col) ^ (rq << 1)) * 4);
  };
  auto mma_chunk = [&]() {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
          for (int j = 0; j < OT; ++j)
            acc[p][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][ks * 2 + p], bv[j][ks * 2 + p], acc[p][i][j], 0, 0, 0);
  };

  // pipeline: iteration q computes chunk q from stage q&1, writes chunk q+1 (registers) into the other stage and issues the
  // loads of chunk q+2; chunks past the slab (or past the tensor) are all-zero and add nothing
  // One copy of the tile loop per staging role, chosen ONCE per wave: inside a copy there is no control flow around the loads, so
  // hipcc's s_waitcnt placement stays exact (vmcnt(N) for the oldest set only) and the PD sets really are in flight -- with
  // `if (this wave stages x)` inside the loop it waited vmcnt(0)/(1) in front of every transform.
  auto tile_loop = [&](auto role_) __attribute__((always_inline)) {
    load_chunk(R[0], role_);
    store_chunk(smem, R[0], role_);
#pragma unroll
    for (int i = 0; i < PD; ++i) load_chunk(R[i], role_);
    __syncthreads();
    for (int q0 = 0; q0 < a.per; q0 += PD) {  // (the last round may run up to PD-1 all-zero chunks)
#pragma unroll
      for (int i = 0; i < PD; ++i) {
        const int q = q0 + i;
        float* cur = smem + (q & 1) * IMG;
        float* nxt = smem + ((q + 1) & 1) * IMG;
        read_operands(cur);
        __builtin_amdgcn_sched_barrier(0);
        store_chunk(nxt, R[i], role_);  // chunk q+1
        load_chunk(R[i], role_);        // chunk q+1+PD takes its place
        mma_chunk();
        __syncthreads();
      }
    }
  };
  if (xw) tile_loop(std::integral_constant<int, 1>{});
  else if (yw) tile_loop(std::integral_constant<int, 2>{});
  else tile_loop(std::integral_constant<int, 0>{});

  // Slab of this split: dW_split = G^T M G per (c, o), 9 planes [split][k][c][o] -- the transform is linear, so it is applied per
  // split and the reduce kernel only sums (9/16 of the bytes, which is what the small-map layers' weight gradients cost: their
  // slabs are larger than their inputs).  A (c, o) pair's 16 components sit in 8 different waves: they meet in LDS (the two
  // stages are free now), 32 in-channels x 64 out-channels x 16 slots = exactly its 128 KB, in two passes over the in-channel
  // tiles; the out-channel tile index is XOR-ed with the row group so that the four row groups of a wave hit disjoint banks.
  float* G = smem;  // [slot 16][cc 16][o 64] = the 64 KB of the two stages: one in-channel tile per pass
#pragma unroll
  for (int h = 0; h < CT; ++h) {
    __syncthreads();  // MFMA loop / previous pass done with the buffer
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int j = 0; j < OT; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) G[((2 * wave + p) * 16 + rq * 4 + g) * 64 + ((j ^ rq) * 16 + col)] = acc[p][h][j][g];
    __syncthreads();
    constexpr int SL[4] = {0, 2, 3, 1};  // slot of column nu within a row of components
#pragma unroll
    for (int k4 = 0; k4 < 2; ++k4) {
      const int cc = (tid >> 6) + 8 * k4;  // wave-uniform row of the in-channel tile
      const int ol = tid & 63;
      const int c = c0 + h * 16 + cc, o = o0 + ol;
      if (ol < OT * 16 && c < a.CinP && o < a.CoutP) {
        const int osw = ((ol >> 4) ^ ((cc >> 2) & 3)) * 16 + (ol & 15);
        float M[4][4];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi)
#pragma unroll
          for (int nu = 0; nu < 4; ++nu) M[xi][nu] = G[((4 * xi + SL[nu]) * 16 + cc) * 64 + osw];
        float hh[3][4];  // G^T M
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
          hh[0][nu] = M[0][nu] + 0.5f * (M[1][nu] + M[2][nu]);
          hh[1][nu] = 0.5f * (M[1][nu] - M[2][nu]);
          hh[2][nu] = 0.5f * (M[1][nu] + M[2][nu]) + M[3][nu];
        }
        float* sl = a.slab + (size_t)split * 9 * a.CinP * a.CoutP + (size_t)c * a.CoutP + o;
        const size_t plane = (size_t)a.CinP * a.CoutP;
#pragma unroll
        for (int aa = 0; aa < 3; ++aa) {
          sl[(size_t)(aa * 3 + 0) * plane] = hh[aa][0] + 0.5f * (hh[aa][1] + hh[aa][2]);
          sl[(size_t)(aa * 3 + 1) * plane] = 0.5f * (hh[aa][1] - hh[aa][2]);
          sl[(size_t)(aa * 3 + 2) * plane] = 0.5f * (hh[aa][1] + hh[aa][2]) + hh[aa][3];
        }
      }
    }
  }
  // bias gradient: the 8 tile lanes of a channel slot, then one value per (split, out-channel); in-channel block 0 only
  bsum += __shfl_xor(bsum, 1);
  bsum += __shfl_xor(bsum, 2);
  bsum += __shfl_xor(bsum, 4);
  if (cb == 0 && t == 0 && ys >= 0 && ys < OT * 16 && o0 + ys < a.CoutP) a.slab_b[(size_t)split * a.CoutP + o0 + ys] = bsum;
}

// Sum the split-K slabs (already transformed to the 9 taps by the partial kernel) in a fixed order.  Block = EL consecutive (c, o)
// pairs x KL split-lanes, EL * KL = 512 (a thread sums every KL-th split of the 9 taps, four slabs of loads in flight: this is a
// pure latency problem), LDS-combined as a fixed tree => deterministic.  KL = 8 normally; 32 for the layers with few filter
// elements and hundreds of splits (16..48-channel layers on 256x256 / 512x512 maps): at 8 lanes their 8 workgroups walked 64 slabs
// each while the chip waited -- and their bias sum, 512 loads four at a time in ONE thread per out-channel, took 86 us at level 7.
// The bias gradient is summed the same way by the threads of in-channel 0.
__host__ __device__ inline int ww_reduce_lanes(int nsplit, int total) { return (nsplit >= 128 && total <= 8192) ? 32 : 8; }

template <int KL>
__device__ __forceinline__ void wino_wgrad_reduce_lanes(const float* __restrict__ slab, const float* __restrict__ slab_b, int nsplit,
                                                        float* __restrict__ gw, float* __restrict__ gb, int Cout, int Cin,
                                                        int CoutP, int CinP, int accumulate, int block, float (*red)[512]) {
  constexpr int EL = 512 / KL;
  const int el = threadIdx.x % EL, kl = threadIdx.x / EL;
  const int e = block * EL + el;  // e = c * CoutP + o over the padded block
  const int total = CinP * CoutP;
  float m[10];  // 9 taps + the bias gradient (threads of in-channel 0: e = o)
#pragma unroll
  for (int s = 0; s < 10; ++s) m[s] = 0.f;
  if (e < total) {
    // four slabs' worth of loads (36) in flight, then their adds in slab order: as a plain loop hipcc waits for each slab's nine
    // loads before it requests the next one -- nsplit / KL memory round trips in a row
    int k = kl;
    for (; k + 3 * KL < nsplit; k += 4 * KL) {
      float v[4][9];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* src = slab + (size_t)(k + KL * u) * 9 * total + e;
#pragma unroll
        for (int s = 0; s < 9; ++s) v[u][s] = src[(size_t)s * total];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int s = 0; s < 9; ++s) m[s] += v[u][s];
    }
    for (; k < nsplit; k += KL) {
      const float* src = slab + (size_t)k * 9 * total + e;
#pragma unroll
      for (int s = 0; s < 9; ++s) m[s] += src[(size_t)s * total];
    }
    if (gb != nullptr && e < CoutP) {  // c == 0: this thread's share of the bias slabs of out-channel o = e
      int kb = kl;
      for (; kb + 7 * KL < nsplit; kb += 8 * KL) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = slab_b[(size_t)(kb + KL * u) * CoutP + e];
#pragma unroll
        for (int u = 0; u < 8; ++u) m[9] += v[u];
      }
      for (; kb < nsplit; kb += KL) m[9] += slab_b[(size_t)kb * CoutP + e];
    }
  }
#pragma unroll
  for (int s = 0; s < 10; ++s) red[s][kl * EL + el] = m[s];
  __syncthreads();
  if (kl != 0 || e >= total) return;
  const int c = e / CoutP, o = e % CoutP;
  auto lanes = [&](int s) {  // fixed pairwise tree over the KL split-lanes
    float t[KL];
#pragma unroll
    for (int q = 0; q < KL; ++q) t[q] = red[s][q * EL + el];
#pragma unroll
    for (int w = 1; w < KL; w *= 2)
#pragma unroll
      for (int q = 0; q < KL; q += 2 * w) t[q] += t[q + w];
    return t[0];
  };
  if (c < Cin && o < Cout) {
    float* dst = gw + ((size_t)o * Cin + c) * 9;
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      const float w = lanes(s);
      dst[s] = accumulate ? dst[s] + w : w;
    }
  }
  if (gb != nullptr && c == 0 && o < Cout) {
    const float sb = lanes(9);
    gb[o] = accumulate ? gb[o] + sb : sb;
  }
}

__device__ __forceinline__ void wino_wgrad_reduce_body(const float* __restrict__ slab, const float* __restrict__ slab_b, int nsplit,
                                                       float* __restrict__ gw, float* __restrict__ gb, int Cout, int Cin,
                                                       int CoutP, int CinP, int accumulate, int block) {
  __shared__ float red[10][512];
  if (ww_reduce_lanes(nsplit, CinP * CoutP) == 32)
    wino_wgrad_reduce_lanes<32>(slab, slab_b, nsplit, gw, gb, Cout, Cin, CoutP, CinP, accumulate, block, red);
  else
    wino_wgrad_reduce_lanes<8>(slab, slab_b, nsplit, gw, gb, Cout, Cin, CoutP, CinP, accumulate, block, red);
}

// One launch for the reduce of SEVERAL layers (the weight-gradient sweep of an update ends with one of these per layer: 8 launches of
// ~20 us each, latency-bound, at level 5): jobs travel by value, blockIdx.y selects the job, blocks past a job's extent return.
constexpr int WW_JOBS = 40;
struct WwJobs {
  int n;
  int first[WW_JOBS + 1];  // prefix sums of the jobs' block counts: workgroup b belongs to the job with first[i] <= b < first[i + 1]
  mg_wgrad_job_t j[WW_JOBS];
};
__global__ void __launch_bounds__(512) wino_wgrad_reduce_multi(const WwJobs jobs) {
  // (a 2-D grid of "largest job x jobs" launched ~10x the workgroups a sweep needs: 19 us for the slabs of level 3)
  int i = 0;
#pragma unroll 1
  while (i + 1 < jobs.n && (int)blockIdx.x >= jobs.first[i + 1]) ++i;
  const mg_wgrad_job_t j = jobs.j[i];
  wino_wgrad_reduce_body(j.slab, j.slab_b, j.nsplit, j.gw, j.gb, j.Cout, j.Cin, j.CoutP, j.CinP, j.accumulate,
                         (int)blockIdx.x - jobs.first[i]);
}

struct WwPlan {
  WwArgs a;
  int CT, OT, ncb, nsplit;
  size_t ws_floats;
};

int blocks_of(int tiles) {  // channel tiles -> blocks of <= 4 tiles, balanced
  return mg_cdiv(tiles, 4);
}

void plan_ww(int N, int Cin, int Cout, int H, int W, WwPlan& pl) {
  WwArgs& a = pl.a;
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  const int Ht = H / 2, Wt = W / 2;
  a.TBW = mg_pow2_ceil(Wt) < KT ? mg_pow2_ceil(Wt) : KT;
  a.TBH = mg_pow2_ceil(Ht) < KT / a.TBW ? mg_pow2_ceil(Ht) : KT / a.TBW;
  a.TBN = KT / (a.TBW * a.TBH);
  a.lgTBW = mg_ilog2(a.TBW); a.lgTBH = mg_ilog2(a.TBH);
  a.blocks_x = mg_cdiv(Wt, a.TBW); a.blocks_y = mg_cdiv(Ht, a.TBH); a.blocks_n = mg_cdiv(N, a.TBN);
  a.nblk = a.blocks_x * a.blocks_y * a.blocks_n;
  const int ct = mg_cdiv(Cin, 16), ot = mg_cdiv(Cout, 16);
  pl.ncb = blocks_of(ct);
  a.nob = blocks_of(ot);
  pl.CT = mg_cdiv(ct, pl.ncb);  // 1..4 channel tiles per block: the kernel is instantiated for each (no MFMAs on padding tiles)
  pl.OT = mg_cdiv(ot, a.nob);
  a.CinP = ct * 16; a.CoutP = ot * 16;
  const int n_cu = mg_cu_count();
  const int ny = pl.ncb * a.nob;
  // one 8-wave workgroup (128 KB of LDS) per CU; the narrow form (64 KB) runs two
  // (when that still leaves every workgroup >= 96 chunks: each slab costs a G^T M G pass and a share of the reduction)
  const int per_cu = (pl.CT + pl.OT <= 4 && (long long)a.nblk * ny >= 96ll * 2 * n_cu) ? 2 : 1;
  int ns = per_cu * n_cu / ny > 0 ? per_cu * n_cu / ny : 1;
  if (ns > a.nblk) ns = a.nblk;
  a.per = mg_cdiv(a.nblk, ns);
  pl.nsplit = mg_cdiv(a.nblk, a.per);
  pl.ws_floats = (size_t)pl.nsplit * (9 * (size_t)a.CinP * a.CoutP + a.CoutP);
}

template <int CT, int OT, bool UPS, bool FAST>
int launch_ww(const WwArgs& a, dim3 grid, hipStream_t s) {
  constexpr size_t lds = (size_t)2 * (CT + OT <= 4 ? IMG : STAGE) * sizeof(float);
  static MgPerDevice once;  // the LDS limit is a per-device function attribute
  if constexpr (CT + OT <= 4) {
    if (mg_first_use_on_device(once)) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_wgrad_narrow_mfma<CT, OT, UPS, FAST>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    hipLaunchKernelGGL((wino_wgrad_narrow_mfma<CT, OT, UPS, FAST>), grid, dim3(512), lds, s, a);
  } else {
    if (mg_first_use_on_device(once)) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_wgrad_mfma<CT, OT, UPS, FAST>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024);
    }
    hipLaunchKernelGGL((wino_wgrad_mfma<CT, OT, UPS, FAST>), grid, dim3(512), lds, s, a);
  }
  MG_CHECK_LAUNCH("mg_wino3x3_wgrad");
  return MG_OK;
}

// the scalar-addressed form (ww_body FAST): chunks of 8 x 1 x 1 tiles, whole chunks per tile row; MG_WGRAD_FAST=0: never
bool ww_fast(const WwArgs& a) {
  const char* e = getenv("MG_WGRAD_FAST");  // (read per call: tests and A/B runs switch it inside one process)
  const bool on = e == nullptr || atoi(e) != 0;
  return on && a.TBW == 8 && a.TBH == 1 && a.TBN == 1 && (a.W % 16) == 0;
}

template <bool UPS, bool FAST>
int dispatch_ww_f(int CT, int OT, const WwArgs& a, dim3 grid, hipStream_t s) {
  switch (CT * 10 + OT) {
    case 11: return launch_ww<1, 1, UPS, FAST>(a, grid, s);
    case 12: return launch_ww<1, 2, UPS, FAST>(a, grid, s);
    case 13: return launch_ww<1, 3, UPS, FAST>(a, grid, s);
    case 14: return launch_ww<1, 4, UPS, FAST>(a, grid, s);
    case 21: return launch_ww<2, 1, UPS, FAST>(a, grid, s);
    case 22: return launch_ww<2, 2, UPS, FAST>(a, grid, s);
    case 23: return launch_ww<2, 3, UPS, FAST>(a, grid, s);
    case 24: return launch_ww<2, 4, UPS, FAST>(a, grid, s);
    case 31: return launch_ww<3, 1, UPS, FAST>(a, grid, s);
    case 32: return launch_ww<3, 2, UPS, FAST>(a, grid, s);
    case 33: return launch_ww<3, 3, UPS, FAST>(a, grid, s);
    case 34: return launch_ww<3, 4, UPS, FAST>(a, grid, s);
    case 41: return launch_ww<4, 1, UPS, FAST>(a, grid, s);
    case 42: return launch_ww<4, 2, UPS, FAST>(a, grid, s);
    case 43: return launch_ww<4, 3, UPS, FAST>(a, grid, s);
    case 44: return launch_ww<4, 4, UPS, FAST>(a, grid, s);
  }
  mg_set_error("mg_wino3x3_wgrad: internal tile error (CT=%d, OT=%d)", CT, OT);
  return MG_EINVAL;
}

template <bool UPS>
int dispatch_ww(int CT, int OT, const WwArgs& a, dim3 grid, hipStream_t s) {
  return ww_fast(a) ? dispatch_ww_f<UPS, true>(CT, OT, a, grid, s) : dispatch_ww_f<UPS, false>(CT, OT, a, grid, s);
}

int launch_ww_group(const WwGroup& g, hipStream_t s) {
  static MgPerDevice once;
  if (mg_first_use_on_device(once)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_wgrad_group_mfma), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
  }
  hipLaunchKernelGGL(wino_wgrad_group_mfma, dim3(g.first[g.n]), dim3(512), (size_t)2 * STAGE * sizeof(float), s, g);
  MG_CHECK_LAUNCH("mg_wino3x3_wgrad_partial_multi");
  return MG_OK;
}

// the block shapes inlined in wino_wgrad_group_mfma
bool ww_groupable(int CT, int OT) { return CT >= 3 && OT >= 3; }

// arguments of one layer -> plan + kernel arguments (pointers, byte limits); shared by the single and the grouped entry point
int prepare_ww(const float* x, const float* gy, const float* gw, const void* ws, size_t ws_bytes, int N, int Cin, int Cout, int H,
               int W, int flags, int bias_n, WwPlan& pl) {
  MG_CHECK_ARG(x && gy && gw && ws && N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "mg_wino3x3_wgrad: bad arguments");
  MG_CHECK_ARG((H % 2 == 0) && (W % 2 == 0), "mg_wino3x3_wgrad: H=%d W=%d must be even", H, W);
  MG_CHECK_ARG(!(flags & ~MG_CONV_UPS_IN), "mg_wino3x3_wgrad: unknown flag");
  const bool ups = (flags & MG_CONV_UPS_IN) != 0;
  MG_CHECK_ARG((long long)N * Cin * H * W < (1ll << 29) && (long long)N * Cout * H * W < (1ll << 29),
               "mg_wino3x3_wgrad: tensor too large for 32-bit byte offsets");
  plan_ww(N, Cin, Cout, H, W, pl);
  if (ws_bytes < pl.ws_floats * sizeof(float)) {
    mg_set_error("mg_wino3x3_wgrad: workspace %zu < %zu bytes", ws_bytes, pl.ws_floats * sizeof(float));
    return MG_EWORKSPACE;
  }
  WwArgs& a = pl.a;
  a.x = x; a.gy = gy;
  a.slab = reinterpret_cast<float*>(const_cast<void*>(ws));
  a.slab_b = a.slab + (size_t)pl.nsplit * 9 * a.CinP * a.CoutP;
  a.bias_n = (bias_n <= 0 || bias_n > N) ? N : bias_n;
  a.x_bytes = (unsigned)((size_t)N * Cin * (ups ? (H / 2) * (W / 2) : H * W) * 4);
  a.gy_bytes = (unsigned)((size_t)N * Cout * H * W * 4);
  return MG_OK;
}

void fill_job(const WwPlan& pl, float* gw, float* gb, int accumulate, mg_wgrad_job_t* job) {
  const WwArgs& a = pl.a;
  job->slab = a.slab; job->slab_b = a.slab_b; job->gw = gw; job->gb = gb;
  job->nsplit = pl.nsplit; job->Cout = a.Cout; job->Cin = a.Cin; job->CoutP = a.CoutP; job->CinP = a.CinP; job->accumulate = accumulate;
}

int launch_single_ww(const WwPlan& pl, bool ups, hipStream_t s) {
  dim3 grid(pl.nsplit, pl.ncb * pl.a.nob);
  return ups ? dispatch_ww<true>(pl.CT, pl.OT, pl.a, grid, s) : dispatch_ww<false>(pl.CT, pl.OT, pl.a, grid, s);
}

}  // namespace

extern "C" size_t mg_wino3x3_wgrad_ws_bytes(int N, int Cin, int Cout, int H, int W) {
  WwPlan pl;
  plan_ww(N, Cin, Cout, H, W, pl);
  return pl.ws_floats * sizeof(float);
}

extern "C" int mg_wino3x3_wgrad_partial(const float* x, const float* gy, float* gw, float* gb, void* ws, size_t ws_bytes, int N,
                                        int Cin, int Cout, int H, int W, int flags, int accumulate, int bias_n,
                                        mg_wgrad_job_t* job, mg_stream_t stream) {
  MG_CHECK_ARG(job, "mg_wino3x3_wgrad: bad arguments");
  WwPlan pl;
  int rc = prepare_ww(x, gy, gw, ws, ws_bytes, N, Cin, Cout, H, W, flags, bias_n, pl);
  if (rc != MG_OK) return rc;
  rc = launch_single_ww(pl, (flags & MG_CONV_UPS_IN) != 0, (hipStream_t)stream);
  if (rc != MG_OK) return rc;
  fill_job(pl, gw, gb, accumulate, job);
  return MG_OK;
}

extern "C" int mg_wino3x3_wgrad_partial_multi(const mg_wgrad_desc_t* d, int n, int group_max_chunks, mg_wgrad_job_t* jobs,
                                              mg_stream_t stream) {
  MG_CHECK_ARG(d && jobs && n > 0 && n <= 64, "mg_wino3x3_wgrad_partial_multi: bad arguments (n = %d, at most 64 layers)", n);
  hipStream_t s = (hipStream_t)stream;
  WwPlan pl[64];
  int key[64];  // > 0: candidate for a grouped launch, layers with equal keys share one
  const int n_cu = mg_cu_count();
  for (int i = 0; i < n; ++i) {
    const int rc = prepare_ww(d[i].x, d[i].gy, d[i].gw, d[i].ws, d[i].ws_bytes, d[i].N, d[i].Cin, d[i].Cout, d[i].H, d[i].W, d[i].flags,
                              d[i].bias_n, pl[i]);
    if (rc != MG_OK) return rc;
    const long long work = (long long)pl[i].a.nblk * pl[i].ncb * pl[i].a.nob;  // chunks x channel blocks
    const bool small = group_max_chunks > 0 && ww_groupable(pl[i].CT, pl[i].OT) && work <= (long long)group_max_chunks * n_cu;
    key[i] = small ? (pl[i].CT * 10 + pl[i].OT) * 2 + ((d[i].flags & MG_CONV_UPS_IN) ? 1 : 0) : 0;
  }
  for (;;) {
    int idx[WW_GROUP], m = 0;
    for (int j = 0; j < n && m < WW_GROUP; ++j)
      if (key[j] > 0) idx[m++] = j;
    if (m < 2) break;
    // The splits of the group: about the same TIME per workgroup (a chunk of a <CT, OT> block costs ~ CT * OT MFMA groups + its
    // staging), at most one workgroup per CU over the whole group -- a 257th workgroup would run alone after the others.
    int fixed = 6;
    {
      const char* e = getenv("MG_WGRAD_GROUP_FIXED");  // measurement switch: the per-chunk staging term of the cost model
      if (e != nullptr && atoi(e) >= 0) fixed = atoi(e);
    }
    auto cost = [&](int i) { return pl[i].CT * pl[i].OT + fixed; };
    long long work = 0;
    for (int k = 0; k < m; ++k) work += (long long)pl[idx[k]].a.nblk * pl[idx[k]].ncb * pl[idx[k]].a.nob * cost(idx[k]);
    int slots = n_cu;
    {
      const char* e = getenv("MG_WGRAD_GROUP_SLOTS");  // measurement switch: workgroups per group launch in units of 1/4 of the CUs
      if (e != nullptr && atoi(e) > 0) slots = n_cu * atoi(e) / 4;
    }
    long long budget = (work + slots - 1) / slots;  // cost units per workgroup
    int ns[WW_GROUP], total;
    for (;;) {
      total = 0;
      bool floor_reached = true;  // every layer at one split: nothing left to shrink
      for (int k = 0; k < m; ++k) {
        const WwPlan& q = pl[idx[k]];
        long long per = budget / cost(idx[k]);
        if (per < 1) per = 1;
        int v = (int)((q.a.nblk + per - 1) / per);
        if (v > q.nsplit) v = q.nsplit;  // (the workspace was sized for the single-layer plan)
        ns[k] = v;
        floor_reached = floor_reached && v == 1;
        total += v * q.ncb * q.a.nob;
      }
      if (total <= slots || floor_reached) break;
      budget += (budget + 15) / 16;
    }
    WwGroup g;
    g.n = m;
    g.first[0] = 0;
    for (int k = 0; k < m; ++k) {
      WwPlan& q = pl[idx[k]];
      q.a.per = mg_cdiv(q.a.nblk, ns[k]);
      q.nsplit = mg_cdiv(q.a.nblk, q.a.per);
      q.a.slab_b = q.a.slab + (size_t)q.nsplit * 9 * q.a.CinP * q.a.CoutP;
      g.a[k] = q.a;
      g.nsplit[k] = q.nsplit;
      g.var[k] = key[idx[k]];
      g.first[k + 1] = g.first[k] + q.nsplit * q.ncb * q.a.nob;
    }
    const int rc = launch_ww_group(g, s);
    if (rc != MG_OK) return rc;
    for (int k = 0; k < m; ++k) key[idx[k]] = -1;
  }
  for (int i = 0; i < n; ++i) {
    if (key[i] >= 0) {  // large, of a block shape that is not inlined in the group kernel, or alone
      const int rc = launch_single_ww(pl[i], (d[i].flags & MG_CONV_UPS_IN) != 0, s);
      if (rc != MG_OK) return rc;
    }
  }
  for (int i = 0; i < n; ++i) fill_job(pl[i], d[i].gw, d[i].gb, d[i].accumulate, &jobs[i]);
  return MG_OK;
}

extern "C" int mg_wino3x3_wgrad_reduce(const mg_wgrad_job_t* jobs, int n, mg_stream_t stream) {
  MG_CHECK_ARG(jobs && n > 0, "mg_wino3x3_wgrad_reduce: bad arguments");
  for (int first = 0; first < n; first += WW_JOBS) {
    const int m = n - first < WW_JOBS ? n - first : WW_JOBS;
    WwJobs c;
    c.n = m;
    c.first[0] = 0;
    for (int i = 0; i < m; ++i) {
      c.j[i] = jobs[first + i];
      MG_CHECK_ARG(c.j[i].slab && c.j[i].gw && c.j[i].nsplit > 0 && c.j[i].CinP > 0 && c.j[i].CoutP > 0,
                   "mg_wino3x3_wgrad_reduce: bad job %d", first + i);
      const int total = c.j[i].CinP * c.j[i].CoutP;
      c.first[i + 1] = c.first[i] + mg_cdiv(total, 512 / ww_reduce_lanes(c.j[i].nsplit, total));
    }
    hipLaunchKernelGGL(wino_wgrad_reduce_multi, dim3(c.first[m]), dim3(512), 0, (hipStream_t)stream, c);
    MG_CHECK_LAUNCH("mg_wino3x3_wgrad_reduce");
  }
  return MG_OK;
}

extern "C" int mg_wino3x3_wgrad(const float* x, const float* gy, float* gw, float* gb, void* ws, size_t ws_bytes, int N, int Cin,
                                int Cout, int H, int W, int flags, int accumulate, int bias_n, mg_stream_t stream) {
  mg_wgrad_job_t job;
  const int rc = mg_wino3x3_wgrad_partial(x, gy, gw, gb, ws, ws_bytes, N, Cin, Cout, H, W, flags, accumulate, bias_n, &job, stream);
  return rc != MG_OK ? rc : mg_wino3x3_wgrad_reduce(&job, 1, stream);
}
