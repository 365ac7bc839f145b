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

// ---- row-staged form (round 6): the operands are transformed IN REGISTERS, in MFMA operand layout, from RAW rows that the memory
// system writes straight into LDS (LDS-DMA, `buffer_load_dwordx4 ... lds`).
// The two kernels above transform on the way INTO LDS: a thread owns a (tile, channel) item, computes all 16 Winograd components of it
// and scatters them over eight component-pair images (64 B per item and operand), so a stage holds only 8 tiles, every 8 tiles cost a
// workgroup barrier, and the per-item work (halo exchange by DPP + selects, 16 packed adds, 8 LDS writes, the loads' address
// arithmetic) is 1.6 vector + 1.7 scalar instructions per MFMA on the widest block shape and 3.7 + 4.2 on the narrow ones
// (profiles/r06_pmc_w20n_base.txt, r06_pmc_ww16_base.txt: 64 % of the wave cycles waiting).  Here
//   * a stage is 16 horizontally adjacent tiles (one k-step = 4 tiles, 4 k-steps) of all channels of the block as RAW pixels:
//     x rows 2by-1 .. 2by+2, pixels -1 .. 34 of the stage (nine 16-byte pieces per row), gy rows 2by, 2by+1 (eight pieces) -- 8 bytes
//     per item and row instead of 64: half the barriers in 45 % of the LDS.  The stage image is a sequence of 16-byte pieces and wave
//     w copies pieces 64 (w + 8 m) .. + 63 with ONE LDS-DMA instruction per m (5-7 per wave and stage): no staging registers, no LDS
//     store instructions (`ds_write_b128` moves 79 B per clock: measured, the register-staged version of this kernel spent 12 % of
//     its time in them and another 13 % around the loads), lane offsets fixed for the kernel, the stage in a scalar offset.  Pieces of
//     rows above / below the image, of channels beyond the tensor and the padding pieces have an out-of-range offset: the bounds
//     check writes ZEROS for them, per dword (tools/hwtests/lds_dma_oob.hip), which is also what clips the last row's pixels behind the
//     tensor's end.  Pixel -1 of a row's first stage and pixel 32 of its last one are fetched (they exist: the neighbouring row's) and
//     overwritten with zeros by the wave that copied them; x[-1] itself, before the tensor, is never touched (the first stage's first
//     piece is patched from pixel 0 on);
//   * an x row is stored from pixel -1 on, so the 4-pixel patch row of tile T (pixels 2T-1 .. 2T+2) starts at an even word:
//     [e0, p0 | p1, e1]; wave w owns component pair w as before -- row i = w / 2 of the component matrix, columns (0, 3) for even and
//     (1, 2) for odd w -- and lane (rq, col) reads, for tile 4 ks + rq and channel 16 i + col, exactly the TWO patch rows its component
//     row needs (B^T d: d0 - d2, d1 + d2, d2 - d1, d1 - d3): two packed adds for the row step, and the column step is ONE packed add of
//     the two halves -- (u0 - u2, u1 - u3) = (v0, v3) for even waves, (u1 + u2, u2 - u1) = (v1, v2) for odd ones -- whose result IS the
//     A operand pair; the gy side is 0-2 packed adds per out-channel tile (A t A^T: row i is t0, t0 + t1, t0 - t1 or t1; the signs of
//     the components that carry a minus are applied once, to the accumulators, after the tile loop);
//   * nothing else in the loop: no DPP, no selects, no per-lane address arithmetic (LDS offsets are immediates);
//   * every operand read is a `ds_read_b64` (two 32-lane groups, bank = word mod 64, 256 B per clock): lane (rq, col) reads words
//     col * stride + 2 rq + {0, 1}, so channel strides of 4 x odd modulo 64 (148, 68 words) give the 16 channels x 2 tiles of a group
//     64 different banks.  The reads are inline assembly: hipcc merges adjacent 8-byte LDS reads into `ds_read2_b64`, which this LDS
//     serves at HALF the rate with banks modulo 32 (MI355X_MICROARCH.md, LDS).  Measured on the way (profiles/r06_wgrad_rows_steps.txt,
//     48 x 64 @128 x 192 images, chunk-staged kernel 914 us): strides of 16 modulo 64 with 4-byte halo reads 1 306 us
//     (SQ_LDS_BANK_CONFLICT 84 % of the LDS cycles); 16-byte patch rows as ds_read2_b64 on strides of 24 / 8: 1 075 us (72 %).
// Each wave runs ONE of eight specialisations of the tile loop (component row x parity), chosen once; all of them execute the
// same barriers.  Accumulator layout, G^T M G slab pass and reduce kernel are those of wino_wgrad_mfma.
constexpr int RW_RS = 36;              // words per staged x row: pixel p (-1 .. 34) at word p + 1 = 9 pieces
constexpr int RW_CSX = 4 * RW_RS + 4;  // words per x channel (4 rows + one padding piece = 37 pieces); 148 = 4 * 5 mod 64
constexpr int RW_CSY = 68;             // words per gy channel: 2 rows x 32 pixels + one padding piece = 17 pieces (= 4 mod 64)
// UPS (the convolution input is the nearest x2 up-sampling of x, generator.py:24-25): the patch of tile (TY, TX) is the 3x3 low-res
// neighbourhood with its centre row / column doubled, so a stage holds low-res rows TY-1 .. TY+1, pixels -1 .. 18 (five pieces)
constexpr int RW_RSU = 20;
constexpr int RW_CSXU = 3 * RW_RSU + 8;  // 3 rows + two padding pieces = 17 pieces; 68 = 4 mod 32 (4-byte reads: banks modulo 32)

template <int CT, int OT, bool UPS = false>
struct RwGeom {
  static constexpr int RS = UPS ? RW_RSU : RW_RS, CSX = UPS ? RW_CSXU : RW_CSX;
  static constexpr int PPR = UPS ? 5 : 9, ROWS = UPS ? 3 : 4, PPC = CSX / 4;  // pieces per row, rows and pieces per x channel
  static constexpr int XI = (16 * CT * PPC + 63) / 64;  // LDS-DMA instructions (64 pieces each) of the x part, of the gy part
  static constexpr int YI = (16 * OT * 17 + 63) / 64;
  static constexpr int NI = XI + YI;
  static constexpr int NM = (NI + 7) / 8;  // per wave
  static constexpr int YB = XI * 256;      // words
  static constexpr int STG = NI * 256;
  static constexpr size_t lds_bytes() {
    const size_t stages = (size_t)2 * STG * sizeof(float), slab = (size_t)16 * 32 * 64 * sizeof(float);
    return stages > slab ? stages : slab;
  }
};

template <class F, int... Is>
__device__ __forceinline__ void rw_static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void rw_static_for(F&& f) {
  rw_static_for_impl(f, std::make_integer_sequence<int, N>{});
}
// 8 bytes of LDS at byte address `addr` + OFF.  NOT tracked by hipcc's s_waitcnt insertion: rw_lds_wait() before the first use.
template <int OFF>
__device__ __forceinline__ f32x2 rw_lds64(unsigned addr) {
  f32x2 v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF>
__device__ __forceinline__ float rw_lds32(unsigned addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ void rw_lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void rw_tie(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void rw_tie(f32x2& v) { asm volatile("" : "+v"(v)); }  // orders the uses of v behind the wait
typedef __attribute__((address_space(3))) void* rw_lds_ptr;

template <int CT, int OT, bool UPS>
__global__ void __launch_bounds__(512) wino_wgrad_rows_mfma(const WwArgs a) {
  using GEO = RwGeom<CT, OT, UPS>;
  constexpr int YB = GEO::YB, STG = GEO::STG, XI = GEO::XI, NI = GEO::NI, NM = GEO::NM;
  constexpr int RS = GEO::RS, CSX = GEO::CSX, PPR = GEO::PPR, PPC = GEO::PPC, XROWS = GEO::ROWS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // = component pair
  const int col = lane & 15, rq = lane >> 4;
  const int split = blockIdx.x;
  const int cb = blockIdx.y / a.nob, ob = blockIdx.y % a.nob;
  const int c0 = cb * CT * 16, o0 = ob * OT * 16;
  const int HW = a.H * a.W;
  const int Ht = a.H >> 1;
  const int Wx = UPS ? a.W >> 1 : a.W, HWx = UPS ? HW >> 2 : HW;  // the x tensor's row and plane
  constexpr unsigned INV = 0x80000000u;

  // ---- loader: slot m of this wave is LDS-DMA instruction k = wave + 8 m of a stage, this lane's piece P = 64 k + lane.
  // x part (k < XI): P = 37 ch + 9 r + seg -- pixels 4 seg - 1 .. 4 seg + 2 of patch row r of channel ch (P mod 37 = 36: padding);
  // gy part: P - 64 XI = 17 ch + 8 r + seg.  x is addressed relative to x - (W + 4) floats, so that row -1 and pixel -1 have
  // non-negative offsets.  voff[m]: byte offset of the piece inside a stage, INV = write zeros.  cls: per slot, bit 4m: the piece is
  // patch row 0 (outside the image in the top tile row), 4m+1: patch row 3 (bottom), 4m+2: its first word is pixel -1, 4m+3: its
  // second word is pixel 32.
  unsigned voff[NM];
  unsigned cls = 0;
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    const int k = wave + 8 * m;
    const int P = 64 * k + lane;
    unsigned v = INV;
    if (k < XI) {
      const int ch = P / PPC, rem = P - PPC * ch, r = rem / PPR, sg = rem - PPR * r;
      if (ch < 16 * CT && rem < XROWS * PPR && c0 + ch < a.Cin) {
        v = (unsigned)((ch * HWx + r * Wx + 4 * sg + 3) * 4);
        cls |= (unsigned)(r == 0) << (4 * m) | (unsigned)(r == XROWS - 1) << (4 * m + 1) | (unsigned)(sg == 0) << (4 * m + 2) |
               (unsigned)(sg == PPR - 1) << (4 * m + 3);
      }
    } else if (k < NI) {
      const int Q = P - 64 * XI;
      const int ch = Q / 17, rem = Q - 17 * ch, r = rem >> 3, sg = rem & 7;
      if (ch < 16 * OT && rem < 16 && o0 + ch < a.Cout) v = (unsigned)((ch * HW + r * a.W + 4 * sg) * 4);
    }
    voff[m] = v;
  }
  const unsigned xshift = (unsigned)(Wx + 4) * 4u;
  const char* xbase = reinterpret_cast<const char*>(a.x) - xshift;
  // The piece that begins at x[-1], BEFORE the tensor, must not be fetched: image row 0 of channel 0 of image 0, piece 0 -- patch row 1
  // of the tensor's first stage (lane PPR of wave 0's slot 0) and, in the up-sampled form, also patch row 0 of the stage below it
  // (lane 0).  That lane's piece is zero-filled and x[0 .. 2] are written behind it once the stage has landed.

  bool zl = false, zr = false;  // the stage copied last begins / ends at the image's left / right edge
  int nq = 0, bx, by, bn;
  {
    const int b0 = split * a.per;
    bx = b0 % a.blocks_x;
    const int t2 = b0 / a.blocks_x;
    by = t2 % a.blocks_y;
    bn = t2 / a.blocks_y;
  }
  // the slab's next stage -> LDS at word offset so (asynchronous: landed() before the barrier that precedes its first read).  The
  // stage's 5-7 copy instructions per wave are issued a few at a time (issue_begin, issue_slots<LO, HI>, ...): all 45-54 of a
  // workgroup in one burst right behind the barrier fill the CU's memory pipeline, and every wave sits at its last copy instruction
  // instead of issuing MFMAs (measured: the burst cost 13 % of the kernel's cycles).
  __amdgpu_buffer_rsrc_t st_xs, st_ys;
  int st_sx = 0, st_sy = 0, st_so = 0;
  bool st_top = false, st_bot = false, st_patch = false;
  int st_lowlane = -1;
  auto issue_begin = [&](int so) __attribute__((always_inline)) {
    const bool ok = nq < a.per && split * a.per + nq < a.nblk;
    st_top = by == 0;
    st_bot = by == Ht - 1;
    st_xs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xbase), 0, ok ? (int)(a.x_bytes + xshift) : 0, 0x00020000);
    st_ys = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.gy), 0, ok ? (int)a.gy_bytes : 0, 0x00020000);
    st_sx = UPS ? ((bn * a.Cin + c0) * HWx + by * Wx + 16 * bx) * 4 : ((bn * a.Cin + c0) * HW + 2 * by * a.W + 32 * bx) * 4;
    st_sy = ((bn * a.Cout + o0) * HW + 2 * by * a.W + 32 * bx) * 4;
    if (!UPS && (a.TBN & 16)) { st_sx = (c0 * HW + (2 + (split & 31) * 2) * a.W) * 4; st_sy = (o0 * HW + (2 + (split & 31) * 2) * a.W) * 4; }  // (rows >= 1: inside the tensor)
    st_lowlane = (bn == 0 && c0 == 0 && bx == 0 && ok) ? (by == 0 ? PPR : ((UPS && by == 1) ? 0 : -1)) : -1;
    st_patch = wave == 0 && st_lowlane >= 0;
    st_so = so;
    zl = bx == 0;
    zr = bx == a.blocks_x - 1;
    ++nq;
    ++bx;
    const int wx = bx == a.blocks_x ? 1 : 0;
    bx = wx ? 0 : bx;
    by += wx;
    const int wy = by == a.blocks_y ? 1 : 0;
    by = wy ? 0 : by;
    bn += wy;
  };
  auto issue_slots = [&](auto lo_, auto hi_) __attribute__((always_inline)) {
    constexpr int LO = decltype(lo_)::value, HI = decltype(hi_)::value < NM ? decltype(hi_)::value : NM;
#pragma unroll
    for (int m = LO; m < HI; ++m) {
      const int k = wave + 8 * m;
      if (k < NI) {  // (wave-uniform)
        float* dst = smem + st_so + 256 * k;
        if (k < XI) {
          if ((a.TBN & 64) && (k & 1)) continue;  // (measurement: every second x copy instruction dropped)
          unsigned v = voff[m];
          if (st_top || st_bot || st_patch) {  // (wave-uniform; the common stage takes the offsets as they are)
            const bool kill = (st_top && ((cls >> (4 * m)) & 1u)) || (st_bot && ((cls >> (4 * m + 1)) & 1u)) || (st_patch && m == 0 && lane == st_lowlane);
            v = kill ? INV : v;
          }
          __builtin_amdgcn_raw_ptr_buffer_load_lds(st_xs, (rw_lds_ptr)dst, 16, (int)v, st_sx, 0, 0);
        } else {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(st_ys, (rw_lds_ptr)dst, 16, (int)voff[m], st_sy, 0, 0);
        }
      }
    }
  };
  using M0_ = std::integral_constant<int, 0>; using M2_ = std::integral_constant<int, 2>; using M4_ = std::integral_constant<int, 4>;
  using M6_ = std::integral_constant<int, 6>; using M8_ = std::integral_constant<int, 8>;
  auto issue_stage = [&](int so) __attribute__((always_inline)) {
    issue_begin(so);
    issue_slots(M0_{}, M8_{});
  };
  // the stage at word offset so has landed (this wave's pieces): zero the pixels beside the image that this wave copied
  auto landed = [&](int so) __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (st_patch && lane == st_lowlane) {  // (the piece that begins at x[-1]: zero-filled, now x[0 .. 2] behind the zero)
      float* pc = smem + so + 4 * lane;
      pc[1] = a.x[0];
      pc[2] = a.x[1];
      pc[3] = a.x[2];
    }
    if (zl || zr) {  // (wave-uniform)
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const int k = wave + 8 * m;
        if (k < XI) {
          float* pc = smem + so + 256 * k + 4 * lane;
          if (zl && ((cls >> (4 * m + 2)) & 1u)) pc[0] = 0.f;
          if (zr && ((cls >> (4 * m + 3)) & 1u)) pc[1] = 0.f;
        }
      }
    }
  };

  f32x4 acc[2][CT][OT];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
      for (int j = 0; j < OT; ++j) acc[p][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[OT];
#pragma unroll
  for (int j = 0; j < OT; ++j) bsum[j] = 0.f;

  // LDS byte addresses of this lane's operand reads (the low 32 bits of a shared-aperture address are the LDS offset)
  // (UPS: tile T's low-res pixels T-1, T, T+1 are words T, T+1, T+2 of a row)
  const unsigned xrd_a = (unsigned)reinterpret_cast<size_t>(smem + col * CSX + (UPS ? 1 : 2) * rq);  // + 16 i CSX + row RS + 8 ks: [e0, p0 | p1, e1] of tile 4 ks + rq
  const unsigned yrd_a = (unsigned)reinterpret_cast<size_t>(smem + YB + col * RW_CSY + 2 * rq);  // + 16 j CSY + row 32 + 8 ks: (t.0, t.1)

  auto tile_loop = [&](auto i_, auto odd_) __attribute__((always_inline)) {
    constexpr int I = decltype(i_)::value;
    constexpr bool ODD = decltype(odd_)::value;
    constexpr int RA = I == 0 ? 0 : (I == 2 ? 2 : 1), RB = I == 0 ? 2 : (I == 1 ? 2 : (I == 2 ? 1 : 3));  // u = d[RA] +- d[RB]
    constexpr bool PLUS = I == 1;
    constexpr bool Y0 = I != 3, Y1 = I != 0;  // which gy rows the component row needs
    // UPS: the component row is  i = 0: l[TY-1] - l[TY];  1: 2 l[TY];  2: ZERO;  3: l[TY] - l[TY+1]  over the low-res pixels (L, C, R) =
    // (TX-1, TX, TX+1), the columns  0: uL - uC;  1: 2 uC;  2: ZERO;  3: uC - uR -- 9 of the 16 components.  Row 2 has no wave work
    // at all (IDLE), an odd wave computes column 1 only (NP = 1); the factors of two are applied once, to the accumulators.
    constexpr bool IDLE = UPS && I == 2;
    constexpr int NP = (UPS && ODD) ? 1 : 2;                        // components this wave accumulates
    constexpr int UA = I == 0 ? 0 : 1, UB = I == 0 ? 1 : 2;         // UPS: low-res rows of u = l[UA] - l[UB]  (I = 1: l[1] alone)
    struct Raw {
      f32x2 al[CT], ah[CT], bl[CT], bh[CT], t0[OT], t1[OT];  // rows RA / RB as (e0, p0) | (p1, e1); gy rows
      float ua[CT][3], ub[CT][3];                             // UPS: (L, C, R) of the two low-res rows
    };
    struct Ops {
      f32x2 av[CT], bv[OT];  // the wave's two x components per in-channel tile / two gy components (unsigned) per out-channel tile
    };
    auto read_x = [&](Raw& r, unsigned xa, auto ks_) __attribute__((always_inline)) {
      constexpr int KS = decltype(ks_)::value;
      rw_static_for<CT>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        if constexpr (UPS) {
          constexpr int o = (16 * i * CSX + 4 * KS) * 4;
          if constexpr (!ODD) {
            r.ua[i][0] = rw_lds32<o + UA * RS * 4>(xa);
            r.ua[i][2] = rw_lds32<o + UA * RS * 4 + 8>(xa);
          }
          r.ua[i][1] = rw_lds32<o + UA * RS * 4 + 4>(xa);
          if constexpr (I != 1) {
            if constexpr (!ODD) {
              r.ub[i][0] = rw_lds32<o + UB * RS * 4>(xa);
              r.ub[i][2] = rw_lds32<o + UB * RS * 4 + 8>(xa);
            }
            r.ub[i][1] = rw_lds32<o + UB * RS * 4 + 4>(xa);
          }
        } else {
          constexpr int o = (16 * i * CSX + 8 * KS) * 4;
          r.al[i] = rw_lds64<o + RA * RS * 4>(xa);
          r.ah[i] = rw_lds64<o + RA * RS * 4 + 8>(xa);
          r.bl[i] = rw_lds64<o + RB * RS * 4>(xa);
          r.bh[i] = rw_lds64<o + RB * RS * 4 + 8>(xa);
        }
      });
    };
    auto read_y = [&](Raw& r, unsigned ya, auto ks_) __attribute__((always_inline)) {
      constexpr int KS = decltype(ks_)::value;
      rw_static_for<OT>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        constexpr int o = (16 * j * RW_CSY + 8 * KS) * 4;
        if constexpr (Y0) r.t0[j] = rw_lds64<o>(ya);
        if constexpr (Y1) r.t1[j] = rw_lds64<o + 128>(ya);
      });
    };
    auto wait_raw = [&](Raw& r) __attribute__((always_inline)) {
      rw_lds_wait();
#pragma unroll
      for (int i = 0; i < CT; ++i) {
        if constexpr (UPS) {
          if constexpr (!ODD) { rw_tie(r.ua[i][0]); rw_tie(r.ua[i][2]); }
          rw_tie(r.ua[i][1]);
          if constexpr (I != 1) {
            if constexpr (!ODD) { rw_tie(r.ub[i][0]); rw_tie(r.ub[i][2]); }
            rw_tie(r.ub[i][1]);
          }
        } else {
          rw_tie(r.al[i]); rw_tie(r.ah[i]); rw_tie(r.bl[i]); rw_tie(r.bh[i]);
        }
      }
#pragma unroll
      for (int j = 0; j < OT; ++j) {
        if constexpr (Y0) rw_tie(r.t0[j]);
        if constexpr (Y1) rw_tie(r.t1[j]);
      }
    };
    // raw rows -> MFMA operands
    auto transform_x = [&](const Raw& r, Ops& o) __attribute__((always_inline)) {
      if constexpr (UPS) {
#pragma unroll
        for (int i = 0; i < CT; ++i) {
          float uL = 0.f, uC, uR = 0.f;
          if constexpr (I == 1) {
            uC = r.ua[i][1];
            if constexpr (!ODD) { uL = r.ua[i][0]; uR = r.ua[i][2]; }
          } else {
            uC = r.ua[i][1] - r.ub[i][1];
            if constexpr (!ODD) { uL = r.ua[i][0] - r.ub[i][0]; uR = r.ua[i][2] - r.ub[i][2]; }
          }
          if constexpr (!ODD) o.av[i] = f32x2{uL - uC, uC - uR};
          else o.av[i] = f32x2{uC, 0.f};
        }
        return;
      }
#pragma unroll
      for (int i = 0; i < CT; ++i) {
        const f32x2 ul = PLUS ? r.al[i] + r.bl[i] : pk_sub(r.al[i], r.bl[i]);  // (u0, u1)
        const f32x2 uh = PLUS ? r.ah[i] + r.bh[i] : pk_sub(r.ah[i], r.bh[i]);  // (u2, u3)
        if constexpr (!ODD) asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(o.av[i]) : "v"(ul), "v"(uh));  // (u0 - u2, u1 - u3)
        else asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[1,0]" : "=v"(o.av[i]) : "v"(ul), "v"(uh));  // (u1 + u2, u2 - u1)
      }
    };
    auto transform_y = [&](const Raw& r, Ops& o, bool bias_on) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < OT; ++j) {
        f32x2 rr;
        if constexpr (I == 0) rr = r.t0[j];
        else if constexpr (I == 1) rr = r.t0[j] + r.t1[j];
        else if constexpr (I == 2) rr = pk_sub(r.t0[j], r.t1[j]);
        else rr = r.t1[j];
        if constexpr (I == 1 && !ODD) {
          if (bias_on) bsum[j] += rr[0] + rr[1];  // t00 + t10 + t01 + t11: the bias gradient rides in wave 2
        }
        if constexpr (!ODD) o.bv[j] = rr;
        else asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(o.bv[j]) : "v"(rr));  // (r0 + r1, r0 - r1)
      }
    };
    // (gfx950: registers written by vector instructions inside inline assembly and read as MFMA sources right behind them need wait
    // states hipcc only inserts for instructions it schedules itself -- wino_strip.hip; the fence names them)
    auto fence = [&](Ops& o) __attribute__((always_inline)) {
      if constexpr (CT == 1) asm volatile("s_nop 1" : "+v"(o.av[0]));
      if constexpr (CT == 2) asm volatile("s_nop 1" : "+v"(o.av[0]), "+v"(o.av[1]));
      if constexpr (CT == 3) asm volatile("s_nop 1" : "+v"(o.av[0]), "+v"(o.av[1]), "+v"(o.av[2]));
      if constexpr (CT == 4) asm volatile("s_nop 1" : "+v"(o.av[0]), "+v"(o.av[1]), "+v"(o.av[2]), "+v"(o.av[3]));
      if constexpr (OT == 1) asm volatile("" : "+v"(o.bv[0]));
      if constexpr (OT == 2) asm volatile("" : "+v"(o.bv[0]), "+v"(o.bv[1]));
      if constexpr (OT == 3) asm volatile("" : "+v"(o.bv[0]), "+v"(o.bv[1]), "+v"(o.bv[2]));
      if constexpr (OT == 4) asm volatile("" : "+v"(o.bv[0]), "+v"(o.bv[1]), "+v"(o.bv[2]), "+v"(o.bv[3]));
    };
    // MFMAs LO .. HI - 1 of a k-step, n = (p CT + i) OT + j
    constexpr int NMF = NP * CT * OT;
    auto mma = [&](const Ops& o, auto lo_, auto hi_) __attribute__((always_inline)) {
      constexpr int LO = decltype(lo_)::value, HI = decltype(hi_)::value;
      rw_static_for<HI - LO>([&](auto nc) __attribute__((always_inline)) {
        constexpr int n = LO + decltype(nc)::value;
        constexpr int p = n / (CT * OT), i = (n / OT) % CT, j = n % OT;
        acc[p][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.av[i][p], o.bv[j][p], acc[p][i][j], 0, 0, 0);
      });
      __builtin_amdgcn_sched_barrier(0);
    };
    // One k-step.  Its MFMAs carry, in their shadow, the operand reads and the transform of the NEXT k-step (vector and LDS
    // instructions issued right behind an MFMA of the same wave cost 2-4 cycles each instead of 8.5, tools/hwtests/
    // valu_latency_under_mfma.hip; in a phase of their own both waves of a SIMD sit in it at the same time and the matrix pipe idles:
    // measured, 24 % of the kernel).  The LAST k-step of a stage also carries the stage change: wait for the own pieces of the next
    // stage, barrier, request the stage after it into the buffer just left -- all between its first MFMAs and the reads.
    constexpr int C1 = NMF / 4, C2 = (3 * NMF) / 8, C3 = NMF / 2, C4 = (3 * NMF) / 4, C5 = (7 * NMF) / 8;
    using N0_ = std::integral_constant<int, 0>; using N1_ = std::integral_constant<int, C1>; using N2_ = std::integral_constant<int, C2>;
    using N3_ = std::integral_constant<int, C3>; using N4_ = std::integral_constant<int, C4>; using N5_ = std::integral_constant<int, C5>;
    using N6_ = std::integral_constant<int, NMF>;
    auto step = [&](const Ops& cur, Ops& nxt, Raw& r, unsigned xa, unsigned ya, auto nks_, bool bias_on, auto&& between) __attribute__((always_inline)) {
      if constexpr (IDLE) {  // (copies its share of the stages and keeps the barriers; its accumulators stay zero)
        between();
        return;
      }
      mma(cur, N0_{}, N1_{});
      between();
      __builtin_amdgcn_sched_barrier(0);
      mma(cur, N1_{}, N2_{});
      read_x(r, xa, nks_);
      __builtin_amdgcn_sched_barrier(0);
      mma(cur, N2_{}, N3_{});
      read_y(r, ya, nks_);
      __builtin_amdgcn_sched_barrier(0);
      mma(cur, N3_{}, N4_{});
      wait_raw(r);
      transform_x(r, nxt);
      __builtin_amdgcn_sched_barrier(0);
      mma(cur, N4_{}, N5_{});
      transform_y(r, nxt, bias_on);
      fence(nxt);
      __builtin_amdgcn_sched_barrier(0);
      mma(cur, N5_{}, N6_{});
    };

    using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>; using K3 = std::integral_constant<int, 3>;
    // the image of a stage (bias gradient: images below bias_n count)
    const int per_img = a.blocks_x * a.blocks_y;
    int img = (split * a.per) / per_img, img_left = per_img - (split * a.per) % per_img;
    const bool staging = !(a.TBN & 2);
    Raw r;
    Ops o0, o1;
    // prologue: stage 0 in buffer 0, stage 1 on its way into buffer 1, the operands of stage 0's first k-step
    issue_stage(0);
    landed(0);
    __syncthreads();
    if (a.per > 1 && staging) issue_stage(STG);
    if constexpr (!IDLE) {
      read_x(r, xrd_a, K0{});
      read_y(r, yrd_a, K0{});
      wait_raw(r);
      transform_x(r, o0);
      transform_y(r, o0, img < a.bias_n);
      fence(o0);
    }
    __builtin_amdgcn_sched_barrier(0);
    for (int q = 0; q < a.per; ++q) {
      const int so = (q & 1) * STG, sn = STG - so;
      const unsigned xa = xrd_a + (unsigned)so * 4u, ya = yrd_a + (unsigned)so * 4u;
      const unsigned xn = xrd_a + (unsigned)sn * 4u, yn = yrd_a + (unsigned)sn * 4u;
      const bool bias_on = img < a.bias_n;
      const bool dma = staging && q > 0 && q + 1 < a.per;  // stage q + 1 is being copied: its slots 2 .. go out under k-steps 0 .. 2
      step(o0, o1, r, xa, ya, K1{}, bias_on, [&]() __attribute__((always_inline)) { if (dma) issue_slots(M2_{}, M4_{}); });
      step(o1, o0, r, xa, ya, K2{}, bias_on, [&]() __attribute__((always_inline)) { if (dma) issue_slots(M4_{}, M6_{}); });
      step(o0, o1, r, xa, ya, K3{}, bias_on, [&]() __attribute__((always_inline)) { if (dma) issue_slots(M6_{}, M8_{}); });
      if (--img_left == 0) { img_left = per_img; ++img; }
      step(o1, o0, r, xn, yn, K0{}, q + 1 < a.per && img < a.bias_n, [&]() __attribute__((always_inline)) {  // (behind the last stage: stale LDS, unused)
        if (staging && !(a.TBN & 8)) landed(sn);  // stage q + 1: this wave's pieces are in LDS (and its edge pixels zeroed)
        if (!(a.TBN & 4)) __syncthreads();       // ... everybody's; and nobody reads buffer `so` any more
        if (staging && q + 2 < a.per) {          // stage q + 2: its first copy instructions
          issue_begin(so);
          issue_slots(M0_{}, M2_{});
        }
      });
    }
    // signs of the gy components computed unsigned: (i, 3) for i < 3, (3, 0), (3, 1), (3, 2)
    // (UPS: x 2 for component row 1, x 2 for column 1 -- the transforms above leave those factors out)
    constexpr float F0 = UPS ? (I == 1 ? 2.f : 1.f) * (ODD ? 2.f : 1.f) : 1.f, F1 = UPS ? (I == 1 ? 2.f : 1.f) : 1.f;
    constexpr float S0 = (I == 3 ? -1.f : 1.f) * F0;                                  // p = 0: column 0 (even) / 1 (odd)
    constexpr float S1 = (ODD ? (I == 3 ? -1.f : 1.f) : (I == 3 ? 1.f : -1.f)) * F1;  // p = 1: column 3 (even) / 2 (odd)
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
      for (int j = 0; j < OT; ++j) {
        if constexpr (S0 != 1.f) acc[0][i][j] = acc[0][i][j] * S0;
        if constexpr (S1 != 1.f && NP == 2) acc[1][i][j] = acc[1][i][j] * S1;
      }
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  // component pair of this wave: rw = 2 i + parity.  UPS: row 2 is idle and an odd wave has half the work, and waves w and w + 4 of a
  // workgroup share a SIMD (MI355X_MICROARCH.md: cyclic placement), so the pairs there are (row even, row odd) for rows 0, 1, 3 and
  // (idle, idle): three component-units on three SIMDs instead of four on each
  const int rw = UPS ? 2 * ((wave & 3) == 2 ? 3 : ((wave & 3) == 3 ? 2 : (wave & 3))) + (wave >> 2) : wave;
  switch (rw) {
    case 0: tile_loop(I0{}, std::false_type{}); break;
    case 1: tile_loop(I0{}, std::true_type{}); break;
    case 2: tile_loop(I1{}, std::false_type{}); break;
    case 3: tile_loop(I1{}, std::true_type{}); break;
    case 4: tile_loop(I2{}, std::false_type{}); break;
    case 5: tile_loop(I2{}, std::true_type{}); break;
    case 6: tile_loop(I3{}, std::false_type{}); break;
    default: tile_loop(I3{}, std::true_type{}); break;
  }

  // slab of this split: G^T M G per (c, o) -- wino_wgrad_mfma's pass, the same accumulator layout
  float* G = smem;  // [slot 16][cc 32][o 64]
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    __syncthreads();
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
              G[((2 * rw + p) * 32 + ii * 16 + rq * 4 + g) * 64 + ((j ^ rq) * 16 + col)] = acc[p][i][j][g];
        }
      }
    __syncthreads();
    constexpr int SL[4] = {0, 2, 3, 1};
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      const int cc = (tid >> 6) + 8 * k4;
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
        float hh[3][4];
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
  // bias gradient: the wave of component pair 2 holds, per out-channel tile, lane (rq, col) = its tiles' sums of channel 16 j + col
  if (rw == 2 && cb == 0) {
#pragma unroll
    for (int j = 0; j < OT; ++j) {
      float v = bsum[j];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (rq == 0 && o0 + 16 * j + col < a.CoutP) a.slab_b[(size_t)split * a.CoutP + o0 + 16 * j + col] = v;
    }
  }
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
  bool rows = false;  // the row-staged kernel (wino_wgrad_rows_mfma) and its stage geometry
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

// ---- the row-staged form: stages of 16 x 1 x 1 tiles.  MG_WGRAD_ROWS=0: never; 2: also the narrow block shapes (measurements)
bool ww_rows_takes(const WwPlan& pl, bool ups) {
  const char* e = getenv("MG_WGRAD_ROWS");  // (read per call: tests and A/B runs switch it inside one process)
  const int mode = e == nullptr ? 1 : atoi(e);
  if (mode == 0 || (pl.a.W % 32) != 0 || (pl.a.H % 2) != 0) return false;
  if (ups) {
    const char* u = getenv("MG_WGRAD_ROWS_UPS");  // (measurement switch: 0 = the up-sampled-input layers stay on the chunk-staged kernels)
    if (u != nullptr && atoi(u) == 0) return false;
  }
  if (pl.CT == 4 && pl.OT == 4) return false;  // (256 accumulators + raw set + two operand sets: 68 bytes of scratch per lane; not instantiated)
  return mode >= 2 || pl.CT * pl.OT >= 4;  // (blocks of one channel tile on either side: 2 MFMAs per wave and k-step -- the chunk-staged narrow form is as fast or faster, profiles/r06_wgrad_rows_steps.txt)
}

void plan_rows(WwPlan& pl) {
  WwArgs& a = pl.a;
  a.TBW = 16; a.TBH = 1; a.TBN = 1; a.lgTBW = 4; a.lgTBH = 0;
  {
    const char* e = getenv("MG_WGRAD_ROWS_ABLATE");  // measurement switch (wrong results): 2 = no staging, 4 = no barriers, 8 = staging never waited for, 16 = every stage re-reads the slab's first one, 64 = every second x copy instruction dropped
    if (e != nullptr) a.TBN |= atoi(e) & 126;
  }
  a.blocks_x = a.W / 32; a.blocks_y = a.H / 2; a.blocks_n = a.N;
  a.nblk = a.blocks_x * a.blocks_y * a.blocks_n;
  const int ny = pl.ncb * a.nob;
  int ns = mg_cu_count() / ny > 0 ? mg_cu_count() / ny : 1;  // one 8-wave workgroup per CU
  if (ns > a.nblk) ns = a.nblk;
  a.per = mg_cdiv(a.nblk, ns);
  pl.nsplit = mg_cdiv(a.nblk, a.per);
  pl.ws_floats = (size_t)pl.nsplit * (9 * (size_t)a.CinP * a.CoutP + a.CoutP);
  a.slab_b = a.slab + (size_t)pl.nsplit * 9 * a.CinP * a.CoutP;
  pl.rows = true;
}

template <int CT, int OT, bool UPS>
int launch_rows(const WwArgs& a, dim3 grid, hipStream_t s) {
  constexpr size_t lds = RwGeom<CT, OT, UPS>::lds_bytes();
  static MgPerDevice once;
  if (mg_first_use_on_device(once)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_wgrad_rows_mfma<CT, OT, UPS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
  }
  hipLaunchKernelGGL((wino_wgrad_rows_mfma<CT, OT, UPS>), grid, dim3(512), lds, s, a);
  MG_CHECK_LAUNCH("mg_wino3x3_wgrad (rows)");
  return MG_OK;
}

template <bool UPS>
int dispatch_rows(int CT, int OT, const WwArgs& a, dim3 grid, hipStream_t s) {
  switch (CT * 10 + OT) {
    case 11: return launch_rows<1, 1, UPS>(a, grid, s);
    case 12: return launch_rows<1, 2, UPS>(a, grid, s);
    case 13: return launch_rows<1, 3, UPS>(a, grid, s);
    case 14: return launch_rows<1, 4, UPS>(a, grid, s);
    case 21: return launch_rows<2, 1, UPS>(a, grid, s);
    case 22: return launch_rows<2, 2, UPS>(a, grid, s);
    case 23: return launch_rows<2, 3, UPS>(a, grid, s);
    case 24: return launch_rows<2, 4, UPS>(a, grid, s);
    case 31: return launch_rows<3, 1, UPS>(a, grid, s);
    case 32: return launch_rows<3, 2, UPS>(a, grid, s);
    case 33: return launch_rows<3, 3, UPS>(a, grid, s);
    case 34: return launch_rows<3, 4, UPS>(a, grid, s);
    case 41: return launch_rows<4, 1, UPS>(a, grid, s);
    case 42: return launch_rows<4, 2, UPS>(a, grid, s);
    case 43: return launch_rows<4, 3, UPS>(a, grid, s);
  }
  mg_set_error("mg_wino3x3_wgrad: internal tile error (CT=%d, OT=%d)", CT, OT);
  return MG_EINVAL;
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
  if (pl.rows) return ups ? dispatch_rows<true>(pl.CT, pl.OT, pl.a, grid, s) : dispatch_rows<false>(pl.CT, pl.OT, pl.a, grid, s);
  return ups ? dispatch_ww<true>(pl.CT, pl.OT, pl.a, grid, s) : dispatch_ww<false>(pl.CT, pl.OT, pl.a, grid, s);
}

}  // namespace

extern "C" int mg_wino3x3_wgrad_form(int N, int Cin, int Cout, int H, int W, int flags, int group_max_chunks) {
  if (N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || (H % 2) || (W % 2)) return -1;
  WwPlan pl;
  plan_ww(N, Cin, Cout, H, W, pl);
  const bool ups = (flags & MG_CONV_UPS_IN) != 0;
  const long long work = (long long)pl.a.nblk * pl.ncb * pl.a.nob;
  const bool small = group_max_chunks > 0 && ww_groupable(pl.CT, pl.OT) && work <= (long long)group_max_chunks * mg_cu_count();
  if (small || !ww_rows_takes(pl, ups)) return 0;
  return ups ? 2 : 1;
}

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
  if (ww_rows_takes(pl, (flags & MG_CONV_UPS_IN) != 0)) plan_rows(pl);  // (never more splits than the chunk plan the workspace is sized for)
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
    if (!small && ww_rows_takes(pl[i], (d[i].flags & MG_CONV_UPS_IN) != 0)) plan_rows(pl[i]);
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
