// Winograd F(2x2,3x3) convolution with one wave per tile block.  Written for the FEW-CHANNEL layers on large maps (16 .. 48 channels at
// 128x128 .. 512x512: where the reference trains, batch 6 at levels 6-7 -- /root/reference/music_gan/networks/generator.py:67-76,
// discriminator.py:60-70, train.py:43,101-109), it now carries every layer of 16 .. 160 out-channels from 2 048 tile blocks on.  Same arithmetic and the same packed filters as wino3x3.hip, bit for bit the same results; what differs
// is who builds the operands and what the vector unit has to do besides.  On gfx950 an fp32 MFMA and a vector instruction
// cost the SAME issue slots (157.3 TFLOP/s is also the v_pk_fma_f32 peak, and the two add up: profiles/r01_hw_valu_under_mfma.txt),
// so a layer with 2-6 channel chunks per tile block is bound by its vector instructions: wino3x3.hip spends 8.8-10 of them per MFMA
// there (staging through LDS, per-element predication, 64-bit address arithmetic, a twelve-variant generic epilogue).  Here
//   * the whole transformed filter bank of the workgroup's out-channels sits in LDS for the workgroup's lifetime (one copy at
//     start, 8 KB per 8-channel chunk and 16 out-channels; persistent workgroups, one per CU);
//   * ONE WAVE owns a tile block -- 16 horizontally adjacent tiles x all input channels x NIW x 16 out-channels -- and there is
//     no barrier after the start: lane (rq, col) holds the 4x4 patch of tile `col` for channel 8 ch + 2 rq + ks -- its own pixel pair
//     per row straight from global memory (one 8-byte load, 128 contiguous bytes per 16 lanes), the two halo columns from the
//     neighbouring lanes' pairs by DPP; only lanes 0 and 15 of a 16-lane row fetch the one halo pixel that belongs to the next tile
//     block (a 4-byte load whose offset is out of range for every other lane and where the image ends: the hardware bounds check
//     returns the zero padding without an access) -- transforms it IN REGISTERS with 16 packed adds, and the 16 components are
//     exactly the B operand of v_mfma_f32_16x16x4_f32 for k-index rq: the activations never touch LDS, no selects;
//   * shapes are restricted to what those layers have (W a multiple of 32, Cout and Cin of 16), so a tile block is valid or not
//     as a whole (one scalar branch) and every global access is a buffer access: 32-bit lane offset computed once per block +
//     scalar offset per plane / row;
//   * each epilogue kind is written for its own outputs: packed output transform, LeakyReLU mask bits by integer clamps of the
//     sign, mask selects as bit-field inserts, pooled outputs summed in the lane (a lane's 2x2 tile IS one pooled pixel);
//   * the eight waves of a workgroup take vertically adjacent tile rows (the two halo rows they share come out of the CU's own
//     cache), consecutive workgroups of an XCD horizontally adjacent blocks; the next chunk's rows -- at a block's last chunk:
//     the first chunk of the wave's next block -- are requested before the current chunk's MFMAs are issued; the kernel is
//     specialised per epilogue kind so that its block loop is straight-line code.
// A hardware rule found on the way (gfx950): a VGPR written by a vector instruction inside INLINE ASSEMBLY and read as an MFMA
// source by the next instructions needs wait states the compiler only inserts for instructions it scheduled itself -- without
// them the MFMA reads the old register contents (wrong sums, no fault).  The packed transforms below therefore end in `s_nop`.
// The same holds the other way round (inline-assembly vector instructions that READ MFMA results: the output transform), and a
// third rule: 16-byte buffer stores with a scalar soffset need a wait state before their data registers are overwritten (below).
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "mg_common.h"
#include "pack_kernels.h"
#include "wino_common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

enum StripKind {
  SK_ACT = 0,        // bias + LeakyReLU -> y (+ pooled)
  SK_ACT_POOL_MOUT,  // bias + LeakyReLU -> pooled + tile-mask bytes (y never written)
  SK_MB_POOL,        // x tile-mask bytes -> pooled
  SK_MASKF,          // x fp32 LeakyReLU mask of aux -> y (+ pooled)
  SK_UNPOOL,         // 0.25 * up2(result) x mask bytes of the layer below -> y at twice the size
  SK_BLEND_FWD,      // fade-in forward: act, tile mask out, alpha * new + (1 - alpha) * old -> y
  SK_BLEND_TAN,      // fade-in tangent: x mask bytes, blend -> y
  SK_BLEND_BWD,      // fade-in backward: two masked, scaled copies -> y, p
  SK_PN,             // bias + LeakyReLU + PixelNorm -> p, rn (+ y)
  SK_MB_Y,           // x tile-mask bytes -> y (a data gradient times the LeakyReLU derivative of the layer below, kept as bytes)
};

// (by value on purpose: __builtin_bit_cast applied to an ext-vector ELEMENT lvalue, e.g. bit_cast(unsigned, v4[g]), reads element 0
// whatever g is -- clang 19 / ROCm 7.2; found as "every out-channel of a lane gets channel 0's pooled value")
__device__ __forceinline__ unsigned f2u(float v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ int f2i(float v) { return __builtin_bit_cast(int, v); }
__device__ __forceinline__ int clamp01(int v) { return v < 0 ? 0 : (v > 1 ? 1 : v); }  // v_med3_i32
// (t ? x : y) for t in {0, -1}: one v_bfi_b32
__device__ __forceinline__ float sel_bits(int t, float x, float y) {
  return __builtin_bit_cast(float, (t & f2i(x)) | (~t & f2i(y)));
}

template <class F, int... Is>
__device__ __forceinline__ void sk_static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void sk_static_for(F&& f) {
  sk_static_for_impl(f, std::make_integer_sequence<int, N>{});
}
// 16 bytes of LDS at byte address addr + OFF as ONE ds_read_b128 the compiler does not schedule or count (sk_wait before the first use)
template <int OFF>
__device__ __forceinline__ f32x4 sk_lds128(unsigned addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ void sk_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void sk_tie(f32x4& v) { asm volatile("" : "+v"(v)); }

// KIND (StripKind) and POOL (SK_ACT / SK_MASKF: also write the pooled tensor) are compile-time: with the epilogue selected by a
// run-time branch the block loop has a control-flow join behind stores whose number the compiler cannot count, and gfx9 has ONE
// counter for loads and stores -- it then waits `vmcnt(0)` for the next block's prefetched rows, i.e. for the previous block's
// stores to be acknowledged (measured: every wave 57 % of its time in s_waitcnt).  One straight-line loop per kind instead.
// The body of a workgroup: NIW = out-channel tiles this workgroup computes (accumulators, MFMAs, epilogue), NLDS = tiles per chunk in
// its LDS filter image (>= NIW: a row whose second tile is padding keeps the two-tile image and computes one), ct0 = its first tile,
// wg / G = its index among / the number of the workgroups that share its tiles (they walk all tile blocks with stride G).
template <int NIW, int NLDS, int NWAVE, int KIND, bool POOL>
__device__ __forceinline__ void strip_body(const WinoArgs& a, const int ct0, const int wg, const int G) {
  constexpr int kind = KIND;
  constexpr int NTHR = 64 * NWAVE;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Us = smem;  // [chunk][NIW tiles][8 component pairs][64 lanes][4]: a verbatim copy of the packed global layout
  const int tid = threadIdx.x;
  const int lane = tid & 63, col = lane & 15, rq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int HW = a.H * a.W;
  const int Ht = a.H >> 1, Wt = a.W >> 1;
  const int PP = Ht * Wt;  // one pooled plane

  {  // the filter bank of this workgroup's out-channels, once
    const int n4 = a.nchunk * NLDS * 512;  // 16-byte pieces
    const f32x4* src = reinterpret_cast<const f32x4*>(a.up);
    f32x4* dst = reinterpret_cast<f32x4*>(Us);
    for (int i = tid; i < n4; i += NTHR) {
      const int ch = i / (NLDS * 512), r = i - ch * (NLDS * 512);
      dst[i] = src[((size_t)ch * a.NT + ct0) * 512 + r];
    }
  }
  __syncthreads();

  // Addressing: every global access is  base(image) + [lane part: VGPR, fixed for the kernel] + [block / row / plane part: SGPR].
  // Work items: groups of NWAVE vertically adjacent tile blocks (16 tiles x 1 tile row), group g = (image * groups_y + gy) *
  // blocks_x + bx, walked with stride gridDim.x; (bx, gy, image) advance by precomputed deltas with carries -- scalar adds, no
  // division per block (a division by a run-time value costs ~20 vector instructions even for a uniform operand).  The two input
  // rows a block shares with the block below are read by two waves of one workgroup at about the same time.  (Measured instead: a
  // wave walking a vertical segment top to bottom, re-reading the shared rows one block time later -- by then they have left the
  // 4 MB L2 of the XCD: rocprofv3 FETCH_SIZE 2.5x the input instead of 1.7x, 32 -> 16 @512 x 18 images, and 15 % slower.)
  const int groups_y = Ht / NWAVE;
  const int nitems = a.N * groups_y * a.blocks_x;
  const int first = mg_xcd_remap(wg, G);  // (row sizes are multiples of 8: workgroup wg of a row sits on XCD wg % 8)
  if (first >= nitems) return;
  const int dbx = G % a.blocks_x, dgy = (G / a.blocks_x) % groups_y, dn = G / (a.blocks_x * groups_y);
  int bx = first % a.blocks_x, gy = (first / a.blocks_x) % groups_y, n0 = first / (a.blocks_x * groups_y);
  int item = first;
  int by = gy * NWAVE + wave;  // = tile row of this wave's block
  auto step = [&](int& item_, int& bx_, int& gy_, int& n_) __attribute__((always_inline)) {
    item_ += G;
    bx_ += dbx;
    const int c1 = bx_ >= a.blocks_x ? 1 : 0;
    bx_ -= c1 * a.blocks_x;
    gy_ += dgy + c1;
    const int c2 = gy_ >= groups_y ? 1 : 0;
    gy_ -= c2 * groups_y;
    n_ += dn + c2;
  };
  auto advance = [&]() __attribute__((always_inline)) {
    step(item, bx, gy, n0);
    by = gy * NWAVE + wave;
  };
  // (Measured and not kept: a second walker two blocks ahead pulling that block's two new rows into the L2 with 16-byte loads nobody
  // waits for -- +3 % on 16 -> 32 @512, +15 % on the 32 -> 16 data gradient: what the memory system costs these launches is not
  // the latency of L2 misses, profiles/r05_ab_wino_strip.txt.)

  // input rows: lane part = own pixel pair of channel 2 rq; the halo pixels one float to the left / right.  Rows outside the image
  // (uniform per wave: the first row of tile row 0, the last of tile row Ht - 1, everything of a tile row beyond the image) are read
  // through a descriptor of zero records, the image's left / right edge through an out-of-range lane offset: both return 0.0.
  const int lp = ((2 * rq) * HW + 2 * col) * 4;
  unsigned vP, vX;  // own pixel pair; the halo pixel only the first / last lane of a 16-lane row has to fetch (see load_rows)
  int rowoff[4];                     // scalar: (row * W + 32 bx) * 4, row = 2 by - 1 + r
  int rowrec[4];                     // scalar: num_records of the row's descriptor (0 = reads as zero)
  const float* img_base;
  auto block_geometry = [&]() __attribute__((always_inline)) {
    img_base = a.x + (size_t)n0 * a.Cin * HW;
    const int xoff = bx * 128;
    vP = (unsigned)(lp + xoff);
    vX = col == 0 ? (bx == 0 ? 0x80000000u : vP - 4u) : ((col == 15 && bx != a.blocks_x - 1) ? vP + 8u : 0x80000000u);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int Y = 2 * by - 1 + r;
      rowoff[r] = Y * a.W * 4;
      rowrec[r] = (item < nitems && Y >= 0 && Y < a.H && !(a.TBN & 2)) ? a.Cin * HW * 4 : 0;  // (behind the last item: nothing is fetched)
    }
  };

  f32x4 acc[16][NIW];
  // Per row and channel a lane fetches its own pixel pair and -- lanes 0 and 15 of a 16-lane row only, every other lane's offset is
  // out of range and costs no fetch -- the one halo pixel that belongs to a neighbouring tile BLOCK; the halo pixels inside the block
  // are the neighbouring lanes' own pixels and come by DPP in the transform (two loads per row and channel instead of three, 8
  // registers fewer: -2..-6 % per launch, -10 % on the one-tile data gradient, bit-identical).
  f32x2 rP[2][4];  // [k-step][row]: own pixel pair
  float rX[2][4];  // the block-edge halo pixel (left for lane 0, right for lane 15 of a row)
  f32x2 V[2][8];   // [k-step][component pair]: the B operands of this lane

  auto load_rows = [&](int ch) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(img_base), 0, rowrec[r], 0x00020000);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int soff = (ch * WCC + ks) * HW * 4 + rowoff[r];
        rP[ks][r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)vP, soff, 0));
        rX[ks][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)vX, soff, 0));
      }
    }
  };

  // B^T d B of the two patches (channels 8 ch + 2 rq, + 1) this lane holds -- wino3x3.hip's staging arithmetic, into registers.
  // Patch row r = {left halo | own pair P | right halo} as two register pairs E = (left, right), P = (own x, own y); rows act
  // element-wise on the pairs, columns are (v0, v3) = (e0 - p1, p0 - e1), (v1, v2) = (p0 + p1, p1 - p0).
  auto transform_rows = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f32x2 rE[4];  // (left, right) halo pixels: lane - 1's right pixel / lane + 1's left pixel; at the ends of a 16-lane row the DPP
                    // source is invalid and the destination keeps `old` = the fetched block-edge pixel (0.0 at the image edge)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int old = __builtin_bit_cast(int, rX[ks][r]);
        rE[r][0] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(old, __builtin_bit_cast(int, f2i(rP[ks][r][1])), 0x111, 0xf, 0xf, false));  // row_shr:1
        rE[r][1] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(old, __builtin_bit_cast(int, f2i(rP[ks][r][0])), 0x101, 0xf, 0xf, false));  // row_shl:1
      }
      f32x2 UE[4], UP[4];
      UE[0] = pk_sub(rE[0], rE[2]);  UP[0] = pk_sub(rP[ks][0], rP[ks][2]);
      UE[1] = rE[1] + rE[2];         UP[1] = rP[ks][1] + rP[ks][2];
      UE[2] = pk_sub(rE[2], rE[1]);  UP[2] = pk_sub(rP[ks][2], rP[ks][1]);
      UE[3] = pk_sub(rE[1], rE[3]);  UP[3] = pk_sub(rP[ks][1], rP[ks][3]);
      // (v0, v3) and (v1, v2) of the four rows: eight packed adds with swizzle / negate modifiers in ONE statement, and one wait state
      // behind the last of them (the hardware rule in the header: these registers are MFMA sources)
      asm("v_pk_add_f32 %0, %8, %12 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,0]\n\t"
          "v_pk_add_f32 %1, %12, %12 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]\n\t"
          "v_pk_add_f32 %2, %9, %13 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,0]\n\t"
          "v_pk_add_f32 %3, %13, %13 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]\n\t"
          "v_pk_add_f32 %4, %10, %14 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,0]\n\t"
          "v_pk_add_f32 %5, %14, %14 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]\n\t"
          "v_pk_add_f32 %6, %11, %15 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,0]\n\t"
          "v_pk_add_f32 %7, %15, %15 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]\n\t"
          "s_nop 1"
          : "=&v"(V[ks][0]), "=&v"(V[ks][1]), "=&v"(V[ks][2]), "=&v"(V[ks][3]), "=&v"(V[ks][4]), "=&v"(V[ks][5]), "=&v"(V[ks][6]), "=&v"(V[ks][7])
          : "v"(UE[0]), "v"(UE[1]), "v"(UE[2]), "v"(UE[3]), "v"(UP[0]), "v"(UP[1]), "v"(UP[2]), "v"(UP[3]));
    }
  };

  // `first_`: the block's first chunk starts its sums from the zero constant (no 64 x NIW register clears per block)
  // The filters of component pair cp + 1 are REQUESTED before the MFMAs of pair cp are issued and waited for after them: the reads are
  // inline assembly (outside hipcc's s_waitcnt bookkeeping: sk_wait + the register ties below), fenced against the scheduler, which
  // otherwise moves them down to just above their first use -- an LDS round trip in front of every group of MFMAs.
  const unsigned us_lane = (unsigned)reinterpret_cast<size_t>(Us + lane * 4);  // (low 32 bits of a shared-aperture address = LDS offset)
  auto mfma_chunk = [&](int ch, auto first_) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_)::value;
    const unsigned ub = us_lane + (unsigned)(ch * NLDS) * 8192u;
    f32x4 bv[2][NIW];
    sk_static_for<NIW>([&](auto ni_) __attribute__((always_inline)) {
      constexpr int ni = decltype(ni_)::value;
      bv[0][ni] = sk_lds128<ni * 8192>(ub);
    });
    sk_static_for<8>([&](auto cp_) __attribute__((always_inline)) {
      constexpr int cp = decltype(cp_)::value;
      constexpr int cur = cp & 1, nxt = cur ^ 1;
      sk_wait();
#pragma unroll
      for (int ni = 0; ni < NIW; ++ni) sk_tie(bv[cur][ni]);
      if constexpr (cp + 1 < 8) {
        sk_static_for<NIW>([&](auto ni_) __attribute__((always_inline)) {
          constexpr int ni = decltype(ni_)::value;
          bv[nxt][ni] = sk_lds128<ni * 8192 + (cp + 1) * 1024>(ub);
        });
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int par = 0; par < 2; ++par)
#pragma unroll
          for (int ni = 0; ni < NIW; ++ni) {
            const f32x4 c0 = (FIRST && ks == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[2 * cp + par][ni];
            acc[2 * cp + par][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[cur][ni][ks * 2 + par], V[ks][cp][par], c0, 0, 0, 0);
          }
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // ------------------------------------------------------------------ epilogue pieces
  const bool lrelu = (a.flags & MG_CONV_LRELU) != 0;
  const float slope = a.slope;
  const float slope_eff = lrelu ? slope : 1.0f;
  // (the lane's 4 x NIW bias values are re-read per block -- 16 bytes per tile out of the CU's cache -- rather than held in 4 x NIW
  // registers across the MFMA loop: the two-tile kernels sit at the 256-register limit of two waves per SIMD)
  auto bias4 = [&](int ni) __attribute__((always_inline)) {
    f32x4 b = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.bias != nullptr) b = *reinterpret_cast<const f32x4*>(a.bias + (ct0 + ni) * 16 + rq * 4);
    return b;
  };
  // A^T M A of out-channel tile ni -> r4[2*i + j] = output pixel (i, j) of the lane's tile, one f32x4 over its 4 out-channels g.
  // The accumulator of component (xi, nu) is acc[4*xi + slot(nu)], slots [nu0, nu3, nu1, nu2]; wino_epilogue.h's arithmetic.
  auto xform = [&](int ni, f32x4 (&r4)[4]) __attribute__((always_inline)) {
    // The packed subtractions below are inline assembly and read MFMA results: the wait states a vector instruction needs behind
    // the MFMA that wrote its source (hipcc inserts them for instructions it knows) are spent here, in two statements that name
    // the tile's 16 accumulators -- without them the first subtractions read registers the matrix pipe has not written yet
    // (seen as wrong odd output rows in out-channels 4 rq + {2, 3}, run-to-run different).
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[0][ni]), "+v"(acc[1][ni]), "+v"(acc[2][ni]), "+v"(acc[3][ni]), "+v"(acc[4][ni]),
                 "+v"(acc[5][ni]), "+v"(acc[6][ni]), "+v"(acc[7][ni]));
    asm volatile("" : "+v"(acc[8][ni]), "+v"(acc[9][ni]), "+v"(acc[10][ni]), "+v"(acc[11][ni]), "+v"(acc[12][ni]), "+v"(acc[13][ni]),
                 "+v"(acc[14][ni]), "+v"(acc[15][ni]));
    constexpr int SL[4] = {0, 2, 3, 1};
    f32x4 s0[4], s1[4];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      const f32x4 m0 = acc[SL[nu]][ni], m1 = acc[4 + SL[nu]][ni], m2 = acc[8 + SL[nu]][ni], m3 = acc[12 + SL[nu]][ni];
      s0[nu] = (m0 + m1) + m2;
      s1[nu] = pk_sub4(pk_sub4(m1, m2), m3);
    }
    r4[0] = (s0[0] + s0[1]) + s0[2];
    r4[1] = pk_sub4(pk_sub4(s0[1], s0[2]), s0[3]);
    r4[2] = (s1[0] + s1[1]) + s1[2];
    r4[3] = pk_sub4(pk_sub4(s1[1], s1[2]), s1[3]);
  };
  auto activate = [&](int ni, f32x4 (&r4)[4]) __attribute__((always_inline)) {  // bias + LeakyReLU: max(v, slope * v), 0 < slope <= 1
    const f32x4 b4 = bias4(ni);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = r4[q] + b4;
      const f32x4 w = v * slope_eff;
#pragma unroll
      for (int g = 0; g < 4; ++g) r4[q][g] = fmaxf(v[g], w[g]);
    }
  };
  // bit 2i+j <-> pixel (i, j) > 0, from the IEEE sign / zero pattern: clamp(int bits, 0, 1) is 1 exactly for positive values
  auto mask_bits = [&](const f32x4 (&r4)[4], int g) __attribute__((always_inline)) {
    const int b0 = clamp01(f2i(r4[0][g])), b1 = clamp01(f2i(r4[1][g])), b2 = clamp01(f2i(r4[2][g])), b3 = clamp01(f2i(r4[3][g]));
    return ((b0 | (b1 << 1)) | (b2 << 2)) | (b3 << 3);
  };
  // r4 *= (bit ? 1 : slope) for the 16 values of tile ni, mb[g] = mask byte of out-channel g
  auto apply_mask_bytes = [&](f32x4 (&r4)[4], const unsigned (&mb)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 os = r4[q] * slope;
#pragma unroll
      for (int g = 0; g < 4; ++g) r4[q][g] = sel_bits(__builtin_amdgcn_sbfe((int)mb[g], q, 1), r4[q][g], os[g]);
    }
  };

  auto acc_fence = [&]() __attribute__((always_inline)) {};
  // lane parts of the epilogue's offsets (bytes inside one image of Cout x H x W / Cout x Ht x Wt)
  const int ly = ((rq * 4) * HW + 2 * col) * 4;
  const int lpo = ((rq * 4) * PP + col) * 4;
  using T_ = std::true_type;
  using F_ = std::false_type;

  // Loop shape (gfx9 has ONE counter, in issue order, for loads and stores): the rows of a block's first chunk are requested under
  // the previous block's last MFMAs, the epilogue's stores follow them, and the transform that consumes them sits at the BOTTOM of
  // the loop body, behind those stores in straight-line code -- there the compiler can wait `vmcnt(number of stores)`.  With that
  // transform at the top of the body the loop header joins "entered from the prologue, no stores behind the loads" with "the back
  // edge, stores behind them" and the wait becomes vmcnt(0): every wave then sits out the write acknowledgements of its own stores.
  // (Measured and not kept, with the second row set this leaves room for: a chunk's rows requested TWO chunks ahead -- 256 registers with
  // 2-10 spilled, +5..+12 % on every shape, profiles/r05_ab_wino_strip.txt: what the memory system costs is not an HBM round trip
  // that one more chunk of MFMAs would cover.)
  const int nch = a.nchunk;
  block_geometry();
  load_rows(0);
  transform_rows();
#pragma nounroll
  for (;;) {
    const int ebx = bx, eby = by, en0 = n0;  // this block, for the epilogue
    load_rows(1);
    mfma_chunk(0, T_{});
    for (int ch = 1; ch + 1 < nch; ++ch) {
      transform_rows();
      load_rows(ch + 1);  // in flight during the MFMAs below
      mfma_chunk(ch, F_{});
    }
    transform_rows();
    // the first chunk of this wave's next block rides under the last chunk's MFMAs and the epilogue (requested unconditionally,
    // through zero-record descriptors behind the last block: the loop body stays free of joins)
    advance();
    block_geometry();
    load_rows(0);
    mfma_chunk(nch - 1, F_{});
    // the next block's first chunk is transformed BEFORE this block's stores are issued: a wait for loads that have stores behind
    // them is a wait for those stores too (one counter on gfx9; the compiler waits vmcnt(0) whenever both kinds are pending)
    __builtin_amdgcn_sched_barrier(0);
    acc_fence();
    transform_rows();
    __builtin_amdgcn_sched_barrier(0);

    // ---------------------------------------------------------------- epilogue
    {
      const int sy = ((2 * eby) * a.W + 32 * ebx) * 4;  // scalar parts of this block
      const int sp = (eby * Wt + 16 * ebx) * 4;
      const int yoff = ly, poff = lpo;
      const int TX = ebx * 16 + col;
      const int W4 = a.W * 4;
      auto rs = [&](const void* base, size_t img_elems, int esize) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(base)) + (size_t)en0 * img_elems * esize,
                                                 0, (a.TBN & 4) ? 0 : (int)(img_elems * esize), 0x00020000);
      };
      const size_t full = (size_t)a.Cout * HW, pooled_n = (size_t)a.Cout * PP;
      auto store_rows = [&](__amdgpu_buffer_rsrc_t r, const f32x4 (&r4)[4], int ni) __attribute__((always_inline)) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int so = ((ct0 + ni) * 16 + g4) * HW * 4 + sy;
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{r4[0][g4], r4[1][g4]}), r, yoff, so, 0);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{r4[2][g4], r4[3][g4]}), r, yoff, so + W4, 0);
        }
      };
      auto load_rows_f = [&](__amdgpu_buffer_rsrc_t r, f32x4 (&o4)[4], int ni) __attribute__((always_inline)) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int so = ((ct0 + ni) * 16 + g4) * HW * 4 + sy;
          const f32x2 t0 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, yoff, so, 0));
          const f32x2 t1 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, yoff, so + W4, 0));
          o4[0][g4] = t0[0]; o4[1][g4] = t0[1]; o4[2][g4] = t1[0]; o4[3][g4] = t1[1];
        }
      };
      auto store_pooled = [&](__amdgpu_buffer_rsrc_t r, const f32x4 (&r4)[4], int ni) __attribute__((always_inline)) {
        const f32x4 pl = ((r4[0] + r4[1]) + (r4[2] + r4[3])) * 0.25f;  // the lane's 2x2 tile is one pooled pixel
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
          __builtin_amdgcn_raw_buffer_store_b32(f2u(pl[g4]), r, poff, ((ct0 + ni) * 16 + g4) * PP * 4 + sp, 0);
      };
      auto load_mask_bytes = [&](__amdgpu_buffer_rsrc_t r, unsigned (&mb)[4], int ni) __attribute__((always_inline)) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) mb[g4] = __builtin_amdgcn_raw_buffer_load_b8(r, poff >> 2, ((ct0 + ni) * 16 + g4) * PP + (sp >> 2), 0);
      };
      auto store_mask_bytes = [&](__amdgpu_buffer_rsrc_t r, const f32x4 (&r4)[4], int ni) __attribute__((always_inline)) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
          __builtin_amdgcn_raw_buffer_store_b8((unsigned char)mask_bits(r4, g4), r, poff >> 2, ((ct0 + ni) * 16 + g4) * PP + (sp >> 2), 0);
      };
      constexpr bool act_on = false;  // (the masked kinds are bias-free here -- mgi_wino_strip_takes -- : plain A^T M A)

      if constexpr (kind == SK_ACT_POOL_MOUT) {
        const auto rp = rs(a.p, pooled_n, 4), rm = rs(a.mo, pooled_n, 1);
#pragma unroll
        for (int ni = 0; ni < NIW; ++ni) {
          f32x4 r4[4];
          xform(ni, r4);
          activate(ni, r4);
          store_mask_bytes(rm, r4, ni);
          store_pooled(rp, r4, ni);
        }
      } else if constexpr (kind == SK_MB_POOL) {
        const auto rp = rs(a.p, pooled_n, 4), rm = rs(a.mi, pooled_n, 1);
#pragma unroll
        for (int ni = 0; ni < NIW; ++ni) {
          unsigned mb[4];
          load_mask_bytes(rm, mb, ni);
          f32x4 r4[4];
          xform(ni, r4);
          if (act_on) activate(ni, r4);
          apply_mask_bytes(r4, mb);
          store_pooled(rp, r4, ni);
        }
      } else if constexpr (kind == SK_ACT) {
        const auto ry = rs(a.y, full, 4), rp = rs(a.p, pooled_n, 4);
#pragma unroll
        for (int ni = 0; ni < NIW; ++ni) {
          f32x4 r4[4];
          xform(ni, r4);
          activate(ni, r4);
          store_rows(ry, r4, ni);
          if (POOL) store_pooled(rp, r4, ni);
        }
      } else if constexpr (kind == SK_MASKF) {
        const auto ry = rs(a.y, full, 4), rx = rs(a.aux, full, 4), rp = rs(a.p, pooled_n, 4);
#pragma unroll
        for (int ni = 0; ni < NIW; ++ni) {
          f32x4 ax[4];
          load_rows_f(rx, ax, ni);
          f32x4 r4[4];
          xform(ni, r4);
          if (act_on) activate(ni, r4);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 os = r4[q] * slope;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) r4[q][g4] = ax[q][g4] > 0.f ? r4[q][g4] : os[g4];
          }
          store_rows(ry, r4, ni);
          if (POOL) store_pooled(rp, r4, ni);
        }
      } else if constexpr (kind == SK_UNPOOL) {
        // y (N, Cout, 2H, 2W): pixel (Y, X) of the result spreads over y[2Y + r][2X + jj] * 0.25, times lrelu'(mask byte bit 2r + jj)
        const auto ry = rs(a.y, 4 * full, 4), rm = rs(a.aux, full, 1);
        const int uoff = yoff >> 2, su = sy >> 2;                   // mask bytes: one per result pixel
        const int y2off = ((rq * 4) * 4 * HW + 4 * col) * 4;       // y: 4 floats per (pixel row, r): 4x the plane, 2x the row, 2x the column
        const int sy2 = ((4 * eby) * (2 * a.W) + 64 * ebx) * 4;
        const float qh = 0.25f, ql = 0.25f * slope;
#pragma unroll
        for (int ni = 0; ni < NIW; ++ni) {
          unsigned mw[4][2];
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
            for (int i = 0; i < 2; ++i)
              mw[g4][i] = __builtin_amdgcn_raw_buffer_load_b16(rm, uoff, ((ct0 + ni) * 16 + g4) * HW + i * a.W + su, 0);
          f32x4 r4[4];
          xform(ni, r4);
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const float vh[2] = {r4[2 * i][g4] * qh, r4[2 * i + 1][g4] * qh}, vl[2] = {r4[2 * i][g4] * ql, r4[2 * i + 1][g4] * ql};
#pragma unroll
              for (int r = 0; r < 2; ++r) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                  for (int jj = 0; jj < 2; ++jj)
                    o[e * 2 + jj] = sel_bits(__builtin_amdgcn_sbfe((int)mw[g4][i], 8 * e + 2 * r + jj, 1), vh[e], vl[e]);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ry, y2off,
                                                       (((ct0 + ni) * 16 + g4) * 4 * HW + (2 * i + r) * 2 * a.W) * 4 + sy2, 0);
                // gfx950: a buffer store of more than 8 bytes still reads its data registers when the next instructions
                // issue -- also with a scalar soffset, for which hipcc assumes no hazard and lets the very next vector
                // instruction overwrite them (seen: lanes 12-15 of a row group storing the NEXT channel's values)
                // (the asm names the data registers, so that nothing that overwrites them can be scheduled in front of it)
                asm volatile("s_nop 1" : "+v"(o) : : "memory");
              }
            }
        }
      } else if constexpr (kind == SK_BLEND_FWD || kind == SK_BLEND_TAN) {
        const auto ry = rs(a.y, full, 4), ro = rs(a.other, full, 4);
        const auto rm = rs(kind == SK_BLEND_FWD ? static_cast<const void*>(a.mo) : static_cast<const void*>(a.mi), pooled_n, 1);
        const float bca = a.coef[0], bcb = a.coef[1];
#pragma unroll
        for (int ni = 0; ni < NIW; ++ni) {
          f32x4 o4[4];
          load_rows_f(ro, o4, ni);
          unsigned mb[4] = {15u, 15u, 15u, 15u};
          if (kind == SK_BLEND_TAN) load_mask_bytes(rm, mb, ni);
          f32x4 r4[4];
          xform(ni, r4);
          if (kind == SK_BLEND_FWD || act_on) activate(ni, r4);
          if (kind == SK_BLEND_TAN) apply_mask_bytes(r4, mb);
          else store_mask_bytes(rm, r4, ni);
#pragma unroll
          for (int q = 0; q < 4; ++q) r4[q] = bca * r4[q] + bcb * o4[q];
          store_rows(ry, r4, ni);
        }
      } else if constexpr (kind == SK_BLEND_BWD) {
        // y = (alpha * acc) * lrelu'(new branch: tile-mask bytes a.mi),  p = ((1 - alpha) * acc) * lrelu'(old branch activation a.other)
        const auto ry = rs(a.y, full, 4), rp = rs(a.p, full, 4), ro = rs(a.other, full, 4), rm = rs(a.mi, pooled_n, 1);
        const float ca = a.coef[0], cb = a.coef[1];
#pragma unroll
        for (int ni = 0; ni < NIW; ++ni) {
          f32x4 o4[4];
          load_rows_f(ro, o4, ni);
          unsigned mb[4];
          load_mask_bytes(rm, mb, ni);
          f32x4 r4[4], ra[4], rb[4];
          xform(ni, r4);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            ra[q] = ca * r4[q];
            rb[q] = cb * r4[q];
            const f32x4 bs = rb[q] * slope;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) rb[q][g4] = o4[q][g4] > 0.f ? rb[q][g4] : bs[g4];
          }
          apply_mask_bytes(ra, mb);
          store_rows(ry, ra, ni);
          store_rows(rp, rb, ni);
        }
      } else if constexpr (kind == SK_MB_Y) {
        const auto ry = rs(a.y, full, 4), rm = rs(a.mi, pooled_n, 1);
#pragma unroll
        for (int ni = 0; ni < NIW; ++ni) {
          unsigned mb[4];
          load_mask_bytes(rm, mb, ni);
          f32x4 r4[4];
          xform(ni, r4);
          apply_mask_bytes(r4, mb);
          store_rows(ry, r4, ni);
        }
      } else {  // SK_PN: all channels of a pixel are in this wave (NIW tiles x 4 row groups x 4)
        const auto rp = rs(a.p, full, 4);  // (y is not written: mgi_wino_strip_takes)
        f32x4 o[NIW][4];
#pragma unroll
        for (int ni = 0; ni < NIW; ++ni) {
          xform(ni, o[ni]);
          activate(ni, o[ni]);
        }
        float rnv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          // summed in wino3x3.hip's order (there every 16-channel tile is a wave group of its own: per-tile sums over the lane's
          // 4 channels, then the 4 row groups, then the tiles) -- the same bits
          float t = 0.f;
#pragma unroll
          for (int ni = 0; ni < NIW; ++ni) {
            float tn = 0.f;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) tn += o[ni][q][g4] * o[ni][q][g4];
            tn += __shfl_xor(tn, 16);
            tn += __shfl_xor(tn, 32);
            t += tn;
          }
          rnv[q] = 1.0f / sqrtf(t / (float)a.Cout + PN_EPS);
        }
#pragma unroll
        for (int ni = 0; ni < NIW; ++ni) {
#pragma unroll
          for (int q = 0; q < 4; ++q) o[ni][q] = o[ni][q] * rnv[q];
          store_rows(rp, o[ni], ni);
        }
        if (rq == 0 && a.rn != nullptr) {
          float* rn = a.rn + ((size_t)en0 * a.H + 2 * eby) * a.W + 2 * TX;
          *reinterpret_cast<float2*>(rn) = make_float2(rnv[0], rnv[1]);
          *reinterpret_cast<float2*>(rn + a.W) = make_float2(rnv[2], rnv[3]);
        }
      }
    }
    if (item >= nitems) break;
  }
}

// Workgroup rows: row r owns out-channel tiles r * NIW .. r * NIW + NIW - 1 and walks ALL tile blocks.  The grid is one-dimensional, rows
// one after the other: a.TBW workgroups per full row, a.TBH for the last row when its second tile is a padding tile (an odd tile
// count) -- that row runs the ONE-tile body on the two-tile filter image (no MFMAs, no epilogue for the padding tile) and gets fewer
// workgroups, in proportion to its work.
template <int NIW, int NWAVE, int KIND, bool POOL>
// (one tile per wave: at most 168 registers, i.e. three waves per SIMD -- except the un-pooling and PixelNorm epilogues, which do not fit)
__global__ void __launch_bounds__(64 * NWAVE, (NIW == 1 && KIND != SK_UNPOOL && KIND != SK_PN) ? 3 : 1) wino3x3_strip(const WinoArgs a) {
  const int g_full = a.TBW, g_pad = a.TBH, nrows = a.lgTBW;
  const int row = ((int)blockIdx.x < (nrows - (g_pad ? 1 : 0)) * g_full) ? (int)blockIdx.x / g_full : nrows - 1;
  const int wg = (int)blockIdx.x - row * g_full;
  if (NIW == 2 && g_pad != 0 && row == nrows - 1) strip_body<1, NIW, NWAVE, KIND, POOL>(a, row * NIW, wg, g_pad);
  else strip_body<NIW, NIW, NWAVE, KIND, POOL>(a, row * NIW, wg, g_full);
}

template <int NIW, int NWAVE, int KIND, bool POOL>
int launch_strip(const WinoArgs& a, dim3 grid, hipStream_t s) {
  const size_t lds = (size_t)a.nchunk * NIW * 2048 * sizeof(float);
  static MgPerDevice once;
  const void* fn = reinterpret_cast<const void*>(&wino3x3_strip<NIW, NWAVE, KIND, POOL>);
  if (mg_first_use_on_device(once)) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  // persistent workgroups: as many per CU as registers and the filter bank's LDS allow (the one-tile kernels: two, i.e. four waves
  // per SIMD -- the waves are independent, occupancy is what hides a block's load latency), never more than there are groups
  // (the occupancy query is a runtime call of microseconds: asked once per (device, LDS size) of this instantiation -- benign race:
  // every thread computes the same value)
  static struct { int dev; size_t lds; int v; } occ[8];
  static int nocc = 0;
  int per_cu = 0;
  const int dev = mg_current_device();
  for (int i = 0; i < nocc && i < 8; ++i)
    if (occ[i].dev == dev && occ[i].lds == lds) per_cu = occ[i].v;
  if (per_cu == 0) {
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64 * NWAVE, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    if (per_cu > 16 / NWAVE) per_cu = 16 / NWAVE;
    if (nocc < 8) { occ[nocc].dev = dev; occ[nocc].lds = lds; occ[nocc].v = per_cu; ++nocc; }
  }
  {
    const char* e = getenv("MG_WINO_STRIP_WGS");  // measurement switch (tools/ab_wino_strip.py flips it inside one process): workgroups per CU
    if (e != nullptr && atoi(e) >= 1) per_cu = atoi(e);
  }
  // rows of workgroups (grid.y of the caller = number of rows) laid out in a 1-D grid; a last row whose second tile is padding does
  // ~0.7 of a full row's work (half the MFMAs and epilogue, the same input loads and transform; swept 0.6 .. 1.0 on 64->48@128,
  // 64->80@64, 48->48@128: 0.7) and gets that share of the workgroups
  const int rows = (int)grid.y, nt = a.Cout / 16;
  const bool has_pad = NIW == 2 && (nt & 1) != 0 && rows > 1;
  const int total = per_cu * mg_cu_count();
  static const double pad_w = getenv("MG_WINO_STRIP_PADW") ? atof(getenv("MG_WINO_STRIP_PADW")) : 0.7;
  int g_pad = 0, g_full;
  if (has_pad) {
    g_pad = (int)(total * pad_w / ((rows - 1) + pad_w)) & ~7;
    if (g_pad < 8) g_pad = 8;
    g_full = ((total - g_pad) / (rows - 1)) & ~7;
  } else {
    g_full = (total / rows) & ~7;
  }
  if (g_full < 8) g_full = 8;
  const int nitems = a.N * (a.blocks_y / NWAVE) * a.blocks_x;
  if (g_full > nitems) g_full = nitems;  // (small launches: fewer workgroups than CUs; the XCD mapping then is whatever it is)
  if (g_pad > nitems) g_pad = nitems;
  WinoArgs b = a;
  b.TBW = g_full; b.TBH = g_pad; b.lgTBW = rows;
  grid = dim3((unsigned)((rows - (g_pad ? 1 : 0)) * g_full + g_pad), 1, 1);
  hipLaunchKernelGGL((wino3x3_strip<NIW, NWAVE, KIND, POOL>), grid, dim3(64 * NWAVE), lds, s, b);
  MG_CHECK_LAUNCH("mg_wino3x3 (strip)");
  return MG_OK;
}

template <int NIW, int NWAVE>
int launch_strip_kind(const WinoArgs& a, int kind, dim3 grid, hipStream_t s) {
  const bool pool = (a.flags & MG_CONV_POOL_OUT) != 0;
  switch (kind) {
    case SK_ACT: return pool ? launch_strip<NIW, NWAVE, SK_ACT, true>(a, grid, s) : launch_strip<NIW, NWAVE, SK_ACT, false>(a, grid, s);
    case SK_ACT_POOL_MOUT: return launch_strip<NIW, NWAVE, SK_ACT_POOL_MOUT, true>(a, grid, s);
    case SK_MB_POOL: return launch_strip<NIW, NWAVE, SK_MB_POOL, true>(a, grid, s);
    case SK_MASKF: return pool ? launch_strip<NIW, NWAVE, SK_MASKF, true>(a, grid, s) : launch_strip<NIW, NWAVE, SK_MASKF, false>(a, grid, s);
    case SK_UNPOOL: return launch_strip<NIW, NWAVE, SK_UNPOOL, false>(a, grid, s);
    case SK_BLEND_FWD: return launch_strip<NIW, NWAVE, SK_BLEND_FWD, false>(a, grid, s);
    case SK_BLEND_TAN: return launch_strip<NIW, NWAVE, SK_BLEND_TAN, false>(a, grid, s);
    case SK_BLEND_BWD: return launch_strip<NIW, NWAVE, SK_BLEND_BWD, false>(a, grid, s);
    case SK_MB_Y: return launch_strip<NIW, NWAVE, SK_MB_Y, false>(a, grid, s);
    default:
      if constexpr (NIW <= 2) return launch_strip<NIW, NWAVE, SK_PN, false>(a, grid, s);
      mg_set_error("mg_wino3x3 (strip): PixelNorm with three out-channel tiles");
      return MG_EINVAL;
  }
}

// Out-channel tiles per wave for a layer of nt tiles, or 0 = leave the call to wino3x3.hip.  Two wherever there are two or more (128
// accumulators, two waves per SIMD; further tile pairs go to further workgroup rows, which read and transform the input again; an odd
// count ends in a row that computes ONE tile on the two-tile filter image and gets a smaller share of the grid), one for 16
// out-channels.  The choice follows tools/ab_wino_strip.py (profiles/r05_ab_wino_strip.txt), us per launch against wino3x3.hip:
// 32 channels 0.72-0.92;  64 channels (the dominant launches of level 5: 48->64 @128 x 192 images 929 -> 782) 0.81-0.94;  48 channels
// 0.73-0.96 (64->48 @128 x 192, the largest launch of a level-5 step: 1016 -> 903);  80 / 96 channels 0.78-0.97 -- each from ~2 000
// tile blocks on (whole steps: 4096 -> 2048 is -0.3 % at level 5, -0.5 % at level 4, nothing at levels 3 / 6 / 7; 1536 is +1.3 % at level 6);  16 channels (one tile per wave: the 32 -> 16 data gradient of level 7) 0.81-0.91 since the in-block halo pixels come
// by DPP (1.01-1.06 before).  `force` (MG_WINO_STRIP=2: tests, A/B): whatever the shape allows.  MG_WINO_STRIP_NIW (1 / 2 / 3) overrides the tile count per wave (3: the 192-accumulator, one-wave-per-SIMD
// form for exactly 48 channels; measurements).
long long strip_min_blocks() {
  static const long long v = getenv("MG_WINO_STRIP_MIN_BLOCKS") ? atoll(getenv("MG_WINO_STRIP_MIN_BLOCKS")) : 2048;
  return v;
}

int strip_plan(const WinoArgs& a, bool pn, bool force) {
  const int nt = a.Cout / 16;
  const long long blocks = (long long)a.N * (a.H / 2) * (a.W / 32);
  int niw = 0;
  niw = (force || blocks >= strip_min_blocks()) ? (nt == 1 ? 1 : 2) : 0;
  // 96 .. 160 input channels: the filter bank of TWO out-channel tiles does not fit the LDS (8 KB per 8-channel chunk and tile), that
  // of one does -- one tile per wave then (the input is read and transformed once per tile, and still 0.82 of the staged kernel's time:
  // 96 -> 80 @32 x 192 images 168 -> 138 us, profiles/r06_ab_strip_small.txt)
  if (niw == 2 && (size_t)(a.Cin / WCC) * 2 * 8192 > 160 * 1024) niw = 1;
  if (pn) niw = nt <= 2 ? (niw ? nt : 0) : 0;  // PixelNorm: all channels of a pixel in one wave
  const char* e = getenv("MG_WINO_STRIP_NIW");
  if (e != nullptr && !pn && niw != 0) {
    const int v = atoi(e);
    if (v == 1 || (v == 2 && nt >= 2) || (v == 3 && nt == 3)) niw = v;
  }
  return niw;
}

int strip_kind(const WinoArgs& a) {  // the dispatch of wino_epilogue.h, by name
  if (a.flags & MG_CONV_PIXNORM) return SK_PN;
  if (a.flags & MG_CONV_UNPOOL) return SK_UNPOOL;
  if (a.flags & WF_BLEND_BWD) return SK_BLEND_BWD;
  if (a.flags & WF_BLEND) return (a.flags & MG_CONV_MASK_BYTES) ? SK_BLEND_TAN : SK_BLEND_FWD;
  if (a.flags & MG_CONV_MASK_AUX) return (a.flags & MG_CONV_MASK_BYTES) ? ((a.flags & MG_CONV_POOL_OUT) ? SK_MB_POOL : SK_MB_Y) : SK_MASKF;
  return (a.flags & MG_CONV_MASK_OUT) ? SK_ACT_POOL_MOUT : SK_ACT;
}

}  // namespace

// Whether wino_run (wino3x3.hip) hands this call to the strip kernel.  Shape: whole 16-channel chunk pairs and out tiles, whole 16-tile
// blocks per tile row, whole groups of 8 tile rows, the filter bank of a workgroup's tiles within LDS, every plane within 32-bit byte
// offsets, masked kinds bias-free, PixelNorm without y; then strip_plan's choice.  MG_WINO_STRIP=0: never; =2: whenever the shape allows.
bool mgi_wino_strip_takes(const WinoArgs& a, bool pn) {
  const char* e = getenv("MG_WINO_STRIP");
  if (e != nullptr && atoi(e) == 0) return false;
  const int nt = a.Cout / 16;
  if ((a.Cin % 16) != 0 || (a.Cout % 16) != 0 || nt > 10 || (a.W % 32) != 0 || ((a.H / 2) % 8) != 0 || (a.H % 2) != 0) return false;
  if (pn && (nt > 2 || a.y != nullptr)) return false;  // (PixelNorm: all channels of a pixel in one wave, p and rn only)
  if ((a.flags & (MG_CONV_MASK_AUX | MG_CONV_UNPOOL | WF_BLEND_BWD)) && a.bias != nullptr) return false;  // masked kinds: bias-free
  if ((long long)a.Cout * a.H * a.W * 16 >= (1ll << 31) || (long long)a.Cin * a.H * a.W * 4 >= (1ll << 31)) return false;
  const int niw = strip_plan(a, pn, e != nullptr && atoi(e) > 1);
  return niw != 0 && (size_t)(a.Cin / WCC) * niw * 8192 <= 160 * 1024;  // the filter bank of a workgroup's tiles fits LDS
}

int mgi_wino_strip_run(WinoArgs& a, hipStream_t s) {
  const int nt = a.Cout / 16;
  a.nchunk = a.Cin / WCC;
  a.NT = pack_wino_nt_padded(a.Cout);
  a.TBW = 16; a.TBH = 1; a.TBN = 1; a.lgTBW = 4; a.lgTBH = 0;
  {
    // measurement switch (ablation): 2 = every input row through a zero-record descriptor (reads return 0, nothing is fetched),
    // 4 = every epilogue access likewise (stores dropped, masks read as 0), 6 = both: what is left is issue time
    const char* e = getenv("MG_WINO_STRIP_ABLATE");
    if (e != nullptr) a.TBN |= atoi(e) & 6;
  }
  a.blocks_x = a.W / 32; a.blocks_y = a.H / 2; a.blocks_n = a.N;
  // out-channel tiles per wave: all of them (NIW = nt: the input transform is done once) or ONE with the tiles on grid.y (half
  // the registers, twice the waves per SIMD, the input read and transformed once per tile); PixelNorm needs all channels in a wave
  const int kind = strip_kind(a);
  const char* e = getenv("MG_WINO_STRIP");
  const int niw = strip_plan(a, kind == SK_PN, e != nullptr && atoi(e) > 1);
  dim3 grid(1, mg_cdiv(nt, niw));  // (an odd tile count: the last workgroup row carries one padding tile of zero filters)
  switch (niw) {
    case 1: return launch_strip_kind<1, 8>(a, kind, grid, s);  // (four-wave workgroups, i.e. up to four waves per SIMD: no effect,
                                                               //  profiles/r05_ab_wino_strip_nt1_occupancy.txt)
    case 2: return launch_strip_kind<2, 8>(a, kind, grid, s);
    default: return launch_strip_kind<3, 4>(a, kind, grid, s);
  }
}
