// 3x3 / stride 1 / pad 1 convolution in Winograd F(2x2,3x3) form on the exact-fp32 matrix cores of gfx950: 16 multiplies per
// 2x2 output tile and channel pair instead of 36, i.e. 2.25x fewer MFMA cycles than the implicit GEMM of conv3x3.hip for the
// same nn.Conv2d(3x3) + LeakyReLU (+ PixelNorm) (+ AvgPool2d) of
//   /root/reference/music_gan/networks/generator.py:9-40 and discriminator.py:8-34
// (forward, data gradient and the tangent pass of the gradient penalty all go through it, the latter two with re-packed weights).
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A      d: 4x4 input patch (stride 2), g: 3x3 filter, Y: 2x2 outputs
//
// Per Winograd component xi (16 of them) this is a GEMM  M_xi[out-ch, tile] = U_xi[c, out-ch]^T * V_xi[tile, c]:
//   M (A rows)  = NIW x 16 output channels per wave, WC channel groups per workgroup
//   N (B cols)  = 16 tiles per wave (one 16x16x4 MFMA tile), WT = 2 or 4 tile groups per workgroup = 32 / 64 tiles
//   K           = input channels, 8 per LDS chunk (two MFMA k-steps: lane k-index rq holds channels 2rq and 2rq+1)
// A wave keeps all 16 components of its (channel, tile) block in registers (16 x NIW accumulator tiles), so the output
// transform, bias, LeakyReLU, PixelNorm, mask and 2x2 average pool are in-register epilogues, and because the filter fragment
// is the A operand a lane ends up with 4 out-channels of ONE tile: the 16 lanes of a row group hold 16 consecutive tiles, and
// every global store / mask load is a run of contiguous bytes per out-channel row.  The input transform is done once per
// (tile, channel) by the staging threads on the way from HBM to LDS; U = G g G^T is pre-computed by the pack kernel.
// LDS images are laid out in operand order ([component pair][lane][k-step][component parity]) so one conflict-free
// ds_read_b128 feeds the operands of two components, and the weight image is a verbatim copy of the packed global layout.
// Input patches come in through buffer loads: positions in the zero padding (and ragged tiles / channels) carry an
// out-of-range offset and read back as 0.0 from the hardware bounds check, so the staging code has no masks or branches.
// Two workgroup sizes: WT = 4 (64 tiles, 4 x WC waves, one workgroup per CU) when the grid gives every CU at least two of
// them, else WT = 2 (32 tiles, two workgroups per CU).  The big one wins wherever it fills the chip because vector work of
// one wave is starved (70-140 cycles per instruction, tools/hwtests/valu_latency_under_mfma.hip) while another wave of its
// SIMD streams fp32 MFMAs, so co-resident workgroups in different phases do not overlap; with one workgroup per CU all waves
// are in the same phase and the filters are staged once per 64 tiles.  Inside each: "issue early / write late" register
// prefetch of the next chunk, and LDS operands of the next component pair requested before the current pair's MFMAs.
// fp32 throughout; rounding differs from the direct form by ~1.2x rms (measured against fp64, tests/test_ops_gpu.py).
#include <cstdlib>
#include <type_traits>

#include "mg_common.h"
#include "pack_kernels.h"
#include "wino_common.h"

namespace {

// WT = tile groups (16 tiles each) per workgroup: 2 (several workgroups per CU) or 4 (one 8..12-wave workgroup per CU); layers of
// at most 16 out-channels take WC = 1 (no all-padding second channel tile) with WT = 2 (4 and 8: measurement variants)
template <int NIW, int WC, int WT>
__global__ void __launch_bounds__(64 * WT * WC, WT == 2 ? 2 : 1) wino3x3_mfma(const WinoArgs a) {
  constexpr int TPB = WT * 16;  // tiles per workgroup
  constexpr int NTHR = 64 * WT * WC;
  constexpr int NITEMS = TPB * WCC;                    // (tile, channel) items per chunk
  constexpr int NITEM = (NITEMS + NTHR - 1) / NTHR;    // ... per thread
  constexpr int V_FLOATS = 16 * TPB * WCC;             // [WT][8 comp pairs][64 lanes][4]
  constexpr int U4 = WC * NIW * 512;                   // 16-byte pieces of the weight image
  constexpr int NU4 = (U4 + NTHR - 1) / NTHR;          // ... per thread
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Vs = smem;
  float* Us = smem + V_FLOATS;  // [WC*NIW tiles][8 comp pairs][64 lanes][4]
  float* red = Us + U4 * 4;     // PixelNorm cross-wave partial sums [WC][TPB][4]

  // Persistent form: a workgroup walks the tile blocks item, item + gridDim.x, ... (gridDim.x = workgroups the chip holds at
  // once) and requests the first chunk of its NEXT block while the last chunk of the current one is in the matrix pipe, so that
  // the ~2 us from "addresses known" to "first operands in LDS" are hidden under the epilogue instead of opening every block.
  // The thread index is laundered once per block: what derives from it is then recomputed per block rather than hoisted out of
  // the loop and kept live across it (the hoisted form costs ~60 registers and spills).
  int tid = threadIdx.x, lane, wave, col, rq, wt, wc;
  auto thread_coords = [&]() __attribute__((always_inline)) {
    asm volatile("" : "+v"(tid));
    lane = tid & 63;
    wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    col = lane & 15;
    rq = lane >> 4;
    wt = wave % WT;
    wc = wave / WT;
  };
  thread_coords();
  const int ct0 = blockIdx.y * (WC * NIW);
  const int HW = a.H * a.W;
  const int Ht = a.H >> 1, Wt = a.W >> 1;
  const int nblk = a.blocks_x * a.blocks_y * a.blocks_n;

  // staging geometry: item it = tid + k*NTHR -> tile it % TPB, chunk-local channel it / TPB, so lanes l and l+1 of a
  // half-wave hold horizontally adjacent tiles of one channel.  Each item loads only its OWN two pixel columns of the 4 patch
  // rows (one aligned 8-byte load per row: 16 lanes = 128 contiguous bytes) and takes the left / right halo column from
  // the neighbour lane (DPP wave shift); the lanes on the left / right edge of the tile block load theirs from memory.
  // Offsets are bytes from the block's first image; 0x80000000 (beyond any num_records) marks the zero padding, tiles that
  // do not exist, item indices past the chunk and -- for the halo loads -- every lane that has a neighbour.
  unsigned voffP[NITEM][4], voffE[NITEM][4];  // own pair; halo column of an edge lane (left OR right: a lane is at most one)
  bool ledge[NITEM], redge[NITEM];
  int vdst[NITEM];  // LDS float offset of the item's first component pair
  int bx, by, n0;   // tile block of the staging geometry (the epilogue keeps its own copy: the geometry runs one block ahead)
  const float* img_base;
  unsigned img_bytes;
  auto block_geometry = [&](int item) __attribute__((always_inline)) {
    const int bid = mg_xcd_remap(item, nblk);
    bx = bid % a.blocks_x;
    const int t2 = bid / a.blocks_x;
    by = t2 % a.blocks_y;
    n0 = (t2 / a.blocks_y) * a.TBN;
    const int nimg = (a.N - n0) < a.TBN ? (a.N - n0) : a.TBN;
    img_base = a.x + (size_t)n0 * a.Cin * HW;
    img_bytes = (unsigned)nimg * (unsigned)(a.Cin * HW) * 4u;
#pragma unroll
    for (int k = 0; k < NITEM; ++k) {
      const int it = tid + k * NTHR;
      const int tl = it % TPB, cl = it / TPB;
      const int txl = tl & (a.TBW - 1);
      const int tyl = (tl >> a.lgTBW) & (a.TBH - 1);
      const int nl = tl >> (a.lgTBW + a.lgTBH);
      const int TY = by * a.TBH + tyl, TX = bx * a.TBW + txl;
      const bool ok = (it < NITEMS) && (nl < nimg) && (TY < Ht) && (TX < Wt);
      const int base = (nl * a.Cin + cl) * HW + 2 * TX;
      ledge[k] = txl == 0;
      redge[k] = txl == a.TBW - 1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int Y = 2 * TY - 1 + r;
        const bool rv = ok && (Y >= 0) && (Y < a.H);
        const unsigned o = (unsigned)(base + Y * a.W) * 4u;
        voffP[k][r] = rv ? o : 0x80000000u;
        // a block one tile wide has both halos outside the image (W == 2), so one offset per lane is enough
        voffE[k][r] = (rv && ledge[k] && TX > 0) ? o - 4u : ((rv && redge[k] && 2 * TX + 2 < a.W) ? o + 8u : 0x80000000u);
      }
      // [tile group][component pair][lane' = (cl>>1)*16 + tile%16][k-step = cl&1][parity]
      vdst[k] = (it < NITEMS) ? (((tl >> 4) * 8 * 64 + (cl >> 1) * 16 + (tl & 15)) * 4 + (cl & 1) * 2) : -1;
    }
  };

  f32x4 acc[16][NIW];

  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 rP[NITEM][4];
  float rE[NITEM][4];
  f32x4 rw[NU4];

  auto load_chunk = [&](int ch) __attribute__((always_inline)) {
    const int soff = ch * WCC * HW * 4;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(img_base), 0, (int)img_bytes, 0x00020000);
#pragma unroll
    for (int k = 0; k < NITEM; ++k) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        rP[k][r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)voffP[k][r], soff, 0));
        rE[k][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voffE[k][r], soff, 0));
      }
    }
    const f32x4* src = reinterpret_cast<const f32x4*>(a.up + ((size_t)ch * a.NT + ct0) * 2048);
#pragma unroll
    for (int j = 0; j < NU4; ++j)
      if (NU4 * NTHR == U4 || tid + NTHR * j < U4) rw[j] = src[tid + NTHR * j];
  };

  auto store_chunk = [&](int ch) __attribute__((always_inline)) {
    if (a.Cin - ch * WCC < WCC) {  // ragged last chunk: channels that do not exist must read as zero (the loads see real data)
      const int lim = a.Cin - ch * WCC;
#pragma unroll
      for (int k = 0; k < NITEM; ++k)
        if ((tid + k * NTHR) / TPB >= lim) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            rP[k][r] = f32x2{0.f, 0.f};
            rE[k][r] = 0.f;
          }
        }
    }
#pragma unroll
    for (int k = 0; k < NITEM; ++k) {
      // patch row r = {left halo | own pair P | right halo}; kept as two register pairs E = (left, right), P = (own x, own y) so
      // that B^T d B is 16 packed adds: rows act element-wise on the pairs, columns are
      //   (v0, v3) = (e0 - p1, p0 - e1)      (v1, v2) = (p0 + p1, p1 - p0)
      f32x2 E[4], P[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        P[r] = rP[k][r];
        const float own_x = rP[k][r][0], own_y = rP[k][r][1];
        const float fl = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, own_y), 0x138, 0xf, 0xf, false));  // lane-1
        const float fr = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, own_x), 0x130, 0xf, 0xf, false));  // lane+1
        E[r] = f32x2{ledge[k] ? (a.TBW > 1 ? rE[k][r] : 0.f) : fl, redge[k] ? (a.TBW > 1 ? rE[k][r] : 0.f) : fr};
      }
      f32x2 UE[4], UP[4];  // B^T d (rows)
      UE[0] = pk_sub(E[0], E[2]);  UP[0] = pk_sub(P[0], P[2]);
      UE[1] = E[1] + E[2];         UP[1] = P[1] + P[2];
      UE[2] = pk_sub(E[2], E[1]);  UP[2] = pk_sub(P[2], P[1]);
      UE[3] = pk_sub(E[1], E[3]);  UP[3] = pk_sub(P[1], P[3]);
      if (NITEM * NTHR == NITEMS || vdst[k] >= 0) {
        float* dst = Vs + vdst[k];
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // (B^T d) B (columns); component slots of row i: [v0, v3 | v1, v2]
          // one v_pk_add_f32 per result pair, swaps and negations in the operand modifiers (hipcc builds them with v_mov / v_xor)
          f32x2 v03, v12;
          asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(v03) : "v"(UE[i]), "v"(UP[i]));  // (e0 - p1, p0 - e1)
          asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(v12) : "v"(UP[i]));             // (p0 + p1, p1 - p0)
          *reinterpret_cast<f32x2*>(dst + (2 * i) * 256) = v03;
          *reinterpret_cast<f32x2*>(dst + (2 * i + 1) * 256) = v12;
        }
      }
    }
    f32x4* dstw = reinterpret_cast<f32x4*>(Us);
#pragma unroll
    for (int j = 0; j < NU4; ++j)
      if (NU4 * NTHR == U4 || tid + NTHR * j < U4) dstw[tid + NTHR * j] = rw[j];
  };

  // operands of component pair cp+1 are requested from LDS before the MFMAs of pair cp are issued (two register sets): a wave
  // that issues its reads only after its MFMAs leaves the matrix pipe idle for an LDS round trip every 8 instructions
  auto compute_chunk = [&]() __attribute__((always_inline)) {
    const float* va = Vs + wt * 2048 + lane * 4;
    const float* ub = Us + (wc * NIW) * 2048 + lane * 4;
    f32x4 av[2], bv[2][NIW];
    av[0] = *reinterpret_cast<const f32x4*>(va);  // {c0 k0, c1 k0, c0 k1, c1 k1}
#pragma unroll
    for (int ni = 0; ni < NIW; ++ni) bv[0][ni] = *reinterpret_cast<const f32x4*>(ub + ni * 2048);
    __builtin_amdgcn_sched_group_barrier(0x100, 1 + NIW, 0);
#pragma unroll
    for (int cp = 0; cp < 8; ++cp) {
      const int cur = cp & 1, nxt = cur ^ 1;
      if (cp + 1 < 8) {
        av[nxt] = *reinterpret_cast<const f32x4*>(va + (cp + 1) * 256);
#pragma unroll
        for (int ni = 0; ni < NIW; ++ni) bv[nxt][ni] = *reinterpret_cast<const f32x4*>(ub + ni * 2048 + (cp + 1) * 256);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int par = 0; par < 2; ++par)
#pragma unroll
          for (int ni = 0; ni < NIW; ++ni)
            acc[2 * cp + par][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[cur][ni][ks * 2 + par], av[cur][ks * 2 + par], acc[2 * cp + par][ni], 0, 0, 0);
      // pin that order (the machine scheduler otherwise sinks the reads below the MFMAs again to shorten live ranges)
      if (cp + 1 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1 + NIW, 0);  // DS reads of pair cp+1
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * NIW, 0);                  // MFMAs of pair cp
    }
  };

  block_geometry(blockIdx.x);
  load_chunk(0);
#pragma nounroll
  for (int item = blockIdx.x;;) {
  const int next = item + (int)gridDim.x;
  const int ebx = bx, eby = by, en0 = n0;  // this block, for the epilogue
#pragma unroll
  for (int c = 0; c < 16; ++c)
#pragma unroll
    for (int ni = 0; ni < NIW; ++ni) {  // cleared with 64-bit moves: fp32 MFMA and VALU time add up on gfx950, every VALU counts
      typedef double f64x2 __attribute__((ext_vector_type(2)));
      acc[c][ni] = __builtin_bit_cast(f32x4, f64x2{0.0, 0.0});
    }
  for (int ch = 0; ch + 1 < a.nchunk; ++ch) {
    __syncthreads();  // previous chunk's operand reads done
    store_chunk(ch);
    __syncthreads();
    load_chunk(ch + 1);  // in flight during the MFMA phase below
    compute_chunk();
  }
  // last chunk (peeled, so that the block geometry stays out of the loop above): the loads in flight during its MFMA phase are
  // chunk 0 of the workgroup's next tile block -- or, for its last block, a harmless repeat
  __syncthreads();
  store_chunk(a.nchunk - 1);
  __syncthreads();
  if (WT >= 4 && next < nblk) block_geometry(next);  // (the 32-tile form is never launched persistent: two workgroups share a CU)
  load_chunk(0);
  compute_chunk();

#include "wino_epilogue.h"
  if (WT < 4 || next >= nblk) break;
  item = next;
  thread_coords();
  }
}

// U = G g G^T for every (out, in) channel pair, written in MFMA operand order (pack_kernels.h):
//   up[(((ch*NT + ct)*8 + slot/2)*64 + lane)*4 + ks*2 + slot%2],   slot = 4*xi + {0, 2, 3, 1}[nu]  (pairs (nu0,nu3), (nu1,nu2):
// the order in which the packed input transform produces them)
// with  in-channel = ch*8 + 2*(lane>>4) + ks,  out-channel = ct*16 + (lane&15)
__global__ void wino3x3_pack_kernel(const float* __restrict__ w, float* __restrict__ up, int Co, int Ci, int dgrad, int NT,
                                    size_t total) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < total) pack_wino3x3_elem(e, w, up, Co, Ci, dgrad, NT);
}

template <int NIW, int WC, int WT>
int launch_wino(const WinoArgs& a, dim3 grid, hipStream_t s) {
  constexpr int TPB = WT * 16;
  constexpr size_t lds = (size_t)(16 * TPB * WCC + WC * NIW * 2048 + WC * TPB * 4) * sizeof(float);
  static MgPerDevice once;  // the LDS limit is a per-device function attribute
  if (mg_first_use_on_device(once)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino3x3_mfma<NIW, WC, WT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  hipLaunchKernelGGL((wino3x3_mfma<NIW, WC, WT>), grid, dim3(64 * WT * WC), lds, s, a);
  MG_CHECK_LAUNCH("mg_wino3x3");
  return MG_OK;
}

int wino_nt_padded(int Cout) { return pack_wino_nt_padded(Cout); }
int wino_run(WinoArgs& a, bool pn, hipStream_t s);  // tile geometry + variant dispatch (below)

}  // namespace

extern "C" size_t mg_wino3x3_packed_floats(int Cin, int Cout) {
  return (size_t)mg_cdiv(Cin, WCC) * wino_nt_padded(Cout) * 2048;
}

extern "C" int mg_wino3x3_pack(const float* w, float* up, int Co, int Ci, int dgrad, mg_stream_t stream) {
  MG_CHECK_ARG(w && up && Co > 0 && Ci > 0, "mg_wino3x3_pack: bad arguments");
  const int NT = wino_nt_padded(dgrad ? Ci : Co);
  const size_t total = pack_wino3x3_threads(Co, Ci, dgrad);  // one thread per (chunk, tile, lane, k-step)
  hipLaunchKernelGGL(wino3x3_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, up, Co,
                     Ci, dgrad, NT, total);
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
  const bool mask_bytes = flags & MG_CONV_MASK_BYTES, mask_out = flags & MG_CONV_MASK_OUT, unpool = flags & MG_CONV_UNPOOL;
  MG_CHECK_ARG(pn || y || mask_bytes, "mg_wino3x3: y is NULL");
  MG_CHECK_ARG(!mask_out || ((flags & MG_CONV_POOL_OUT) && (flags & MG_CONV_LRELU) && !(flags & MG_CONV_MASK_AUX) && !unpool),
               "mg_wino3x3: MASK_OUT needs POOL_OUT and LRELU");
  MG_CHECK_ARG(!mask_bytes || ((flags & MG_CONV_MASK_AUX) && ((flags & MG_CONV_POOL_OUT) || y) && !unpool),
               "mg_wino3x3: MASK_BYTES needs MASK_AUX and an output (POOL_OUT: the pooled result only; else y)");
  MG_CHECK_ARG(!unpool || (aux && y && !(flags & ~MG_CONV_UNPOOL) && !bias), "mg_wino3x3: UNPOOL takes aux and y and no other flag");
  MG_CHECK_ARG(!(flags & MG_CONV_POOL_OUT) || (!pn && p), "mg_wino3x3: POOL_OUT needs p and no PIXNORM");
  MG_CHECK_ARG(!((flags & MG_CONV_MASK_AUX) && (flags & (MG_CONV_LRELU | MG_CONV_PIXNORM))),
               "mg_wino3x3: MASK_AUX excludes LRELU/PIXNORM");
  MG_CHECK_ARG(!(flags & ~(MG_CONV_LRELU | MG_CONV_PIXNORM | MG_CONV_MASK_AUX | MG_CONV_POOL_OUT | MG_CONV_MASK_OUT |
                           MG_CONV_MASK_BYTES | MG_CONV_UNPOOL)), "mg_wino3x3: unknown flag");
  MG_CHECK_ARG((long long)N * Cin * H * W < (1ll << 40) && (long long)N * Cout * H * W < (1ll << 40), "mg_wino3x3: tensor too large");
  MG_CHECK_ARG(!pn || mg_cdiv(Cout, 16) <= 4, "mg_wino3x3: PIXNORM needs Cout <= 64 (all channels of a pixel in one workgroup)");

  WinoArgs a;
  a.x = x; a.up = up; a.bias = bias; a.aux = aux; a.y = y; a.p = p; a.rn = rn;
  a.mi = reinterpret_cast<const unsigned char*>(aux);
  a.mo = reinterpret_cast<unsigned char*>(y);
  a.other = nullptr; a.coef = nullptr;
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  a.flags = flags; a.slope = slope;
  return wino_run(a, pn, (hipStream_t)stream);
}

extern "C" int mg_wino3x3_mask_bytes_y_supported(int N, int Cin, int Cout, int H, int W) {
  if (N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || (H % 2) || (W % 2)) return 0;
  WinoArgs a = {};
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  a.flags = MG_CONV_MASK_AUX | MG_CONV_MASK_BYTES;
  return mgi_wino_strip_takes(a, false) ? 1 : 0;
}

namespace {
int wino_run(WinoArgs& a, bool pn, hipStream_t s) {
  const int N = a.N, Cin = a.Cin, Cout = a.Cout, H = a.H, W = a.W;
  const int nt = mg_cdiv(Cout, 16);
  a.nchunk = mg_cdiv(Cin, WCC);
  a.NT = wino_nt_padded(Cout);
  const int Ht = H / 2, Wt = W / 2;
  // few channels on a large map: one wave per tile block, operands built in registers, filters resident in LDS (wino_strip.hip)
  if (mgi_wino_strip_takes(a, pn)) return mgi_wino_strip_run(a, s);
  // (tile-mask bytes applied to a full-resolution result: an epilogue of the strip kernel only -- mg_wino3x3_mask_bytes_y_supported)
  MG_CHECK_ARG(!(a.flags & MG_CONV_MASK_BYTES) || (a.flags & (MG_CONV_POOL_OUT | WF_BLEND)),
               "mg_wino3x3: MASK_BYTES without POOL_OUT needs a shape the strip kernel takes (N=%d %d->%d @%dx%d)", N, Cin, Cout, H, W);

  // out-channel tiles per workgroup (wave groups x tiles per wave): 4 = 2x2, 3 = 3x1, 2 = 2x1 -- least padding wins
  int cfg = 4, best = mg_cdiv(nt, 4) * 4;
  if (mg_cdiv(nt, 3) * 3 < best) { cfg = 3; best = mg_cdiv(nt, 3) * 3; }
  if (mg_cdiv(nt, 2) * 2 < best) { cfg = 2; best = mg_cdiv(nt, 2) * 2; }
  // five or more channel tiles (80 .. 160 out-channels: the 64x64 .. 8x8 layers): two tiles per workgroup in the 32-tile form --
  // four-wave workgroups, several per CU -- beat the 12- and 16-wave tilings by 6-8 % at 64x64 / 32x32 and by ~20 % at 16x16
  // (tools/tune_wino.py, every (cfg, wt) per layer shape of a level-5 step); 48 and 64 out-channels stay on their one
  // workgroup per CU (3x1 and 2x2 tiles)
  if (nt >= 5) cfg = 2;
  if (pn) cfg = nt <= 2 ? 2 : (nt == 3 ? 3 : 4);
  // at most 16 out-channels: ONE channel tile per workgroup and 128 tiles instead of a second, all-padding channel tile
  // (the 32 -> 16 data gradient at 512x512 spent half its MFMAs on zero filters)
  const bool narrow = nt == 1 && !pn && getenv("MG_WINO_CFG") == nullptr && getenv("MG_WINO_WT") == nullptr &&
                      getenv("MG_WINO_NARROW") == nullptr;  // (MG_WINO_NARROW=0: measurement switch, the two-tile forms)
  if (narrow) cfg = 1;
  {
    const char* e = getenv("MG_WINO_CFG");  // measurement override: 2, 3 or 4 out-channel tiles per workgroup
    if (e != nullptr && !pn) {
      const int v = atoi(e);
      if (v == 2 || v == 4 || (v == 3 && mg_cdiv(nt, 3) * 3 <= a.NT)) cfg = v;
    }
  }
  MG_CHECK_ARG(mg_cdiv(nt, cfg) * cfg <= a.NT, "mg_wino3x3: internal tile error");
  // tiles per workgroup: 64 (one 8..12-wave workgroup per CU: every wave of the CU is in the same phase, so the VALU staging
  // and epilogue never queue behind another workgroup's matrix instructions -- measured 70..140 cycles per VALU instruction
  // when they do, tools/hwtests/valu_latency_under_mfma.hip -- and the filters are staged once per 64 tiles) when that still
  // gives every CU two or more workgroups, else 32 (two workgroups per CU).
  const int n_cu = mg_cu_count();
  int wt = ((long long)N * Ht * Wt / 64) * mg_cdiv(nt, cfg) >= 2ll * n_cu ? 4 : 2;
  if (narrow) {
    // 32 tiles per two-wave workgroup, ~6 of them per CU: with 16 out-channels a block is a few hundred cycles of MFMAs between
    // memory round trips, and occupancy hides those where one 128-tile workgroup per CU (the first form of this variant) waited:
    // 32->16@512, 18 images: 0.70 (two-tile form) -> 0.49 (128 tiles) -> 0.40 ms
    wt = 2;
    const char* e = getenv("MG_WINO_NARROW_WT");  // measurement override: 2, 4 or 8 tile groups per workgroup
    if (e != nullptr && (atoi(e) == 4 || atoi(e) == 8)) wt = atoi(e);
    if (wt == 8 && (long long)N * Ht * Wt / 128 < 2ll * n_cu) wt = 2;
  }
  // two out-channel tiles (<= 32 channels): few MFMAs per staged item and per epilogue, so a block is mostly memory latency and
  // vector work -- four-wave workgroups, several per CU, overlap that where one eight-wave workgroup per CU waits (measured
  // 11-16 % on 16->32@512, 32->32@256, 48->32@256; the opposite of the 48- and 64-channel tilings, tools/bench_l67.py)
  if (cfg == 2 && !narrow && !pn && getenv("MG_WINO_WT") == nullptr && getenv("MG_WINO_CFG") == nullptr) wt = 2;
  {
    const char* e = getenv("MG_WINO_WT");  // measurement override
    if (e != nullptr && (atoi(e) == 2 || atoi(e) == 4)) wt = atoi(e);
  }
  // a grid that leaves CUs idle even in the 32-tile form takes two out-channel tiles per workgroup instead of three or four: more,
  // lighter workgroups (the input tile is staged by more of them, which a half-empty chip does not notice)
  if (wt == 2 && !pn && cfg > 2 && getenv("MG_WINO_CFG") == nullptr &&
      ((long long)N * Ht * Wt + 31) / 32 * mg_cdiv(nt, cfg) < n_cu && mg_cdiv(nt, 2) * 2 <= a.NT)
    cfg = 2;
  // Few 64-tile blocks per CU (the 48- and 64-channel layers at the reference's batch of 6: 128x128 / 64x64 maps at levels 6-7):
  // the one-workgroup-per-CU tilings idle through most of their last round and have nobody to overlap their vector phases with;
  // two out-channel tiles per four-wave workgroup, several per CU, are 5-18 % ahead there (tools/tune_wino.py 6 6 and 7 6:
  // 64 channels up to 4.5 blocks per CU, 48 channels at 1.5 but not at 4.5; at 16 per CU -- level 5, batch 64 -- the large tilings win)
  const char* few = getenv("MG_WINO_FEW");  // measurement switch: 0 = keep the large tilings
  if (!pn && !narrow && getenv("MG_WINO_CFG") == nullptr && getenv("MG_WINO_WT") == nullptr && (few == nullptr || atoi(few) != 0)) {
    const long long b64 = (long long)N * Ht * Wt / 64;
    if (((cfg == 4 && b64 < 6ll * n_cu) || (cfg == 3 && 2 * b64 < 5ll * n_cu)) && mg_cdiv(nt, 2) * 2 <= a.NT) {
      cfg = 2;
      wt = 2;
    }
  }
  const int tpb = wt * 16;
  a.TBW = mg_pow2_ceil(Wt) < 16 ? mg_pow2_ceil(Wt) : 16;  // 16 tiles = 32 pixels = one 128-byte line per row and channel
  a.TBH = mg_pow2_ceil(Ht) < tpb / a.TBW ? mg_pow2_ceil(Ht) : tpb / a.TBW;
  a.TBN = tpb / (a.TBW * a.TBH);
  a.lgTBW = mg_ilog2(a.TBW); a.lgTBH = mg_ilog2(a.TBH);
  a.blocks_x = mg_cdiv(Wt, a.TBW); a.blocks_y = mg_cdiv(Ht, a.TBH); a.blocks_n = mg_cdiv(N, a.TBN);
  MG_CHECK_ARG((long long)a.TBN * Cin * H * W < (1ll << 29), "mg_wino3x3: image block too large for 32-bit offsets");
  dim3 grid(a.blocks_x * a.blocks_y * a.blocks_n, mg_cdiv(nt, cfg));
  {  // persistent launch of the 64-tile form: one workgroup per CU, each walking its tile blocks
    const char* e = getenv("MG_WINO_PERSIST");  // measurement switch: 0 = one workgroup per tile block
    if (wt >= 4 && (e == nullptr || atoi(e) != 0) && (int)grid.x > n_cu) grid.x = n_cu;
  }
  switch (cfg * 10 + wt) {
    case 18: return launch_wino<1, 1, 8>(a, grid, s);
    case 14: return launch_wino<1, 1, 4>(a, grid, s);
    case 12: return launch_wino<1, 1, 2>(a, grid, s);
    case 44: return launch_wino<2, 2, 4>(a, grid, s);
    case 42: return launch_wino<2, 2, 2>(a, grid, s);
    case 34: return launch_wino<1, 3, 4>(a, grid, s);
    case 32: return launch_wino<1, 3, 2>(a, grid, s);
    case 24: return launch_wino<1, 2, 4>(a, grid, s);
    default: return launch_wino<1, 2, 2>(a, grid, s);
  }
}
}  // namespace

extern "C" int mg_wino3x3_fade(const float* x, const float* up, const float* bias, const unsigned char* mask_in, const float* other,
                               const float* coef, float* y, void* out2, int N, int Cin, int Cout, int H, int W, int mode,
                               float slope, mg_stream_t stream) {
  MG_CHECK_ARG(x && up && other && coef && y && N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "mg_wino3x3_fade: bad arguments");
  MG_CHECK_ARG((H % 2 == 0) && (W % 2 == 0), "mg_wino3x3_fade: H=%d W=%d must be even", H, W);
  MG_CHECK_ARG(Cout <= 160, "mg_wino3x3_fade: Cout=%d > 160 unsupported", Cout);
  MG_CHECK_ARG(mode == MG_FADE_FWD || mode == MG_FADE_TANGENT || mode == MG_FADE_BWD, "mg_wino3x3_fade: unknown mode %d", mode);
  MG_CHECK_ARG(mode == MG_FADE_FWD ? (out2 != nullptr && mask_in == nullptr) : mask_in != nullptr,
               "mg_wino3x3_fade: FWD writes a tile mask to out2, TANGENT / BWD read mask_in");
  MG_CHECK_ARG(mode != MG_FADE_BWD || (out2 != nullptr && bias == nullptr), "mg_wino3x3_fade: BWD needs out2 and no bias");
  MG_CHECK_ARG((long long)N * Cin * H * W < (1ll << 40) && (long long)N * Cout * H * W < (1ll << 40), "mg_wino3x3_fade: tensor too large");
  WinoArgs a;
  a.x = x; a.up = up; a.bias = bias; a.aux = nullptr; a.y = y; a.rn = nullptr;
  a.p = mode == MG_FADE_BWD ? static_cast<float*>(out2) : nullptr;
  a.mi = mask_in;
  a.mo = mode == MG_FADE_FWD ? static_cast<unsigned char*>(out2) : nullptr;
  a.other = other; a.coef = coef;
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  a.flags = mode == MG_FADE_FWD ? (WF_BLEND | MG_CONV_LRELU)
          : mode == MG_FADE_TANGENT ? (WF_BLEND | MG_CONV_MASK_AUX | MG_CONV_MASK_BYTES) : WF_BLEND_BWD;
  a.slope = slope;
  return wino_run(a, false, (hipStream_t)stream);
}
