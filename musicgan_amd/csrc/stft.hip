// Framed, Hann-windowed 1024-point real FFT (the STFT of /root/reference/music_gan/audio/functions.py:38-62):
//   X[k,t] = 1/sqrt(sum w^2) * sum_{n<1024} w[n] xpad[256 t + n] e^{-2 pi i k n / 1024},  k = 0..511 (Nyquist dropped),
//   xpad = reflect-pad(x, 512), w = periodic Hann, T = 1 + L/256.
//
// One wave transforms one frame: the 1024 real samples are packed as 512 complex points (8 per lane), three radix-8
// Stockham passes (two exchanges through a per-wave 4 KiB LDS buffer) leave Z[j + 64 r] in lane j / register r; the
// real-FFT untangling needs Z[512-k], which lives in lane (64-j)&63 -> one wavefront shuffle per value, no LDS.
// A workgroup (8 waves, persistent: one per CU) covers 16 consecutive frames per tile and transposes its 512x16 output tile
// through LDS so that global stores are 128-byte runs along t (the output is frequency-major).  Twiddles and the window are
// built once per workgroup into LDS with sincospi (no global tables, no hidden state).
#include <cmath>

#include "mg_common.h"

namespace {

constexpr int NFFT = 1024, HOP = 256, NB = 512;  // NB = complex points = output bins
constexpr int NWAVE = 8;                          // waves per workgroup
constexpr int FPW = 2;                            // frames per wave and tile
constexpr int FPB = NWAVE * FPW;                  // frames per tile: 8 x 8 bytes = one 64-byte run per bin row
// per-frame LDS column (float2 units).  XSTR = 2 (mod 32): in the transposed read-out the 16 frames of one bin row land on the
// even 8-byte slots of the 256-byte bank row and the next bin row's on the odd ones, so a 32-lane ds_read_b64 group is conflict-free
constexpr int XSTR = NB + 2;
// twiddle tables, each laid out in the order its pass reads it (lane-contiguous or broadcast: no bank conflicts):
//   tw[k]       = e^{-2 pi i k / 1024}, k < 512                     (untangling; lane k & 63)
//   tw1[r][k]   = e^{-2 pi i r k / 64},  r, k < 8                    (pass 1; 8 distinct addresses per instruction)
//   tw2[r][j]   = e^{-2 pi i r j / 512}, r < 8, j < 64               (pass 2; lane j)
constexpr int TW_FLOATS = (NB + 64 + 512) * 2;
constexpr size_t STFT_LDS = (size_t)(TW_FLOATS + NFFT + FPB * XSTR * 2) * sizeof(float);

// Complex numbers as register pairs.  Every swap / negate of a component rides on the operand-select and negate modifiers of
// the packed instruction that consumes it (hipcc builds such vectors with v_mov / v_xor instead: a third of the kernel's
// vector instructions), so a complex multiply is 2 instructions and a multiply by -i is free.
typedef float c2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ c2 csub(c2 a, c2 b) {  // a - b
  c2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ c2 cmul(c2 a, c2 b) {  // (a.x b.x - a.y b.y, a.x b.y + a.y b.x)
  c2 t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));  // (a.x b.x, a.x b.y)
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,0,0]"
      : "=v"(r) : "v"(a), "v"(b), "v"(t));  // (a.y * -b.y + t.x, a.y * b.x + t.y)
  return r;
}
__device__ __forceinline__ c2 add_mi(c2 s, c2 d) {  // s + (-i) d = (s.x + d.y, s.y - d.x)
  c2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(r) : "v"(s), "v"(d));
  return r;
}
__device__ __forceinline__ c2 sub_mi(c2 s, c2 d) {  // s - (-i) d = (s.x - d.y, s.y + d.x)
  c2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(r) : "v"(s), "v"(d));
  return r;
}
__device__ __forceinline__ c2 add_conj(c2 a, c2 z) {  // a + conj z
  c2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(z));
  return r;
}
__device__ __forceinline__ c2 sub_conj(c2 a, c2 z) {  // a - conj z
  c2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "=v"(r) : "v"(a), "v"(z));
  return r;
}

// One 8-byte LDS read as `ds_read_b64`.  hipcc merges two such reads at a common base into `ds_read2_b64` / `ds_read2st64_b64`, which
// this LDS serves at HALF the rate with banks modulo 32 (8 cycles per wave-instruction against 2 x 2; MI355X_MICROARCH.md, LDS) -- the
// column strides here are laid out for the 64-bank rule of `ds_read_b64`.  A volatile access is not merged (and stays under the
// compiler's s_waitcnt bookkeeping).
typedef const volatile __attribute__((address_space(3))) c2* lds_c2_ptr;
__device__ __forceinline__ c2 lds_c2(const c2* p) { return *(lds_c2_ptr)p; }  // (the explicit LDS address space: a volatile generic load is a flat_load)

// 4-point DFT; MI2: y2 is handed over without its pending factor -i
template <bool MI2>
__device__ __forceinline__ void dft4(c2 y0, c2 y1, c2 y2, c2 y3, c2& q0, c2& q1, c2& q2, c2& q3) {
  const c2 s0 = MI2 ? add_mi(y0, y2) : y0 + y2, s1 = MI2 ? sub_mi(y0, y2) : csub(y0, y2);
  const c2 s2 = y1 + y3, t = csub(y1, y3);  // s3 = -i t
  q0 = s0 + s2;
  q2 = csub(s0, s2);
  q1 = add_mi(s1, t);
  q3 = sub_mi(s1, t);
}

// in-place 8-point DFT, natural order in and out
__device__ __forceinline__ void dft8(c2 (&v)[8]) {
  const float h = 0.70710678118654752440f;
  const c2 a0 = v[0] + v[4], a1 = v[1] + v[5], a2 = v[2] + v[6], a3 = v[3] + v[7];
  c2 d0 = csub(v[0], v[4]), d1 = csub(v[1], v[5]), d2 = csub(v[2], v[6]), d3 = csub(v[3], v[7]);
  d1 = add_mi(d1, d1) * h;           // * W8^1 = (1 - i)/sqrt2 : (d.x + d.y, d.y - d.x) h
  {                                  // * W8^3 = (-1 - i)/sqrt2 : (d.y - d.x, -d.x - d.y) h   (W8^2 = -i of d2 rides into dft4)
    c2 r;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[1,1]" : "=v"(r) : "v"(d3));
    d3 = r * h;
  }
  dft4<false>(a0, a1, a2, a3, v[0], v[2], v[4], v[6]);
  dft4<true>(d0, d1, d2, d3, v[1], v[3], v[5], v[7]);
}

// Persistent workgroups of 8 waves (two per CU: one's write-out runs under the other's FFTs; <= 128 registers): twiddles and window are built once
// per workgroup, then each tile of 16 consecutive frames is transformed -- one wave = 2 frames, each in its own 4 KiB LDS column
// that serves the two Stockham exchanges and finally holds the frame's 512 output bins, so there is no workgroup barrier inside
// a frame -- and the 16 columns are read out transposed so that every global store instruction writes four 128-byte runs along t.
// PCM: what `wav` holds -- 0: float32 samples, 1: int16 (scaled by 1/32768 as torchaudio.load(normalize=True) does);  CH: 1 = mono,
// 2 = interleaved stereo frames as a WAV file stores them, averaged on the way in (functions.py:49: raw_audio.mean(0)) -- a
// file's bytes go to the GPU as they are and the de-interleave / scale / mono mean cost no pass of their own.
template <int PCM, int CH>
struct PcmSrc {
  // two consecutive mono samples s, s + 1 (s even: aligned loads of 4 .. 16 bytes)
  static __device__ __forceinline__ c2 pair(const void* __restrict__ w, long long s) {
    if constexpr (PCM == 0 && CH == 1) {
      return *reinterpret_cast<const c2*>(reinterpret_cast<const float*>(w) + s);
    } else if constexpr (PCM == 0) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(w) + 2 * s);
      return c2{(v[0] + v[1]) / 2.0f, (v[2] + v[3]) / 2.0f};
    } else if constexpr (CH == 1) {
      const short2 v = *reinterpret_cast<const short2*>(reinterpret_cast<const short*>(w) + s);
      return c2{(float)v.x / 32768.0f, (float)v.y / 32768.0f};
    } else {
      const short4 v = *reinterpret_cast<const short4*>(reinterpret_cast<const short*>(w) + 2 * s);
      return c2{((float)v.x / 32768.0f + (float)v.y / 32768.0f) / 2.0f, ((float)v.z / 32768.0f + (float)v.w / 32768.0f) / 2.0f};
    }
  }
  static __device__ __forceinline__ float one(const void* __restrict__ w, long long s) {
    if constexpr (PCM == 0 && CH == 1) {
      return reinterpret_cast<const float*>(w)[s];
    } else if constexpr (PCM == 0) {
      const float2 v = *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(w) + 2 * s);
      return (v.x + v.y) / 2.0f;
    } else if constexpr (CH == 1) {
      return (float)reinterpret_cast<const short*>(w)[s] / 32768.0f;
    } else {
      const short2 v = *reinterpret_cast<const short2*>(reinterpret_cast<const short*>(w) + 2 * s);
      return ((float)v.x / 32768.0f + (float)v.y / 32768.0f) / 2.0f;
    }
  }
};

template <int PCM, int CH>
__global__ void __launch_bounds__(64 * NWAVE, 2) stft1024_kernel(const void* __restrict__ wav, float* __restrict__ out_re,
                                                              float* __restrict__ out_im, long long L, int T, int ntiles) {
  using Src = PcmSrc<PCM, CH>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  c2* tw = reinterpret_cast<c2*>(smem);                       // tw[k], k < 512
  c2* tw1 = tw + NB;                                           // tw1[r * 8 + k]
  c2* tw2 = tw1 + 64;                                          // tw2[r * 64 + j]
  float* win = smem + TW_FLOATS;                               // Hann / sqrt(sum w^2)
  c2* xbuf = reinterpret_cast<c2*>(win + NFFT);                // [FPB][XSTR] one column per frame of the tile

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int m = tid; m < NFFT; m += 64 * NWAVE) {
    float s, c;
    sincospif((float)m * (1.0f / 512.0f), &s, &c);  // angle = 2 pi m / 1024 = pi * m / 512
    if (m < NB) tw[m] = c2{c, -s};
    // Hann / sqrt(384) (sum of hann^2 over 1024 = 384), times the 1/2 of the real-FFT untangling: a power-of-two scale
    // commutes exactly with every rounding of the (linear) transform, so it costs nothing here instead of 16 multiplies there
    win[m] = (0.5f - 0.5f * c) * (0.5f * 0.05103103630798288f);
    // the pass twiddles are the same 1024-th roots (identical bits to indexing one table): tw1[r][k] = root 16 r k, tw2[r][j] = root 2 r j
    if (m < 64) {
      float s1, c1;
      sincospif((float)(((m >> 3) * (m & 7) * 16) & (NFFT - 1)) * (1.0f / 512.0f), &s1, &c1);
      tw1[m] = c2{c1, -s1};
    }
    if (m < 512) {
      float s2, c2_;
      sincospif((float)(((m >> 6) * (m & 63) * 2) & (NFFT - 1)) * (1.0f / 512.0f), &s2, &c2_);
      tw2[m] = c2{c2_, -s2};
    }
  }
  __syncthreads();

  // Samples of a frame, 8 complex pairs per lane: requested one frame ahead (the next frame of the tile, or the first frame
  // of the workgroup's next tile before the write-out of this one), so the HBM latency runs under a whole frame of FFT work
  // instead of in front of it (a wave has nothing else to do: 4 waves per SIMD do not hide ~2500 cycles).
  auto load_frame = [&](int t, c2 (&x)[8]) {
    if (t < T) {
      const long long base = (long long)t * HOP - NFFT / 2;  // sample index of padded position 0 of this frame
      const bool interior = (base >= 0) && (base + NFFT <= L);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int n = 2 * (lane + 64 * r);
        if (interior) {
          x[r] = Src::pair(wav, base + n);
        } else {
          long long s0 = base + n, s1 = base + n + 1;
          if (s0 < 0) s0 = -s0;
          if (s1 < 0) s1 = -s1;
          if (s0 >= L) s0 = 2 * (L - 1) - s0;
          if (s1 >= L) s1 = 2 * (L - 1) - s1;
          x[r] = c2{Src::one(wav, s0), Src::one(wav, s1)};
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) x[r] = c2{0.f, 0.f};
    }
  };

  c2 xn[8];  // the frame in flight
  if ((int)blockIdx.x < ntiles) load_frame(blockIdx.x * FPB + wave * FPW, xn);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int t0 = tile * FPB;
#pragma unroll 1
    for (int f = 0; f < FPW; ++f) {
      const int fl = wave * FPW + f;  // frame slot in the tile
      c2* xb = xbuf + fl * XSTR;
      c2 v[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = xn[r] * lds_c2(reinterpret_cast<const c2*>(win + 2 * (lane + 64 * r)));
      {  // one call site: next frame of this tile, first frame of the next tile, or nothing (t = T reads as zeros)
        const int tnext = tile + (int)gridDim.x < ntiles ? (tile + (int)gridDim.x) * FPB + wave * FPW : T;
        load_frame(f + 1 < FPW ? t0 + fl + 1 : tnext, xn);
      }
      // pass 0 (Ns = 1): no twiddles; out[8 j + r]
      dft8(v);
      __builtin_amdgcn_wave_barrier();
      // Exchange 1: element e = 8 j + r is stored at e ^ ((e >> 4) & 7).  A ds_write_b64 is served in groups of 16 contiguous
      // lanes over 32 banks (16 eight-byte slots): unswizzled, 8 j + r puts the 16 lanes on 2 slots (8-way conflict); the XOR
      // with j >> 1 spreads them over all 16.  The reader's e = j + 64 r has (e >> 4) & 7 = ((j >> 4) + 4 r) & 7: constant
      // per 16 lanes, so its 32-lane groups still read 32 distinct slots.
#pragma unroll
      for (int r = 0; r < 8; ++r) xb[(8 * lane + r) ^ ((lane >> 1) & 7)] = v[r];
      __builtin_amdgcn_wave_barrier();
      // pass 1 (Ns = 8): in[j + 64 r] * exp(-2 pi i r k / 64), k = j & 7; out[(j>>3)*64 + k + 8 r]
      {
        const int k = lane & 7;
#pragma unroll
        for (int r = 0; r < 8; ++r)  // ((lane >> 4) + 4 r) & 7 = (lane >> 4) ^ 4 (r & 1): two base addresses + immediate offsets
          v[r] = cmul(lds_c2(xb + (((lane ^ (lane >> 4)) ^ (4 * (r & 1))) + 64 * r)), lds_c2(tw1 + r * 8 + k));
        dft8(v);
        __builtin_amdgcn_wave_barrier();
        // Exchange 2: element e = 64 g + k + 8 r (g = j >> 3) is stored at e ^ (8 * (g & 1)): the two g of a 16-lane write group
        // would share their 8 slots, bit 3 separates them.  The reader's e = j + 64 r sees the constant 8 * (r & 1).
        const int j0 = (lane >> 3) * 64 + k, swz = lane & 8;  // 8 * (g & 1)
#pragma unroll
        for (int r = 0; r < 8; ++r) xb[(j0 + 8 * r) ^ swz] = v[r];
        __builtin_amdgcn_wave_barrier();
      }
      // pass 2 (Ns = 64): in[j + 64 r] * exp(-2 pi i r j / 512); result Z[j + 64 r] stays in lane j, register r
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = cmul(lds_c2(xb + ((lane + 64 * r) ^ (8 * (r & 1)))), lds_c2(tw2 + r * 64 + lane));
      dft8(v);
      __builtin_amdgcn_wave_barrier();
      // untangle: X[k] = (Z[k] + conj Z[512-k])/2 - i/2 * e^{-2 pi i k/1024} * (Z[k] - conj Z[512-k]),  k = lane + 64 r.
      // Z[512-k] sits in lane (64-lane)&63, register 7-r (lane != 0) or 8-r (lane == 0; r == 0 -> Z[0] itself).
      const int src = (64 - lane) & 63;
      c2 zc[8];  // Z[512-k]
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        // value this lane must SEND for the receiver's register r: receiver lane l' = (64-lane)&63 wants register
        // (l' == 0 ? 8-r : 7-r) of its source lane; every lane != 0 is read by a lane != 0 (7-r); lane 0 reads itself.
        const c2 send_n0 = v[7 - r];
        const c2 send_0 = v[(8 - r) & 7];
        const c2 send = (lane == 0) ? send_0 : send_n0;
        zc[r] = c2{__shfl(send.x, src), __shfl(send.y, src)};  // Z[512-k] (conjugated by the consumers; the 1/2 is in the window)
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int k = lane + 64 * r;
        const c2 wd = cmul(lds_c2(tw + k), sub_conj(v[r], zc[r]));
        xb[k] = add_mi(add_conj(v[r], zc[r]), wd);  // e - i * wd  (pass 2 has read the column: the exchanges are done with it)
      }
    }
    __syncthreads();
    // transposed write-out: the 16 frames of one bin row per 8 lanes, two frames (16 bytes) per lane -- stores are issue-bound per
    // instruction, so half as many twice as wide; a store instruction writes eight 128-byte runs along t.  Non-temporal stores:
    // the bins are written once and not read back by this kernel, and keeping them out of the L2's way is worth 6-8 %
    // (0.146-0.166 -> 0.136-0.152 ms per 10-minute file, same box, alternating libraries)
    {
      const int fp = tid & (FPB / 2 - 1);  // frame pair
      const int t = t0 + 2 * fp;
      if (out_im == nullptr && (T & 1) == 0) {  // interleaved complex output, rows 16-byte aligned
        if (t < T) {  // T even: t + 1 < T too
#pragma unroll 8
          for (int k = tid / (FPB / 2); k < NB; k += 64 * NWAVE / (FPB / 2)) {
            const c2 o0 = lds_c2(xbuf + (2 * fp) * XSTR + k), o1 = lds_c2(xbuf + (2 * fp + 1) * XSTR + k);
            __builtin_nontemporal_store(f32x4{o0.x, o0.y, o1.x, o1.y}, reinterpret_cast<f32x4*>(out_re + 2 * ((size_t)k * T + t)));
          }
        }
      } else {
        const int f = tid & (FPB - 1);
        const int tt = t0 + f;
        if (tt < T) {
#pragma unroll 4
          for (int k = tid / FPB; k < NB; k += 64 * NWAVE / FPB) {
            const c2 o = xbuf[f * XSTR + k];
            const size_t idx = (size_t)k * T + tt;
            if (out_im != nullptr) {
              __builtin_nontemporal_store(o.x, out_re + idx);
              __builtin_nontemporal_store(o.y, out_im + idx);
            } else {
              __builtin_nontemporal_store(o, reinterpret_cast<c2*>(out_re + 2 * idx));
            }
          }
        }
      }
    }
    __syncthreads();
  }
}

// Other sample formats / channel counts: one pass to mono float32 first (rare: 8-bit, 32-bit integer, more than two channels).
__global__ void __launch_bounds__(256) pcm_to_mono_k(const void* __restrict__ pcm, float* __restrict__ mono, long long L, int C, int kind) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < L; i += (long long)gridDim.x * 256) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) {
      const long long e = i * C + c;
      float v;
      if (kind == MG_PCM_F32) v = reinterpret_cast<const float*>(pcm)[e];
      else if (kind == MG_PCM_I16) v = (float)reinterpret_cast<const short*>(pcm)[e] / 32768.0f;
      else if (kind == MG_PCM_I32) v = (float)reinterpret_cast<const int*>(pcm)[e] / 2147483648.0f;
      else v = ((float)reinterpret_cast<const unsigned char*>(pcm)[e] - 128.0f) / 128.0f;
      s += v;
    }
    mono[i] = C > 1 ? s / (float)C : s;
  }
}

template <int PCM, int CH>
static int launch_stft(const void* wav, float* out_re, float* out_im, long long L, hipStream_t stream) {
  const int T = (int)(L / HOP) + 1;
  const int ntiles = (T + FPB - 1) / FPB;
  const int n_cu = mg_cu_count();
  static MgPerDevice once;
  if (mg_first_use_on_device(once))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stft1024_kernel<PCM, CH>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
  const int blocks = ntiles < 2 * n_cu ? ntiles : 2 * n_cu;  // two 78 KB workgroups per CU, persistent over the tiles
  hipLaunchKernelGGL((stft1024_kernel<PCM, CH>), dim3(blocks), dim3(64 * NWAVE), STFT_LDS, stream, wav, out_re, out_im, L, T, ntiles);
  MG_CHECK_LAUNCH("mg_stft_1024");
  return MG_OK;
}

// Any other (n_fft, hop) the reference's signature admits (functions.py:38-41: wav_to_stft(wav_p, nperseg, stride)): a plain
// radix-2 transform in LDS, one frame per workgroup -- not a tuned path (the drivers only ever use 1024 / 256), same definition:
// periodic Hann, centre reflect padding of n_fft / 2, division by sqrt(sum w^2), bins 0 .. n_fft/2 - 1.
__global__ void __launch_bounds__(256) stft_generic_k(const float* __restrict__ wav, float* __restrict__ out, long long L, int T, int N,
                                                      int lgN, int hop, float inv_norm) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  c2* buf = reinterpret_cast<c2*>(smem);
  const int t = blockIdx.x;
  const long long base = (long long)t * hop - N / 2;
  for (int n = threadIdx.x; n < N; n += 256) {
    long long s = base + n;
    if (s < 0) s = -s;
    if (s >= L) s = 2 * (L - 1) - s;
    float sn, cs;
    sincospif(2.0f * (float)n / (float)N, &sn, &cs);
    const float w = (0.5f - 0.5f * cs) * inv_norm;
    // bit-reversed placement: the stages below are decimation in time
    buf[__brev((unsigned)n) >> (32 - lgN)] = c2{wav[s] * w, 0.f};
  }
  __syncthreads();
  for (int st = 0; st < lgN; ++st) {
    const int half = 1 << st;
    for (int i = threadIdx.x; i < N / 2; i += 256) {
      const int k = i & (half - 1), j = ((i >> st) << (st + 1)) + k;
      float sn, cs;
      sincospif((float)k / (float)half, &sn, &cs);  // e^{-i pi k / half}
      const c2 a = buf[j], b = buf[j + half];
      const c2 wb = c2{b.x * cs + b.y * sn, b.y * cs - b.x * sn};
      buf[j] = c2{a.x + wb.x, a.y + wb.y};
      buf[j + half] = c2{a.x - wb.x, a.y - wb.y};
    }
    __syncthreads();
  }
  for (int k = threadIdx.x; k < N / 2; k += 256) *reinterpret_cast<c2*>(out + 2 * ((size_t)k * T + t)) = buf[k];
}

}  // namespace

extern "C" int mg_pcm_to_mono(const void* pcm, int kind, int channels, float* mono, int64_t L, mg_stream_t stream) {
  MG_CHECK_ARG(pcm && mono && L > 0 && channels >= 1 && channels <= 64 && kind >= MG_PCM_F32 && kind <= MG_PCM_U8,
               "mg_pcm_to_mono: bad arguments");
  const long long nb = (L + 255) / 256;
  hipLaunchKernelGGL(pcm_to_mono_k, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, (hipStream_t)stream, pcm, mono,
                     (long long)L, channels, kind);
  MG_CHECK_LAUNCH("mg_pcm_to_mono");
  return MG_OK;
}

extern "C" int mg_stft_generic(const float* wav, float* out_c64, int64_t L, int n_fft, int hop, mg_stream_t stream) {
  MG_CHECK_ARG(wav && out_c64 && hop >= 1, "mg_stft_generic: bad arguments");
  MG_CHECK_ARG(n_fft >= 64 && n_fft <= 8192 && (n_fft & (n_fft - 1)) == 0, "mg_stft_generic: n_fft=%d is not a power of two in [64, 8192]",
               n_fft);
  MG_CHECK_ARG(L > n_fft / 2, "mg_stft_generic: reflect padding needs L > n_fft / 2 (got %lld)", (long long)L);
  MG_CHECK_ARG(L / hop + 1 < (1ll << 30), "mg_stft_generic: too many frames");
  const int T = (int)(L / hop) + 1;
  // periodic Hann: sum w^2 = 3 N / 8
  const float inv_norm = (float)(1.0 / sqrt(0.375 * (double)n_fft));
  hipLaunchKernelGGL(stft_generic_k, dim3((unsigned)T), dim3(256), (size_t)n_fft * sizeof(c2), (hipStream_t)stream, wav, out_c64,
                     (long long)L, T, n_fft, mg_ilog2(n_fft), hop, inv_norm);
  MG_CHECK_LAUNCH("mg_stft_generic");
  return MG_OK;
}

extern "C" int mg_stft_1024(const float* wav, float* out_re, float* out_im, int64_t L, mg_stream_t stream) {
  MG_CHECK_ARG(wav && out_re, "mg_stft_1024: bad arguments");
  MG_CHECK_ARG(L > NFFT / 2, "mg_stft_1024: reflect padding needs L > 512 (got %lld)", (long long)L);
  MG_CHECK_ARG(L / HOP + 1 < (1ll << 30), "mg_stft_1024: too many frames");
  return launch_stft<0, 1>(wav, out_re, out_im, (long long)L, (hipStream_t)stream);
}

extern "C" size_t mg_stft_1024_pcm_ws_bytes(int64_t L, int channels, int kind) {
  const bool direct = (kind == MG_PCM_F32 || kind == MG_PCM_I16) && (channels == 1 || channels == 2);
  return direct ? 0 : (size_t)L * sizeof(float);
}

extern "C" int mg_stft_1024_pcm(const void* pcm, int kind, int channels, float* out_re, float* out_im, void* ws, size_t ws_bytes,
                                int64_t L, mg_stream_t stream) {
  MG_CHECK_ARG(pcm && out_re && channels >= 1 && channels <= 64 && kind >= MG_PCM_F32 && kind <= MG_PCM_U8,
               "mg_stft_1024_pcm: bad arguments");
  MG_CHECK_ARG(L > NFFT / 2, "mg_stft_1024_pcm: reflect padding needs L > 512 (got %lld)", (long long)L);
  MG_CHECK_ARG(L / HOP + 1 < (1ll << 30), "mg_stft_1024_pcm: too many frames");
  hipStream_t s = (hipStream_t)stream;
  if (kind == MG_PCM_F32 && channels == 1) return launch_stft<0, 1>(pcm, out_re, out_im, (long long)L, s);
  if (kind == MG_PCM_F32 && channels == 2) return launch_stft<0, 2>(pcm, out_re, out_im, (long long)L, s);
  if (kind == MG_PCM_I16 && channels == 1) return launch_stft<1, 1>(pcm, out_re, out_im, (long long)L, s);
  if (kind == MG_PCM_I16 && channels == 2) return launch_stft<1, 2>(pcm, out_re, out_im, (long long)L, s);
  if (ws == nullptr || ws_bytes < (size_t)L * sizeof(float)) {
    mg_set_error("mg_stft_1024_pcm: workspace of %zu bytes needed", (size_t)L * sizeof(float));
    return MG_EWORKSPACE;
  }
  const long long nb = (L + 255) / 256;
  hipLaunchKernelGGL(pcm_to_mono_k, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, s, pcm, reinterpret_cast<float*>(ws),
                     (long long)L, channels, kind);
  MG_CHECK_LAUNCH("mg_stft_1024_pcm(mono)");
  return launch_stft<0, 1>(ws, out_re, out_im, (long long)L, s);
}
