// Shared host/device helpers for libmusicgan_hip.so (gfx950 only).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/musicgan_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

void mg_set_error(const char* fmt, ...);

#define MG_CHECK_ARG(cond, ...)    \
  do {                             \
    if (!(cond)) {                 \
      mg_set_error(__VA_ARGS__);   \
      return MG_EINVAL;            \
    }                              \
  } while (0)

#define MG_CHECK_LAUNCH(name)                                              \
  do {                                                                     \
    hipError_t e_ = hipGetLastError();                                     \
    if (e_ != hipSuccess) {                                                \
      mg_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));  \
      return MG_ELAUNCH;                                                   \
    }                                                                      \
  } while (0)

// One-time per-DEVICE setup (function attributes such as the dynamic-LDS limit are per device; a process may drive several GPUs).
#define MG_MAX_DEVICES 64
struct MgPerDevice {
  bool done[MG_MAX_DEVICES] = {};
};
static inline int mg_current_device() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= MG_MAX_DEVICES) d = 0;
  return d;
}
// true exactly once per device (benign race: the guarded setup is idempotent)
static inline bool mg_first_use_on_device(MgPerDevice& f) {
  const int d = mg_current_device();
  if (f.done[d]) return false;
  f.done[d] = true;
  return true;
}
int mg_cu_count();  // compute units of the current device (cached per device, core.hip)

static inline int mg_ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}
static inline int mg_pow2_ceil(int v) { return 1 << mg_ilog2(v); }
__host__ __device__ static inline int mg_cdiv(int a, int b) { return (a + b - 1) / b; }

// XCD-aware, bijective block-id remap: blocks b and b+8 share an XCD (observed round-robin dispatch), so give each of the
// 8 residue classes a contiguous chunk of the logical grid => neighbouring tiles (shared halos / weights) hit one L2.
__device__ __forceinline__ int mg_xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Sum over the 64 lanes of a wave with DPP adds (6 vector instructions, no LDS round trips); the total ends up in LANE 63
// (every lane of the last row of 16, in fact).  A __shfl_xor butterfly is 6 dependent ds_bpermute round trips (~100 cycles
// each): fine for one value per wave, 25 us for the 84 partial sums of conv1x1_wgrad_part.  Fixed order => deterministic.
__device__ __forceinline__ float mg_wave_sum_to_lane63(float v) {
  auto dpp_add = [](float x, auto ctrl_, auto rmask_) {
    constexpr int CTRL = decltype(ctrl_)::value, RMASK = decltype(rmask_)::value;
    return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, RMASK, 0xf, true));
  };
  using I = int;
  v = dpp_add(v, std::integral_constant<I, 0xB1>{}, std::integral_constant<I, 0xf>{});   // quad_perm [1,0,3,2]
  v = dpp_add(v, std::integral_constant<I, 0x4E>{}, std::integral_constant<I, 0xf>{});   // quad_perm [2,3,0,1]
  v = dpp_add(v, std::integral_constant<I, 0x141>{}, std::integral_constant<I, 0xf>{});  // row_half_mirror
  v = dpp_add(v, std::integral_constant<I, 0x140>{}, std::integral_constant<I, 0xf>{});  // row_mirror: every lane = its row's sum
  v = dpp_add(v, std::integral_constant<I, 0x142>{}, std::integral_constant<I, 0xa>{});  // row_bcast15 into rows 1, 3
  v = dpp_add(v, std::integral_constant<I, 0x143>{}, std::integral_constant<I, 0xc>{});  // row_bcast31 into rows 2, 3
  return v;
}

// LeakyReLU for 0 < slope <= 1 (every caller: 0.2, or 1 = none): max(v, slope * v) -- one v_mul + one v_max instead of v_cmp +
// v_mul + v_cndmask (the select alone was measured at 20 cycles against 7.5 for v_max, tools/hwtests/valu_cost.hip); the same bits
// as `v > 0 ? v : v * slope` for every input incl. signed zeros.
__device__ __forceinline__ float mg_lrelu(float v, float slope) { return fmaxf(v, v * slope); }
// 1 for positive values, else 0, from the IEEE bit pattern (v_med3_i32): the LeakyReLU mask bit of an activation
__device__ __forceinline__ unsigned mg_pos_bit(float v) {
  const int b = __builtin_bit_cast(int, v);
  return (unsigned)(b < 0 ? 0 : (b > 1 ? 1 : b));
}
__device__ __forceinline__ float mg_lrelu_mask(float act, float slope) { return act > 0.f ? 1.f : slope; }
