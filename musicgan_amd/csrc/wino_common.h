// Shared pieces of the Winograd F(2x2,3x3) convolution kernels (wino3x3.hip: LDS-staged tile blocks; wino_strip.hip: one wave per
// tile block, operands built in registers): launch arguments, packed-subtraction helper, internal epilogue selectors.
#pragma once
#include "mg_common.h"

// (global: wino3x3.hip hands its arguments to wino_strip.hip)
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
  int TBW, TBH, TBN, lgTBW, lgTBH;  // tile-block geometry in TILES: TBW * TBH * TBN == TPB
  int blocks_x, blocks_y, blocks_n;
  int nchunk;
  int NT;  // out-channel tiles in the packed weights (padded)
  const unsigned char* mi;  // tile mask read by the epilogue (MG_CONV_MASK_BYTES, fade-in tangent / backward forms)
  unsigned char* mo;        // tile mask written by the epilogue (MG_CONV_MASK_OUT, fade-in forward form)
  const float* other;       // fade-in forms: the old branch (forward / tangent) or its activation (backward)
  const float* coef;        // fade-in forms: {alpha, 1 - alpha} in device memory
};

bool mgi_wino_strip_takes(const WinoArgs& a, bool pn);  // wino_strip.hip
int mgi_wino_strip_run(WinoArgs& a, hipStream_t s);

namespace {

constexpr int WCC = 8;  // input channels per LDS chunk
constexpr float PN_EPS = 1e-8f;

// Packed fp32 subtraction as ONE instruction (hipcc lowers vector subtraction to one v_sub_f32 per element; fp32 MFMA and VALU
// time add up on gfx950, so every vector instruction of the staging code and the epilogue is matrix time lost).
typedef float wf32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ wf32x2 pk_sub(wf32x2 x, wf32x2 y) {
  wf32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(x), "v"(y));
  return d;
}
__device__ __forceinline__ f32x4 pk_sub4(f32x4 x, f32x4 y) {
  const wf32x2 lo = pk_sub(__builtin_shufflevector(x, x, 0, 1), __builtin_shufflevector(y, y, 0, 1));
  const wf32x2 hi = pk_sub(__builtin_shufflevector(x, x, 2, 3), __builtin_shufflevector(y, y, 2, 3));
  return f32x4{lo[0], lo[1], hi[0], hi[1]};
}


// internal epilogue selectors of mg_wino3x3_fade (above the public MG_CONV_* bits)
constexpr int WF_BLEND = 1 << 8;      // y = coef[0] * result + coef[1] * other
constexpr int WF_BLEND_BWD = 1 << 9;  // y = (coef[0] * acc) * lrelu'(mi),  p = (coef[1] * acc) * lrelu'(other)

}  // namespace
