// atan2f and hypotf with the bits of torch's CPU `th.angle` / `th.abs` on complex64.
//
// The reference's codec (/root/reference/music_gan/audio/functions.py:69-70) calls `th.abs(complex_values)` and
// `th.angle(complex_values)` on CPU tensors.  ATen evaluates both through the SLEEF vector math library it vendors
// (third_party/sleef; `Sleef_hypotf{8,16}_u05`, `Sleef_atan2f{8,16}_u10` in aten/src/ATen/cpu/vec/vec{256,512}/*complex_float.h,
// the FMA builds) -- a third-party dependency that is not in /root/reference.  Its algorithm is published (N. Shibata, F. Petrogalli,
// "SLEEF: A Portable Vectorized Library of C Standard Mathematical Functions", IEEE TPDS 2020; sleefsimdsp.c `xatan2f_u1`,
// `xhypotf_u05`): double-float ("df") arithmetic on (hi, lo) pairs of float32 built from error-free FMA products, a degree-7
// minimax polynomial for atan on [0, 1] after an octant reduction, and sqrt by one Newton step in df.  This file restates it as
// scalar code; every operation below is an IEEE float32 add / mul / fma / div / sqrt in the library's order, so the result is the
// library's bit for bit -- which matters because the unwrapped phase is DISCONTINUOUS in angle(X): one ulp of atan2f moves the
// float32 rounding of a 1e5-rad running sum (rocm's ocml atan2f agrees with torch's on 60-65 % of random inputs, this on 100 %).
// Pinned by tests/test_host_cpu.py (this header compiled by g++, 4 M random + special inputs against torch.angle / torch.abs on the
// CPU, bitwise) and by tests/test_audio_gpu.py (the device build against the reference's golden codec output).
//
// Device code for the codec kernel; the CPU pin compiles the same text with g++ (SLF_FN defined as `static inline`).  Floating-point
// contraction must be OFF (musicgan_amd/_build.py passes -ffp-contract=off; the test passes it to g++).
#pragma once
#include <math.h>

#ifndef SLF_FN  // the CPU pin defines it as `static inline` before including this file
#define SLF_FN __device__ __forceinline__
#endif

namespace slf {

struct f2 { float x, y; };  // hi, lo

SLF_FN float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
SLF_FN f2 mk(float x, float y) { f2 r; r.x = x; r.y = y; return r; }
SLF_FN f2 neg(f2 a) { return mk(-a.x, -a.y); }
SLF_FN f2 normalize(f2 t) { const float s = t.x + t.y; return mk(s, (t.x - s) + t.y); }
SLF_FN f2 scale(f2 d, float s) { return mk(d.x * s, d.y * s); }
SLF_FN f2 add_f_f(float x, float y) { const float s = x + y; return mk(s, (x - s) + y); }                       // |x| >= |y|
SLF_FN f2 add_f_f2(float x, f2 y) { const float s = x + y.x; return mk(s, ((x - s) + y.x) + y.y); }              // |x| >= |y|
SLF_FN f2 add_f2_f2(f2 x, f2 y) { const float s = x.x + y.x; return mk(s, (((x.x - s) + y.x) + x.y) + y.y); }   // |x| >= |y|
SLF_FN f2 add2_f2_f(f2 x, float y) {
  const float s = x.x + y, v = s - x.x;
  return mk(s, ((x.x - (s - v)) + (y - v)) + x.y);
}
SLF_FN f2 add2_f2_f2(f2 x, f2 y) {
  const float s = x.x + y.x, v = s - x.x;
  return mk(s, ((x.x - (s - v)) + (y.x - v)) + (x.y + y.y));
}
SLF_FN f2 mul_f_f(float x, float y) { const float s = x * y; return mk(s, fma_(x, y, -s)); }
SLF_FN f2 mul_f2_f(f2 x, float y) { const float s = x.x * y; return mk(s, fma_(x.y, y, fma_(x.x, y, -s))); }
SLF_FN f2 mul_f2_f2(f2 x, f2 y) { const float s = x.x * y.x; return mk(s, fma_(x.x, y.y, fma_(x.y, y.x, fma_(x.x, y.x, -s)))); }
SLF_FN f2 squ(f2 x) { const float s = x.x * x.x; return mk(s, fma_(x.x + x.x, x.y, fma_(x.x, x.x, -s))); }
SLF_FN f2 rec_f(float d) { const float s = 1.0f / d; return mk(s, s * fma_(-d, s, 1.0f)); }
SLF_FN f2 div(f2 n, f2 d) {
  const float t = 1.0f / d.x;  // correctly rounded (hipcc: -fhip-fp32-correctly-rounded-divide-sqrt is the default)
  const float s = n.x * t;
  const float u = fma_(t, n.x, -s);
  const float v = fma_(-d.y, t, fma_(-d.x, t, 1.0f));
  return mk(s, fma_(s, v, fma_(n.y, t, u)));
}
SLF_FN f2 sqrt2(f2 d) {
  const float t = sqrtf(d.x + d.y);
  return scale(mul_f2_f2(add2_f2_f2(d, mul_f_f(t, t)), rec_f(t)), 0.5f);
}

SLF_FN float mulsign(float x, float y) {
  return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) ^ (__builtin_bit_cast(unsigned, y) & 0x80000000u));
}
SLF_FN bool is_inf(float x) { return x == INFINITY || x == -INFINITY; }

// atan2 of non-negative y (as a df) and any x: octant reduction to s = min / max in [0, 1], atan(s) = s + s * t * P(t), t = s^2,
// plus q * pi/2 (pi/2 as a df).
SLF_FN f2 atan2k(f2 y, f2 x) {
  int q = 0;
  if (x.x < 0.f) { x = neg(x); q = -2; }
  f2 s, t;
  if (x.x < y.x) { s = neg(x); t = y; q += 1; } else { s = y; t = x; }
  s = div(s, t);
  t = normalize(squ(s));
  float u = -0.00176397908944636583328247f;
  u = fma_(u, t.x, 0.0107900900766253471374512f);
  u = fma_(u, t.x, -0.0309564601629972457885742f);
  u = fma_(u, t.x, 0.0577365085482597351074219f);
  u = fma_(u, t.x, -0.0838950723409652709960938f);
  u = fma_(u, t.x, 0.109463557600975036621094f);
  u = fma_(u, t.x, -0.142626821994781494140625f);
  u = fma_(u, t.x, 0.199983194470405578613281f);
  t = mul_f2_f2(t, add_f_f(-0.333332866430282592773438f, u * t.x));
  t = mul_f2_f2(s, add_f_f2(1.0f, t));
  return add_f2_f2(mul_f2_f(mk(1.5707963705062866211f, -4.3711388286737928865e-08f), (float)q), t);
}

// th.angle(complex(x, y)) = atan2(y, x), float32
SLF_FN float atan2f_u10(float y, float x) {
  if (fabsf(x) < 2.9387372783541830947e-39f) { y *= 16777216.0f; x *= 16777216.0f; }
  const f2 d = atan2k(mk(fabsf(y), 0.f), mk(x, 0.f));
  float r = mulsign(d.x + d.y, x);
  const float PI2 = 1.57079637050628662109375f, PI4 = 0.785398185253143310546875f, PI1 = 3.1415927410125732421875f;
  if (is_inf(x) || x == 0.f) r = PI2 - (is_inf(x) ? mulsign(1.0f, x) * PI2 : 0.0f);
  if (is_inf(y)) r = PI2 - (is_inf(x) ? mulsign(1.0f, x) * PI4 : 0.0f);
  if (y == 0.f) r = (mulsign(1.0f, x) == -1.0f) ? PI1 : 0.0f;
  return (x != x || y != y) ? NAN : mulsign(r, y);
}

// th.abs(complex(x, y)) = hypot(x, y), float32: max * sqrt(1 + (min / max)^2) in df
SLF_FN float hypotf_u05(float x, float y) {
  x = fabsf(x);
  y = fabsf(y);
  const float mn = fminf(x, y), mx = fmaxf(x, y);
  float n = mn, d = mx;
  if (mx < 1.17549435e-38f) { n *= 16777216.0f; d *= 16777216.0f; }
  f2 t = div(mk(n, 0.f), mk(d, 0.f));
  t = mul_f2_f(sqrt2(add2_f2_f(squ(t), 1.0f)), mx);
  float r = t.x + t.y;
  if (r != r) r = INFINITY;
  if (mn == 0.f) r = mx;
  if (x != x || y != y) r = NAN;
  if (x == INFINITY || y == INFINITY) r = INFINITY;
  return r;
}

// Both at once for one bin (re, im).  hypot's quotient min / max IS |s| of atan2's octant reduction (the same df division of the
// same two floats), so for bins in the range where neither routine rescales or special-cases its arguments the division and its
// square are evaluated once; negating a df quotient is exact, and the signs of zero low parts -- the only thing that can differ
// from the library's own order -- do not change a non-zero sum.  Everything else (zeros, denormals, huge values, inf / nan) takes
// the two routines above unchanged.
SLF_FN void abs_angle(float re, float im, float& mag, float& ang) {
  const float ax = fabsf(re), ay = fabsf(im);
  const float mn = fminf(ax, ay), mx = fmaxf(ax, ay);
  if (!(ax >= 1e-30f && ay >= 1e-30f && ax <= 1e30f && ay <= 1e30f)) {  // (comparisons with a NaN are false: slow path)
    mag = hypotf_u05(re, im);
    ang = atan2f_u10(im, re);
    return;
  }
  const f2 tq = div(mk(mn, 0.f), mk(mx, 0.f));
  const f2 sq = squ(tq);
  const f2 h = mul_f2_f(sqrt2(add2_f2_f(sq, 1.0f)), mx);
  mag = h.x + h.y;
  int q = re < 0.f ? -2 : 0;
  f2 s = tq;
  if (ax < ay) { s = neg(tq); q += 1; }
  f2 t = normalize(sq);
  float u = -0.00176397908944636583328247f;
  u = fma_(u, t.x, 0.0107900900766253471374512f);
  u = fma_(u, t.x, -0.0309564601629972457885742f);
  u = fma_(u, t.x, 0.0577365085482597351074219f);
  u = fma_(u, t.x, -0.0838950723409652709960938f);
  u = fma_(u, t.x, 0.109463557600975036621094f);
  u = fma_(u, t.x, -0.142626821994781494140625f);
  u = fma_(u, t.x, 0.199983194470405578613281f);
  t = mul_f2_f2(t, add_f_f(-0.333332866430282592773438f, u * t.x));
  t = mul_f2_f2(s, add_f_f2(1.0f, t));
  const f2 d = add_f2_f2(mul_f2_f(mk(1.5707963705062866211f, -4.3711388286737928865e-08f), (float)q), t);
  ang = mulsign(mulsign(d.x + d.y, re), im);
}

}  // namespace slf
