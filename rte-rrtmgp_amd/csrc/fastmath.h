// fastmath.h -- fp64 exp / reciprocal / division / square root for the solver kernels (gfx950).
//
// The solvers are fp64-issue bound where they are not HBM bound, and the device libm / IEEE-division
// sequences the compiler emits are long: exp() ~35 instructions (Horner steps as v_mov + v_fmac pairs, range
// selects), a/b 12, sqrt ~12.  The replacements below are written for the argument ranges these kernels
// actually have and stay within ~1-2 ulp of the correctly rounded result (the tests assert 1e-10 on fluxes
// against the reference kernels, whose libm is itself only faithful to ~1 ulp):
//   exp_nonpos(x)  x <= 0 (optical depths): no overflow path, underflow through v_ldexp          18 instr
//   rcp_nr(d), div_nr(n, d)  d normal and away from the exponent range limits                       6 / 8
//   sqrt_nr(x), rsqrt_nr(x)  x >= 0 normal or zero                                                   8
// Single-precision builds (RTE_USE_SP) map to the plain functions.
#pragma once
#include <math.h>

#include "common.h"

namespace rte {

#ifdef RTE_USE_SP
__device__ __forceinline__ float exp_nonpos(float x) { return expf(x); }
__device__ __forceinline__ float rcp_nr(float d) { return 1.0f / d; }
__device__ __forceinline__ float div_nr(float n, float d) { return n / d; }
__device__ __forceinline__ float sqrt_nr(float x) { return sqrtf(x); }
__device__ __forceinline__ float sqrt_pos(float x) { return sqrtf(x); }
__device__ __forceinline__ float sqrt_cr0(float x) { return sqrtf(x); }
#else
// Horner step with the coefficient as the addend; the constant sits in a register pair of its own (VGPRs: these kernels have no scalar registers to spare),
// so a step is ONE v_fma_f64 instead of the v_mov_b64 + v_fmac_f64 pair the generic code gets
__device__ __forceinline__ double fma_c(double p, double r, double c) {
  asm("" : "+s"(c));  // the constant in a scalar register pair: the FMA takes it as its SGPR operand
  return __builtin_fma(p, r, c);
}

__device__ __forceinline__ double exp_nonpos(double x) {
  // x = k ln2 + r, |r| <= ln2/2; exp(r) by a polynomial of degree 11 (below)
  {
    // exp underflows to 0 long before -1100; the clamp keeps k and r finite for any input and leaves a NaN a NaN (fmax would
    // turn it into a silent 0 where the reference's exp propagates it).  Only the HIGH word is replaced (one v_cndmask
    // instead of two): any low word under -1100's high word is a value in [-1100.001, -1100], as good a clamp as -1100
    unsigned long long b = __builtin_bit_cast(unsigned long long, x);
    const unsigned hi = (unsigned)(b >> 32);
    const unsigned hi_c = x < -1100.0 ? 0xC0913000u : hi;  // high word of -1100.0 (0xC091300000000000)
    b = ((unsigned long long)hi_c << 32) | (unsigned)b;
    x = __builtin_bit_cast(double, b);
  }
  // k = nearest integer to x / ln2 through the 1.5 * 2^52 shift: the sum's low word IS k as a 32-bit integer (|k| < 2^31), and
  // subtracting the shift again gives k as a double -- fma + add instead of mul + rndne + cvt
  const double shift = 6755399441055744.0;
  const double t = __builtin_fma(x, 1.4426950408889634074, shift);
  const double k = t - shift;
  const int ki = (int)(unsigned)__builtin_bit_cast(unsigned long long, t);
  double r = __builtin_fma(k, -6.93147180369123816490e-01, x);  // ln2 high part: k * hi is exact
  r = __builtin_fma(k, -1.90821492927058770002e-10, r);         // ln2 low part
  // exp(r) = 1 + r + r^2 q(r), q of degree 9: the minimax coefficients of tools/exp_minimax.py (relative truncation error
  // 3.6e-18 on |r| <= ln2 / 2; the double-precision evaluation is as close to exp as that of the degree-13 Taylor polynomial
  // it replaces, 1.12e-16 against 1.11e-16 worst case, in two steps less)
  double p = 0x1.ad7f6c51b1da1p-26;
  p = fma_c(p, r, 0x1.28ad72cedc06bp-22);
  p = fma_c(p, r, 0x1.71df2553d8691p-19);
  p = fma_c(p, r, 0x1.a0199a0c64c3ep-16);
  p = fma_c(p, r, 0x1.a01a012a57075p-13);
  p = fma_c(p, r, 0x1.6c16c1842a12ap-10);
  p = fma_c(p, r, 0x1.1111111127be7p-7);
  p = fma_c(p, r, 0x1.555555555087cp-5);
  p = fma_c(p, r, 0x1.55555555554fap-3);
  p = fma_c(p, r, 0x1.000000000000ap-1);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_amdgcn_ldexp(p, ki);  // gradual underflow / 0 for large |x|
}

// sqrt(x) for x > 0 and normal (the caller has clamped it from below): sqrt_nr without the select that guards x = 0
__device__ __forceinline__ double sqrt_pos(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double s = x * y;
  double h = 0.5 * y;
  double e = __builtin_fma(-h, s, 0.5);
  s = __builtin_fma(s, e, s);
  h = __builtin_fma(h, e, h);
  e = __builtin_fma(-s, s, x);
  return __builtin_fma(e, h, s);
}

// 1/d: hardware estimate + two Newton steps (quadratic: 2^-26 -> 2^-52 -> rounding level).  Deviation from IEEE division:
// d = 0 gives NaN (inf * 0 in the Newton step), not +-inf; the solvers' denominators are bounded away from 0 by the
// reference's own guards (k floors, eps thresholds: SURVEY section 9-11) wherever these sequences are used.
__device__ __forceinline__ double rcp_nr(double d) {
  double x = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, x, 1.0);
  x = __builtin_fma(x, e, x);
  e = __builtin_fma(-d, x, 1.0);
  x = __builtin_fma(x, e, x);
  return x;
}
// n/d: reciprocal, then one residual correction of the quotient (result within 1 ulp)
__device__ __forceinline__ double div_nr(double n, double d) {
  const double x = rcp_nr(d);
  const double q = n * x;
  const double res = __builtin_fma(-d, q, n);
  return __builtin_fma(res, x, q);
}
// sqrt(x), x >= 0: hardware rsq estimate, one Newton step on y = 1/sqrt(x), then Heron correction of s = x*y
__device__ __forceinline__ double sqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x > 0.0 ? x : 1.0);
  double s = x * y;               // ~ sqrt(x)
  double h = 0.5 * y;
  double e = __builtin_fma(-h, s, 0.5);
  s = __builtin_fma(s, e, s);
  h = __builtin_fma(h, e, h);
  e = __builtin_fma(-s, s, x);    // residual
  return __builtin_fma(e, h, s);
}
// sqrt(x), correctly rounded, for x = 0 or x >= 2^-767 and finite: the sequence the compiler emits for sqrt() without its range
// scaling (which only acts below 2^-767) and without its inf / NaN / zero selects -- the same bits as sqrt() on that domain,
// 12 instructions instead of 19.  A zero argument zeroes the estimate (rsq(0) = +inf has a zero low word: only the high word
// is replaced), and the sequence then returns 0.
__device__ __forceinline__ double sqrt_cr0(double x) {
  double y = __builtin_amdgcn_rsq(x);
  {
    unsigned long long b = __builtin_bit_cast(unsigned long long, y);
    const unsigned hi = x == 0.0 ? 0u : (unsigned)(b >> 32);
    b = ((unsigned long long)hi << 32) | (unsigned)b;
    y = __builtin_bit_cast(double, b);
  }
  double g = x * y;
  double h = y * 0.5;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  return __builtin_fma(d, h, g);
}
#endif

}  // namespace rte
