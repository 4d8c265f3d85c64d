// fdiv.hpp — IEEE-identical f64 division with a shared divisor.
//
// hipcc lowers a / b on gfx950 to
//     bs = v_div_scale(b,b,a); as = v_div_scale(a,b,a); y = v_rcp(bs);
//     y = fma(y, fma(-bs,y,1), y)  (twice);  q0 = as*y;  r = fma(-bs,q0,as);
//     q = v_div_fmas(r,y,q0);  result = v_div_fixup(q,b,a)
// (11 VALU instructions).  v_div_scale / v_div_fmas / v_div_fixup are the identity unless an
// operand is 0/inf/nan/denormal or the exponents are extreme, so for moderate magnitudes
// the quotient is  fma(fma(-b, a*y, a), y, a*y)  with y depending on b only.  When several
// numbers are divided by the same b (the t-scale update divides 7 residuals per lane by s^2; the
// Student-t fusion divides three terms by s1^2 + s2^2) the refined reciprocal is computed once
// and each further quotient costs 3 instructions — bit for bit the compiler's result.
// Outside the safe window the plain division is used.  esvo_selftest_division() checks the
// equivalence on the device over random and edge-case operands.
#pragma once
#include <hip/hip_runtime.h>

namespace esvo {

struct Recip {
  double b;   // divisor
  double y;   // refined reciprocal (valid iff fast)
  bool fast;  // |b| in [1e-100, 1e100]
};

// v == 0, or |v| in [2^-332, 2^333) (about 1e-100 .. 1e100): decided on the exponent bits
__device__ inline bool fdiv_ok(double v) {
  const unsigned hi = (unsigned)__double2hiint(v) & 0x7fffffffu;
  return ((hi >> 20) - 691u) <= 664u || (hi | (unsigned)__double2loint(v)) == 0u;
}

// A divisor b and four numerators at once: b, a1, a3, a4 inside the window and a2 inside it or zero -- ONE min / max over the
// five exponent fields instead of five tests whose results are materialised and and-ed (36 -> 14 instructions in the
// regulariser's fusion step).  Stricter than fdiv_ok on a1, a3, a4 (an exact zero there is refused): a refusal only sends
// the step down the literal-division path, which gives the same bits.
__device__ inline bool fdiv_ok_b4(double b, double a1, double a2_or_zero, double a3, double a4) {
  const unsigned eb = ((unsigned)__double2hiint(b) >> 20) & 0x7ffu, e1 = ((unsigned)__double2hiint(a1) >> 20) & 0x7ffu,
                 e2 = ((unsigned)__double2hiint(a2_or_zero) >> 20) & 0x7ffu, e3 = ((unsigned)__double2hiint(a3) >> 20) & 0x7ffu,
                 e4 = ((unsigned)__double2hiint(a4) >> 20) & 0x7ffu;
  const unsigned lo = min(min(eb, e1), min(e3, e4)), hi = max(max(max(eb, e1), max(e3, e4)), e2);
  return (lo - 691u) <= 664u && hi <= 1355u && (e2 >= 691u || a2_or_zero == 0.0);
}

// the refined reciprocal alone, for a divisor the caller vouches for (inside the window above)
__device__ inline double recip_refined(double b) {
  double y = __builtin_amdgcn_rcp(b);
  y = __builtin_fma(y, __builtin_fma(-b, y, 1.0), y);
  return __builtin_fma(y, __builtin_fma(-b, y, 1.0), y);
}

__device__ inline Recip make_recip(double b) {
  Recip R;
  R.b = b;
  const unsigned hi = (unsigned)__double2hiint(b) & 0x7fffffffu;
  R.fast = ((hi >> 20) - 691u) <= 664u;
  double y = __builtin_amdgcn_rcp(b);
  y = __builtin_fma(y, __builtin_fma(-b, y, 1.0), y);
  y = __builtin_fma(y, __builtin_fma(-b, y, 1.0), y);
  R.y = y;
  return R;
}

// quotient for operands already known to be in the safe window (R.fast && fdiv_ok(a))
__device__ inline double div_fast(double a, const Recip& R) {
  const double q0 = a * R.y;
  const double r = __builtin_fma(-R.b, q0, a);
  return __builtin_fma(r, R.y, q0);
}

__device__ inline double div_by(double a, const Recip& R) {
  return (R.fast && fdiv_ok(a)) ? div_fast(a, R) : a / R.b;
}

// IEEE-identical f64 square root for x in [2^-700, 2^700]: hipcc lowers sqrt(x) to
//     scale x by 2^256 if x < 2^-767;  y = v_rsq(x);  g = x*y;  h = 0.5*y;  r = fma(-h,g,0.5);  g = fma(g,r,g);
//     h = fma(h,r,h);  d = fma(-g,g,x);  g = fma(d,h,g);  d = fma(-g,g,x);  g = fma(d,h,g);  unscale;  x if x is 0/inf
// (18 VALU instructions).  For a moderate positive x the scaling and the class select are identities, so the ten
// core operations below give the same bits.  The caller vouches for the range (the LM weights are (nu+1)/(nu + r^2/s^2)
// with the exponent of r^2/s^2 already bounded); esvo_selftest_division() checks the equivalence on the device.
__device__ inline double sqrt_moderate(double x) {
  const double y = __builtin_amdgcn_rsq(x);
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

}  // namespace esvo
