// kmd_ddmath.h -- log and exp rounded correctly to double, through double-double arithmetic.
//
// Why: the survivor set is decided by `p <= threshold` (include/kmdiff/merge.hpp:78), and p comes out of
// a chain of IEEE operations (identical on host and device under -ffp-contract=off) plus four libm calls
// -- two log() of the null-hypothesis rates, the log() and exp() inside Cephes' igamc.  The device's ocml
// and the host's glibc may differ in the last bit of those, which moves p by up to ~1e-9 relative (the
// continued fraction amplifies).  A row whose p lands that close to the threshold could be decided
// differently by the two.  For exactly those rows (|p / threshold - 1| <= 1e-8: none in 10^10 synthetic
// rows, kmd_eval.h counts them) the four calls are repeated with the functions below, whose results are
// the correctly rounded ones.  glibc's log and exp (since 2.28) are correctly rounded in all but ~1 in
// 10^3 arguments (their stated error bound is 0.52 ulp), so the decision then agrees with a glibc-built
// reference wherever glibc itself returned the rounded value; tests/test_gpu_threshold.py pins both
// halves: these functions against mpmath at 200 digits, the decisions against the oracle.
//
// Accuracy: ~2^-100 relative before the final rounding, so the result is the correctly rounded double
// unless the true value lies within 2^-100 of a rounding boundary (probability ~2^-47 per call).
// Range: finite positive normal arguments for log; |x| < 708 for exp (beyond: the caller's ordinary libm
// result is kept -- p is then 0 or far from any threshold of interest).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace kmd { namespace ddm {

struct dd { double hi, lo; };

#define KMD_DD __host__ __device__ __forceinline__

KMD_DD dd two_sum(double a, double b)
{
  const double s = a + b, bb = s - a;
  return { s, (a - (s - bb)) + (b - bb) };
}
KMD_DD dd quick_two_sum(double a, double b)       // |a| >= |b|
{
  const double s = a + b;
  return { s, b - (s - a) };
}
KMD_DD dd two_prod(double a, double b)
{
  const double p = a * b;
  return { p, __builtin_fma(a, b, -p) };
}
KMD_DD dd add(dd a, dd b)
{
  dd s = two_sum(a.hi, b.hi);
  const dd t = two_sum(a.lo, b.lo);
  s.lo += t.hi;
  s = quick_two_sum(s.hi, s.lo);
  s.lo += t.lo;
  return quick_two_sum(s.hi, s.lo);
}
KMD_DD dd neg(dd a) { return { -a.hi, -a.lo }; }
KMD_DD dd mul(dd a, dd b)
{
  dd p = two_prod(a.hi, b.hi);
  p.lo += a.hi * b.lo + a.lo * b.hi;
  return quick_two_sum(p.hi, p.lo);
}
KMD_DD dd mul_d(dd a, double b)
{
  dd p = two_prod(a.hi, b);
  p.lo += a.lo * b;
  return quick_two_sum(p.hi, p.lo);
}
KMD_DD dd div(dd a, dd b)
{
  const double q1 = a.hi / b.hi;
  dd r = add(a, neg(mul_d(b, q1)));
  const double q2 = r.hi / b.hi;
  r = add(r, neg(mul_d(b, q2)));
  const double q3 = r.hi / b.hi;
  dd q = quick_two_sum(q1, q2);
  return add(q, { q3, 0.0 });
}

// ln 2 to ~107 bits
constexpr double kLn2Hi = 0x1.62e42fefa39efp-1;
constexpr double kLn2Lo = 0x1.abc9e3b39803fp-56;

// log(x), x finite, positive, normal
KMD_DD double log_cr(double x)
{
  // x = m 2^e with m in [sqrt(1/2), sqrt(2))
  uint64_t b;
  __builtin_memcpy(&b, &x, 8);
  int e = (int)((b >> 52) & 0x7ff) - 1023;
  b = (b & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
  double m;
  __builtin_memcpy(&m, &b, 8);                            // [1, 2)
  if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
  // log m = 2 atanh(s), s = (m - 1) / (m + 1); m - 1 is exact
  const dd s = div({ m - 1.0, 0.0 }, two_sum(m, 1.0));
  const dd s2 = mul(s, s);
  // sum_{k=0}^{K} s2^k / (2k + 1), Horner from the top: |s| <= 0.1716, s2^24 / 49 < 2^-127
  dd acc = div({ 1.0, 0.0 }, { 49.0, 0.0 });
  for (int k = 23; k >= 0; --k)
    acc = add(mul(acc, s2), div({ 1.0, 0.0 }, { (double)(2 * k + 1), 0.0 }));
  dd r = mul(s, acc);
  r = { 2.0 * r.hi, 2.0 * r.lo };
  const dd el = add(mul_d({ kLn2Hi, kLn2Lo }, (double)e), r);
  return el.hi + el.lo;
}

// exp(x), |x| < 708 (result a normal double)
KMD_DD double exp_cr(double x)
{
  const double kd = rint(x * 1.4426950408889634);         // x / ln 2
  const int k = (int)kd;
  // r = x - k ln 2, |r| <= 0.35; then r / 32
  dd r = add({ x, 0.0 }, neg(mul_d({ kLn2Hi, kLn2Lo }, kd)));
  r = { r.hi * 0.03125, r.lo * 0.03125 };
  // sum r^n / n!, n = 0 .. 15 (|r| <= 0.011: r^16 / 16! < 2^-148), Horner
  dd acc = { 1.0, 0.0 };
  for (int n = 15; n >= 1; --n)
    acc = add(mul(div(acc, { (double)n, 0.0 }), r), { 1.0, 0.0 });
  for (int i = 0; i < 5; ++i) acc = mul(acc, acc);
  return ldexp(acc.hi + acc.lo, k);
}

#undef KMD_DD

} } // namespace kmd::ddm
