// kmd_math.h -- FP64 building blocks of the Poisson likelihood-ratio test, usable from
// host and device code.  Compiled with -ffp-contract=off: the operation order below is the
// reference's and must not be fused or reassociated.
//
// The chi-square upper tail follows the Cephes igamc/igam algorithm as published (Moshier,
// Cephes Math Library 2.8) and as the reference reaches it through
// alglib::chisquarecdistribution(1, 2*LR) = igamc(1/2, LR)
// (thirdparty/alglib/src/specialfunctions.cpp:2750-2770, 9559-9567, 4655-4739, 4578-4619).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "kmd_ddmath.h"

#define KMD_HD __host__ __device__ __forceinline__

namespace kmd {

// ln Gamma(1/2) as Cephes' lgam evaluates it for x = 0.5 (the x < 13 branch:
// specialfunctions.cpp:3789-3841): the argument is shifted to u = 2.5 with z = 1/(0.5*1.5),
// then log(z) + x*B(x)/C(x) at x = 0.5.  Evaluated once on the host (kmd_model_create) in
// this exact operation order and handed to the kernels as a constant.
inline double lngamma_half_host()
{
  double x = 0.5;
  double z = 1, p = 0, u = x;
  while (u < 2) { z = z / u; p = p + 1; u = x + p; }
  p = p - 2;
  x = x + p;
  double b = -1378.25152569120859100;
  b = -38801.6315134637840924 + x * b;
  b = -331612.992738871184744 + x * b;
  b = -1162370.97492762307383 + x * b;
  b = -1721737.00820839662146 + x * b;
  b = -853555.664245765465627 + x * b;
  double c = 1;
  c = -351.815701436523470549 + x * c;
  c = -17064.2106651881159223 + x * c;
  c = -220528.590553854454839 + x * c;
  c = -1139334.44367982507207 + x * c;
  c = -2532523.07177582951285 + x * c;
  c = -2018891.41433532773231 + x * c;
  p = x * b / c;
  return ::log(z) + p;
}

constexpr double kIgamEps = 0.000000000000001;
constexpr double kIgamBig = 4503599627370496.0;
constexpr double kIgamBigInv = 2.22044604925031308085 * 0.0000000000000001;
constexpr double kIgamMinLog = -709.78271289338399;

// The two libm functions of the path: the platform's (glibc on the host, ocml on the device) or the
// correctly rounded ones of kmd_ddmath.h (rows within 1e-8 of the threshold, kmd_eval.h).
struct libm_native
{
  static KMD_HD double log(double x) { return ::log(x); }
  static KMD_HD double exp(double x) { return ::exp(x); }
};
struct libm_rounded
{
  static KMD_HD double log(double x) { return (x > 2.3e-308 && x < 1.7e308) ? ddm::log_cr(x) : ::log(x); }
  static KMD_HD double exp(double x) { return (x > -708.0 && x < 708.0) ? ddm::exp_cr(x) : ::exp(x); }
};

// igamc(1/2, x) -- continued fraction for x >= 1, 1 - power series below.
// lg_half = lngamma_half_host().
template <class M = libm_native>
KMD_HD double igamc_half(double x, double lg_half)
{
  const double a = 0.5;
  if (x <= 0) return 1;
  double ax = a * M::log(x) - x - lg_half;
  if (x < 1)   // (x < 1 || x < a) with a = 1/2
  {
    // igam(a, x): specialfunctions.cpp:4596-4617 (x > 1 && x > a cannot hold here)
    if (ax < kIgamMinLog) return 1 - 0.0;
    ax = M::exp(ax);
    double r = a, c = 1, ans = 1;
    do { r = r + 1; c = c * x / r; ans = ans + c; } while (c / ans > kIgamEps);
    return 1 - ans * ax / a;
  }
  if (ax < kIgamMinLog) return 0;
  ax = M::exp(ax);
  double y = 1 - a, z = x + y + 1, c = 0;
  double pkm2 = 1, qkm2 = x, pkm1 = x + 1, qkm1 = z * x;
  double ans = pkm1 / qkm1, t;
  do
  {
    c = c + 1; y = y + 1; z = z + 2;
    double yc = y * c;
    double pk = pkm1 * z - pkm2 * yc;
    double qk = qkm1 * z - qkm2 * yc;
    if (qk != 0) { double r = pk / qk; t = ::fabs((ans - r) / r); ans = r; }
    else t = 1;
    pkm2 = pkm1; pkm1 = pk; qkm2 = qkm1; qkm1 = qk;
    if (::fabs(pk) > kIgamBig)
    {
      pkm2 *= kIgamBigInv; pkm1 *= kIgamBigInv; qkm2 *= kIgamBigInv; qkm1 *= kIgamBigInv;
    }
  } while (t > kIgamEps);
  return ans * ax;
}

// PoissonLikelihood::poisson_prob (include/kmdiff/model.hpp:133-138) with the table value
// lf[k] already looked up.
#define KMD_LOG(x) ::log(x)
template <class M = libm_native>
KMD_HD double poisson_prob(double k, double lambda, double lf_k)
{
  if (lambda <= 0) return 0;
  return (-lambda + (k * M::log(lambda) - lf_k));
}

// `int k = mean_control` (model.hpp:152-156): double -> int truncation.  Sums >= 2^31 are
// undefined behaviour in the reference; x86-64's cvttsd2si yields INT_MIN, which
// poisson_prob clamps to 0 (model.hpp:136) -- reproduced here.
KMD_HD uint32_t table_index(uint64_t sum)
{
  return sum >= 0x80000000ull ? 0u : (uint32_t)sum;
}

struct lrt_result { double lr; double mean_control; int sign; };

// The likelihood ratio of PoissonLikelihood::process (model.hpp:147-160) from the two count
// sums.  log_sc / log_sk are log(double(sum_c)) / log(double(sum_k)) -- the two logarithms of
// the alternative hypothesis, whose arguments are integers: the kernels read them from a table
// built on the host with the same libm call the reference makes (so they are the reference's
// own values), only the two null-hypothesis logarithms are evaluated per row.
// dT = double(Tc + Tk), dTc = double(Tc), dTk = double(Tk).
template <class M = libm_native>
KMD_HD double lr_from_sums(uint64_t sum_c, uint64_t sum_k, double lf_c, double lf_k,
                           double log_sc, double log_sk, double dT, double dTc, double dTk)
{
  const double sc = (double)sum_c, sk = (double)sum_k;   // exact: sums < 2^53
  const double kc = (double)table_index(sum_c), kk = (double)table_index(sum_k);
  const double mean = (sc + sk) / dT;                                        // :147
  double alt = 0, nul = 0;
  alt += (sc <= 0) ? 0.0 : (-sc + (kc * log_sc - lf_c));                     // :152 (poisson_prob :133-138)
  alt += (sk <= 0) ? 0.0 : (-sk + (kk * log_sk - lf_k));                     // :153
  nul += poisson_prob<M>(kc, mean * dTc, lf_c);                              // :155
  nul += poisson_prob<M>(kk, mean * dTk, lf_k);                              // :156
  double lr = alt - nul;                                                     // :158
  if (lr < 0) lr = 0;                                                        // :160
  return lr;
}

// model.hpp:165-172: normalised control sum and the sign
KMD_HD void sign_of(uint64_t sum_c, uint64_t sum_k, double dTc, double dTk, double& mean_control, int& sign)
{
  const double sc = (double)sum_c, sk = (double)sum_k;
  mean_control = sc * dTk / dTc;                                             // :165
  sign = (mean_control < sk) ? 1 : (mean_control > sk) ? 0 : 2;              // :167-172
}

// everything at once, logarithms evaluated here (k_process_all, host-side uses)
KMD_HD lrt_result lrt_from_sums(uint64_t sum_c, uint64_t sum_k, double lf_c, double lf_k,
                                double dT, double dTc, double dTk)
{
  lrt_result r;
  const double lc = sum_c ? KMD_LOG((double)sum_c) : 0.0, lk = sum_k ? KMD_LOG((double)sum_k) : 0.0;
  r.lr = lr_from_sums(sum_c, sum_k, lf_c, lf_k, lc, lk, dT, dTc, dTk);
  sign_of(sum_c, sum_k, dTc, dTk, r.mean_control, r.sign);
  return r;
}

} // namespace kmd
