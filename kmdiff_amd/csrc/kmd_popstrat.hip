// kmd_popstrat.hip -- K3: population-stratification re-test of the survivors.
//
// Replaces pop_strat_corrector (include/kmdiff/popstrat.hpp:148-367, src/popstrat.cpp:136-370)
// and the IRLS logistic regression it drives (glm_irls, src/linear_model.cpp:297-410, with
// lu_decomposition :94-132, inverse :134-189, sigmoid :191-195):
//   for each survivor: design = [1, PCs, total, kmer_count/total]; fit by IRLS; likelihood
//   ratio against the once-fitted null model; p = chi2(1) tail -> replaces the Poisson p.
//
// Shape: FP64-VALU-bound on survivors only (<< rows), not on the HBM roofline.  One lane per
// survivor: the reference's arithmetic is a chain of ORDERED sums over samples (Hessian
// X^T S X accumulated sample by sample, no-pivot LU, per-column substitutions); keeping a
// whole fit inside one lane preserves that order exactly, and the parallelism comes from
// the number of survivors (10^5..10^6 per run).  The shared design columns are wave-uniform
// (staged once per wave in LDS: n x (F + 1) doubles; from global memory when that does not fit);
// per-sample state (eta, mu) is recomputed on the fly from the current weights instead of being
// stored; the inverse of the Hessian is never stored either (its columns go into the weight update as
// they are solved), so a fit needs F*F + O(F) doubles of registers.  The survivors' own column -- their
// count in every sample -- is fetched one sample ahead.
//
// Several reference quirks are behaviour and are reproduced: the
// standardisation divides column sums by ncols and scales ROW i by the deviation of
// COLUMN i; the k-mer column is never standardised; weights are not copied when the
// iteration limit is hit; `det` is the running product over every column solve.
#include "kmd_internal.h"
#include "kmd_math.h"

#include <cmath>
#include <cstring>
#include <cstdlib>
#include <new>
#include <vector>

struct kmd_popstrat
{
  int n;            // samples
  int f;            // alt feature count = null + 1
  int max_iter;
  std::vector<double> h_alt;      // n x f, row-major (last column is the per-k-mer slot)
  std::vector<double> h_null_model;
  double null_likelihood;
  double lg_half;
  double epsilon;   // pop_strat_corrector::s_epsilon (popstrat.hpp:154,172-173,321)
  double* d_alt;    // device copies
  double* d_y;
  double* d_totals;
  double* d_null_model;
};

namespace {

constexpr double kE = 2.718281828459045235360287471352662498;   // M_E

// linear_model.cpp:191-195: 1 / (1 + pow(e, -x)) with e = M_E, the double nearest to e.  ln(M_E) rounds to
// exactly 1.0 in double precision (it is 1 - 5.3e-17), so pow(M_E, -x) = exp(-x) (1 + 5.3e-17 x): evaluated
// as exp(-x) -- a third of pow's instructions, and the sigmoid is on the critical path of every sample of
// every iteration -- it stays within 4e-14 relative of the reference's value over the whole range where the
// result is not 0 or 1 anyway (|x| < 745); the p-values keep the 1e-7 relative bar of the parity tests.
#ifndef KMD_SIGMOID_POW
#define KMD_SIGMOID_POW 0
#endif
__device__ __forceinline__ double sigmoid_ref(double x)
{
#if KMD_SIGMOID_POW
  return 1.0 / (1.0 + ::pow(kE, -x));
#else
  return 1.0 / (1.0 + ::exp(-x));
#endif
}

// no-pivot Doolittle LU in place (linear_model.cpp:94-132) + per-column solves (:134-189) + the weight
// update w = H^-1 (X^T S z) (:381).  a: F x F (destroyed: L below the diagonal, U on and above).
// Column c of the inverse is used the moment it is solved: w[p] = sum over c, ascending, of
// inv[p][c] * b[c] -- the order multiply() adds them in (:77-86) -- so the inverse is never stored.
// Returns 1 singular (det == 0), 2 NaN det, 0 ok (w valid).  The sums skip the structural zeros of the
// reference's full-width loops (lower[r][c >= r] * y[c] with y[c] still 0): adding +0.0 terms does not
// change a sum.
template <int F>
__device__ __forceinline__ int lu_solve(double (&a)[F][F], const double (&b)[F], double (&w)[F])
{
#pragma unroll
  for (int i = 0; i < F; ++i)
  {
#pragma unroll
    for (int k = i; k < F; ++k)
    {
      double sum = 0.0;
#pragma unroll
      for (int j = 0; j < i; ++j) sum += a[i][j] * a[j][k];
      a[i][k] = a[i][k] - sum;
    }
#pragma unroll
    for (int k = i + 1; k < F; ++k)
    {
      double sum = 0;
#pragma unroll
      for (int j = 0; j < i; ++j) sum += a[k][j] * a[j][i];
      a[k][i] = (a[k][i] - sum) / a[i][i];
    }
  }
  double det = 1;
#pragma unroll
  for (int p = 0; p < F; ++p) w[p] = 0.0;
#pragma unroll
  for (int c = 0; c < F; ++c)
  {
    double y[F], x[F];
    y[0] = (c == 0) ? 1.0 : 0.0;
#pragma unroll
    for (int row = 1; row < F; ++row)
    {
      double sum = 0;
#pragma unroll
      for (int col = 0; col < row; ++col) sum += a[row][col] * y[col];
      y[row] = ((c == row) ? 1.0 : 0.0) - sum;
    }
    x[F - 1] = y[F - 1] / a[F - 1][F - 1];
    det *= a[F - 1][F - 1];
#pragma unroll
    for (int row = F - 2; row > -1; --row)
    {
      double sum = 0;
#pragma unroll
      for (int col = row + 1; col < F; ++col) sum += a[row][col] * x[col];
      x[row] = (y[row] - sum) / a[row][row];
      det *= a[row][row];
    }
#pragma unroll
    for (int p = 0; p < F; ++p) w[p] = w[p] + x[p] * b[c];          // inv[p][c] * b[c]
  }
  if (det == 0) return 1;
  if (det != det) return 2;
  return 0;
}

struct irls_args
{
  const double* alt;      // n x FA row-major design (FA = stride)
  int stride;             // FA
  const double* y;
  const double* totals;
  const double* counts;   // count of sample i for survivor v: counts[i * si + v * ss]
  size_t si, ss;          // sample-major [S][ld]: (ld, 1) -- what a lane per survivor reads coalesced; survivor-major [n][S]: (1, S)
  int n;
  int max_iter;
};

// The shared part of a sample's design row: features 0 .. F-2, phenotype, total.  LDSD: staged in LDS
// as [n][F + 1] (feature 0 .. F-2 | y | total); else read from global memory (wave-uniform addresses).
template <int F, bool LDSD>
struct design_rows
{
  const irls_args& A;
  const double* s_d;
  __device__ __forceinline__ double x(int i, int j) const { return LDSD ? s_d[i * (F + 1) + j] : A.alt[i * A.stride + j]; }
  __device__ __forceinline__ double y(int i) const { return LDSD ? s_d[i * (F + 1) + F - 1] : A.y[i]; }
  __device__ __forceinline__ double total(int i) const { return LDSD ? s_d[i * (F + 1) + F] : A.totals[i]; }
};

// glm_irls (linear_model.cpp:297-410) over F features.  KMER: feature F-1 of each sample is
// counts/totals (popstrat.hpp:254-257), else all F features come from `alt`.
template <int F, bool KMER, bool LDSD>
__device__ __forceinline__ void irls_fit(const irls_args& A, const double* s_d, size_t surv, double (&weight)[F], int* iters_out = nullptr)
{
  const design_rows<F, LDSD> D { A, s_d };
  double w[F];
#pragma unroll
  for (int j = 0; j < F; ++j) { weight[j] = 1; w[j] = 1; }
  double prev_error = 1e18;
  int iter = 0;
  bool first = true;
  for (;;)
  {
    double H[F][F], b[F];
#pragma unroll
    for (int p = 0; p < F; ++p) { b[p] = 0.0;
#pragma unroll
      for (int q = 0; q < F; ++q) H[p][q] = 0.0; }
    double error = 0.0;
    int ng = 0;
    auto sample = [&](int i, double count)
    {
      double x[F];
#pragma unroll
      for (int j = 0; j < F - 1; ++j) x[j] = D.x(i, j);
      x[F - 1] = KMER ? count / D.total(i) : A.alt[i * A.stride + F - 1];
      const double yi = D.y(i);
      double eta, mu;
      if (first)
      {
        mu = (yi + 0.5) / 2;                                     // :314
        eta = ::log(mu / (1 - mu));                              // :315
      }
      else
      {
        eta = 0;                                                 // :400-405
#pragma unroll
        for (int j = 0; j < F; ++j) eta += x[j] * w[j];
        mu = sigmoid_ref(eta);
      }
      const double g = mu * (1.0 - mu);                          // :333
      if (g > 1e-305)
      {
        ++ng;
        const double z = eta + (yi - mu) / (g + 1e-305);         // :338
#pragma unroll
        for (int p = 0; p < F; ++p)
        {
#pragma unroll
          for (int q = 0; q < F; ++q) H[p][q] = H[p][q] + x[p] * (g * x[q]);   // :357-364
          b[p] = b[p] + x[p] * (g * z);                                           // :376-380
        }
      }
      error += (yi - mu) * (yi - mu);                            // :341
    };
    {
      // the count of the next sample is requested before this one is worked on
      double c_next = KMER ? A.counts[surv * A.ss] : 0.0;
      for (int i = 0; i < A.n; ++i)
      {
        const double c = c_next;
        if (KMER) c_next = A.counts[(size_t)(i + 1 < A.n ? i + 1 : i) * A.si + surv * A.ss];
        sample(i, c);
      }
    }
    first = false;
    if (ng == 0) break;                                          // :343
    error /= A.n;
    if (::fabs(error - prev_error) < 1e-6) break;                // :349
    prev_error = error;
    if (lu_solve<F>(H, b, w)) break;                             // :366-381 (w is dead on this way out)
    iter += 1;
    if (iter >= A.max_iter) break;                               // :386-389 (weight NOT updated)
    prev_error = error;
#pragma unroll
    for (int j = 0; j < F; ++j) weight[j] = w[j];                // :394-395
  }
  if (iters_out) *iters_out = iter;                              // (the test hook's: what glm_irls counts, :385)
}

// pop_strat_corrector::apply(KmerSign&) (popstrat.hpp:249-333) for one survivor per lane
#ifndef KMD_PS_WAVES
#define KMD_PS_WAVES 1               // waves per SIMD the register allocator is held to (dev: A/B)
#endif
template <int F, bool LDSD>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(KMD_PS_WAVES)))
k_popstrat_apply(irls_args A, size_t n_surv, double null_likelihood,
                                                       double lg_half, double epsilon, double* __restrict__ out_p)
{
  extern __shared__ double s_d[];
  if constexpr (LDSD)
  {
    // the shared design, once per wave: [n][feature 0 .. F-2 | y | total]
    for (int t = threadIdx.x; t < A.n * (F + 1); t += 64)
    {
      const int i = t / (F + 1), j = t - i * (F + 1);
      s_d[t] = j < F - 1 ? A.alt[i * A.stride + j] : j == F - 1 ? A.y[i] : A.totals[i];
    }
    __syncthreads();
  }
  size_t surv = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = surv < n_surv;
  if (!live) surv = n_surv - 1;                                  // (keeps the wave together; result dropped)
  const design_rows<F, LDSD> D { A, s_d };
  double model[F];
  irls_fit<F, true, LDSD>(A, s_d, surv, model);
  double alt_l = 1.0;                                            // :263-287
  {
    auto sample = [&](int i, double count)
    {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < F; ++j)
      {
        const double xj = (j == F - 1) ? count / D.total(i) : D.x(i, j);
        s += model[j] * xj;
      }
      const double p = sigmoid_ref(s);
      if (D.y(i) == 1) alt_l = alt_l * p; else alt_l *= 1.0 - p;
    };
    double c_next = A.counts[surv * A.ss];
    for (int i = 0; i < A.n; ++i)
    {
      const double c = c_next;
      c_next = A.counts[(size_t)(i + 1 < A.n ? i + 1 : i) * A.si + surv * A.ss];
      sample(i, c);
    }
  }
  double null_l = null_likelihood;                               // :289-310 (same for every k-mer)
  if (null_l == 0.0 && alt_l == 0.0) { null_l = 0.001; alt_l = 1.0; }     // :312-316
  const double ratio = null_l / alt_l;
  double llr = -2.0 * ::log(ratio);                              // :318-319
  if (::fabs(llr) < epsilon || llr < 0.0 || alt_l != alt_l) llr = 0.0;    // :321-326
  if (live) out_p[surv] = kmd::igamc_half(llr / 2.0, lg_half);   // :328 chisquarecdistribution(1, llr)
}

// ---- the same fit with a GROUP of lanes per survivor ---------------------------------------------------
// One lane per survivor runs ~2 x 10^5 dependent instructions per fit: a small batch of survivors (one
// partition's: 10^3..10^4) leaves most SIMDs with one wave or none, and beyond 7 features the F x F Hessian
// does not fit a lane's registers any more (npc = 10: 1.3 KB of scratch per lane, 42 ms for 22 k survivors).
// Here L lanes (32 up to 5 features, else 64) share a survivor:
//   * samples are PREPARED L at a time, one per lane (design row, dot product, sigmoid, g, g z: the long
//     dependent chains, now L-wide) into an LDS chunk;
//   * the F x F + F + 1 running sums -- Hessian entries, X^T S z entries, the squared error -- are spread over
//     the lanes, and every lane adds its entries' terms sample after sample IN SAMPLE ORDER, exactly the sums
//     the reference forms (multiply(), linear_model.cpp:77-86);
//   * the no-pivot LU runs in LDS, one row / column of entries per step across the lanes, the F column solves
//     of inverse() one per lane; det is formed by every lane in the reference's order.
// Every value is computed by the same operations in the same order as in irls_fit above: the two kernels
// return the same bits (tests/test_gpu_popstrat.py::test_group_kernel_equals_lane_kernel).
__device__ __forceinline__ void group_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int F> struct group_shape
{
  static constexpr int kEntries = F * F + F + 1;          // H, X^T S z, squared error
  static constexpr int L = kEntries <= 32 ? 32 : 64;       // lanes per survivor
  static constexpr int kPerLane = (kEntries + L - 1) / L;
  static constexpr int kSample = 2 * F + 3;                // chunk entry: x[F], g x[F], g z, (y - mu)^2, good
  // doubles of LDS per survivor: chunk | H | b | w | weight | inverse columns | scalars
  static constexpr int kLds = L * kSample + F * F + 3 * F + F * F + 8;
};

// The linear algebra of one IRLS step of the group kernel, on the group's LDS: no-pivot LU of Hm in place, the F column
// solves of inverse() one per lane (xc, column-major), det, and wv = H^-1 bv.  `active` goes false on det == 0 / NaN
// (:366-373).  A function of its own so that the reference's vectors (tests/linear_test.cpp:80-151) can be pushed through
// the very code the kernel runs (kmd_test_popstrat_linear).
template <int F>
__device__ __forceinline__ void group_lu_solve(double* const Hm, const double* const bv, double* const xc, double* const wv, bool& active, const int l)
{
  // no-pivot Doolittle LU in place (:94-132): row i of U across the lanes, then column i of L
  for (int i = 0; i < F; ++i)
  {
    if (active && l >= i && l < F)
    {
      const int k = l;
      double sum = 0.0;
      for (int j = 0; j < i; ++j) sum += Hm[i * F + j] * Hm[j * F + k];
      Hm[i * F + k] = Hm[i * F + k] - sum;
    }
    group_sync();
    if (active && l > i && l < F)
    {
      const int k = l;
      double sum = 0;
      for (int j = 0; j < i; ++j) sum += Hm[k * F + j] * Hm[j * F + i];
      Hm[k * F + i] = (Hm[k * F + i] - sum) / Hm[i * F + i];
    }
    group_sync();
  }
  // inverse() (:134-189): lane c solves column c; det is the running product over every column solve
  if (active && l < F)
  {
    const int c = l;
    double y[F], x[F];
    y[0] = (c == 0) ? 1.0 : 0.0;
#pragma unroll
    for (int row = 1; row < F; ++row)
    {
      double sum = 0;
      for (int col = 0; col < row; ++col) sum += Hm[row * F + col] * y[col];
      y[row] = ((c == row) ? 1.0 : 0.0) - sum;
    }
    x[F - 1] = y[F - 1] / Hm[(F - 1) * F + (F - 1)];
#pragma unroll
    for (int row = F - 2; row > -1; --row)
    {
      double sum = 0;
      for (int col = row + 1; col < F; ++col) sum += Hm[row * F + col] * x[col];
      x[row] = (y[row] - sum) / Hm[row * F + row];
    }
#pragma unroll
    for (int p = 0; p < F; ++p) xc[c * F + p] = x[p];
  }
  group_sync();
  if (active)
  {
    double det = 1;
    for (int c = 0; c < F; ++c)
      for (int row = F - 1; row > -1; --row) det *= Hm[row * F + row];
    if (det == 0 || det != det) active = false;                  // :366-373
  }
  if (active && l < F)
  {
    double r = 0.0;
    for (int c = 0; c < F; ++c) r = r + xc[c * F + l] * bv[c];   // :381
    wv[l] = r;
  }
  group_sync();
}

template <int F>
__global__ void __launch_bounds__(64) k_popstrat_group(irls_args A, size_t n_surv, double null_likelihood,
                                                       double lg_half, double epsilon, double* __restrict__ out_p,
                                                       double* __restrict__ out_w = nullptr, int* __restrict__ out_iter = nullptr)
{
  using G = group_shape<F>;
  constexpr int L = G::L, kGroups = 64 / L;
  extern __shared__ double s_all[];
  const int lane = (int)threadIdx.x, grp = lane / L, l = lane % L;
  double* const base = s_all + (size_t)grp * G::kLds;
  double* const chunk = base;                              // [L][kSample]
  double* const Hm = chunk + L * G::kSample;               // [F][F]
  double* const bv = Hm + F * F;                           // [F]
  double* const wv = bv + F;                               // [F]  new weights
  double* const weight = wv + F;                           // [F]  returned weights
  double* const xc = weight + F;                           // [F][F] inverse, column-major: xc[c][p]
  double* const sc = xc + F * F;                           // [0] squared error, [1] rows with g > 1e-305
  size_t surv = (size_t)blockIdx.x * kGroups + grp;
  const bool live = surv < n_surv;
  if (!live) surv = n_surv - 1;

  // the entries this lane owns: e = l + m L
  int ep[G::kPerLane], eq[G::kPerLane], ek[G::kPerLane];   // kind 0 H[p][q], 1 b[p], 2 error, -1 none
#pragma unroll
  for (int m = 0; m < G::kPerLane; ++m)
  {
    const int e = l + m * L;
    ek[m] = e < F * F ? 0 : e < F * F + F ? 1 : e == F * F + F ? 2 : -1;
    ep[m] = e < F * F ? e / F : e - F * F;
    eq[m] = e < F * F ? e % F : 0;
  }
  double w[F];
#pragma unroll
  for (int j = 0; j < F; ++j) w[j] = 1;
  if (l < F) weight[l] = 1;
  double prev_error = 1e18;
  int iter = 0;
  bool first = true, active = true;
  const double* cnt = A.counts;

  // one sample's design row, the k-mer column from the survivor's count
  auto row_of = [&](int i, double (&x)[F])
  {
#pragma unroll
    for (int j = 0; j < F - 1; ++j) x[j] = A.alt[i * A.stride + j];
    x[F - 1] = cnt[(size_t)i * A.si + surv * A.ss] / A.totals[i];
  };

  while (__ballot(active))
  {
    double acc[G::kPerLane];
    int ng = 0;
#pragma unroll
    for (int m = 0; m < G::kPerLane; ++m) acc[m] = 0.0;
    for (int c0 = 0; c0 < A.n; c0 += L)
    {
      const int i = c0 + l;
      if (active && i < A.n)
      {
        double x[F];
        row_of(i, x);
        const double yi = A.y[i];
        double eta, mu;
        if (first)
        {
          mu = (yi + 0.5) / 2;                                     // :314
          eta = ::log(mu / (1 - mu));                              // :315
        }
        else
        {
          eta = 0;                                                 // :400-405
#pragma unroll
          for (int j = 0; j < F; ++j) eta += x[j] * w[j];
          mu = sigmoid_ref(eta);
        }
        const double g = mu * (1.0 - mu);                          // :333
        const double z = eta + (yi - mu) / (g + 1e-305);           // :338
        double* s = chunk + l * G::kSample;
#pragma unroll
        for (int j = 0; j < F; ++j) { s[j] = x[j]; s[F + j] = g * x[j]; }
        s[2 * F] = g * z;
        s[2 * F + 1] = (yi - mu) * (yi - mu);
        s[2 * F + 2] = g > 1e-305 ? 1.0 : 0.0;
      }
      group_sync();
      const int tn = A.n - c0 < L ? A.n - c0 : L;
      if (active)
      {
        // the two factors of every term sit at fixed offsets of a chunk entry: left = x[p] (or (y - mu)^2 for the
        // error entry, times 1), right = g x[q] / g z (or 1); four samples' worth are fetched before the four
        // dependent additions (an LDS round trip per addition made this loop the kernel's time); kU samples at a time
        int off_a[G::kPerLane], off_b[G::kPerLane];
#pragma unroll
        for (int m = 0; m < G::kPerLane; ++m)
        {
          off_a[m] = ek[m] == 2 ? 2 * F + 1 : ep[m];
          off_b[m] = ek[m] == 0 ? F + eq[m] : 2 * F;
        }
        constexpr int kU = G::kPerLane == 1 ? 8 : 4;
        for (int t0 = 0; t0 < tn; t0 += kU)
        {
          double fa[kU][G::kPerLane], fb[kU][G::kPerLane], fg[kU];
#pragma unroll
          for (int u = 0; u < kU; ++u)
          {
            const double* s = chunk + (t0 + u < tn ? t0 + u : tn - 1) * G::kSample;
            fg[u] = s[2 * F + 2];
#pragma unroll
            for (int m = 0; m < G::kPerLane; ++m) { fa[u][m] = s[off_a[m]]; fb[u][m] = s[off_b[m]]; }
          }
#pragma unroll
          for (int u = 0; u < kU; ++u)
          {
            if (t0 + u >= tn) break;
            const bool good = fg[u] != 0.0;
#pragma unroll
            for (int m = 0; m < G::kPerLane; ++m)
            {
              if (ek[m] == 2) { acc[m] += fa[u][m]; ng += good ? 1 : 0; }                    // :341
              else if (ek[m] >= 0 && good) acc[m] = acc[m] + fa[u][m] * fb[u][m];             // :357-364, :376-380
            }
          }
        }
      }
      group_sync();
    }
    first = false;
    // the sums to where every lane can read them
#pragma unroll
    for (int m = 0; m < G::kPerLane; ++m)
    {
      if (ek[m] == 0) Hm[ep[m] * F + eq[m]] = acc[m];
      else if (ek[m] == 1) bv[ep[m]] = acc[m];
      else if (ek[m] == 2) { sc[0] = acc[m]; sc[1] = (double)ng; }
    }
    group_sync();
    if (active)
    {
      double error = sc[0];
      if (sc[1] == 0.0) active = false;                            // :343
      else
      {
        error /= A.n;
        if (::fabs(error - prev_error) < 1e-6) active = false;     // :349
        else prev_error = error;
      }
    }
    group_lu_solve<F>(Hm, bv, xc, wv, active, l);
    if (active)
    {
#pragma unroll
      for (int j = 0; j < F; ++j) w[j] = wv[j];
      iter += 1;
      if (iter >= A.max_iter) active = false;                      // :386-389 (weight NOT updated)
      else if (l < F) weight[l] = w[l];                            // :394-395
    }
    group_sync();
  }

  // the alternative likelihood (popstrat.hpp:263-287): factors prepared L at a time, multiplied in sample order
  double model[F];
#pragma unroll
  for (int j = 0; j < F; ++j) model[j] = weight[j];
  // (kmd_test_popstrat_irls: the loop's own result -- weights as returned, iterations as glm_irls counts them)
  if (out_w && live && l < F) out_w[surv * F + l] = weight[l];
  if (out_iter && live && l == 0) out_iter[surv] = iter;
  double alt_l = 1.0;
  for (int c0 = 0; c0 < A.n; c0 += L)
  {
    const int i = c0 + l;
    if (i < A.n)
    {
      double x[F];
      row_of(i, x);
      double sdot = 0.0;
#pragma unroll
      for (int j = 0; j < F; ++j) sdot += model[j] * x[j];
      const double p = sigmoid_ref(sdot);
      chunk[l] = A.y[i] == 1 ? p : 1.0 - p;
    }
    group_sync();
    const int tn = A.n - c0 < L ? A.n - c0 : L;
    for (int t = 0; t < tn; ++t) alt_l = alt_l * chunk[t];
    group_sync();
  }
  double null_l = null_likelihood;                               // :289-310 (same for every k-mer)
  if (null_l == 0.0 && alt_l == 0.0) { null_l = 0.001; alt_l = 1.0; }     // :312-316
  const double ratio = null_l / alt_l;
  double llr = -2.0 * ::log(ratio);                              // :318-319
  if (::fabs(llr) < epsilon || llr < 0.0 || alt_l != alt_l) llr = 0.0;    // :321-326
  if (live && l == 0) out_p[surv] = kmd::igamc_half(llr / 2.0, lg_half);  // :328
}

// the null model: glm_irls(null features, Y) (popstrat.cpp:316-324) and its likelihood
template <int F>
__global__ void k_popstrat_null(irls_args A, double* __restrict__ out_model, double* __restrict__ out_like)
{
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double model[F];
  irls_fit<F, false, false>(A, nullptr, 0, model);
  double l = 1.0;
  for (int i = 0; i < A.n; ++i)
  {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < F; ++j) s += model[j] * A.alt[i * A.stride + j];
    const double p = sigmoid_ref(s);
    if (A.y[i] == 1) l *= p; else l *= 1.0 - p;
  }
#pragma unroll
  for (int j = 0; j < F; ++j) out_model[j] = model[j];
  *out_like = l;
}

// [n][S] (KmerSign::m_counts_ratio order) -> [S][ld] sample-major
__global__ void __launch_bounds__(256) k_transpose_counts(const double* __restrict__ in, size_t n, int S,
                                                          size_t ld, double* __restrict__ out)
{
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * (size_t)S) return;
  const size_t i = t / (size_t)S;
  const int s = (int)(t - i * (size_t)S);
  out[(size_t)s * ld + i] = in[t];
}

// Which kernel: measured at 100v100 (tools/kbench_popstrat.py, KMD_POPSTRAT_KERNEL=lane|group) -- 22 k survivors:
// F = 5: 0.98 (group) / 1.44 ms (lane), F = 7: 1.72 / 1.77, F = 9: 2.56 / 2.06, F = 13: 4.3 / 6.4; 68 k survivors,
// F = 5: 2.65 / 1.93; 364 k: 25 / 5.7 (F = 5), 66 / 28 (F = 13).  A lane per survivor is the throughput design; the
// group kernel is for batches that cannot fill the chip (one partition's survivors).
inline bool use_group_kernel(int F, size_t n)
{
  if (const char* e = std::getenv("KMD_POPSTRAT_KERNEL")) return e[0] == 'g';   // tests, A/B
  return (F <= 7 && n < 32768) || (F >= 12 && n < 65536);
}

template <int F>
void launch_apply(const irls_args& A, size_t n, double null_like, double lg_half, double epsilon, double* d_p, hipStream_t st)
{
  using G = group_shape<F>;
  const bool group = use_group_kernel(F, n);
  if (group)
  {
    const size_t lds = (size_t)(64 / G::L) * G::kLds * sizeof(double);
    const unsigned grid = (unsigned)((n + 64 / G::L - 1) / (64 / G::L));
    hipLaunchKernelGGL((k_popstrat_group<F>), dim3(grid), dim3(64), lds, st, A, n, null_like, lg_half, epsilon, d_p);
    return;
  }
  // the shared design in LDS when it fits comfortably (n x (F + 1) doubles per wave; 11 KB at 100v100)
  const size_t lds = (size_t)A.n * (F + 1) * sizeof(double);
  const unsigned grid = (unsigned)((n + 63) / 64);
  if (lds <= 40 * 1024)
    hipLaunchKernelGGL((k_popstrat_apply<F, true>), dim3(grid), dim3(64), lds, st, A, n, null_like, lg_half, epsilon, d_p);
  else
    hipLaunchKernelGGL((k_popstrat_apply<F, false>), dim3(grid), dim3(64), 0, st, A, n, null_like, lg_half, epsilon, d_p);
}

template <int F>
void launch_null(const irls_args& A, double* d_model, double* d_like, hipStream_t st)
{
  hipLaunchKernelGGL((k_popstrat_null<F>), dim3(1), dim3(64), 0, st, A, d_model, d_like);
}

} // namespace

extern "C" {

int kmd_popstrat_create(kmd_popstrat** out, int nb_controls, int nb_cases,
                        const uint64_t* total_controls, const uint64_t* total_cases,
                        const double* Z, int z_cols, int npc, const double* Y,
                        int standardize, int max_iter)
{
  KMD_REQUIRE(out && total_controls && total_cases && Z && Y, "kmd_popstrat_create: NULL");
  KMD_REQUIRE(nb_controls > 0 && nb_cases > 0, "kmd_popstrat_create: sample counts");
  KMD_REQUIRE(npc >= 0 && npc <= z_cols && npc <= 10, "kmd_popstrat_create: npc");
  const int n = nb_controls + nb_cases;
  const int fn = 1 + npc + 0 + 1;            // popstrat.cpp:272 (no covariates, sex unknown: m_unkg == n)
  const int fa = fn + 1;                     // :273
  KMD_REQUIRE(fa <= 13, "kmd_popstrat_create: too many features");
  kmd_popstrat* ps = new (std::nothrow) kmd_popstrat();
  if (!ps) return KMD_E_NOMEM;
  ps->n = n; ps->f = fa; ps->max_iter = max_iter > 0 ? max_iter : 100;    // popstrat.hpp:151,162-178
  ps->lg_half = kmd::lngamma_half_host();
  ps->epsilon = 1e-30;                                                      // popstrat.hpp:154
  ps->d_alt = ps->d_y = ps->d_totals = ps->d_null_model = nullptr;
  std::vector<double> totals(n);
  for (int i = 0; i < nb_controls; ++i) totals[i] = (double)total_controls[i];           // popstrat.cpp:144-145
  for (int i = 0; i < nb_cases; ++i) totals[nb_controls + i] = (double)total_cases[i];
  // init_global_features (popstrat.cpp:270-311): null = [1, Z[0..npc), total], alt = null + k-mer slot
  std::vector<double> nul((size_t)n * fn, 0.0);
  ps->h_alt.assign((size_t)n * fa, 0.0);
  for (int i = 0; i < n; ++i)
  {
    nul[(size_t)i * fn + 0] = 1; ps->h_alt[(size_t)i * fa + 0] = 1;
    for (int z = 0; z < npc; ++z)
    {
      nul[(size_t)i * fn + z + 1] = Z[(size_t)i * z_cols + z];
      ps->h_alt[(size_t)i * fa + z + 1] = Z[(size_t)i * z_cols + z];
    }
    nul[(size_t)i * fn + 1 + npc] = totals[i];
    ps->h_alt[(size_t)i * fa + 1 + npc] = totals[i];
  }
  if (standardize)
  {
    // pop_strat_corrector::standardize (popstrat.cpp:327-370), quirks included
    // (the reference sizes stddev by ROWS and fills it by COLUMN, popstrat.cpp:330,349: with more columns than rows --
    // fewer than 4 + npc samples -- it writes past the vector's end, undefined behaviour there; here those entries
    // exist, and nothing reads them: the division below indexes rows.  tests/soak.py found this one: glibc's heap check)
    std::vector<double> means(fn, 0.0), stddev(std::max(n, fn), 0.0);
    for (int i = 0; i < n; ++i) for (int j = 0; j < fn; ++j) means[j] += nul[(size_t)i * fn + j];
    for (int j = 1; j < fn; ++j) means[j] /= fn;                                   // :342 (ncols, not nrows)
    for (int i = 0; i < n; ++i) for (int j = 1; j < fn; ++j)
      stddev[j] += std::pow(nul[(size_t)i * fn + j] - means[j], 2);                // :349 (indexed by column)
    for (int j = 1; j < fn; ++j) { stddev[j] /= n; stddev[j] = std::sqrt(stddev[j]); }
    for (int i = 0; i < n; ++i) for (int j = 1; j < fn; ++j)
      if (std::fabs(stddev[i]) > 1e-305)                                           // :361 (indexed by ROW)
      {
        nul[(size_t)i * fn + j] = (nul[(size_t)i * fn + j] - means[j]) / stddev[i];
        ps->h_alt[(size_t)i * fa + j] = (ps->h_alt[(size_t)i * fa + j] - means[j]) / stddev[i];
      }
  }
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&ps->d_alt), (size_t)n * fa * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ps->d_y), n * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ps->d_totals), n * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ps->d_null_model), (fn + 1) * sizeof(double));
  if (e == hipSuccess) e = hipMemcpy(ps->d_alt, ps->h_alt.data(), (size_t)n * fa * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(ps->d_y, Y, n * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(ps->d_totals, totals.data(), n * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess)
  {
    // null model on the device: same code, fn features read with the alt stride (the null
    // design is the first fn columns of the alt design, popstrat.cpp:279-311)
    irls_args A { ps->d_alt, fa, ps->d_y, ps->d_totals, nullptr, 0, 0, n, ps->max_iter };
    double* d_like = ps->d_null_model + fn;
    switch (fn)
    {
      case 2: launch_null<2>(A, ps->d_null_model, d_like, nullptr); break;
      case 3: launch_null<3>(A, ps->d_null_model, d_like, nullptr); break;
      case 4: launch_null<4>(A, ps->d_null_model, d_like, nullptr); break;
      case 5: launch_null<5>(A, ps->d_null_model, d_like, nullptr); break;
      case 6: launch_null<6>(A, ps->d_null_model, d_like, nullptr); break;
      case 7: launch_null<7>(A, ps->d_null_model, d_like, nullptr); break;
      case 8: launch_null<8>(A, ps->d_null_model, d_like, nullptr); break;
      case 9: launch_null<9>(A, ps->d_null_model, d_like, nullptr); break;
      case 10: launch_null<10>(A, ps->d_null_model, d_like, nullptr); break;
      case 11: launch_null<11>(A, ps->d_null_model, d_like, nullptr); break;
      default: launch_null<12>(A, ps->d_null_model, d_like, nullptr); break;
    }
    e = hipGetLastError();
    ps->h_null_model.assign(fn + 1, 0.0);
    if (e == hipSuccess) e = hipMemcpy(ps->h_null_model.data(), ps->d_null_model, (fn + 1) * sizeof(double), hipMemcpyDeviceToHost);
    ps->null_likelihood = ps->h_null_model[fn];
    ps->h_null_model.resize(fn);
  }
  if (e != hipSuccess)
  {
    int rc = kmd::hip_fail(e, "kmd_popstrat_create", __FILE__, __LINE__);
    kmd_popstrat_destroy(ps);
    return rc;
  }
  *out = ps;
  return KMD_OK;
}

int kmd_popstrat_destroy(kmd_popstrat* ps)
{
  if (!ps) return KMD_OK;
  if (ps->d_alt) (void)hipFree(ps->d_alt);
  if (ps->d_y) (void)hipFree(ps->d_y);
  if (ps->d_totals) (void)hipFree(ps->d_totals);
  if (ps->d_null_model) (void)hipFree(ps->d_null_model);
  delete ps;
  return KMD_OK;
}

// pop_strat_corrector::set_params' epsilon (popstrat.hpp:162-175: only a non-zero value replaces the
// default 1e-30): the bound below which |LLR| counts as zero (popstrat.hpp:321)
int kmd_popstrat_set_epsilon(kmd_popstrat* ps, double epsilon)
{
  KMD_REQUIRE(ps, "kmd_popstrat_set_epsilon: NULL");
  if (epsilon) ps->epsilon = epsilon;
  return KMD_OK;
}

int kmd_popstrat_info(const kmd_popstrat* ps, int* n_samples, int* n_features_alt, double* alt_global,
                      double* null_model, double* null_likelihood)
{
  KMD_REQUIRE(ps, "kmd_popstrat_info: NULL");
  if (n_samples) *n_samples = ps->n;
  if (n_features_alt) *n_features_alt = ps->f;
  if (alt_global) std::memcpy(alt_global, ps->h_alt.data(), ps->h_alt.size() * sizeof(double));
  if (null_model) std::memcpy(null_model, ps->h_null_model.data(), ps->h_null_model.size() * sizeof(double));
  if (null_likelihood) *null_likelihood = ps->null_likelihood;
  return KMD_OK;
}

int kmd_popstrat_apply(const kmd_popstrat* ps, const double* d_counts, int sample_major, size_t ld,
                       size_t n, double* d_pvalue, void* stream)
{
  KMD_REQUIRE(ps && (n == 0 || (d_counts && d_pvalue)), "kmd_popstrat_apply: NULL");
  if (n == 0) return KMD_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // which kernel (launch_apply): a lane per survivor for large batches of few features, a group of lanes per
  // survivor otherwise; the first reads counts sample-major (coalesced across survivors), the second
  // survivor-major (a group reads consecutive samples of its survivor)
  const bool group = use_group_kernel(ps->f, n);
  const double* counts = d_counts;
  double* d_t = nullptr;
  size_t si, ss;
  if (!sample_major)
  {
    KMD_REQUIRE(ld == 0 || ld == (size_t)ps->n, "kmd_popstrat_apply: survivor-major counts must be dense [n][S]");
    si = 1; ss = (size_t)ps->n;
    if (!group)
    {
      // [n][S] as gathered for KmerSign::m_counts_ratio -> sample-major so lanes read coalesced
      KMD_HIP(kmd::scratch_alloc(reinterpret_cast<void**>(&d_t), n * (size_t)ps->n * sizeof(double)));
      const size_t total = n * (size_t)ps->n;
      hipLaunchKernelGGL(k_transpose_counts, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_counts, n, ps->n, n, d_t);
      counts = d_t; si = n; ss = 1;
    }
  }
  else
  {
    KMD_REQUIRE(ld >= n, "kmd_popstrat_apply: ld < n");
    si = ld; ss = 1;
  }
  irls_args A { ps->d_alt, ps->f, ps->d_y, ps->d_totals, counts, si, ss, ps->n, ps->max_iter };
  switch (ps->f)
  {
    case 3: launch_apply<3>(A, n, ps->null_likelihood, ps->lg_half, ps->epsilon, d_pvalue, st); break;
    case 4: launch_apply<4>(A, n, ps->null_likelihood, ps->lg_half, ps->epsilon, d_pvalue, st); break;
    case 5: launch_apply<5>(A, n, ps->null_likelihood, ps->lg_half, ps->epsilon, d_pvalue, st); break;
    case 6: launch_apply<6>(A, n, ps->null_likelihood, ps->lg_half, ps->epsilon, d_pvalue, st); break;
    case 7: launch_apply<7>(A, n, ps->null_likelihood, ps->lg_half, ps->epsilon, d_pvalue, st); break;
    case 8: launch_apply<8>(A, n, ps->null_likelihood, ps->lg_half, ps->epsilon, d_pvalue, st); break;
    case 9: launch_apply<9>(A, n, ps->null_likelihood, ps->lg_half, ps->epsilon, d_pvalue, st); break;
    case 10: launch_apply<10>(A, n, ps->null_likelihood, ps->lg_half, ps->epsilon, d_pvalue, st); break;
    case 11: launch_apply<11>(A, n, ps->null_likelihood, ps->lg_half, ps->epsilon, d_pvalue, st); break;
    case 12: launch_apply<12>(A, n, ps->null_likelihood, ps->lg_half, ps->epsilon, d_pvalue, st); break;
    default: launch_apply<13>(A, n, ps->null_likelihood, ps->lg_half, ps->epsilon, d_pvalue, st); break;
  }
  hipError_t e = hipGetLastError();
  if (d_t)
  {
    hipError_t e2 = hipStreamSynchronize(st);
    kmd::scratch_free(d_t);
    if (e == hipSuccess) e = e2;
  }
  if (e != hipSuccess) return kmd::hip_fail(e, "kmd_popstrat_apply", __FILE__, __LINE__);
  return KMD_OK;
}

// ---- test hooks (include/kmdiff_hip_test.h): the reference's own linear-algebra vectors through the DEVICE routines of
// stage 2 -- lu_solve (the lane kernel's), group_lu_solve (the group kernel's), sigmoid_ref and the dot product of
// eta = X w -- tests/linear_test.cpp:29-31,80-151 are the only numbers of R9 the reference itself holds.
} // extern "C"
namespace {
template <int F>
__global__ void __launch_bounds__(64) k_test_linear(const double* __restrict__ a_in, const double* __restrict__ b_in, double* __restrict__ out)
{
  // out: [0, F*F) LU in place (lane) | [F*F, 2F*F) inverse row-major (lane) | [2F*F, 2F*F+F) w (lane) | status (lane)
  //      | the same four of the group routine
  using G = group_shape<F>;
  __shared__ double s_h[F * F], s_b[F], s_xc[F * F], s_w[F];
  const int lane = (int)threadIdx.x;
  double* lane_out = out;
  double* grp_out = out + 2 * F * F + F + 1;
  if (lane == 0)
  {
    double a[F][F], b[F], w[F];
    for (int c = 0; c < F; ++c)
    {
      // column c of the inverse: the routine's w for b = e_c (w[p] = sum_c inv[p][c] b[c]: the other terms are +-0)
      for (int i = 0; i < F; ++i) { b[i] = i == c ? 1.0 : 0.0; for (int j = 0; j < F; ++j) a[i][j] = a_in[i * F + j]; }
      (void)lu_solve<F>(a, b, w);
      for (int p = 0; p < F; ++p) lane_out[F * F + p * F + c] = w[p];
    }
    for (int i = 0; i < F; ++i) { b[i] = b_in[i]; for (int j = 0; j < F; ++j) a[i][j] = a_in[i * F + j]; }
    const int st = lu_solve<F>(a, b, w);
    for (int i = 0; i < F; ++i) for (int j = 0; j < F; ++j) lane_out[i * F + j] = a[i][j];
    for (int p = 0; p < F; ++p) lane_out[2 * F * F + p] = w[p];
    lane_out[2 * F * F + F] = (double)st;
  }
  // the group routine: lanes [0, L) of the wave are one group
  for (int i = lane; i < F * F; i += 64) s_h[i] = a_in[i];
  if (lane < F) s_b[lane] = b_in[lane];
  group_sync();
  bool active = true;
  if (lane < G::L) group_lu_solve<F>(s_h, s_b, s_xc, s_w, active, lane);
  group_sync();
  if (lane == 0)
  {
    for (int i = 0; i < F * F; ++i) grp_out[i] = s_h[i];
    for (int c = 0; c < F; ++c) for (int p = 0; p < F; ++p) grp_out[F * F + p * F + c] = s_xc[c * F + p];
    for (int p = 0; p < F; ++p) grp_out[2 * F * F + p] = active ? s_w[p] : 0.0;
    grp_out[2 * F * F + F] = active ? 0.0 : 1.0;
  }
}

__global__ void __launch_bounds__(64) k_test_sigmoid(const double* __restrict__ x, size_t n, double* __restrict__ out)
{
  const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i < n) out[i] = sigmoid_ref(x[i]);
}

// glm_irls (linear_model.cpp:297-410) on a design of the caller's, through the lane kernel's loop (irls_fit: every column
// from the design, as the null model is fitted)
template <int F>
__global__ void __launch_bounds__(64) k_test_irls_lane(irls_args A, double* __restrict__ out_w, int* __restrict__ out_iter)
{
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double model[F];
  int it = 0;
  irls_fit<F, false, false>(A, nullptr, 0, model, &it);
#pragma unroll
  for (int j = 0; j < F; ++j) out_w[j] = model[j];
  *out_iter = it;
}

// predict() (linear_model.cpp:197-211): sigmoid of the dot product, summed in index order as irls_fit forms eta
__global__ void __launch_bounds__(64) k_test_predict(const double* __restrict__ w, const double* __restrict__ x, int n, double* __restrict__ out)
{
  if (threadIdx.x != 0) return;
  double eta = 0;
  for (int j = 0; j < n; ++j) eta += x[j] * w[j];
  out[0] = eta; out[1] = sigmoid_ref(eta);
}
} // namespace
extern "C" {

int kmd_test_popstrat_linear(int F, const double* a, const double* b, double* lane_out, double* group_out)
{
  KMD_REQUIRE(a && b && lane_out && group_out, "kmd_test_popstrat_linear: NULL");
  KMD_REQUIRE(F >= 2 && F <= 13, "kmd_test_popstrat_linear: F");
  const size_t each = 2 * (size_t)F * F + F + 1;
  double *d_a = nullptr, *d_b = nullptr, *d_o = nullptr;
  KMD_HIP(hipMalloc(reinterpret_cast<void**>(&d_a), (size_t)F * F * 8));
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_b), (size_t)F * 8);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_o), 2 * each * 8);
  if (e == hipSuccess) e = hipMemcpy(d_a, a, (size_t)F * F * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_b, b, (size_t)F * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess)
  {
    switch (F)
    {
#define KMD_TL(N) case N: hipLaunchKernelGGL((k_test_linear<N>), dim3(1), dim3(64), 0, nullptr, d_a, d_b, d_o); break;
      KMD_TL(2) KMD_TL(3) KMD_TL(4) KMD_TL(5) KMD_TL(6) KMD_TL(7) KMD_TL(8) KMD_TL(9) KMD_TL(10) KMD_TL(11) KMD_TL(12) KMD_TL(13)
#undef KMD_TL
    }
    e = hipGetLastError();
  }
  std::vector<double> h(2 * each);
  if (e == hipSuccess) e = hipMemcpy(h.data(), d_o, 2 * each * 8, hipMemcpyDeviceToHost);
  (void)hipFree(d_a); (void)hipFree(d_b); (void)hipFree(d_o);
  if (e != hipSuccess) return kmd::hip_fail(e, "kmd_test_popstrat_linear", __FILE__, __LINE__);
  std::copy(h.begin(), h.begin() + each, lane_out);
  std::copy(h.begin() + each, h.end(), group_out);
  return KMD_OK;
}

// The IRLS loop itself on a design of the caller's: X (n x f row-major), y (n) -> weights and glm_irls's iteration count
// from (a) the lane kernel's loop and (b) the group kernel's.  The group kernel takes its last column as count / total
// (popstrat.hpp:254-257): it is handed counts = that column and totals of 1.0 -- the division is exact.
int kmd_test_popstrat_irls(const double* X, const double* y, int n, int f, int max_iter, double* w_lane, int* iters_lane,
                           double* w_group, int* iters_group)
{
  KMD_REQUIRE(X && y && w_lane && iters_lane && w_group && iters_group, "kmd_test_popstrat_irls: NULL");
  KMD_REQUIRE(n > 0 && f >= 2 && f <= 13 && max_iter > 0, "kmd_test_popstrat_irls: arguments");
  std::vector<double> last((size_t)n), ones((size_t)n, 1.0);
  for (int i = 0; i < n; ++i) last[(size_t)i] = X[(size_t)i * f + f - 1];
  double *d_x = nullptr, *d_y = nullptr, *d_c = nullptr, *d_t = nullptr, *d_o = nullptr;
  int* d_it = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_x), (size_t)n * f * 8);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_y), (size_t)n * 8);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_c), (size_t)n * 8);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_t), (size_t)n * 8);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_o), (size_t)(2 * f + 1) * 8);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_it), 2 * sizeof(int));
  if (e == hipSuccess) e = hipMemcpy(d_x, X, (size_t)n * f * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_y, y, (size_t)n * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_c, last.data(), (size_t)n * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_t, ones.data(), (size_t)n * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess)
  {
    // one survivor, its counts survivor-major: counts[i * si + v * ss] with (si, ss) = (1, n)
    const irls_args A { d_x, f, d_y, d_t, d_c, 1, (size_t)n, n, max_iter };
    switch (f)
    {
#define KMD_TI(N) case N: \
        hipLaunchKernelGGL((k_test_irls_lane<N>), dim3(1), dim3(64), 0, nullptr, A, d_o, d_it); \
        hipLaunchKernelGGL((k_popstrat_group<N>), dim3(1), dim3(64), (size_t)(64 / group_shape<N>::L) * group_shape<N>::kLds * sizeof(double), nullptr, \
                           A, (size_t)1, 1.0, 0.0, 1e-30, d_o + 2 * f, d_o + f, d_it + 1); break;
      KMD_TI(2) KMD_TI(3) KMD_TI(4) KMD_TI(5) KMD_TI(6) KMD_TI(7) KMD_TI(8) KMD_TI(9) KMD_TI(10) KMD_TI(11) KMD_TI(12) KMD_TI(13)
#undef KMD_TI
    }
    e = hipGetLastError();
  }
  std::vector<double> h((size_t)(2 * f + 1));
  int h_it[2] = { 0, 0 };
  if (e == hipSuccess) e = hipMemcpy(h.data(), d_o, h.size() * 8, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(h_it, d_it, sizeof h_it, hipMemcpyDeviceToHost);
  (void)hipFree(d_x); (void)hipFree(d_y); (void)hipFree(d_c); (void)hipFree(d_t); (void)hipFree(d_o); (void)hipFree(d_it);
  if (e != hipSuccess) return kmd::hip_fail(e, "kmd_test_popstrat_irls", __FILE__, __LINE__);
  std::copy(h.begin(), h.begin() + f, w_lane);
  std::copy(h.begin() + f, h.begin() + 2 * f, w_group);
  *iters_lane = h_it[0]; *iters_group = h_it[1];
  return KMD_OK;
}

int kmd_test_popstrat_sigmoid(const double* x, size_t n, double* out)
{
  KMD_REQUIRE(x && out, "kmd_test_popstrat_sigmoid: NULL");
  if (n == 0) return KMD_OK;
  double *d_x = nullptr, *d_o = nullptr;
  KMD_HIP(hipMalloc(reinterpret_cast<void**>(&d_x), n * 8));
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_o), n * 8);
  if (e == hipSuccess) e = hipMemcpy(d_x, x, n * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) { hipLaunchKernelGGL(k_test_sigmoid, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, nullptr, d_x, n, d_o); e = hipGetLastError(); }
  if (e == hipSuccess) e = hipMemcpy(out, d_o, n * 8, hipMemcpyDeviceToHost);
  (void)hipFree(d_x); (void)hipFree(d_o);
  if (e != hipSuccess) return kmd::hip_fail(e, "kmd_test_popstrat_sigmoid", __FILE__, __LINE__);
  return KMD_OK;
}

int kmd_test_popstrat_predict(const double* w, const double* x, int n, double* eta_out, double* p_out)
{
  KMD_REQUIRE(w && x && eta_out && p_out && n > 0, "kmd_test_popstrat_predict: arguments");
  double *d_w = nullptr, *d_x = nullptr, *d_o = nullptr;
  KMD_HIP(hipMalloc(reinterpret_cast<void**>(&d_w), (size_t)n * 8));
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_x), (size_t)n * 8);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_o), 16);
  if (e == hipSuccess) e = hipMemcpy(d_w, w, (size_t)n * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_x, x, (size_t)n * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) { hipLaunchKernelGGL(k_test_predict, dim3(1), dim3(64), 0, nullptr, d_w, d_x, n, d_o); e = hipGetLastError(); }
  double h[2] = { 0, 0 };
  if (e == hipSuccess) e = hipMemcpy(h, d_o, 16, hipMemcpyDeviceToHost);
  (void)hipFree(d_w); (void)hipFree(d_x); (void)hipFree(d_o);
  if (e != hipSuccess) return kmd::hip_fail(e, "kmd_test_popstrat_predict", __FILE__, __LINE__);
  *eta_out = h[0]; *p_out = h[1];
  return KMD_OK;
}

} // extern "C"
