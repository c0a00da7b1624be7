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
  const double* counts;   // [n_samples][ld] (sample-major): counts[i*ld + survivor]
  size_t ld;
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
__device__ __forceinline__ void irls_fit(const irls_args& A, const double* s_d, size_t surv, double (&weight)[F])
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
      double c_next = KMER ? A.counts[surv] : 0.0;
      for (int i = 0; i < A.n; ++i)
      {
        const double c = c_next;
        if (KMER) c_next = A.counts[(size_t)(i + 1 < A.n ? i + 1 : i) * A.ld + surv];
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
}

// pop_strat_corrector::apply(KmerSign&) (popstrat.hpp:249-333) for one survivor per lane
template <int F, bool LDSD>
__global__ void __launch_bounds__(64) k_popstrat_apply(irls_args A, size_t n_surv, double null_likelihood,
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
    double c_next = A.counts[surv];
    for (int i = 0; i < A.n; ++i)
    {
      const double c = c_next;
      c_next = A.counts[(size_t)(i + 1 < A.n ? i + 1 : i) * A.ld + surv];
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

template <int F>
void launch_apply(const irls_args& A, size_t n, double null_like, double lg_half, double epsilon, double* d_p, hipStream_t st)
{
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
    std::vector<double> means(fn, 0.0), stddev(n, 0.0);
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
    irls_args A { ps->d_alt, fa, ps->d_y, ps->d_totals, nullptr, 0, n, ps->max_iter };
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
  const double* counts = d_counts;
  double* d_t = nullptr;
  if (!sample_major)
  {
    // [n][S] as gathered for KmerSign::m_counts_ratio -> sample-major so lanes read coalesced
    KMD_REQUIRE(ld == 0 || ld == (size_t)ps->n, "kmd_popstrat_apply: survivor-major counts must be dense [n][S]");
    KMD_HIP(kmd::scratch_alloc(reinterpret_cast<void**>(&d_t), n * (size_t)ps->n * sizeof(double)));
    const size_t total = n * (size_t)ps->n;
    hipLaunchKernelGGL(k_transpose_counts, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_counts, n, ps->n, n, d_t);
    counts = d_t; ld = n;
  }
  else KMD_REQUIRE(ld >= n, "kmd_popstrat_apply: ld < n");
  irls_args A { ps->d_alt, ps->f, ps->d_y, ps->d_totals, counts, ld, ps->n, ps->max_iter };
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

} // extern "C"
