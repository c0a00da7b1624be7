// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE, not product code.
//
// C-ABI wrapper around the pieces of the reference (tlemane/kmdiff v1.1.0) that compile
// from their own sources with no stand-in headers:
//   * src/log_factorial_table.cpp            (LogFactorialTable, R5)
//   * src/corrector.cpp, src/correction.cpp  (ICorrector family, R8)
//   * thirdparty/alglib/src/{ap,alglibinternal,specialfunctions}.cpp
//                                            (alglib::chisquarecdistribution, R6)
// The sources are compiled where they lie under /root/reference by oracle/Makefile; the
// output goes to oracle/_ref/libkmdiff_ref.so (git-ignored). Nothing from the reference is
// copied into this repository.
//
// include/kmdiff/model.hpp (PoissonLikelihood::process, R4) cannot be compiled here: it
// includes <kmtricks/utils.hpp> and <spdlog/spdlog.h>, which are empty submodules in
// /root/reference. kmdref_poisson_process below therefore RESTATES its 20 lines of glue
// (model.hpp:133-176) around the REAL reference callees (LogFactorialTable::operator[]
// and alglib::chisquarecdistribution), so the only restated arithmetic is the sum and the
// four-term likelihood expression.
#include <cstddef>
#include <cstdint>
#include <cmath>
#include <memory>
#include <vector>

#include <kmdiff/log_factorial_table.hpp>
#include <kmdiff/corrector.hpp>
#include <specialfunctions.h>

extern "C" {

// alglib::chisquarecdistribution (specialfunctions.cpp:2750-2770); NaN on alglib exception.
double kmdref_chisqc(double v, double x)
{
  try { return alglib::chisquarecdistribution(v, x); }
  catch (...) { return std::nan(""); }
}

// LogFactorialTable(size)[idx[i]] (log_factorial_table.hpp:14-18, .cpp:5-22)
void kmdref_lf(size_t size, const uint64_t* idx, size_t n, double* out)
{
  kmdiff::LogFactorialTable t(size);
  for (size_t i = 0; i < n; i++) out[i] = t[idx[i]];
}

// make_corrector(type, threshold, total) (corrector.cpp:101-116); type = CorrectionType value
void* kmdref_corrector_new(int type, double threshold, uint64_t total)
{
  auto sp = kmdiff::make_corrector(static_cast<kmdiff::CorrectionType>(type), threshold, total);
  return new std::shared_ptr<kmdiff::ICorrector>(sp);
}
int kmdref_corrector_apply(void* h, double p)
{
  return (*static_cast<std::shared_ptr<kmdiff::ICorrector>*>(h))->apply(p) ? 1 : 0;
}
void kmdref_corrector_free(void* h)
{
  delete static_cast<std::shared_ptr<kmdiff::ICorrector>*>(h);
}

// Glue of PoissonLikelihood::process (model.hpp:142-176) around the real reference callees.
// counts: n_rows x (nc+nk) row-major u32. sign: 0 CONTROL, 1 CASE, 2 NO (kmer.hpp:33-38).
struct kmdref_model
{
  kmdiff::LogFactorialTable lf;
  size_t nc, nk, sum_controls, sum_cases;
  kmdref_model(size_t preload, size_t nc_, size_t nk_, size_t tc, size_t tk)
    : lf(preload), nc(nc_), nk(nk_), sum_controls(tc), sum_cases(tk) {}
  double poisson_prob(int k, double lambda)
  {
    if (lambda <= 0) return 0;
    if (k < 0) k = 0;
    return (-lambda + (k * log(lambda) - lf[k]));
  }
};

void* kmdref_model_new(size_t preload, size_t nc, size_t nk, uint64_t tc, uint64_t tk)
{
  return new kmdref_model(preload, nc, nk, tc, tk);
}
void kmdref_model_free(void* h) { delete static_cast<kmdref_model*>(h); }

void kmdref_model_process(void* h, const uint32_t* counts, size_t n_rows,
                          double* p, int32_t* sign, double* mean_ctrl, double* mean_case)
{
  kmdref_model& m = *static_cast<kmdref_model*>(h);
  const size_t S = m.nc + m.nk;
  for (size_t r = 0; r < n_rows; r++)
  {
    const uint32_t* row = counts + r * S;
    double mean_control = 0, mean_cas = 0;
    for (size_t i = 0; i < m.nc; i++) mean_control += row[i];
    for (size_t i = 0; i < m.nk; i++) mean_cas += row[m.nc + i];
    double mean = (mean_control + mean_cas) / static_cast<double>(m.sum_controls + m.sum_cases);
    double null_h = 0, alt_h = 0;
    alt_h += m.poisson_prob(mean_control, mean_control);
    alt_h += m.poisson_prob(mean_cas, mean_cas);
    null_h += m.poisson_prob(mean_control, mean * m.sum_controls);
    null_h += m.poisson_prob(mean_cas, mean * m.sum_cases);
    double lr = alt_h - null_h;
    if (lr < 0) lr = 0;
    p[r] = alglib::chisquarecdistribution(1, 2 * lr);
    mean_control = mean_control * m.sum_cases / m.sum_controls;
    sign[r] = (mean_control < mean_cas) ? 1 : (mean_control > mean_cas) ? 0 : 2;
    mean_ctrl[r] = mean_control;
    mean_case[r] = mean_cas;
  }
}

} // extern "C"
