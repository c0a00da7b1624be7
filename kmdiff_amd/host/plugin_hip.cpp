// plugin_hip.cpp -> libkmdiff_hip_model.so: the Poisson model of libkmdiff_hip.so behind the
// reference's plugin loader (`kmdiff diff --cmodel libkmdiff_hip_model.so --config "..."`,
// cmd/diff.hpp:268-271, model_manager.hpp:58-63).
//
// This is the COMPATIBILITY surface, not the fast path: IModel::process is called once per
// row from the merge threads (merge.hpp:73), so each call pushes one row through
// kmd_poisson_process -- exact, but it pays a kernel launch per row.  The batch observer of
// INTEGRATION.md section 2 is how the kernels are meant to be fed.
//
// --config "controls=N;cases=M;total_controls=a,b,..;total_cases=c,d,..[;log_factorial=10000]"
// (a plugin model does not receive the per-sample totals from the host, imodel.hpp:34).
#include <mutex>
#include <sstream>
#include <stdexcept>

#include "../../include/kmdiff_hip.h"
#include "imodel_abi.hpp"

namespace {

std::vector<uint64_t> parse_list(const std::string& v)
{
  std::vector<uint64_t> out;
  std::stringstream ss(v);
  for (std::string item; std::getline(ss, item, ',');) if (!item.empty()) out.push_back(std::stoull(item));
  return out;
}

template <size_t MAX_C>
class PoissonLikelihoodHip : public kmdiff::IModel<MAX_C>
{
  using count_type = typename kmdiff::IModel<MAX_C>::count_type;
 public:
  ~PoissonLikelihoodHip() override
  {
    if (m_model) kmd_model_destroy(m_model);
    for (void* p : { m_d_row, m_d_p, m_d_sign, m_d_mc, m_d_mk }) if (p) kmd_free(p);
  }

  void configure(const std::string& config) override
  {
    size_t nc = 0, nk = 0, log_size = 10000;
    std::vector<uint64_t> tc, tk;
    std::stringstream ss(config);
    for (std::string kv; std::getline(ss, kv, ';');)
    {
      const auto eq = kv.find('=');
      if (eq == std::string::npos) continue;
      const std::string k = kv.substr(0, eq), v = kv.substr(eq + 1);
      if (k == "controls") nc = std::stoull(v);
      else if (k == "cases") nk = std::stoull(v);
      else if (k == "total_controls") tc = parse_list(v);
      else if (k == "total_cases") tk = parse_list(v);
      else if (k == "log_factorial") log_size = std::stoull(v);
    }
    if (!nc || !nk || tc.size() != nc || tk.size() != nk)
      throw std::runtime_error("kmdiff_hip model: --config needs controls, cases, total_controls, total_cases");
    int ndev = 0;
    if (kmd_device_count(&ndev) != KMD_OK || ndev < 1) throw std::runtime_error("kmdiff_hip model: no HIP device");
    if (kmd_model_create(&m_model, (int)nc, (int)nk, tc.data(), tk.data(), log_size) != KMD_OK)
      throw std::runtime_error(kmd_last_error());
    m_nc = nc; m_nk = nk;
    bool ok = kmd_malloc(&m_d_row, (nc + nk) * sizeof(count_type)) == KMD_OK && kmd_malloc(&m_d_p, 8) == KMD_OK &&
              kmd_malloc(&m_d_sign, 4) == KMD_OK && kmd_malloc(&m_d_mc, 8) == KMD_OK && kmd_malloc(&m_d_mk, 8) == KMD_OK;
    if (!ok) throw std::runtime_error(kmd_last_error());
  }

  kmdiff::model_ret_t process(const kmdiff::Range<count_type>& controls,
                              const kmdiff::Range<count_type>& cases) override
  {
    if (!m_model) throw std::runtime_error("kmdiff_hip model: configure() was not called");
    if (controls.size() != m_nc || cases.size() != m_nk) throw std::runtime_error("kmdiff_hip model: sample count mismatch");
    std::lock_guard<std::mutex> lock(m_mu);            // one instance is shared by all merge threads (merge.hpp:418)
    std::vector<count_type> row(m_nc + m_nk);
    for (size_t i = 0; i < m_nc; ++i) row[i] = controls[i];
    for (size_t i = 0; i < m_nk; ++i) row[m_nc + i] = cases[i];
    double p = 0, mc = 0, mk = 0; int32_t sign = 2;
    kmd_tile t { m_d_row, (int)sizeof(count_type), KMD_LAYOUT_ROWS, m_nc + m_nk, nullptr, nullptr, 1, 0 };
    int rc = kmd_memcpy_h2d(m_d_row, row.data(), row.size() * sizeof(count_type), nullptr);
    if (rc == KMD_OK) rc = kmd_poisson_process(m_model, &t, (double*)m_d_p, (int32_t*)m_d_sign, (double*)m_d_mc, (double*)m_d_mk, nullptr);
    // the reference decides `p <= threshold` on this number itself (merge.hpp:78): give it glibc's bits
    if (rc == KMD_OK) rc = kmd_pvalues_refine(m_model, 1, (const double*)m_d_mc, (const double*)m_d_mk, (double*)m_d_p, nullptr);
    if (rc == KMD_OK) rc = kmd_memcpy_d2h(&p, m_d_p, 8, nullptr);
    if (rc == KMD_OK) rc = kmd_memcpy_d2h(&sign, m_d_sign, 4, nullptr);
    if (rc == KMD_OK) rc = kmd_memcpy_d2h(&mc, m_d_mc, 8, nullptr);
    if (rc == KMD_OK) rc = kmd_memcpy_d2h(&mk, m_d_mk, 8, nullptr);
    if (rc != KMD_OK) throw std::runtime_error(kmd_last_error());
    return std::make_tuple(p, static_cast<kmdiff::Significance>(sign), mc, mk);
  }

 private:
  kmd_model* m_model = nullptr;
  size_t m_nc = 0, m_nk = 0;
  void *m_d_row = nullptr, *m_d_p = nullptr, *m_d_sign = nullptr, *m_d_mc = nullptr, *m_d_mk = nullptr;
  std::mutex m_mu;
};

} // namespace

// plugins/ex_model.cpp:29-32 -- C linkage, C++ types (same compiler ABI as the host required)
extern "C" std::string plugin_name() { return "kmdiff_hip_poisson"; }
extern "C" kmdiff::IModel<kmdiff::maxc8>* create8() { return new PoissonLikelihoodHip<kmdiff::maxc8>(); }
extern "C" kmdiff::IModel<kmdiff::maxc16>* create16() { return new PoissonLikelihoodHip<kmdiff::maxc16>(); }
extern "C" kmdiff::IModel<kmdiff::maxc32>* create32() { return new PoissonLikelihoodHip<kmdiff::maxc32>(); }
