// kmd_api.hip -- C-ABI plumbing of libkmdiff_hip.so: error reporting, device memory and
// event helpers, the model object, the synthetic-matrix generator and the small support
// kernels (column sums, copy probe).
#include "kmd_internal.h"
#include "../../include/kmdiff_hip_test.h"
#include "kmd_math.h"
#include "../../include/kmdiff_synth_tables.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <vector>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

namespace {
thread_local std::string g_last_error;
}

void kmd::set_error(const std::string& msg) { g_last_error = msg; }

namespace {
struct scratch_cache
{
  std::mutex mu;
  std::multimap<std::pair<int, size_t>, void*> parked;      // (device, size class) -> block
  std::map<void*, std::pair<int, size_t>> live;             // block -> (device, size class)
  size_t parked_bytes = 0;
};
scratch_cache& cache() { static scratch_cache c; return c; }
// Parked beyond this, blocks are freed for real: an eighth of the device's memory (36 GB of an MI355X's 288), at least
// 8 GB.  (Six whole configs[2] partitions in flight hold ~2 GB of scratch each: with a fixed 8 GB every free past the
// fourth partition was a hipFree -- a device-wide synchronisation -- and every allocation a hipMalloc: 44 ms per
// partition instead of 2.5.)
size_t max_parked()
{
  static const size_t limit = []
  {
    if (const char* e = std::getenv("KMD_SCRATCH_PARK_MB")) return (size_t)std::strtoull(e, nullptr, 10) << 20;     // dev: a small cache (evictions all the time)
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); total_b = 0; }
    return std::max<size_t>((size_t)8 << 30, total_b / 8);
  }();
  return limit;
}
}

hipError_t kmd::scratch_alloc(void** p, size_t bytes)
{
  // size classes: powers of two up to 64 MB; above, eighths of an octave (at most 12.5 % over the request --
  // a plain power of two nearly doubled the footprint of the sort path's n x 8-byte arrays -- and still few
  // enough classes that the blocks of one partition are found again by the next)
  size_t cls = 256;
  while (cls < bytes && cls < ((size_t)64 << 20)) cls <<= 1;
  if (cls < bytes)
  {
    size_t octave = cls;
    while ((octave << 1) <= bytes) octave <<= 1;          // largest power of two <= bytes
    const size_t step = octave >> 3;
    cls = (bytes + step - 1) / step * step;
  }
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  scratch_cache& c = cache();
  {
    std::lock_guard<std::mutex> lock(c.mu);
    auto it = c.parked.find({ dev, cls });
    if (it != c.parked.end())
    {
      *p = it->second;
      c.parked.erase(it);
      c.parked_bytes -= cls;
      c.live[*p] = { dev, cls };
      return hipSuccess;
    }
  }
  e = hipMalloc(p, cls);
  if (e != hipSuccess)
  {
    kmd::scratch_release_all();                              // give parked memory back and retry once
    e = hipMalloc(p, cls);
    if (e != hipSuccess) return e;
  }
  std::lock_guard<std::mutex> lock(c.mu);
  c.live[*p] = { dev, cls };
  return hipSuccess;
}

void kmd::scratch_free(void* p)
{
  if (!p) return;
  scratch_cache& c = cache();
  std::unique_lock<std::mutex> lock(c.mu);
  auto it = c.live.find(p);
  if (it == c.live.end()) { lock.unlock(); (void)hipFree(p); return; }
  const auto key = it->second;
  c.live.erase(it);
  const size_t limit = max_parked();
  if (key.second > limit) { lock.unlock(); (void)hipFree(p); return; }
  if (c.parked_bytes + key.second > limit)
  {
    // make room: the parked blocks go (largest classes first: they are what filled the cache), this one stays -- it is
    // the size the caller works with NOW (a cache stuck full of an earlier job's large blocks turned every later
    // small allocation into a hipMalloc / hipFree pair)
    std::vector<void*> out;
    for (auto it = c.parked.end(); it != c.parked.begin() && c.parked_bytes + key.second > limit;)
    {
      --it;
      out.push_back(it->second);
      c.parked_bytes -= it->first.second;
      it = c.parked.erase(it);
    }
    c.parked.insert({ key, p });
    c.parked_bytes += key.second;
    lock.unlock();
    for (void* q : out) (void)hipFree(q);
    return;
  }
  c.parked.insert({ key, p });
  c.parked_bytes += key.second;
}

void kmd::scratch_release_all()
{
  scratch_cache& c = cache();
  std::lock_guard<std::mutex> lock(c.mu);
  for (auto& kv : c.parked) (void)hipFree(kv.second);
  c.parked.clear();
  c.parked_bytes = 0;
}

int kmd::hip_fail(hipError_t e, const char* what, const char* file, int line)
{
  char buf[512];
  std::snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
  g_last_error = buf;
  return e == hipErrorOutOfMemory ? KMD_E_NOMEM : KMD_E_HIP;
}

extern "C" {

const char* kmd_status_string(int status)
{
  switch (status)
  {
    case KMD_OK: return "ok";
    case KMD_E_INVALID: return "invalid argument";
    case KMD_E_HIP: return "HIP runtime error";
    case KMD_E_NO_DEVICE: return "no HIP device";
    case KMD_E_OVERFLOW: return "survivor capacity exceeded";
    case KMD_E_NOMEM: return "out of device memory";
    default: return "unknown status";
  }
}

const char* kmd_last_error(void) { return g_last_error.c_str(); }
int kmd_abi_version(void) { return KMD_ABI_VERSION; }

// test hooks (host side of kmd_ddmath.h / kmd_math.h, the code the device runs for rows within 1e-8 of the
// threshold): tests/test_rounded_math.py holds them against mpmath
double kmd_test_log_rounded(double x) { return kmd::libm_rounded::log(x); }
double kmd_test_exp_rounded(double x) { return kmd::libm_rounded::exp(x); }
double kmd_test_igamc_half_rounded(double x) { return kmd::igamc_half<kmd::libm_rounded>(x, kmd::lngamma_half_host()); }
// the p-value of a row with these two count sums the way the device decides a near-threshold row (and kmd_pvalues_refine
// rewrites a survivor's): table terms as the model holds them -- beyond the table the reference's running sum
// (log_factorial_table.cpp:13-22), for sums below 2^20 -- and the four libm calls correctly rounded.  -1: not such a row
double kmd_test_row_pvalue_rounded(const kmd_model* m, uint64_t sum_c, uint64_t sum_k)
{
  if (!m || ((sum_c >= m->lf_n || sum_k >= m->lf_n) && (sum_c >= (1ull << 20) || sum_k >= (1ull << 20)))) return -1.0;
  auto lf_of = [&](uint64_t k) {
    if (k < m->lf_n) return m->h_lf[k];
    double res = 0;
    for (; k > 1; --k) res += kmd::libm_rounded::log((double)k);
    return res;
  };
  auto log_of = [&](uint64_t k) { return k < m->lf_n ? (k ? ::log((double)k) : 0.0) : kmd::libm_rounded::log((double)k); };   // the table's second column
  const double lr = kmd::lr_from_sums<kmd::libm_rounded>(sum_c, sum_k, lf_of(sum_c), lf_of(sum_k), log_of(sum_c), log_of(sum_k), m->dT, m->dTc, m->dTk);
  return kmd::igamc_half<kmd::libm_rounded>(lr, m->lg_half);
}

int kmd_device_count(int* n)
{
  KMD_REQUIRE(n, "kmd_device_count: NULL");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) { *n = 0; (void)hipGetLastError(); kmd::set_error("no HIP device"); return KMD_E_NO_DEVICE; }
  *n = c;
  return c > 0 ? KMD_OK : KMD_E_NO_DEVICE;
}

int kmd_set_device(int device) { KMD_HIP(hipSetDevice(device)); return KMD_OK; }

int kmd_device_name(char* buf, size_t len)
{
  KMD_REQUIRE(buf && len, "kmd_device_name: NULL");
  int dev = 0;
  KMD_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  KMD_HIP(hipGetDeviceProperties(&prop, dev));
  std::snprintf(buf, len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return KMD_OK;
}

int kmd_malloc(void** d_ptr, size_t bytes)
{
  KMD_REQUIRE(d_ptr, "kmd_malloc: NULL");
  KMD_HIP(hipMalloc(d_ptr, bytes ? bytes : 1));
  return KMD_OK;
}
int kmd_free(void* d_ptr) { if (d_ptr) KMD_HIP(hipFree(d_ptr)); return KMD_OK; }
int kmd_malloc_host(void** h_ptr, size_t bytes)
{
  KMD_REQUIRE(h_ptr, "kmd_malloc_host: NULL");
  KMD_HIP(hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocPortable));   // usable from any device of the process
  return KMD_OK;
}
int kmd_free_host(void* h_ptr) { if (h_ptr) KMD_HIP(hipHostFree(h_ptr)); return KMD_OK; }
int kmd_memcpy_h2d(void* d_dst, const void* src, size_t bytes, void* stream)
{
  if (!bytes) return KMD_OK;
  KMD_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
  KMD_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  return KMD_OK;
}
int kmd_stream_create(void** stream)
{
  KMD_REQUIRE(stream, "kmd_stream_create: NULL argument");
  hipStream_t st = nullptr;
  KMD_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  *stream = st;
  return KMD_OK;
}
int kmd_stream_destroy(void* stream)
{
  if (stream)
  {
    KMD_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    kmd::near_list_forget(static_cast<hipStream_t>(stream));     // (a later stream may get the same handle)
    kmd::unpack_tables_forget(static_cast<hipStream_t>(stream), false);
    KMD_HIP(hipStreamDestroy(static_cast<hipStream_t>(stream)));
  }
  return KMD_OK;
}
int kmd_memcpy_h2d_async(void* d_dst, const void* src, size_t bytes, void* stream)
{
  if (!bytes) return KMD_OK;
  KMD_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
  return KMD_OK;
}
int kmd_memcpy_d2h(void* dst, const void* d_src, size_t bytes, void* stream)
{
  if (!bytes) return KMD_OK;
  KMD_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
  KMD_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  return KMD_OK;
}
int kmd_memset(void* d_dst, int value, size_t bytes, void* stream)
{
  if (!bytes) return KMD_OK;
  KMD_HIP(hipMemsetAsync(d_dst, value, bytes, static_cast<hipStream_t>(stream)));
  return KMD_OK;
}
int kmd_release_cache(void) { kmd::scratch_release_all(); kmd::near_lists_release(); kmd::unpack_tables_forget(nullptr, true); return KMD_OK; }
int kmd_stream_sync(void* stream) { KMD_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream))); return KMD_OK; }

int kmd_event_create(void** ev)
{
  KMD_REQUIRE(ev, "kmd_event_create: NULL");
  hipEvent_t e;
  KMD_HIP(hipEventCreate(&e));
  *ev = e;
  return KMD_OK;
}
int kmd_event_destroy(void* ev) { if (ev) KMD_HIP(hipEventDestroy(static_cast<hipEvent_t>(ev))); return KMD_OK; }
int kmd_event_record(void* ev, void* stream)
{
  KMD_HIP(hipEventRecord(static_cast<hipEvent_t>(ev), static_cast<hipStream_t>(stream)));
  return KMD_OK;
}
int kmd_stream_wait_event(void* stream, void* ev)
{
  KMD_REQUIRE(ev, "kmd_stream_wait_event: NULL event");
  KMD_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(ev), 0));
  return KMD_OK;
}
int kmd_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms)
{
  KMD_REQUIRE(ms, "kmd_event_elapsed_ms: NULL");
  KMD_HIP(hipEventSynchronize(static_cast<hipEvent_t>(ev_stop)));
  KMD_HIP(hipEventElapsedTime(ms, static_cast<hipEvent_t>(ev_start), static_cast<hipEvent_t>(ev_stop)));
  return KMD_OK;
}

// ---- model ---------------------------------------------------------------------------------

int kmd_model_create(kmd_model** out, int nb_controls, int nb_cases,
                     const uint64_t* total_controls, const uint64_t* total_cases,
                     size_t log_factorial_size)
{
  KMD_REQUIRE(out, "kmd_model_create: NULL out");
  KMD_REQUIRE(nb_controls > 0 && nb_cases > 0, "kmd_model_create: need >= 1 control and >= 1 case");
  KMD_REQUIRE(total_controls && total_cases, "kmd_model_create: NULL totals");
  KMD_REQUIRE(log_factorial_size < 0x80000000ull, "kmd_model_create: log-factorial table too large");
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); kmd::set_error("no HIP device"); return KMD_E_NO_DEVICE; }
  hipDeviceProp_t prop;
  KMD_HIP(hipGetDeviceProperties(&prop, dev));

  kmd_model* m = new (std::nothrow) kmd_model();
  if (!m) return KMD_E_NOMEM;
  m->device = dev;
  m->nc = nb_controls; m->nk = nb_cases;
  // model.hpp:185-188: std::accumulate over the per-sample totals
  m->tc = 0; m->tk = 0;
  for (int i = 0; i < nb_controls; ++i) m->tc += total_controls[i];
  for (int i = 0; i < nb_cases; ++i) m->tk += total_cases[i];
  m->dT = static_cast<double>(m->tc + m->tk);
  m->dTc = static_cast<double>(m->tc);
  m->dTk = static_cast<double>(m->tk);
  m->lg_half = kmd::lngamma_half_host();
  m->lf_n = log_factorial_size;
  m->n_cu = prop.multiProcessorCount;
  size_t lds = prop.sharedMemPerBlock;
  int optin = 0;
  if (hipDeviceGetAttribute(&optin, hipDeviceAttributeSharedMemPerBlockOptin, dev) == hipSuccess &&
      (size_t)optin > lds)
    lds = (size_t)optin;
  (void)hipGetLastError();
  m->lds_per_block_max = lds;
  m->cut_valid = false; m->cut_threshold_bits = 0; m->cut_value = 0;

  // LogFactorialTable::LogFactorialTable (src/log_factorial_table.cpp:5-22): entry i is the
  // descending sum log(i) + log(i-1) + ... + log(2), each entry summed from scratch.
  const size_t n = log_factorial_size ? log_factorial_size : 1;
  m->h_lf = static_cast<double*>(std::malloc(n * sizeof(double)));
  if (!m->h_lf) { delete m; return KMD_E_NOMEM; }
  m->h_lf[0] = 0;
  for (size_t i = 0; i < log_factorial_size; ++i)
  {
    double res = 0;
    for (size_t k = i; k > 1; --k) res += std::log(static_cast<double>(k));
    m->h_lf[i] = res;
  }
  m->d_lf = nullptr; m->d_tab = nullptr;
  // the kernels read {lf[k], log(k)} pairs: log(double(k)) is what the reference evaluates for
  // the alternative hypothesis (model.hpp:152-153, lambda = the integer count sum), taken here
  // from the same host libm
  std::vector<double> tab(2 * n, 0.0);
  for (size_t i = 0; i < log_factorial_size; ++i)
  {
    tab[2 * i] = m->h_lf[i];
    tab[2 * i + 1] = i ? std::log(static_cast<double>(i)) : 0.0;
  }
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&m->d_lf), n * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&m->d_tab), 2 * n * sizeof(double));
  if (e == hipSuccess) e = hipMemcpy(m->d_lf, m->h_lf, n * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(m->d_tab, tab.data(), 2 * n * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipStreamSynchronize(nullptr);     // (the tables are read from streams that are not ordered against the null stream)
  if (e != hipSuccess)
  {
    if (m->d_lf) (void)hipFree(m->d_lf);
    if (m->d_tab) (void)hipFree(m->d_tab);
    std::free(m->h_lf); delete m;
    return kmd::hip_fail(e, "upload log-factorial table", __FILE__, __LINE__);
  }
  m->d_log_int = nullptr;
  const int rc_log = kmd::log_int_table(&m->d_log_int);      // (the running sums beyond the table: near-threshold rows, kmd_pvalues_refine)
  if (rc_log != KMD_OK)
  {
    (void)hipFree(m->d_lf); (void)hipFree(m->d_tab);
    std::free(m->h_lf); delete m;
    return rc_log;
  }
  *out = m;
  return KMD_OK;
}

int kmd_model_destroy(kmd_model* m)
{
  if (!m) return KMD_OK;
  if (m->d_lf) (void)hipFree(m->d_lf);
  if (m->d_tab) (void)hipFree(m->d_tab);
  std::free(m->h_lf);
  delete m;
  return KMD_OK;
}

int kmd_model_info(const kmd_model* m, int* nc, int* nk, uint64_t* tc, uint64_t* tk, size_t* lf_n)
{
  KMD_REQUIRE(m, "kmd_model_info: NULL");
  if (nc) *nc = m->nc;
  if (nk) *nk = m->nk;
  if (tc) *tc = m->tc;
  if (tk) *tk = m->tk;
  if (lf_n) *lf_n = m->lf_n;
  return KMD_OK;
}

int kmd_model_lf_table(const kmd_model* m, double* out, size_t n)
{
  KMD_REQUIRE(m && out, "kmd_model_lf_table: NULL");
  KMD_REQUIRE(n <= m->lf_n, "kmd_model_lf_table: n > table size");
  KMD_HIP(hipMemcpy(out, m->d_lf, n * sizeof(double), hipMemcpyDeviceToHost));
  return KMD_OK;
}

} // extern "C"

// ---- synthetic matrices ----------------------------------------------------------------------
// Definition (also restated, independently, by the CPU oracle):
//   h_row   = mix(mix(seed ^ C_PART*(part+1)) ^ C_ROW*(row+1))
//   class   = rate class from h_row[0,16) with weights .40 .30 .15 .10 .04 .01 -> lambda index
//             base {1,3,5,7,11,17}  (lambda_j = 0.5 * 2^(j/2), tables in include/kmdiff_synth_tables.h)
//   big     = (h_row >> 16) % 1e6 == 1   -> base index 24 (count sums beyond the lf table)
//   planted = (h_row >> 36) % 1e4 == 0   -> +6 steps (x8) on cases, on controls if h_row bit 63
//   depth_s = mix(seed ^ C_DEPTH*(s+1)) % 3 -> +0/+1/+2 steps per sample
//   cell    = hc = mix(h_row ^ C_CELL*(s+1)); classes 0,1 zero-inflated (p = 0.3) on
//             hc[32,48); else inverse-CDF Poisson draw of the low 32 bits of hc
//   all-zero rows get count 1 in sample h_row % S
//   kmer    = part*2^54 + row*2^21 + 1 + (mix(h_row ^ C_KMER) & 0xFFFFF); for k > 32 that is
//             the high limb and the low limb is mix(h_row ^ C_KMER2)
namespace {

constexpr uint64_t C_PART = 0xA0761D6478BD642Full, C_ROW = 0xE7037ED1A0B428DBull,
                   C_DEPTH = 0x8EBC6AF09C88C6E3ull, C_CELL = 0x589965CC75374CC3ull,
                   C_KMER = 0x1D8E4E27C47D124Full, C_KMER2 = 0xEB44ACCAB455D165ull;

__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

struct synth_tables { const uint32_t* off; const uint32_t* c0; const uint32_t* len; const uint32_t* thr; };
synth_tables g_tables[16] = {};          // per device
bool g_tables_ready[16] = {};

int ensure_tables(synth_tables& t)
{
  int dev = 0;
  KMD_HIP(hipGetDevice(&dev));
  KMD_REQUIRE(dev < 16, "kmd_synth: device index >= 16");
  static std::mutex mu;                                  // (bench's ranks-in-one-process generate from several threads)
  std::lock_guard<std::mutex> lock(mu);
  if (!g_tables_ready[dev])
  {
    uint32_t* d = nullptr;
    const size_t n = 3 * KMD_SYNTH_NJ + KMD_SYNTH_NTHR;
    KMD_HIP(hipMalloc(reinterpret_cast<void**>(&d), n * sizeof(uint32_t)));
    KMD_HIP(hipMemcpy(d, KMD_SYNTH_OFF, KMD_SYNTH_NJ * 4, hipMemcpyHostToDevice));
    KMD_HIP(hipMemcpy(d + KMD_SYNTH_NJ, KMD_SYNTH_C0, KMD_SYNTH_NJ * 4, hipMemcpyHostToDevice));
    KMD_HIP(hipMemcpy(d + 2 * KMD_SYNTH_NJ, KMD_SYNTH_LEN, KMD_SYNTH_NJ * 4, hipMemcpyHostToDevice));
    KMD_HIP(hipMemcpy(d + 3 * KMD_SYNTH_NJ, KMD_SYNTH_THR, KMD_SYNTH_NTHR * 4, hipMemcpyHostToDevice));
    KMD_HIP(hipStreamSynchronize(nullptr));
    g_tables[dev] = synth_tables{ d, d + KMD_SYNTH_NJ, d + 2 * KMD_SYNTH_NJ, d + 3 * KMD_SYNTH_NJ };
    g_tables_ready[dev] = true;
  }
  t = g_tables[dev];
  return KMD_OK;
}

template <typename CT>
__global__ void __launch_bounds__(256) k_synth(uint64_t seed, uint32_t part, uint64_t row0, size_t n_rows,
                                               int nc, int nk, int layout, size_t ld, CT* __restrict__ counts,
                                               uint64_t* __restrict__ kmer_lo, uint64_t* __restrict__ kmer_hi,
                                               synth_tables T)
{
  const int S = nc + nk;
  constexpr uint32_t cmax = sizeof(CT) == 1 ? 0xFFu : sizeof(CT) == 2 ? 0xFFFFu : 0xFFFFFFFFu;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  // bits 8.. of `part`: the presence profile (include/kmdiff_hip.h, kmd_synth_fill).  1 = MIXED: every second row (by a
  // bit of its hash) is RARE -- present in one or two samples, picked by the hash -- the others are COMMON: present in
  // 95 % of the samples.  The counts are the default profile's draws (at least 1 where the row is present).
  const uint32_t profile = part >> 8;
  part &= 0xFFu;
  for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += stride)
  {
    const uint64_t row = row0 + r;
    const uint64_t h = mix64(mix64(seed ^ (C_PART * ((uint64_t)part + 1))) ^ (C_ROW * (row + 1)));
    const bool rare = profile == 1 && ((h >> 41) & 1ull) != 0;
    const int rare_s0 = (int)((h >> 8) % (uint64_t)S), rare_s1 = ((h >> 42) & 1ull) ? (int)((h >> 24) % (uint64_t)S) : rare_s0;
    const uint32_t u16 = (uint32_t)(h & 0xFFFF);
    int cls = u16 < 26214 ? 0 : u16 < 45875 ? 1 : u16 < 55705 ? 2 : u16 < 62259 ? 3 : u16 < 64880 ? 4 : 5;
    int jbase = cls == 0 ? 1 : cls == 1 ? 3 : cls == 2 ? 5 : cls == 3 ? 7 : cls == 4 ? 11 : 17;
    if ((h >> 16) % 1000000ull == 1) { jbase = 24; cls = 6; }
    // (a rare row is a low-abundance one: the two low rate classes, no planted signal -- a k-mer seen 300 times in one
    // sample and nowhere else would be "significant", and one row in forty would be)
    if (rare && cls > 1) { cls = 1; jbase = 3; }
    const bool planted = !rare && (h >> 36) % 10000ull == 0;
    const int boost_controls = (int)(h >> 63);
    bool any = false;
    for (int s = 0; s < S; ++s)
    {
      const uint64_t hc = mix64(h ^ (C_CELL * ((uint64_t)s + 1)));
      uint32_t v = 0;
      const bool absent = profile == 1 ? (rare ? (s != rare_s0 && s != rare_s1) : ((hc >> 48) % 20ull) == 0)
                                       : (cls <= 1 && ((hc >> 32) & 0xFFFF) < 19661);
      if (!absent)
      {
        int j = jbase + (int)(mix64(seed ^ (C_DEPTH * ((uint64_t)s + 1))) % 3);
        const int is_case = s >= nc;
        if (planted && boost_controls != is_case) j += 6;
        if (j > KMD_SYNTH_NJ - 1) j = KMD_SYNTH_NJ - 1;
        const uint32_t* thr = T.thr + T.off[j];
        const uint32_t u = (uint32_t)hc;
        uint32_t lo = 0, hi = T.len[j];
        while (lo < hi)
        {
          const uint32_t mid = (lo + hi) >> 1;
          if (thr[mid] <= u) lo = mid + 1; else hi = mid;
        }
        v = T.c0[j] + lo;
        if (profile == 1 && v == 0) v = 1;                    // (mixed profile: presence is the profile's, not the draw's)
      }
      any |= (v != 0);
      if (v > cmax) v = cmax;
      const size_t idx = kmd::count_index(layout, ld, S, r, s);
      counts[idx] = (CT)v;
    }
    if (!any)
    {
      const int s = (int)(h % (uint64_t)S);
      const size_t idx = kmd::count_index(layout, ld, S, r, s);
      counts[idx] = (CT)1;
    }
    if (kmer_lo)
    {
      const uint64_t v = ((uint64_t)part << 54) + (row << 21) + 1 + (mix64(h ^ C_KMER) & 0xFFFFF);
      if (kmer_hi) { kmer_hi[r] = v; kmer_lo[r] = mix64(h ^ C_KMER2); }
      else kmer_lo[r] = v;
    }
  }
}

// per-sample totals: d_totals[s] += sum_rows counts[.][s]
template <typename CT>
__global__ void __launch_bounds__(256) k_column_sums(const CT* __restrict__ counts, int layout, size_t ld,
                                                     size_t n_rows, int n_samples,
                                                     unsigned long long* __restrict__ totals)
{
  __shared__ unsigned long long s_part[4];
  // blockIdx.y = sample; blockIdx.x strides rows
  const int s = blockIdx.y;
  unsigned long long acc = 0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += stride)
    acc += counts[kmd::count_index(layout, ld, n_samples, r, s)];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0)
    atomicAdd(&totals[s], s_part[0] + s_part[1] + s_part[2] + s_part[3]);
}

// ---- the same synthetic partition as per-sample streams (what kmtricks hands the merge) ---------------
// Records of sample s = the rows with a non-zero count in column s, (k-mer, count), in row order (= ascending
// k-mer).  Built chunk by chunk from k_synth's matrix (SoA scratch of `chunk` rows): per (sample, block of 1024
// rows) the non-zero counts are counted, scanned per sample, and scattered behind the sample's cursor.
constexpr int kStreamBlockRows = 1024;       // 256 threads x 4 rows

__global__ void __launch_bounds__(256) k_streams_count(const uint32_t* __restrict__ counts, size_t ld, size_t n_rows,
                                                       uint32_t n_blocks, uint32_t* __restrict__ cnt)
{
  __shared__ uint32_t s_part[4];
  const uint32_t b = blockIdx.x, s = blockIdx.y;
  const size_t r0 = (size_t)b * kStreamBlockRows + (size_t)threadIdx.x * 4;
  uint32_t mine = 0;
  for (int u = 0; u < 4; ++u) mine += (r0 + u < n_rows && counts[(size_t)s * ld + r0 + u] != 0) ? 1u : 0u;
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) cnt[(size_t)s * n_blocks + b] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

// one workgroup per sample: cnt[s][b] -> where block b's records of sample s begin (cursor[s] + exclusive
// prefix); cursor[s] += the chunk's records of sample s
__global__ void __launch_bounds__(256) k_streams_scan(uint32_t n_blocks, const uint32_t* __restrict__ cnt,
                                                      unsigned long long* __restrict__ blk_off, unsigned long long* __restrict__ cursor)
{
  __shared__ unsigned long long s_sum[256];
  const uint32_t s = blockIdx.x, t = threadIdx.x;
  const uint32_t per = (n_blocks + 255u) / 256u;
  const uint32_t lo = t * per, hi = lo + per < n_blocks ? lo + per : n_blocks;
  unsigned long long acc = 0;
  for (uint32_t b = lo; b < hi; ++b) acc += cnt[(size_t)s * n_blocks + b];
  s_sum[t] = acc;
  __syncthreads();
  if (t == 0)
  {
    unsigned long long run = cursor[s];
    for (int i = 0; i < 256; ++i) { const unsigned long long v = s_sum[i]; s_sum[i] = run; run += v; }
    cursor[s] = run;
  }
  __syncthreads();
  unsigned long long run = s_sum[t];
  for (uint32_t b = lo; b < hi; ++b) { blk_off[(size_t)s * n_blocks + b] = run; run += cnt[(size_t)s * n_blocks + b]; }
}

__global__ void __launch_bounds__(256) k_streams_scatter(const uint32_t* __restrict__ counts, size_t ld, size_t n_rows, uint32_t n_blocks,
                                                         const uint64_t* __restrict__ kmer_lo, const uint64_t* __restrict__ kmer_hi,
                                                         const unsigned long long* __restrict__ blk_off, uint64_t* __restrict__ out_k,
                                                         uint64_t* __restrict__ out_kh, uint32_t* __restrict__ out_c)
{
  __shared__ uint32_t s_wave[4];
  const uint32_t b = blockIdx.x, s = blockIdx.y, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const size_t r0 = (size_t)b * kStreamBlockRows + (size_t)threadIdx.x * 4;
  uint32_t c[4], mine = 0;
  for (int u = 0; u < 4; ++u) { c[u] = r0 + u < n_rows ? counts[(size_t)s * ld + r0 + u] : 0u; mine += c[u] != 0 ? 1u : 0u; }
  uint32_t incl = mine;
  for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, (unsigned)o, 64); incl += lane >= (uint32_t)o ? v : 0u; }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  uint32_t before = incl - mine;
  for (uint32_t w = 0; w < wave; ++w) before += s_wave[w];
  unsigned long long at = blk_off[(size_t)s * n_blocks + b] + before;
  for (int u = 0; u < 4; ++u)
    if (c[u] != 0)
    {
      out_k[at] = kmer_lo[r0 + u];
      if (out_kh) out_kh[at] = kmer_hi[r0 + u];
      out_c[at] = c[u];
      ++at;
    }
}

__global__ void __launch_bounds__(256) k_copy_probe(const float4* __restrict__ src, float4* __restrict__ dst, size_t n)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

// streaming-read probes of a known byte count, one per load width: calibrates the FETCH_SIZE
// PMC counter for the access pattern of the filter kernel (MI355X_MICROARCH.md, HBM section)
template <typename V, bool NT>
__global__ void __launch_bounds__(256) k_read_probe(const V* __restrict__ src, size_t n,
                                                    unsigned long long* __restrict__ sink)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned int acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
  {
    if constexpr (sizeof(V) == 4)
    {
      acc += NT ? __builtin_nontemporal_load(src + i) : src[i];
    }
    else if constexpr (sizeof(V) == 8)
    {
      typedef uint32_t n2 __attribute__((ext_vector_type(2)));
      const n2 v = NT ? __builtin_nontemporal_load(reinterpret_cast<const n2*>(src + i)) : *reinterpret_cast<const n2*>(src + i);
      acc += v.x ^ v.y;
    }
    else
    {
      typedef uint32_t n4 __attribute__((ext_vector_type(4)));
      const n4 v = NT ? __builtin_nontemporal_load(reinterpret_cast<const n4*>(src + i)) : *reinterpret_cast<const n4*>(src + i);
      acc += v.x ^ v.y ^ v.z ^ v.w;
    }
  }
  if (acc == 0x9E3779B9u) atomicAdd(sink, 1ull);     // keeps the loads alive
}

} // namespace

extern "C" {

int kmd_read_probe(const void* d_src, size_t bytes, int width_bytes, uint64_t* d_sink, void* stream)
{
  KMD_REQUIRE(d_src && d_sink, "kmd_read_probe: NULL");
  const bool nt = (width_bytes & 64) != 0;                // + 64: non-temporal loads (the filter kernel's)
  width_bytes &= ~64;
  KMD_REQUIRE(width_bytes == 4 || width_bytes == 8 || width_bytes == 16, "kmd_read_probe: width");
  KMD_REQUIRE(bytes % 16 == 0, "kmd_read_probe: bytes % 16");
  const size_t n = bytes / (size_t)width_bytes;
  if (!n) return KMD_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  unsigned long long* sink = reinterpret_cast<unsigned long long*>(d_sink);
  const dim3 grid(256 * 8), block(256);
  if (width_bytes == 4 && !nt) hipLaunchKernelGGL((k_read_probe<uint32_t, false>), grid, block, 0, st, static_cast<const uint32_t*>(d_src), n, sink);
  else if (width_bytes == 4) hipLaunchKernelGGL((k_read_probe<uint32_t, true>), grid, block, 0, st, static_cast<const uint32_t*>(d_src), n, sink);
  else if (width_bytes == 8 && !nt) hipLaunchKernelGGL((k_read_probe<uint2, false>), grid, block, 0, st, static_cast<const uint2*>(d_src), n, sink);
  else if (width_bytes == 8) hipLaunchKernelGGL((k_read_probe<uint2, true>), grid, block, 0, st, static_cast<const uint2*>(d_src), n, sink);
  else if (!nt) hipLaunchKernelGGL((k_read_probe<uint4, false>), grid, block, 0, st, static_cast<const uint4*>(d_src), n, sink);
  else hipLaunchKernelGGL((k_read_probe<uint4, true>), grid, block, 0, st, static_cast<const uint4*>(d_src), n, sink);
  KMD_HIP(hipGetLastError());
  return KMD_OK;
}

int kmd_synth_fill(uint64_t seed, uint32_t partition, uint64_t row0, size_t n_rows, int nc,
                   int nk, int count_bytes, int layout, size_t ld, void* d_counts,
                   uint64_t* d_kmer_lo, uint64_t* d_kmer_hi, void* stream)
{
  KMD_REQUIRE(d_counts || n_rows == 0, "kmd_synth_fill: NULL counts");
  KMD_REQUIRE(nc > 0 && nk > 0, "kmd_synth_fill: nc, nk must be positive");
  KMD_REQUIRE(count_bytes == 1 || count_bytes == 2 || count_bytes == 4, "kmd_synth_fill: count_bytes");
  KMD_REQUIRE(kmd::layout_ok(layout), "kmd_synth_fill: layout");
  KMD_REQUIRE(layout != KMD_LAYOUT_TILED || (ld > 0 && ld % 4096 == 0), "kmd_synth_fill: tiled ld % 4096");
  KMD_REQUIRE((partition >> 8) <= 1, "kmd_synth_fill: partition >= 256 or an unknown presence profile");
  if (n_rows == 0) return KMD_OK;
  synth_tables T;
  int rc = ensure_tables(T);
  if (rc != KMD_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  size_t grid = (n_rows + 255) / 256;
  if (grid > 65535 * 4) grid = 65535 * 4;
  switch (count_bytes)
  {
    case 1: hipLaunchKernelGGL((k_synth<uint8_t>), dim3((unsigned)grid), dim3(256), 0, st, seed, partition, row0, n_rows, nc, nk, layout, ld, static_cast<uint8_t*>(d_counts), d_kmer_lo, d_kmer_hi, T); break;
    case 2: hipLaunchKernelGGL((k_synth<uint16_t>), dim3((unsigned)grid), dim3(256), 0, st, seed, partition, row0, n_rows, nc, nk, layout, ld, static_cast<uint16_t*>(d_counts), d_kmer_lo, d_kmer_hi, T); break;
    default: hipLaunchKernelGGL((k_synth<uint32_t>), dim3((unsigned)grid), dim3(256), 0, st, seed, partition, row0, n_rows, nc, nk, layout, ld, static_cast<uint32_t*>(d_counts), d_kmer_lo, d_kmer_hi, T); break;
  }
  KMD_HIP(hipGetLastError());
  return KMD_OK;
}

int kmd_synth_streams(uint64_t seed, uint32_t partition, uint64_t row0, size_t n_rows, int nc, int nk,
                      uint64_t* offsets, uint64_t* d_kmers, uint64_t* d_kmers_hi, uint32_t* d_counts,
                      uint64_t* d_totals, void* stream)
{
  KMD_REQUIRE(offsets, "kmd_synth_streams: NULL offsets");
  KMD_REQUIRE(nc > 0 && nk > 0 && nc + nk <= 1024, "kmd_synth_streams: nc, nk");
  KMD_REQUIRE((partition >> 8) <= 1, "kmd_synth_streams: partition >= 256 or an unknown presence profile");
  KMD_REQUIRE((d_kmers == nullptr) == (d_counts == nullptr), "kmd_synth_streams: d_kmers and d_counts go together");
  KMD_REQUIRE(d_kmers || !d_kmers_hi, "kmd_synth_streams: d_kmers_hi without d_kmers");
  const int S = nc + nk;
  const bool fill = d_kmers != nullptr;
  if (!fill) for (int s = 0; s <= S; ++s) offsets[s] = 0;
  if (n_rows == 0) return KMD_OK;
  synth_tables T;
  int rc = ensure_tables(T);
  if (rc != KMD_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // rows per chunk: ~512 MB of SoA scratch, whole blocks
  size_t chunk = ((size_t)512 << 20) / (4 * (size_t)S) / kStreamBlockRows * kStreamBlockRows;
  chunk = std::max<size_t>(kStreamBlockRows, std::min<size_t>(chunk, (size_t)1 << 22));
  if (const char* e = std::getenv("KMD_SYNTH_CHUNK"))                 // dev: rows per chunk (tests: many chunks on a small partition)
    chunk = std::max<size_t>(kStreamBlockRows, (size_t)std::strtoull(e, nullptr, 10) / kStreamBlockRows * kStreamBlockRows);
  chunk = std::min(chunk, (n_rows + kStreamBlockRows - 1) / kStreamBlockRows * kStreamBlockRows);
  const uint32_t nb_max = (uint32_t)(chunk / kStreamBlockRows);
  struct bufs
  {
    void* p[6] = {};
    ~bufs() { for (void* q : p) if (q) (void)hipFree(q); }
  } B;
  KMD_HIP(hipMalloc(&B.p[0], chunk * (size_t)S * 4));                 // counts[s][row]
  KMD_HIP(hipMalloc(&B.p[1], chunk * 8));                             // k-mers of the chunk's rows
  KMD_HIP(hipMalloc(&B.p[2], chunk * 8));                             // (high limbs)
  KMD_HIP(hipMalloc(&B.p[3], (size_t)S * nb_max * 4));                // records per (sample, block)
  KMD_HIP(hipMalloc(&B.p[4], (size_t)S * nb_max * 8));                // where they go
  KMD_HIP(hipMalloc(&B.p[5], (size_t)S * 8));                         // cursor per sample
  uint32_t* d_mat = static_cast<uint32_t*>(B.p[0]);
  uint64_t *d_lo = static_cast<uint64_t*>(B.p[1]), *d_hi = static_cast<uint64_t*>(B.p[2]);
  uint32_t* d_cnt = static_cast<uint32_t*>(B.p[3]);
  unsigned long long *d_off = static_cast<unsigned long long*>(B.p[4]), *d_cur = static_cast<unsigned long long*>(B.p[5]);
  if (fill) KMD_HIP(hipMemcpyAsync(d_cur, offsets, (size_t)S * 8, hipMemcpyHostToDevice, st));
  else KMD_HIP(hipMemsetAsync(d_cur, 0, (size_t)S * 8, st));
  for (size_t r = 0; r < n_rows; r += chunk)
  {
    const size_t m = std::min(chunk, n_rows - r);
    const uint32_t nb = (uint32_t)((m + kStreamBlockRows - 1) / kStreamBlockRows);
    size_t grid = (m + 255) / 256;
    if (grid > 65535 * 4) grid = 65535 * 4;
    // (two limbs: the high limb carries the row, the low one is a hash -- as kmd_synth_fill writes them)
    hipLaunchKernelGGL((k_synth<uint32_t>), dim3((unsigned)grid), dim3(256), 0, st, seed, partition, row0 + r, m, nc, nk, (int)KMD_LAYOUT_SOA, chunk,
                       d_mat, fill ? d_lo : nullptr, fill && d_kmers_hi ? d_hi : nullptr, T);
    if (!fill && d_totals)
      hipLaunchKernelGGL((k_column_sums<uint32_t>), dim3((unsigned)std::min<size_t>((m + 255) / 256, 1024), (unsigned)S), dim3(256), 0, st, d_mat,
                         (int)KMD_LAYOUT_SOA, chunk, m, S, reinterpret_cast<unsigned long long*>(d_totals));
    hipLaunchKernelGGL(k_streams_count, dim3(nb, (unsigned)S), dim3(256), 0, st, d_mat, chunk, m, nb, d_cnt);
    hipLaunchKernelGGL(k_streams_scan, dim3((unsigned)S), dim3(256), 0, st, nb, d_cnt, d_off, d_cur);
    if (fill)
      hipLaunchKernelGGL(k_streams_scatter, dim3(nb, (unsigned)S), dim3(256), 0, st, d_mat, chunk, m, nb, d_lo, d_kmers_hi ? d_hi : nullptr, d_off,
                         d_kmers, d_kmers_hi, d_counts);
    KMD_HIP(hipGetLastError());
  }
  if (!fill)
  {
    std::vector<uint64_t> per((size_t)S);
    KMD_HIP(hipMemcpyAsync(per.data(), d_cur, (size_t)S * 8, hipMemcpyDeviceToHost, st));
    KMD_HIP(hipStreamSynchronize(st));
    for (int s = 0; s < S; ++s) offsets[s + 1] = offsets[s] + per[(size_t)s];
  }
  else KMD_HIP(hipStreamSynchronize(st));
  return KMD_OK;
}

int kmd_column_sums(const void* d_counts, int count_bytes, int layout, size_t ld,
                    size_t n_rows, int n_samples, uint64_t* d_totals, void* stream)
{
  KMD_REQUIRE(d_totals && (d_counts || n_rows == 0), "kmd_column_sums: NULL");
  KMD_REQUIRE(count_bytes == 1 || count_bytes == 2 || count_bytes == 4, "kmd_column_sums: count_bytes");
  KMD_REQUIRE(n_samples > 0 && n_samples <= 65535, "kmd_column_sums: n_samples");
  if (n_rows == 0) return KMD_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  size_t gx = (n_rows + 255) / 256;
  if (gx > 1024) gx = 1024;
  dim3 grid((unsigned)gx, (unsigned)n_samples);
  unsigned long long* tot = reinterpret_cast<unsigned long long*>(d_totals);
  switch (count_bytes)
  {
    case 1: hipLaunchKernelGGL((k_column_sums<uint8_t>), grid, dim3(256), 0, st, static_cast<const uint8_t*>(d_counts), layout, ld, n_rows, n_samples, tot); break;
    case 2: hipLaunchKernelGGL((k_column_sums<uint16_t>), grid, dim3(256), 0, st, static_cast<const uint16_t*>(d_counts), layout, ld, n_rows, n_samples, tot); break;
    default: hipLaunchKernelGGL((k_column_sums<uint32_t>), grid, dim3(256), 0, st, static_cast<const uint32_t*>(d_counts), layout, ld, n_rows, n_samples, tot); break;
  }
  KMD_HIP(hipGetLastError());
  return KMD_OK;
}

int kmd_copy_probe(void* d_dst, const void* d_src, size_t bytes, void* stream)
{
  KMD_REQUIRE(d_dst && d_src, "kmd_copy_probe: NULL");
  KMD_REQUIRE(bytes % 16 == 0, "kmd_copy_probe: bytes % 16");
  const size_t n = bytes / 16;
  if (!n) return KMD_OK;
  size_t grid = (n + 255) / 256;
  if (grid > 256 * 16) grid = 256 * 16;
  hipLaunchKernelGGL(k_copy_probe, dim3((unsigned)grid), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const float4*>(d_src), static_cast<float4*>(d_dst), n);
  KMD_HIP(hipGetLastError());
  return KMD_OK;
}

} // extern "C"

// dev (KMD_ABORT_TRACE=1 | <file> | fd:<n>): the native stack of whoever calls abort() -- the HIP runtime, libstdc++'s
// terminate, an assert -- written to stderr, and to <file> (appended) or descriptor <n> if one is named (a test runner may
// have fd 2 captured), before the handler that was there before (Python's faulthandler, say) or the default takes over.
#include <fcntl.h>
namespace {
int g_abort_trace_fd = -1;
struct sigaction g_abort_prev;
void abort_trace(int sig, siginfo_t* info, void* ctx)
{
  void* frames[64];
  const int n = backtrace(frames, 64);
  const char msg[] = "\n[kmdiff_hip] SIGABRT: native backtrace of the aborting thread:\n";
  for (int fd : { 2, g_abort_trace_fd })
  {
    if (fd < 0) continue;
    (void)!write(fd, msg, sizeof msg - 1);
    backtrace_symbols_fd(frames, n, fd);
  }
  if ((g_abort_prev.sa_flags & SA_SIGINFO) && g_abort_prev.sa_sigaction) { g_abort_prev.sa_sigaction(sig, info, ctx); }
  else if (!(g_abort_prev.sa_flags & SA_SIGINFO) && g_abort_prev.sa_handler != SIG_DFL && g_abort_prev.sa_handler != SIG_IGN) g_abort_prev.sa_handler(sig);
  signal(sig, SIG_DFL);
  raise(sig);
}
struct abort_trace_installer
{
  abort_trace_installer()
  {
    const char* e = std::getenv("KMD_ABORT_TRACE");
    if (!e || !*e) return;
    if (std::strncmp(e, "fd:", 3) == 0)
    {
      g_abort_trace_fd = std::atoi(e + 3);
      // (a child process inherits the variable, not the descriptor: only one that is open NOW is written to)
      if (g_abort_trace_fd <= 2 || fcntl(g_abort_trace_fd, F_GETFD) == -1) g_abort_trace_fd = -1;
    }
    else if (e[0] == '/' || e[0] == '.') g_abort_trace_fd = open(e, O_WRONLY | O_CREAT | O_APPEND, 0644);
    void* warm[4];
    (void)backtrace(warm, 4);                                  // (loads libgcc now, not inside the handler)
    struct sigaction sa;
    std::memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = abort_trace;
    sa.sa_flags = SA_SIGINFO;
    sigemptyset(&sa.sa_mask);
    sigaction(SIGABRT, &sa, &g_abort_prev);
  }
} g_abort_trace_installer;
} // namespace
