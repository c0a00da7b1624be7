// kmd_internal.h -- shared between the translation units of libkmdiff_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <mutex>
#include <string>

#include "../../include/kmdiff_hip.h"

// The model: device-resident state of PoissonLikelihood + LogFactorialTable.
struct kmd_model
{
  int device;
  int nc, nk;
  uint64_t tc, tk;          // sums of the per-sample totals (model.hpp:185-188)
  double dT, dTc, dTk;      // double(Tc+Tk), double(Tc), double(Tk)
  double lg_half;           // Cephes lgam(1/2)
  size_t lf_n;              // --log-factorial
  double* h_lf;             // host copy of the table
  double* d_lf;             // device copy
  double* d_tab;            // device: pairs { lf[k], log(double(k)) }, k < lf_n (16 B each)
  const double* d_log_int;  // device: correctly rounded log(j), j < 2^20 (the device's one copy, kmd::log_int_table)
  int n_cu;                 // multiProcessorCount
  size_t lds_per_block_max; // sharedMemPerBlock
  // cache of lr_cut_for_threshold (host-side constant of the last threshold used); one model
  // is shared by all partition tasks in the reference (merge.hpp:418), so guard it
  mutable std::mutex cut_mu;
  mutable bool cut_valid;
  mutable uint64_t cut_threshold_bits;
  mutable double cut_value;
};

namespace kmd {

// element index of (row, sample) in a count matrix of S samples.  The tiled layout's block
// size is a power of two in practice (4096): shifts instead of 64-bit divisions, which cost
// hundreds of instructions per element on the device.
__host__ __device__ inline size_t count_index(int layout, size_t ld, int S, size_t row, int s)
{
  if (layout == KMD_LAYOUT_ROWS) return row * ld + (size_t)s;
  if (layout == KMD_LAYOUT_SOA) return (size_t)s * ld + row;
  if ((ld & (ld - 1)) == 0)
  {
    const int sh = __builtin_ctzll((unsigned long long)ld);
    return ((row >> sh) * (size_t)S + (size_t)s) * ld + (row & (ld - 1));
  }
  return (row / ld) * ((size_t)S * ld) + (size_t)s * ld + (row % ld);   // KMD_LAYOUT_TILED, any T
}

inline bool layout_ok(int layout)
{
  return layout == KMD_LAYOUT_ROWS || layout == KMD_LAYOUT_SOA || layout == KMD_LAYOUT_TILED;
}

void set_error(const std::string& msg);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define KMD_HIP(call)                                                       \
  do {                                                                      \
    hipError_t e__ = (call);                                                \
    if (e__ != hipSuccess) return kmd::hip_fail(e__, #call, __FILE__, __LINE__); \
  } while (0)

#define KMD_REQUIRE(cond, msg)                                              \
  do { if (!(cond)) { kmd::set_error(msg); return KMD_E_INVALID; } } while (0)

// Caching device allocator for the library's internal scratch (sort buffers, flags, tallies):
// hipMalloc / hipFree cost 0.1-1 ms each and hipFree synchronises the device, which showed up
// as several ms per job in bench.py.  Blocks are rounded to a power of two and parked on free;
// every internal user synchronises its stream before freeing.  kmd_release_cache() trims.
hipError_t scratch_alloc(void** p, size_t bytes);
void scratch_free(void* p);
void scratch_release_all();
// the near-threshold lists kept per stream (kmd_filter.hip): one stream's (it is being destroyed), all of them
void near_list_forget(hipStream_t stream);
void near_lists_release();
// the table of correctly rounded log(j), j < 2^20 (8 MB), behind the running sums of kmd_filter.hip: the current device's,
// built at the first call and kept for the life of the process (models point into it)
int log_int_table(const double** out);
// the page-locked table rings of kmd_unpack_streams (kmd_pack.hip): one stream's, or all of them
void unpack_tables_forget(hipStream_t stream, bool all);

// smallest LR at which igamc(1/2, LR) <= threshold, minus a safety margin; rows with a
// likelihood ratio below it cannot pass `p <= threshold` (kmd_filter.hip).
double lr_cut_for_threshold(double threshold, double lg_half);

inline uint64_t bits_of(double x) { uint64_t b; __builtin_memcpy(&b, &x, 8); return b; }

} // namespace kmd
