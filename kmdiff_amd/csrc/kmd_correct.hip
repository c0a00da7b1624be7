// kmd_correct.hip -- K4: significance correction over the survivor list, survivor ordering
// and the survivor count gather.  Survivors are << rows, so none of this is on the HBM
// roofline; it exists so that stages 1-3 stay on the device with no host round trip of
// per-survivor data.
//
// Replaces make_corrector / ICorrector::apply (src/corrector.cpp:6-116) as driven by
// aggregator::worker (include/kmdiff/aggregator.hpp:137-171) and sorted_aggregator::run
// (aggregator.hpp:240-322).
#include <cstring>
#include <string.h>

#include "kmd_internal.h"

#include <rocprim/rocprim.hpp>

#include <cmath>
#include <utility>

namespace {

struct corr_params
{
  int type;
  double threshold;
  uint64_t total;
  double bonf_cut;    // threshold / total             (corrector.cpp:11)
  double sidak_cut;   // 1 - pow(1 - threshold, 1/N)   (corrector.cpp:52)
  uint64_t rank0;     // BH / Holm: survivors already accepted before this list (sharded runs)
};

// stateless correctors: one predicate per survivor (aggregator.hpp:146-166)
__global__ void __launch_bounds__(256) k_correct_stateless(corr_params C, const double* __restrict__ p,
                                                           const int32_t* __restrict__ sign, size_t n,
                                                           uint8_t* __restrict__ keep,
                                                           unsigned long long* __restrict__ tallies)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool k = false, ctrl = false;
  if (i < n)
  {
    const double pv = p[i];
    k = C.type == KMD_CORR_BONFERRONI ? (pv < C.bonf_cut)
      : C.type == KMD_CORR_SIDAK ? (pv < C.sidak_cut)
      : (pv < C.threshold);
    ctrl = k && sign && sign[i] == KMD_SIGN_CONTROL;
    if (keep) keep[i] = k ? 1 : 0;
  }
  const unsigned long long km = __ballot(k), cm = __ballot(ctrl);
  if ((threadIdx.x & 63) == 0 && km)
  {
    atomicAdd(&tallies[0], (unsigned long long)__popcll(km));
    if (cm) atomicAdd(&tallies[1], (unsigned long long)__popcll(cm));
  }
}

// BH / Holm over survivors sorted by ascending p: position j (0-based) is accepted iff
//   BH  : p_j < ((j+1) / double(N)) * fdr      (corrector.cpp:27-35, m_rank starts at 1 and
//                                               has been incremented j times on reaching j)
//   Holm: p_j < threshold / (N - j)            (corrector.cpp:68-71, m_total-- per apply)
// and the walk stops at the first rejection (aggregator.hpp:290-291): find that index.
__global__ void __launch_bounds__(256) k_first_reject(corr_params C, const uint64_t* __restrict__ p_sorted_bits,
                                                      size_t n, unsigned long long* __restrict__ first_reject)
{
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const double pv = __longlong_as_double((long long)p_sorted_bits[j]);
  bool ok;
  const uint64_t r = C.rank0 + j;                      // applies made before this one
  if (C.type == KMD_CORR_BENJAMINI)
    ok = pv < (((double)(r + 1) / (double)C.total) * C.threshold);
  else
    ok = pv < (C.threshold / (double)(C.total - r));
  if (!ok) atomicMin(first_reject, (unsigned long long)j);
}

__global__ void __launch_bounds__(256) k_mark_sorted(const uint32_t* __restrict__ order, size_t n,
                                                     const unsigned long long* __restrict__ first_reject,
                                                     const int32_t* __restrict__ sign, uint8_t* __restrict__ keep,
                                                     unsigned long long* __restrict__ tallies)
{
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool k = false, ctrl = false;
  if (j < n)
  {
    const uint32_t i = order[j];
    k = j < *first_reject;
    ctrl = k && sign && sign[i] == KMD_SIGN_CONTROL;
    if (keep) keep[i] = k ? 1 : 0;
  }
  const unsigned long long km = __ballot(k), cm = __ballot(ctrl);
  if ((threadIdx.x & 63) == 0 && km)
  {
    atomicAdd(&tallies[0], (unsigned long long)__popcll(km));
    if (cm) atomicAdd(&tallies[1], (unsigned long long)__popcll(cm));
  }
}

__global__ void __launch_bounds__(256) k_iota(uint32_t* v, size_t n)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = (uint32_t)i;
}

template <typename T>
__global__ void __launch_bounds__(256) k_permute(const T* __restrict__ src, T* __restrict__ dst,
                                                 const uint32_t* __restrict__ order, size_t n)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[order[i]];
}

template <typename CT>
__global__ void __launch_bounds__(256) k_gather_counts(const CT* __restrict__ counts, int layout, size_t ld,
                                                       uint64_t row_base, int S, const uint64_t* __restrict__ rows,
                                                       size_t n, double* __restrict__ out)
{
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * (size_t)S) return;
  const size_t i = t / (size_t)S;
  const int s = (int)(t - i * (size_t)S);
  const size_t r = (size_t)(rows[i] - row_base);
  const CT v = counts[kmd::count_index(layout, ld, S, r, s)];
  out[t] = (double)v;                                          // merge.hpp:91
}

inline unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256); }

// 4096 log-spaced bins: the top 12 magnitude bits of the double (11 exponent bits + 1 mantissa
// bit); monotone in p for p >= 0
__global__ void __launch_bounds__(256) k_p_histogram(const double* __restrict__ p, size_t n,
                                                     unsigned long long* __restrict__ hist)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long bits = (unsigned long long)__double_as_longlong(p[i]);
  atomicAdd(&hist[(bits >> 51) & 4095ull], 1ull);
}

// First bin that is NOT accepted wholesale by the ascending walk of BH / Holm: a bin whose
// upper bound is <= the cut of the first rank it can occupy passes entirely, whatever the order
// inside it (the cut grows with the rank).  out[0] = bin, out[1] = survivors before it.
__global__ void k_critical_bin(corr_params C, const unsigned long long* __restrict__ hist,
                               unsigned long long* __restrict__ out)
{
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  unsigned long long before = 0;
  unsigned int b = 0;
  for (; b < 4096; ++b)
  {
    const unsigned long long c = hist[b];
    if (c == 0) continue;
    const double hi = __longlong_as_double((long long)(((unsigned long long)b + 1ull) << 51));   // exclusive upper bound
    const unsigned long long r = C.rank0 + before;              // applies made before the bin's first element
    const double cut = (C.type == KMD_CORR_BENJAMINI) ? (((double)(r + 1) / (double)C.total) * C.threshold)
                                                      : (C.threshold / (double)(C.total - r));
    if (!(hi <= cut)) break;
    before += c;
  }
  out[0] = b; out[1] = before;
}

// order[] = permutation sorting keys ascending (stable); keys_sorted optional output
int sort_order_u64(const uint64_t* d_keys, size_t n, uint64_t* d_keys_sorted, uint32_t* d_order,
                   hipStream_t st)
{
  uint32_t* d_iota = nullptr;
  void* d_tmp = nullptr;
  uint64_t* d_ks = d_keys_sorted;
  bool own_ks = false;
  size_t tmp_bytes = 0;
  KMD_HIP(kmd::scratch_alloc(reinterpret_cast<void**>(&d_iota), n * sizeof(uint32_t)));
  if (!d_ks)
  {
    hipError_t e = kmd::scratch_alloc(reinterpret_cast<void**>(&d_ks), n * sizeof(uint64_t));
    if (e != hipSuccess) { kmd::scratch_free(d_iota); return kmd::hip_fail(e, "hipMalloc", __FILE__, __LINE__); }
    own_ks = true;
  }
  hipLaunchKernelGGL(k_iota, dim3(blocks_for(n)), dim3(256), 0, st, d_iota, n);
  hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_keys, d_ks, d_iota, d_order, n, 0, 64, st);
  if (e == hipSuccess) e = kmd::scratch_alloc(&d_tmp, tmp_bytes ? tmp_bytes : 1);
  if (e == hipSuccess) e = rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_keys, d_ks, d_iota, d_order, n, 0, 64, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  kmd::scratch_free(d_iota);
  if (d_tmp) kmd::scratch_free(d_tmp);
  if (own_ks) kmd::scratch_free(d_ks);
  if (e != hipSuccess) return kmd::hip_fail(e, "radix sort", __FILE__, __LINE__);
  return KMD_OK;
}

template <typename T>
int permute_in_place(T* d_arr, const uint32_t* d_order, size_t n, void* d_scratch, hipStream_t st)
{
  if (!d_arr) return KMD_OK;
  hipLaunchKernelGGL((k_permute<T>), dim3(blocks_for(n)), dim3(256), 0, st, d_arr, static_cast<T*>(d_scratch), d_order, n);
  KMD_HIP(hipGetLastError());
  KMD_HIP(hipMemcpyAsync(d_arr, d_scratch, n * sizeof(T), hipMemcpyDeviceToDevice, st));
  return KMD_OK;
}

} // namespace

extern "C" {

int kmd_correct(int correction, double threshold, uint64_t total_kmers,
                const double* d_pvalue, const int32_t* d_sign, size_t n, uint8_t* d_keep,
                uint64_t* n_kept, uint64_t* n_control, uint64_t* n_case, void* stream)
{
  return kmd_correct_from_rank(correction, threshold, total_kmers, 0, d_pvalue, d_sign, n, d_keep,
                               n_kept, n_control, n_case, stream);
}

int kmd_pvalue_histogram(const double* d_pvalue, size_t n, uint64_t* d_hist, void* stream)
{
  KMD_REQUIRE(d_hist && (n == 0 || d_pvalue), "kmd_pvalue_histogram: NULL");
  if (n == 0) return KMD_OK;
  hipLaunchKernelGGL(k_p_histogram, dim3(blocks_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), d_pvalue, n,
                     reinterpret_cast<unsigned long long*>(d_hist));
  KMD_HIP(hipGetLastError());
  return KMD_OK;
}

int kmd_correct_critical_bin(int correction, double threshold, uint64_t total_kmers, const uint64_t* d_hist,
                             uint32_t* bin, uint64_t* n_before, void* stream)
{
  KMD_REQUIRE(correction == KMD_CORR_BENJAMINI || correction == KMD_CORR_HOLM, "kmd_correct_critical_bin: BH or Holm only");
  KMD_REQUIRE(d_hist && bin && n_before, "kmd_correct_critical_bin: NULL");
  hipStream_t st = static_cast<hipStream_t>(stream);
  corr_params C;
  C.type = correction; C.threshold = threshold; C.total = total_kmers; C.bonf_cut = 0; C.sidak_cut = 0; C.rank0 = 0;
  unsigned long long* d_out = nullptr;
  KMD_HIP(kmd::scratch_alloc(reinterpret_cast<void**>(&d_out), 16));
  hipLaunchKernelGGL(k_critical_bin, dim3(1), dim3(64), 0, st, C, reinterpret_cast<const unsigned long long*>(d_hist), d_out);
  unsigned long long h[2] = { 0, 0 };
  hipError_t e = hipMemcpyAsync(h, d_out, 16, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  kmd::scratch_free(d_out);
  if (e != hipSuccess) return kmd::hip_fail(e, "kmd_correct_critical_bin", __FILE__, __LINE__);
  *bin = (uint32_t)h[0]; *n_before = h[1];
  return KMD_OK;
}

int kmd_correct_from_rank(int correction, double threshold, uint64_t total_kmers, uint64_t rank_offset,
                          const double* d_pvalue, const int32_t* d_sign, size_t n, uint8_t* d_keep,
                          uint64_t* n_kept, uint64_t* n_control, uint64_t* n_case, void* stream)
{
  KMD_REQUIRE(correction >= KMD_CORR_NOTHING && correction <= KMD_CORR_HOLM, "kmd_correct: bad correction type");
  KMD_REQUIRE(n == 0 || d_pvalue, "kmd_correct: NULL p-values");
  KMD_REQUIRE(n < 0xFFFFFFFFull, "kmd_correct: too many survivors");
  hipStream_t st = static_cast<hipStream_t>(stream);
  uint64_t h_t[3] = { 0, 0, ~0ull };
  if (n)
  {
    corr_params C;
    C.type = correction; C.threshold = threshold; C.total = total_kmers;
    C.bonf_cut = threshold / (double)total_kmers;                       // corrector.cpp:11
    C.sidak_cut = 1 - std::pow(1 - threshold, 1.0 / (double)total_kmers); // corrector.cpp:52
    C.rank0 = rank_offset;
    unsigned long long* d_t = nullptr;       // [0] kept, [1] kept controls, [2] first reject
    uint32_t* d_order = nullptr; uint64_t* d_ps = nullptr;   // BH / Holm: the ascending-p order
    KMD_HIP(kmd::scratch_alloc(reinterpret_cast<void**>(&d_t), 3 * sizeof(unsigned long long)));
    hipError_t e = hipMemcpyAsync(d_t, h_t, sizeof h_t, hipMemcpyHostToDevice, st);
    int rc = KMD_OK;
    if (e != hipSuccess) rc = kmd::hip_fail(e, "hipMemcpyAsync", __FILE__, __LINE__);
    if (rc == KMD_OK)
    {
      if (correction == KMD_CORR_BENJAMINI || correction == KMD_CORR_HOLM)      // aggregator.hpp:358-360
      {
        e = kmd::scratch_alloc(reinterpret_cast<void**>(&d_order), n * sizeof(uint32_t));
        if (e == hipSuccess) e = kmd::scratch_alloc(reinterpret_cast<void**>(&d_ps), n * sizeof(uint64_t));
        if (e != hipSuccess) rc = kmd::hip_fail(e, "hipMalloc", __FILE__, __LINE__);
        // p >= 0: the IEEE bit pattern orders like the value
        if (rc == KMD_OK) rc = sort_order_u64(reinterpret_cast<const uint64_t*>(d_pvalue), n, d_ps, d_order, st);
        if (rc == KMD_OK)
        {
          hipLaunchKernelGGL(k_first_reject, dim3(blocks_for(n)), dim3(256), 0, st, C, d_ps, n, d_t + 2);
          hipLaunchKernelGGL(k_mark_sorted, dim3(blocks_for(n)), dim3(256), 0, st, d_order, n, d_t + 2, d_sign, d_keep, d_t);
          e = hipGetLastError();
          if (e != hipSuccess) rc = kmd::hip_fail(e, "launch", __FILE__, __LINE__);
        }
      }
      else
      {
        hipLaunchKernelGGL(k_correct_stateless, dim3(blocks_for(n)), dim3(256), 0, st, C, d_pvalue, d_sign, n, d_keep, d_t);
        e = hipGetLastError();
        if (e != hipSuccess) rc = kmd::hip_fail(e, "launch", __FILE__, __LINE__);
      }
    }
    if (rc == KMD_OK)
    {
      e = hipMemcpyAsync(h_t, d_t, sizeof h_t, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      if (e != hipSuccess) rc = kmd::hip_fail(e, "read tallies", __FILE__, __LINE__);
    }
    // the sort buffers go back to the (process-wide) cache only now: the kernels that read them have
    // finished -- or, on an error path, the stream is drained first
    if (rc != KMD_OK) (void)hipStreamSynchronize(st);
    if (d_order) kmd::scratch_free(d_order);
    if (d_ps) kmd::scratch_free(d_ps);
    kmd::scratch_free(d_t);
    if (rc != KMD_OK) return rc;
  }
  if (n_kept) *n_kept = h_t[0];
  if (n_control) *n_control = h_t[1];
  if (n_case) *n_case = h_t[0] - h_t[1];
  return KMD_OK;
}

int kmd_survivors_sort_by_row(const kmd_survivors* s, size_t n, void* stream)
{
  KMD_REQUIRE(s && s->d_row, "kmd_survivors_sort_by_row: needs d_row");
  KMD_REQUIRE(n <= s->capacity, "kmd_survivors_sort_by_row: n > capacity");
  KMD_REQUIRE(n < 0xFFFFFFFFull, "kmd_survivors_sort_by_row: too many survivors");
  if (n < 2) return KMD_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  uint32_t* d_order = nullptr; void* d_scratch = nullptr;
  KMD_HIP(kmd::scratch_alloc(reinterpret_cast<void**>(&d_order), n * sizeof(uint32_t)));
  hipError_t e = kmd::scratch_alloc(&d_scratch, n * sizeof(uint64_t));
  if (e != hipSuccess) { kmd::scratch_free(d_order); return kmd::hip_fail(e, "hipMalloc", __FILE__, __LINE__); }
  int rc = sort_order_u64(s->d_row, n, nullptr, d_order, st);
  if (rc == KMD_OK) rc = permute_in_place(s->d_row, d_order, n, d_scratch, st);
  if (rc == KMD_OK) rc = permute_in_place(s->d_kmer_lo, d_order, n, d_scratch, st);
  if (rc == KMD_OK) rc = permute_in_place(s->d_kmer_hi, d_order, n, d_scratch, st);
  if (rc == KMD_OK) rc = permute_in_place(s->d_pvalue, d_order, n, d_scratch, st);
  if (rc == KMD_OK) rc = permute_in_place(s->d_sign, d_order, n, d_scratch, st);
  if (rc == KMD_OK) rc = permute_in_place(s->d_mean_control, d_order, n, d_scratch, st);
  if (rc == KMD_OK) rc = permute_in_place(s->d_mean_case, d_order, n, d_scratch, st);
  hipError_t e2 = hipStreamSynchronize(st);
  kmd::scratch_free(d_order); kmd::scratch_free(d_scratch);
  if (rc != KMD_OK) return rc;
  if (e2 != hipSuccess) return kmd::hip_fail(e2, "sync", __FILE__, __LINE__);
  return KMD_OK;
}

// The reference's order of a partition's survivors is ascending k-mer (they are pushed as the merge
// emits rows, merge.hpp:100).  For survivors that carry their k-mer but no row index
// (kmd_merge_filter): stable LSD sort on (high limb, low limb).
int kmd_survivors_sort_by_kmer(const kmd_survivors* s, size_t n, void* stream)
{
  KMD_REQUIRE(s && s->d_kmer_lo, "kmd_survivors_sort_by_kmer: needs d_kmer_lo");
  KMD_REQUIRE(n <= s->capacity, "kmd_survivors_sort_by_kmer: n > capacity");
  KMD_REQUIRE(n < 0xFFFFFFFFull, "kmd_survivors_sort_by_kmer: too many survivors");
  if (n < 2) return KMD_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  uint32_t *d_order = nullptr, *d_order2 = nullptr;
  void *d_scratch = nullptr, *d_hs = nullptr, *d_tmp = nullptr;
  auto release = [&]() { for (void* q : { (void*)d_order, (void*)d_order2, d_scratch, d_hs, d_tmp }) if (q) kmd::scratch_free(q); };
  hipError_t e = kmd::scratch_alloc(reinterpret_cast<void**>(&d_order), n * sizeof(uint32_t));
  if (e == hipSuccess) e = kmd::scratch_alloc(&d_scratch, n * sizeof(uint64_t));
  if (e != hipSuccess) { release(); return kmd::hip_fail(e, "hipMalloc", __FILE__, __LINE__); }
  int rc = sort_order_u64(s->d_kmer_lo, n, nullptr, d_order, st);
  if (rc == KMD_OK && s->d_kmer_hi)
  {
    // second pass: high limbs in low-limb order, stable sort carrying the first permutation
    size_t tmp_bytes = 0;
    uint64_t* hi_g = static_cast<uint64_t*>(d_scratch);
    e = kmd::scratch_alloc(reinterpret_cast<void**>(&d_order2), n * sizeof(uint32_t));
    if (e == hipSuccess) e = kmd::scratch_alloc(&d_hs, n * sizeof(uint64_t));
    if (e == hipSuccess)
    {
      hipLaunchKernelGGL((k_permute<uint64_t>), dim3(blocks_for(n)), dim3(256), 0, st, s->d_kmer_hi, hi_g, d_order, n);
      e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, hi_g, static_cast<uint64_t*>(d_hs), d_order, d_order2, n, 0, 64, st);
    }
    if (e == hipSuccess) e = kmd::scratch_alloc(&d_tmp, tmp_bytes ? tmp_bytes : 1);
    if (e == hipSuccess) e = rocprim::radix_sort_pairs(d_tmp, tmp_bytes, hi_g, static_cast<uint64_t*>(d_hs), d_order, d_order2, n, 0, 64, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { release(); return kmd::hip_fail(e, "radix sort", __FILE__, __LINE__); }
    std::swap(d_order, d_order2);
  }
  if (rc == KMD_OK) rc = permute_in_place(s->d_row, d_order, n, d_scratch, st);
  if (rc == KMD_OK) rc = permute_in_place(s->d_kmer_lo, d_order, n, d_scratch, st);
  if (rc == KMD_OK) rc = permute_in_place(s->d_kmer_hi, d_order, n, d_scratch, st);
  if (rc == KMD_OK) rc = permute_in_place(s->d_pvalue, d_order, n, d_scratch, st);
  if (rc == KMD_OK) rc = permute_in_place(s->d_sign, d_order, n, d_scratch, st);
  if (rc == KMD_OK) rc = permute_in_place(s->d_mean_control, d_order, n, d_scratch, st);
  if (rc == KMD_OK) rc = permute_in_place(s->d_mean_case, d_order, n, d_scratch, st);
  const hipError_t e2 = hipStreamSynchronize(st);
  release();
  if (rc != KMD_OK) return rc;
  if (e2 != hipSuccess) return kmd::hip_fail(e2, "sync", __FILE__, __LINE__);
  return KMD_OK;
}

int kmd_survivors_gather_counts(const kmd_tile* tile, int n_samples, const uint64_t* d_rows,
                                size_t n, double* d_out, void* stream)
{
  KMD_REQUIRE(tile && tile->d_counts && d_rows && d_out, "kmd_survivors_gather_counts: NULL");
  KMD_REQUIRE(n_samples > 0, "kmd_survivors_gather_counts: n_samples");
  if (n == 0) return KMD_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t total = n * (size_t)n_samples;
  switch (tile->count_bytes)
  {
    case 1: hipLaunchKernelGGL((k_gather_counts<uint8_t>), dim3(blocks_for(total)), dim3(256), 0, st, static_cast<const uint8_t*>(tile->d_counts), tile->layout, tile->ld, tile->row_base, n_samples, d_rows, n, d_out); break;
    case 2: hipLaunchKernelGGL((k_gather_counts<uint16_t>), dim3(blocks_for(total)), dim3(256), 0, st, static_cast<const uint16_t*>(tile->d_counts), tile->layout, tile->ld, tile->row_base, n_samples, d_rows, n, d_out); break;
    case 4: hipLaunchKernelGGL((k_gather_counts<uint32_t>), dim3(blocks_for(total)), dim3(256), 0, st, static_cast<const uint32_t*>(tile->d_counts), tile->layout, tile->ld, tile->row_base, n_samples, d_rows, n, d_out); break;
    default: kmd::set_error("kmd_survivors_gather_counts: count_bytes"); return KMD_E_INVALID;
  }
  KMD_HIP(hipGetLastError());
  return KMD_OK;
}

} // extern "C"
