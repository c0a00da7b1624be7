// kmd_merge.hip -- K2: the k-way merge of one partition's per-sample k-mer streams into the
// merged count matrix, on the device.
//
// Replaces km::KmerMerger<KSIZE,CMAX>::merge as kmdiff drives it (include/kmdiff/merge.hpp:
// 265-289: paths of one partition, abundance minima all 1, recurrence minimum 1, save_if 0 =>
// every distinct k-mer is emitted, ascending, with the count of each sample or 0).  kmtricks
// itself is not part of the reference tree (empty submodule); the contract restated here is
// the one SURVEY.md 8a R1 derives from the call site and the fixture bytes.
//
// kmd_merge_partition (bottom of the file) is the tile merge of kmd_tilemerge.hip in rows mode + a fill of the matrix
// (merge_tiles, below): ONE merge implementation serves the fused path and the matrix path.  Tiny inputs (< 2^16
// records), and whatever is beyond the tile merge's limits, are sorted: records tagged with their sample id,
// radix-sorted together (rocPRIM), run heads flagged and scanned into row numbers, a scatter kernel writes the matrix
// (first half of the file).  Both write the layout K1 wants and the sorted k-mer column.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string.h>

#include "kmd_internal.h"

#include <algorithm>
#include <vector>
#include <rocprim/rocprim.hpp>

namespace {

inline unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256); }

// vals[i] = sample << 32 | count for the records of one sample
__global__ void __launch_bounds__(256) k_tag(const uint32_t* __restrict__ counts, size_t begin, size_t end,
                                             uint32_t sample, uint64_t* __restrict__ vals)
{
  const size_t i = begin + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < end) vals[i] = ((uint64_t)sample << 32) | counts[i];
}

__global__ void __launch_bounds__(256) k_heads(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ keys_hi,
                                               size_t n, uint32_t* __restrict__ flag)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = (i == 0 || keys[i] != keys[i - 1] || (keys_hi && keys_hi[i] != keys_hi[i - 1])) ? 1u : 0u;
}

__global__ void __launch_bounds__(256) k_iota32(uint32_t* v, size_t n)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = (uint32_t)i;
}

__global__ void __launch_bounds__(256) k_gather64(const uint64_t* __restrict__ src, const uint32_t* __restrict__ idx,
                                                  size_t n, uint64_t* __restrict__ dst)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}

template <typename CT>
__global__ void __launch_bounds__(256) k_scatter(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ keys_hi,
                                                 const uint64_t* __restrict__ vals,
                                                 const uint32_t* __restrict__ rank, size_t n, int layout, size_t ld,
                                                 int S, CT* __restrict__ matrix, uint64_t* __restrict__ kmer_out,
                                                 uint64_t* __restrict__ kmer_hi_out)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t row = rank[i] - 1;                       // inclusive scan of head flags
  const uint64_t v = vals[i];
  const int s = (int)(v >> 32);
  uint32_t c = (uint32_t)v;
  constexpr uint32_t cmax = sizeof(CT) == 1 ? 0xFFu : sizeof(CT) == 2 ? 0xFFFFu : 0xFFFFFFFFu;
  if (c > cmax) c = cmax;
  matrix[kmd::count_index(layout, ld, S, row, s)] = (CT)c;
  if (i == 0 || rank[i] != rank[i - 1])                 // head of its run
  {
    if (kmer_out) kmer_out[row] = keys[i];
    if (kmer_hi_out && keys_hi) kmer_hi_out[row] = keys_hi[i];
  }
}


// ---------------------------------------------------------------------------------------------
// The merge on K2t's tiles (round 6: the ONE merge of the library -- the bucketed LDS merge that stood here, one wave per
// key-range bucket with its own splitters, refinement levels and decoupled look-back, 1 100 lines at 0.10 of the HBM
// peak, is gone):
//   kmd_merge_sums       the tile merge in rows mode (kmd_tilemerge.hip): every distinct k-mer of the partition, once,
//                        in no particular order -- the rows of the matrix;
//   rocprim radix sort   the rows ascending (two limbs: stable by the low limb, then by the high one): the k-mer column;
//   k_row_windows        where every stream meets every 256th row: window[b][s] = first record of stream s whose key is
//                        >= row 256 b -- the records of stream s that belong to rows [256 b, 256 b + 256) are
//                        window[b][s] .. window[b + 1][s], at most 256 of them (a row holds a stream once);
//   k_fill_matrix        one workgroup per 256 rows: their keys in LDS; for 16 samples at a time every record of the
//                        windows finds its row by a bisection in LDS (8 steps) and puts its count into an LDS tile
//                        [sample][row] that was zeroed before; the tile leaves for the matrix with coalesced stores
//                        (any layout, any count width, saturating).  Every cell is written once: no zero pass over the
//                        matrix, no atomics.
// HBM traffic: the streams twice (tiles, windows' records) + the matrix once.
struct scratch
{
  void* p[10] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
  std::vector<void*> more;                               // buffers of a loop (take)
  hipError_t take(void** out, size_t bytes)
  {
    const hipError_t e = kmd::scratch_alloc(out, bytes);
    if (e == hipSuccess) more.push_back(*out);
    return e;
  }
  ~scratch()
  {
    for (void* q : p) if (q) kmd::scratch_free(q);
    for (void* q : more) if (q) kmd::scratch_free(q);
  }
};


constexpr uint32_t kFillRows = 256;                       // rows per workgroup of k_fill_matrix
constexpr uint32_t kFillSamples = 16;                     // samples per LDS tile (16 KB: eight workgroups per CU)

// window[b * S + s] for b = 0 .. n_blocks (b = n_blocks: the stream's end)
template <bool kTwo>
__global__ void __launch_bounds__(256) k_row_windows(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ keys_hi,
                                                     const uint64_t* __restrict__ offs, uint32_t S,
                                                     const uint64_t* __restrict__ rows, const uint64_t* __restrict__ rows_hi,
                                                     size_t n_rows, size_t n_blocks, uint32_t* __restrict__ window)
{
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (n_blocks + 1) * S) return;
  // (consecutive threads: consecutive row blocks of ONE stream -- their answers lie next to each other)
  const uint32_t s = (uint32_t)(t / (n_blocks + 1));
  const size_t b = t - (size_t)s * (n_blocks + 1);
  size_t lo = (size_t)offs[s], hi = (size_t)offs[s + 1];
  if (b < n_blocks)
  {
    const uint64_t k = rows[b * kFillRows], kh = kTwo ? rows_hi[b * kFillRows] : 0ull;
    while (lo < hi)
    {
      const size_t mid = lo + ((hi - lo) >> 1);
      const bool less = kTwo ? (keys_hi[mid] < kh || (keys_hi[mid] == kh && keys[mid] < k)) : keys[mid] < k;
      if (less) lo = mid + 1; else hi = mid;
    }
  }
  else lo = hi;
  window[b * S + s] = (uint32_t)lo;
}

template <typename CT, bool kTwo>
__global__ void __launch_bounds__(256) k_fill_matrix(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ keys_hi,
                                                     const uint32_t* __restrict__ counts, uint32_t S,
                                                     const uint64_t* __restrict__ rows, const uint64_t* __restrict__ rows_hi,
                                                     size_t n_rows, const uint32_t* __restrict__ window, int layout, size_t ld,
                                                     CT* __restrict__ matrix)
{
  __shared__ uint64_t s_row[kFillRows];
  __shared__ uint64_t s_row_hi[kTwo ? kFillRows : 1];
  __shared__ uint32_t s_tile[kFillSamples * kFillRows];                 // [sample of the chunk][row of the block]
  __shared__ uint32_t s_wlo[kFillSamples], s_whi[kFillSamples];         // the chunk's windows
  const uint32_t t = threadIdx.x;
  const size_t b = blockIdx.x, r0 = b * kFillRows;
  const uint32_t nr = (uint32_t)(n_rows - r0 < kFillRows ? n_rows - r0 : kFillRows);
  // (rows beyond the last: a key no record can match below the last real one -- the bisection never lands there)
  s_row[t] = t < nr ? rows[r0 + t] : ~0ull;
  if constexpr (kTwo) s_row_hi[t] = t < nr ? rows_hi[r0 + t] : ~0ull;
  constexpr uint32_t cmax = sizeof(CT) == 1 ? 0xFFu : sizeof(CT) == 2 ? 0xFFFFu : 0xFFFFFFFFu;
  constexpr uint32_t kAhead = 8;                                        // samples whose records are requested before the first is looked at
  for (uint32_t c0 = 0; c0 < S; c0 += kFillSamples)
  {
    const uint32_t cs = S - c0 < kFillSamples ? S - c0 : kFillSamples;
    for (uint32_t i = t; i < cs * kFillRows; i += 256) s_tile[i] = 0u;
    if (t < cs) { s_wlo[t] = window[b * S + c0 + t]; s_whi[t] = window[(b + 1) * S + c0 + t]; }
    __syncthreads();
    // A window holds at most 256 records (a row holds a stream once): thread t takes record t of every window.  The
    // records of kAhead samples are requested back to back and only then looked at -- one sample after the other, each
    // waiting for its window, its k-mer and then its count, the kernel stood at three memory latencies per sample:
    // 1.1 ms for 4 M rows of 40 samples, most of the whole merge.
    for (uint32_t g0 = 0; g0 < cs; g0 += kAhead)
    {
      uint64_t k[kAhead], kh[kTwo ? kAhead : 1];
      uint32_t c[kAhead];
      bool has[kAhead];
#pragma unroll
      for (uint32_t u = 0; u < kAhead; ++u)
      {
        const uint32_t sl = g0 + u;
        const uint32_t i = sl < cs ? s_wlo[sl] + t : 0u;
        has[u] = sl < cs && i < s_whi[sl];
        k[u] = 0; c[u] = 0;
        if constexpr (kTwo) kh[u] = 0;
        if (has[u])
        {
          k[u] = keys[i]; c[u] = counts[i];
          if constexpr (kTwo) kh[u] = keys_hi[i];
        }
      }
#pragma unroll
      for (uint32_t u = 0; u < kAhead; ++u)
      {
        if (!has[u]) continue;
        const uint64_t kk = k[u], kkh = kTwo ? kh[u] : 0ull;
        uint32_t lo = 0, hi = nr;
        while (lo < hi)
        {
          const uint32_t mid = (lo + hi) >> 1;
          const bool less = kTwo ? (s_row_hi[mid] < kkh || (s_row_hi[mid] == kkh && s_row[mid] < kk)) : s_row[mid] < kk;
          if (less) lo = mid + 1; else hi = mid;
        }
        // (every record's k-mer is a row; should the streams not be what the contract says -- unsorted, say -- a record
        // without its row is dropped rather than written somewhere)
        if (lo < nr && s_row[lo] == kk && (!kTwo || s_row_hi[lo] == kkh)) s_tile[(g0 + u) * kFillRows + lo] = c[u] > cmax ? cmax : c[u];
      }
    }
    __syncthreads();
    if (layout == KMD_LAYOUT_ROWS)
    {
      // a row's cs samples are contiguous: consecutive threads take consecutive samples of a row
      for (uint32_t i = t; i < cs * nr; i += 256)
      {
        const uint32_t r = i / cs, sl = i - r * cs;
        matrix[(r0 + r) * ld + c0 + sl] = (CT)s_tile[sl * kFillRows + r];
      }
    }
    else
    {
      // a sample's rows are contiguous (tiled: within a block of ld rows, which 256 divides): consecutive threads, consecutive rows
      for (uint32_t sl = 0; sl < cs; ++sl)
        if (t < nr) matrix[kmd::count_index(layout, ld, (int)S, r0 + t, (int)(c0 + sl))] = (CT)s_tile[sl * kFillRows + t];
    }
    __syncthreads();
  }
}

// *used = false (and nothing written) when the input is beyond the tile merge's limits: the caller then sorts
template <typename CT>
int merge_tiles(int S, const uint64_t* d_kmers_lo, const uint64_t* d_kmers_hi, const uint32_t* d_counts,
                const uint64_t* offsets, int layout, size_t ld, size_t row_capacity, CT* d_matrix,
                uint64_t* d_kmer_out, uint64_t* d_kmer_hi_out, uint64_t* n_rows_out, hipStream_t st, bool* used)
{
  *used = false;
  const size_t n = (size_t)offsets[S];
  const bool two = d_kmers_hi != nullptr;
  if (S > 1024 || n >= 0xFFFFFFFFull - 128ull) return KMD_OK;            // (kmd_merge_sums' limits)
  for (int s = 0; s < S; ++s) if (offsets[s + 1] - offsets[s] >= (1ull << 29)) return KMD_OK;
  const size_t cap = std::min(n, row_capacity);
  if (cap == 0) { *n_rows_out = 1; kmd::set_error("kmd_merge_partition: row capacity exceeded"); return KMD_E_OVERFLOW; }
  scratch sc;
  void *p_lo = nullptr, *p_hi = nullptr, *p_sc = nullptr, *p_sk = nullptr;
  KMD_HIP(sc.take(&p_lo, cap * 8)); KMD_HIP(sc.take(&p_sc, cap * 8)); KMD_HIP(sc.take(&p_sk, cap * 8));
  if (two) KMD_HIP(sc.take(&p_hi, cap * 8));
  uint64_t n_rows64 = 0;
  int rc = kmd_merge_sums(S, S, d_kmers_lo, d_kmers_hi, d_counts, offsets, cap, static_cast<uint64_t*>(p_lo), static_cast<uint64_t*>(p_hi),
                          static_cast<uint64_t*>(p_sc), static_cast<uint64_t*>(p_sk), &n_rows64, st);
  if (rc == KMD_E_OVERFLOW || (rc == KMD_OK && n_rows64 > row_capacity))
  {
    *used = true; *n_rows_out = n_rows64;
    kmd::set_error("kmd_merge_partition: row capacity exceeded");
    return KMD_E_OVERFLOW;
  }
  if (rc != KMD_OK) return rc;
  *used = true;
  const size_t n_rows = (size_t)n_rows64;
  // the rows ascending: the k-mer column (the caller's, or scratch when it wants none)
  void *p_slo = d_kmer_out, *p_shi = d_kmer_hi_out, *p_tmp = nullptr;
  if (!p_slo) KMD_HIP(sc.take(&p_slo, n_rows * 8));
  if (two && !p_shi) KMD_HIP(sc.take(&p_shi, n_rows * 8));
  uint64_t* rows = static_cast<uint64_t*>(p_slo);
  uint64_t* rows_hi = two ? static_cast<uint64_t*>(p_shi) : nullptr;
  size_t tmp_bytes = 0;
  if (!two)
  {
    KMD_HIP(rocprim::radix_sort_keys(nullptr, tmp_bytes, static_cast<const uint64_t*>(p_lo), rows, n_rows, 0, 64, st));
    KMD_HIP(sc.take(&p_tmp, tmp_bytes ? tmp_bytes : 1));
    KMD_HIP(rocprim::radix_sort_keys(p_tmp, tmp_bytes, static_cast<const uint64_t*>(p_lo), rows, n_rows, 0, 64, st));
  }
  else
  {
    // 128-bit keys, least significant limb first: stable by the low limb carrying the row's index, then by the high one
    void *p_i0 = nullptr, *p_i1 = nullptr, *p_i2 = nullptr, *p_k = nullptr, *p_g = nullptr;
    KMD_HIP(sc.take(&p_i0, n_rows * 4)); KMD_HIP(sc.take(&p_i1, n_rows * 4)); KMD_HIP(sc.take(&p_i2, n_rows * 4));
    KMD_HIP(sc.take(&p_k, n_rows * 8)); KMD_HIP(sc.take(&p_g, n_rows * 8));
    uint32_t *i0 = static_cast<uint32_t*>(p_i0), *i1 = static_cast<uint32_t*>(p_i1), *i2 = static_cast<uint32_t*>(p_i2);
    uint64_t *k_s = static_cast<uint64_t*>(p_k), *g = static_cast<uint64_t*>(p_g);
    KMD_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, static_cast<const uint64_t*>(p_lo), k_s, i0, i1, n_rows, 0, 64, st));
    KMD_HIP(sc.take(&p_tmp, tmp_bytes ? tmp_bytes : 1));
    hipLaunchKernelGGL(k_iota32, dim3(blocks_for(n_rows)), dim3(256), 0, st, i0, n_rows);
    KMD_HIP(rocprim::radix_sort_pairs(p_tmp, tmp_bytes, static_cast<const uint64_t*>(p_lo), k_s, i0, i1, n_rows, 0, 64, st));
    hipLaunchKernelGGL(k_gather64, dim3(blocks_for(n_rows)), dim3(256), 0, st, static_cast<const uint64_t*>(p_hi), i1, n_rows, g);
    KMD_HIP(rocprim::radix_sort_pairs(p_tmp, tmp_bytes, g, k_s, i1, i2, n_rows, 0, 64, st));
    hipLaunchKernelGGL(k_gather64, dim3(blocks_for(n_rows)), dim3(256), 0, st, static_cast<const uint64_t*>(p_lo), i2, n_rows, rows);
    hipLaunchKernelGGL(k_gather64, dim3(blocks_for(n_rows)), dim3(256), 0, st, static_cast<const uint64_t*>(p_hi), i2, n_rows, rows_hi);
    KMD_HIP(hipGetLastError());
  }
  // windows, then the matrix
  const size_t n_blocks = (n_rows + kFillRows - 1) / kFillRows;
  void *p_win = nullptr, *p_offs = nullptr;
  KMD_HIP(sc.take(&p_win, (n_blocks + 1) * (size_t)S * 4));
  KMD_HIP(sc.take(&p_offs, ((size_t)S + 1) * 8));
  KMD_HIP(hipMemcpyAsync(p_offs, offsets, ((size_t)S + 1) * 8, hipMemcpyHostToDevice, st));
  const size_t cells = (n_blocks + 1) * (size_t)S;
  if (two)
  {
    hipLaunchKernelGGL((k_row_windows<true>), dim3(blocks_for(cells)), dim3(256), 0, st, d_kmers_lo, d_kmers_hi, static_cast<const uint64_t*>(p_offs), (uint32_t)S,
                       rows, rows_hi, n_rows, n_blocks, static_cast<uint32_t*>(p_win));
    hipLaunchKernelGGL((k_fill_matrix<CT, true>), dim3((unsigned)n_blocks), dim3(256), 0, st, d_kmers_lo, d_kmers_hi, d_counts, (uint32_t)S, rows, rows_hi, n_rows,
                       static_cast<const uint32_t*>(p_win), layout, ld, d_matrix);
  }
  else
  {
    hipLaunchKernelGGL((k_row_windows<false>), dim3(blocks_for(cells)), dim3(256), 0, st, d_kmers_lo, d_kmers_hi, static_cast<const uint64_t*>(p_offs), (uint32_t)S,
                       rows, rows_hi, n_rows, n_blocks, static_cast<uint32_t*>(p_win));
    hipLaunchKernelGGL((k_fill_matrix<CT, false>), dim3((unsigned)n_blocks), dim3(256), 0, st, d_kmers_lo, d_kmers_hi, d_counts, (uint32_t)S, rows, rows_hi, n_rows,
                       static_cast<const uint32_t*>(p_win), layout, ld, d_matrix);
  }
  KMD_HIP(hipGetLastError());
  KMD_HIP(hipStreamSynchronize(st));                     // (scratch and the host's offsets are read until here)
  *n_rows_out = n_rows64;
  return KMD_OK;
}

} // namespace

// KmerSign::m_counts_ratio (merge.hpp:91-92) of survivors that came out of the fused merge: there is no
// matrix to gather from, so every (survivor, sample) pair looks its k-mer up in the sample's sorted
// stream.  Survivors are few: n x S binary searches.
namespace {
__global__ void __launch_bounds__(256) k_gather_from_streams(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ keys_hi,
                                                             const uint32_t* __restrict__ counts,
                                                             const uint64_t* __restrict__ offs, uint32_t S,
                                                             const uint64_t* __restrict__ row_kmer,
                                                             const uint64_t* __restrict__ row_kmer_hi,
                                                             const uint64_t* __restrict__ rows, size_t n,
                                                             double* __restrict__ out)
{
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * S) return;
  const size_t i = t / S;
  const uint32_t s = (uint32_t)(t - i * S);
  const size_t at = rows ? rows[i] : i;
  const uint64_t k = row_kmer[at], kh = keys_hi ? row_kmer_hi[at] : 0ull;
  size_t lo = (size_t)offs[s], hi = (size_t)offs[s + 1];
  const size_t end = hi;
  while (lo < hi)
  {
    const size_t mid = lo + ((hi - lo) >> 1);
    const bool less = keys_hi ? (keys_hi[mid] < kh || (keys_hi[mid] == kh && keys[mid] < k)) : keys[mid] < k;
    if (less) lo = mid + 1; else hi = mid;
  }
  out[t] = (lo < end && keys[lo] == k && (!keys_hi || keys_hi[lo] == kh)) ? (double)counts[lo] : 0.0;
}
} // namespace

extern "C" int kmd_survivors_gather_counts_streams(int n_samples, const uint64_t* d_kmers, const uint64_t* d_kmers_hi,
                                                   const uint32_t* d_counts, const uint64_t* offsets,
                                                   const uint64_t* d_row_kmer, const uint64_t* d_row_kmer_hi,
                                                   const uint64_t* d_rows, size_t n, double* d_out, void* stream)
{
  KMD_REQUIRE(n_samples > 0 && offsets, "kmd_survivors_gather_counts_streams: arguments");
  if (n == 0) return KMD_OK;
  KMD_REQUIRE(d_row_kmer && d_out && (offsets[n_samples] == 0 || (d_kmers && d_counts)), "kmd_survivors_gather_counts_streams: NULL device buffers");
  KMD_REQUIRE(!d_kmers_hi || d_row_kmer_hi, "kmd_survivors_gather_counts_streams: two-limb streams need the survivors' high limbs");
  hipStream_t st = static_cast<hipStream_t>(stream);
  scratch sc;
  void* p_offs = nullptr;
  KMD_HIP(sc.take(&p_offs, ((size_t)n_samples + 1) * 8));
  KMD_HIP(hipMemcpyAsync(p_offs, offsets, ((size_t)n_samples + 1) * 8, hipMemcpyHostToDevice, st));
  const size_t cells = n * (size_t)n_samples;
  hipLaunchKernelGGL(k_gather_from_streams, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, d_kmers, d_kmers_hi, d_counts,
                     static_cast<const uint64_t*>(p_offs), (uint32_t)n_samples, d_row_kmer, d_row_kmer_hi, d_rows, n, d_out);
  KMD_HIP(hipGetLastError());
  KMD_HIP(hipStreamSynchronize(st));                     // the offsets copy reads the caller's host array
  return KMD_OK;
}

extern "C" int kmd_merge_partition(int n_samples, const uint64_t* d_kmers, const uint64_t* d_kmers_hi,
                                   const uint32_t* d_counts, const uint64_t* offsets, int count_bytes,
                                   int layout, size_t ld, size_t row_capacity, void* d_matrix,
                                   uint64_t* d_kmer_out, uint64_t* d_kmer_hi_out, uint64_t* n_rows_out,
                                   void* stream)
{
  KMD_REQUIRE(n_samples > 0 && n_samples <= 65535 && offsets && n_rows_out, "kmd_merge_partition: arguments");
  KMD_REQUIRE(count_bytes == 1 || count_bytes == 2 || count_bytes == 4, "kmd_merge_partition: count_bytes");
  KMD_REQUIRE(kmd::layout_ok(layout), "kmd_merge_partition: layout");
  KMD_REQUIRE(layout != KMD_LAYOUT_TILED || (ld > 0 && ld % 4096 == 0), "kmd_merge_partition: tiled ld % 4096");
  const size_t n = (size_t)offsets[n_samples];
  KMD_REQUIRE(n < 0xFFFFFFFFull, "kmd_merge_partition: more than 2^32-1 records in one partition");
  for (int s = 0; s < n_samples; ++s)
    KMD_REQUIRE(offsets[s] <= offsets[s + 1], "kmd_merge_partition: offsets must be ascending");
  *n_rows_out = 0;
  if (layout == KMD_LAYOUT_SOA) KMD_REQUIRE(ld >= row_capacity, "kmd_merge_partition: SoA ld < row_capacity");
  if (layout == KMD_LAYOUT_ROWS) KMD_REQUIRE(ld >= (size_t)n_samples, "kmd_merge_partition: ld < samples");
  if (n == 0) return KMD_OK;
  KMD_REQUIRE(d_kmers && d_counts && d_matrix, "kmd_merge_partition: NULL device buffers");
  hipStream_t st = static_cast<hipStream_t>(stream);

  // the tile merge from 2^16 records up; below that (and beyond its limits): sort
  const char* force = std::getenv("KMD_MERGE_PATH");            // "sort" | "fast" | "fast-only" (tests, benchmarks)
  const bool want_fast = force ? std::strcmp(force, "sort") != 0 : n >= (1u << 16);
  if (want_fast)
  {
    bool used = false;
    int rc;
    switch (count_bytes)
    {
      case 1: rc = merge_tiles<uint8_t>(n_samples, d_kmers, d_kmers_hi, d_counts, offsets, layout, ld, row_capacity, static_cast<uint8_t*>(d_matrix), d_kmer_out, d_kmer_hi_out, n_rows_out, st, &used); break;
      case 2: rc = merge_tiles<uint16_t>(n_samples, d_kmers, d_kmers_hi, d_counts, offsets, layout, ld, row_capacity, static_cast<uint16_t*>(d_matrix), d_kmer_out, d_kmer_hi_out, n_rows_out, st, &used); break;
      default: rc = merge_tiles<uint32_t>(n_samples, d_kmers, d_kmers_hi, d_counts, offsets, layout, ld, row_capacity, static_cast<uint32_t*>(d_matrix), d_kmer_out, d_kmer_hi_out, n_rows_out, st, &used); break;
    }
    if (rc != KMD_OK || used) return rc;
    // "fast-only" (tests): report instead of quietly sorting
    KMD_REQUIRE(!(force && std::strcmp(force, "fast-only") == 0), "kmd_merge_partition: the tile merge does not take this input");
  }

  scratch sc;   // [0] vals, [1] keys sorted, [2] vals sorted, [3] flags, [4] ranks, [5] rocprim temp
  KMD_HIP(kmd::scratch_alloc(&sc.p[0], n * 8));
  KMD_HIP(kmd::scratch_alloc(&sc.p[1], n * 8));
  KMD_HIP(kmd::scratch_alloc(&sc.p[2], n * 8));
  KMD_HIP(kmd::scratch_alloc(&sc.p[3], n * 4));
  KMD_HIP(kmd::scratch_alloc(&sc.p[4], n * 4));
  uint64_t* vals = static_cast<uint64_t*>(sc.p[0]);
  uint64_t* keys_s = static_cast<uint64_t*>(sc.p[1]);
  uint64_t* vals_s = static_cast<uint64_t*>(sc.p[2]);
  uint32_t* flag = static_cast<uint32_t*>(sc.p[3]);
  uint32_t* rank = static_cast<uint32_t*>(sc.p[4]);

  for (int s = 0; s < n_samples; ++s)
  {
    const size_t b = offsets[s], e = offsets[s + 1];
    if (e > b)
      hipLaunchKernelGGL(k_tag, dim3(blocks_for(e - b)), dim3(256), 0, st, d_counts, b, e, (uint32_t)s, vals);
  }
  KMD_HIP(hipGetLastError());

  size_t tmp_sort = 0, tmp_scan = 0, tmp_sort32 = 0;
  const uint64_t* keys_hi_s = nullptr;
  KMD_HIP(rocprim::radix_sort_pairs(nullptr, tmp_sort, d_kmers, keys_s, vals, vals_s, n, 0, 64, st));
  KMD_HIP(rocprim::inclusive_scan(nullptr, tmp_scan, flag, rank, n, rocprim::plus<uint32_t>(), st));
  if (d_kmers_hi)
    KMD_HIP(rocprim::radix_sort_pairs(nullptr, tmp_sort32, d_kmers, keys_s, flag, rank, n, 0, 64, st));
  size_t tmp = tmp_sort > tmp_scan ? tmp_sort : tmp_scan;
  if (tmp_sort32 > tmp) tmp = tmp_sort32;
  KMD_HIP(kmd::scratch_alloc(&sc.p[5], tmp ? tmp : 1));
  if (!d_kmers_hi)
  {
    KMD_HIP(rocprim::radix_sort_pairs(sc.p[5], tmp_sort, d_kmers, keys_s, vals, vals_s, n, 0, 64, st));
  }
  else
  {
    // 128-bit keys (32 < k <= 64): LSD order -- stable sort by the low limb carrying the
    // record index, then stable sort by the high limb; gather everything by the result
    KMD_HIP(kmd::scratch_alloc(&sc.p[6], n * 8));     // hi gathered by perm1, later lo gathered by perm
    KMD_HIP(kmd::scratch_alloc(&sc.p[7], n * 8));     // hi sorted
    KMD_HIP(kmd::scratch_alloc(&sc.p[8], n * 4));     // perm1
    KMD_HIP(kmd::scratch_alloc(&sc.p[9], n * 4));     // perm
    uint64_t* hi_g = static_cast<uint64_t*>(sc.p[6]);
    uint64_t* hi_s = static_cast<uint64_t*>(sc.p[7]);
    uint32_t* perm1 = static_cast<uint32_t*>(sc.p[8]);
    uint32_t* perm = static_cast<uint32_t*>(sc.p[9]);
    hipLaunchKernelGGL(k_iota32, dim3(blocks_for(n)), dim3(256), 0, st, flag, n);
    KMD_HIP(rocprim::radix_sort_pairs(sc.p[5], tmp_sort32, d_kmers, keys_s, flag, perm1, n, 0, 64, st));
    hipLaunchKernelGGL(k_gather64, dim3(blocks_for(n)), dim3(256), 0, st, d_kmers_hi, perm1, n, hi_g);
    KMD_HIP(rocprim::radix_sort_pairs(sc.p[5], tmp_sort32, hi_g, hi_s, perm1, perm, n, 0, 64, st));
    hipLaunchKernelGGL(k_gather64, dim3(blocks_for(n)), dim3(256), 0, st, d_kmers, perm, n, keys_s);
    hipLaunchKernelGGL(k_gather64, dim3(blocks_for(n)), dim3(256), 0, st, vals, perm, n, vals_s);
    KMD_HIP(hipGetLastError());
    keys_hi_s = hi_s;
  }
  hipLaunchKernelGGL(k_heads, dim3(blocks_for(n)), dim3(256), 0, st, keys_s, keys_hi_s, n, flag);
  KMD_HIP(hipGetLastError());
  KMD_HIP(rocprim::inclusive_scan(sc.p[5], tmp_scan, flag, rank, n, rocprim::plus<uint32_t>(), st));
  uint32_t n_rows32 = 0;
  KMD_HIP(hipMemcpyAsync(&n_rows32, rank + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  KMD_HIP(hipStreamSynchronize(st));
  const size_t n_rows = n_rows32;
  if (n_rows > row_capacity)
  {
    *n_rows_out = n_rows;
    kmd::set_error("kmd_merge_partition: row capacity exceeded");
    return KMD_E_OVERFLOW;
  }
  // zero the part of the matrix the rows occupy, then scatter the counts
  size_t n_el;
  if (layout == KMD_LAYOUT_SOA) { KMD_REQUIRE(ld >= n_rows, "kmd_merge_partition: SoA ld < rows"); n_el = ld * (size_t)n_samples; }
  else if (layout == KMD_LAYOUT_ROWS) { KMD_REQUIRE(ld >= (size_t)n_samples, "kmd_merge_partition: ld < samples"); n_el = ld * n_rows; }
  else n_el = (n_rows + ld - 1) / ld * ld * (size_t)n_samples;
  KMD_HIP(hipMemsetAsync(d_matrix, 0, n_el * (size_t)count_bytes, st));
  switch (count_bytes)
  {
    case 1: hipLaunchKernelGGL((k_scatter<uint8_t>), dim3(blocks_for(n)), dim3(256), 0, st, keys_s, keys_hi_s, vals_s, rank, n, layout, ld, n_samples, static_cast<uint8_t*>(d_matrix), d_kmer_out, d_kmer_hi_out); break;
    case 2: hipLaunchKernelGGL((k_scatter<uint16_t>), dim3(blocks_for(n)), dim3(256), 0, st, keys_s, keys_hi_s, vals_s, rank, n, layout, ld, n_samples, static_cast<uint16_t*>(d_matrix), d_kmer_out, d_kmer_hi_out); break;
    default: hipLaunchKernelGGL((k_scatter<uint32_t>), dim3(blocks_for(n)), dim3(256), 0, st, keys_s, keys_hi_s, vals_s, rank, n, layout, ld, n_samples, static_cast<uint32_t*>(d_matrix), d_kmer_out, d_kmer_hi_out); break;
  }
  KMD_HIP(hipGetLastError());
  KMD_HIP(hipStreamSynchronize(st));
  *n_rows_out = n_rows;
  return KMD_OK;
}
