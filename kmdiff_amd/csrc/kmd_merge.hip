// kmd_merge.hip -- K2: the k-way merge of one partition's per-sample k-mer streams into the
// merged count matrix, on the device.
//
// Replaces km::KmerMerger<KSIZE,CMAX>::merge as kmdiff drives it (include/kmdiff/merge.hpp:
// 265-289: paths of one partition, abundance minima all 1, recurrence minimum 1, save_if 0 =>
// every distinct k-mer is emitted, ascending, with the count of each sample or 0).  kmtricks
// itself is not part of the reference tree (empty submodule); the contract restated here is
// the one SURVEY.md 8a R1 derives from the call site and the fixture bytes.
//
// Round-1 form: correct and device-resident, NOT yet a tuned kernel.  The S sorted streams are
// tagged with their sample id, radix-sorted together (rocPRIM), run heads are flagged and
// scanned into row numbers, and a scatter kernel writes the matrix in the layout K1 wants.
// The sort ignores that the inputs are already sorted; the bucketed LDS merge that uses it
// (sampled splitters -> one workgroup merges one key range in LDS) is the planned
// replacement and keeps this interface.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string.h>

#include "kmd_internal.h"

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
// Fast path (one 64-bit limb, enough records): key-range buckets merged in LDS.
//
//   pass 1  k_bucket_starts : one streaming pass over the keys; because every stream is sorted,
//           the first record of bucket j in stream s is where bucket(key) changes: a
//           [bucket][sample] table of start offsets, no searching.
//   pass 2  k_bucket_merge<false> : one wave per bucket inserts the bucket's keys (S short
//           segments, <= kWaveCap records) into an LDS hash set -> number of distinct k-mers.
//           (exclusive scan over buckets -> first row of every bucket)
//   pass 3  k_bucket_merge<true>  : the same hash set again, distinct keys compacted and sorted
//           in LDS (bitonic), k-mer column written, every record binary-searches its row and
//           scatters its count into the zero-filled matrix.
// Buckets are equal slices of [min key, max key]; a bucket holding more than kWaveCap records
// (heavily clustered keys) raises a flag and the caller falls back to the sort-based path.
constexpr uint64_t kEmpty = ~0ull;

struct bucket_map { uint64_t kmin; uint32_t shift; uint32_t nb; };

__device__ __forceinline__ uint32_t bucket_of(const bucket_map& B, uint64_t key)
{
  return (uint32_t)((key - B.kmin) >> B.shift);
}

// start[j * S + s] = index of the first record of stream s whose bucket is >= j   (j in [0, nb]);
// one launch for all streams: blockIdx.y = stream, grid-stride over its records
__global__ void __launch_bounds__(256) k_bucket_starts(const uint64_t* __restrict__ keys,
                                                       const uint64_t* __restrict__ offs, uint32_t S,
                                                       bucket_map B, uint32_t* __restrict__ start)
{
  const uint32_t s = blockIdx.y;
  const size_t begin = offs[s], end = offs[s + 1];
  if (begin == end)
  {
    // empty stream: every bucket starts (and ends) at its offset
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j <= B.nb; j += (size_t)gridDim.x * blockDim.x)
      start[j * S + s] = (uint32_t)begin;
    return;
  }
  for (size_t i = begin + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < end; i += (size_t)gridDim.x * blockDim.x)
  {
    const uint32_t bi = bucket_of(B, keys[i]);
    const int64_t bprev = (i == begin) ? -1 : (int64_t)bucket_of(B, keys[i - 1]);
    for (int64_t j = bprev + 1; j <= (int64_t)bi; ++j) start[(size_t)j * S + s] = (uint32_t)i;
    if (i == end - 1)
      for (uint32_t j = bi + 1; j <= B.nb; ++j) start[(size_t)j * S + s] = (uint32_t)end;
  }
}

__device__ __forceinline__ uint32_t hash_slot(uint64_t k)
{
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 29;
  return (uint32_t)k;
}

// One WAVE per bucket: a bucket is small (~128 records in S short segments), so a workgroup
// per bucket spends its time in barriers and dependent-load latency.  A wave needs no
// workgroup barrier (its LDS operations execute in order), and 14 KB of LDS per wave keeps
// ~10 buckets in flight per CU.
constexpr uint32_t kMaxFastSamples = 256;    // segment tables of one bucket live in LDS
constexpr uint32_t kWaveCap = 512;           // records per bucket
constexpr uint32_t kWaveSlots = 1024;        // hash slots
constexpr int kWavesPerBlock = 2;

__device__ __forceinline__ void wave_sync()
{
  // LDS traffic of one wave is in order; this only stops the compiler from moving or caching
  // LDS accesses across the point where other lanes' values are consumed
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <bool EMIT, typename CT>
__global__ void __launch_bounds__(64 * kWavesPerBlock) k_bucket_merge(const uint64_t* __restrict__ keys,
                                                              const uint32_t* __restrict__ counts,
                                                              const uint32_t* __restrict__ start, uint32_t S,
                                                              uint32_t nb, uint32_t* __restrict__ distinct,
                                                              const uint32_t* __restrict__ row_base,
                                                              int layout, size_t ld, CT* __restrict__ matrix,
                                                              uint64_t* __restrict__ kmer_out,
                                                              uint32_t* __restrict__ overflow)
{
  __shared__ unsigned long long s_hash_all[kWavesPerBlock][kWaveSlots];
  __shared__ unsigned long long s_keys_all[kWavesPerBlock][EMIT ? kWaveCap : 1];
  __shared__ uint32_t s_beg_all[kWavesPerBlock][kMaxFastSamples];
  __shared__ uint32_t s_pref_all[kWavesPerBlock][kMaxFastSamples + 1];
  constexpr uint32_t cmax = sizeof(CT) == 1 ? 0xFFu : sizeof(CT) == 2 ? 0xFFFFu : 0xFFFFFFFFu;
  const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned long long* s_hash = s_hash_all[w];
  unsigned long long* s_keys = s_keys_all[w];
  uint32_t* s_beg = s_beg_all[w];
  uint32_t* pref = s_pref_all[w];
  const uint32_t n_waves = gridDim.x * kWavesPerBlock;

  for (uint32_t j = blockIdx.x * kWavesPerBlock + w; j < nb; j += n_waves)
  {
    // ---- the S segments of this bucket; exclusive prefix of their lengths (4 samples per lane)
    uint32_t len[4], lsum = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
    {
      const uint32_t s = lane * 4 + q;
      len[q] = 0;
      if (s < S)
      {
        const uint32_t b = start[(size_t)j * S + s];
        s_beg[s] = b;
        len[q] = start[(size_t)(j + 1) * S + s] - b;
      }
      lsum += len[q];
    }
    uint32_t incl = lsum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1)
    {
      const uint32_t up = __shfl_up(incl, o, 64);
      if ((int)lane >= o) incl += up;
    }
    uint32_t run = incl - lsum;                         // records before this lane's first sample
#pragma unroll
    for (int q = 0; q < 4; ++q)
    {
      const uint32_t s = lane * 4 + q;
      if (s < S) pref[s] = run;
      run += len[q];
    }
    if (lane == 63) pref[S] = incl;                     // pref[S] = n
    const uint32_t n = __shfl(incl, 63, 64);
    if (n == 0) { if (!EMIT && lane == 0) distinct[j] = 0; continue; }
    if (n > kWaveCap)
    {
      if (lane == 0) { atomicAdd(overflow, 1u); if (!EMIT) distinct[j] = 0; }
      continue;
    }
    // ---- hash set sized to the bucket (power of two >= 2n)
    uint32_t slots = 64;
    while (slots < 2 * n) slots <<= 1;
    const uint32_t mask = slots - 1;
    for (uint32_t t = lane; t < slots; t += 64) s_hash[t] = kEmpty;
    wave_sync();
    uint32_t d = 0;
    bool has_max_key = false;                           // the key equal to the empty marker, if present
    for (uint32_t f0 = 0; f0 < n; f0 += 64)
    {
      const uint32_t f = f0 + lane;
      bool fresh = false;
      if (f < n)
      {
        uint32_t lo = 0, hi = S;                        // last q with pref[q] <= f
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (pref[mid] <= f) lo = mid; else hi = mid; }
        const uint64_t k = keys[s_beg[lo] + (f - pref[lo])];
        if (k == kEmpty) has_max_key = true;
        else
        {
          uint32_t h = hash_slot(k) & mask;
          for (;;)
          {
            const unsigned long long old = atomicCAS(&s_hash[h], kEmpty, (unsigned long long)k);
            if (old == kEmpty) { fresh = true; break; }
            if (old == k) break;
            h = (h + 1) & mask;
          }
        }
      }
      d += (uint32_t)__popcll(__ballot(fresh));
    }
    const bool any_max = __ballot(has_max_key) != 0;
    d += any_max ? 1u : 0u;
    if (!EMIT) { if (lane == 0) distinct[j] = d; continue; }

    // ---- EMIT: compact the distinct keys (ballot prefix), sort them, write the k-mer column
    wave_sync();
    uint32_t filled = 0;
    for (uint32_t t0 = 0; t0 < slots; t0 += 64)
    {
      const unsigned long long k = s_hash[t0 + lane];
      const bool occ = k != kEmpty;
      const unsigned long long m = __ballot(occ);
      if (occ) s_keys[filled + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = k;
      filled += (uint32_t)__popcll(m);
    }
    if (any_max && lane == 0) s_keys[filled] = kEmpty;              // sorts last
    uint32_t p2 = 1;
    while (p2 < d) p2 <<= 1;
    for (uint32_t t = d + lane; t < p2; t += 64) s_keys[t] = kEmpty;      // padding (>= every key)
    wave_sync();
    for (uint32_t k2 = 2; k2 <= p2; k2 <<= 1)
      for (uint32_t j2 = k2 >> 1; j2 > 0; j2 >>= 1)
      {
        for (uint32_t t = lane; t < p2; t += 64)
        {
          const uint32_t l = t ^ j2;
          if (l > t)
          {
            const unsigned long long a = s_keys[t], b = s_keys[l];
            const bool up = ((t & k2) == 0);
            if ((a > b) == up) { s_keys[t] = b; s_keys[l] = a; }
          }
        }
        wave_sync();
      }
    const uint32_t rb = row_base[j];
    if (kmer_out)
      for (uint32_t t = lane; t < d; t += 64) kmer_out[(size_t)rb + t] = s_keys[t];
    // ---- the bucket's d x S block of the matrix is assembled in LDS (the hash set's memory,
    // free now) and written out whole: no zero-fill pass over the matrix, no 4-byte scatter;
    // a block too large for LDS is zero-filled and scattered in place
    const uint32_t cells = d * S;
    const bool in_lds = (size_t)cells * sizeof(CT) <= sizeof(unsigned long long) * kWaveSlots;
    CT* tile = reinterpret_cast<CT*>(s_hash);                        // [sample][row in bucket]
    if (in_lds)
      for (uint32_t t = lane; t < cells; t += 64) tile[t] = (CT)0;
    else
      for (uint32_t t = lane; t < cells; t += 64)
        matrix[kmd::count_index(layout, ld, (int)S, (size_t)rb + (t % d), (int)(t / d))] = (CT)0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");           // zero-fill before the scatter (global case)
    wave_sync();
    for (uint32_t f = lane; f < n; f += 64)
    {
      uint32_t lo = 0, hi = S;
      while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (pref[mid] <= f) lo = mid; else hi = mid; }
      const uint32_t i = s_beg[lo] + (f - pref[lo]);
      const uint64_t k = keys[i];
      uint32_t a = 0, b = d;                           // first index with s_keys[idx] >= k
      while (a < b) { const uint32_t mid = (a + b) >> 1; if (s_keys[mid] < k) a = mid + 1; else b = mid; }
      uint32_t c = counts[i];
      if (c > cmax) c = cmax;
      if (in_lds) tile[lo * d + a] = (CT)c;
      else matrix[kmd::count_index(layout, ld, (int)S, (size_t)rb + a, (int)lo)] = (CT)c;
    }
    wave_sync();
    if (in_lds)
      for (uint32_t t = lane; t < cells; t += 64)
        matrix[kmd::count_index(layout, ld, (int)S, (size_t)rb + (t % d), (int)(t / d))] = tile[t];
    wave_sync();
  }
}

// min of the first keys / max of the last keys of the non-empty streams
__global__ void k_key_range(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ offs, uint32_t S,
                            uint64_t* __restrict__ out)
{
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint64_t lo = ~0ull, hi = 0;
  for (uint32_t s = 0; s < S; ++s)
    if (offs[s + 1] > offs[s])
    {
      const uint64_t a = keys[offs[s]], b = keys[offs[s + 1] - 1];
      if (a < lo) lo = a;
      if (b > hi) hi = b;
    }
  out[0] = lo; out[1] = hi;
}


struct scratch
{
  void* p[10] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
  ~scratch() { for (void* q : p) if (q) kmd::scratch_free(q); }
};

// The bucketed LDS merge.  *used = false (and nothing written) when the input does not suit
// it (clustered keys overflow a bucket): the caller then takes the sort-based path.
template <typename CT>
int merge_fast(int S, const uint64_t* d_kmers, const uint32_t* d_counts, const uint64_t* offsets,
               int layout, size_t ld, size_t row_capacity, CT* d_matrix, uint64_t* d_kmer_out,
               uint64_t* n_rows_out, int n_cu, hipStream_t st, bool* used)
{
  *used = false;
  const size_t n = (size_t)offsets[S];
  const bool dbg = std::getenv("KMD_DEBUG") != nullptr;
#define KMD_DBG(msg) do { if (dbg) { hipError_t e_ = hipStreamSynchronize(st); std::fprintf(stderr, "[merge_fast] %s: %s\n", msg, hipGetErrorString(e_)); } } while (0)
  scratch sc;   // [0] device offsets + key range, [1] start table, [2] distinct, [3] row_base, [4] overflow, [5] temp
  KMD_HIP(kmd::scratch_alloc(&sc.p[0], ((size_t)S + 1 + 2) * 8));
  uint64_t* d_offs = static_cast<uint64_t*>(sc.p[0]);
  uint64_t* d_range = d_offs + S + 1;
  KMD_HIP(hipMemcpyAsync(d_offs, offsets, ((size_t)S + 1) * 8, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_key_range, dim3(1), dim3(64), 0, st, d_kmers, d_offs, (uint32_t)S, d_range);
  uint64_t range[2];
  KMD_HIP(hipMemcpyAsync(range, d_range, 16, hipMemcpyDeviceToHost, st));
  KMD_HIP(hipStreamSynchronize(st));
  const uint64_t span = range[1] - range[0];                    // kmax - kmin
  // ~128 records per bucket on average (one wave each); the start table is capped at 1 GiB
  uint64_t nb_target = n / 128 + 1;
  const uint64_t table_cap = (1ull << 30) / (4ull * (uint64_t)S);
  if (nb_target > table_cap) nb_target = table_cap;
  bucket_map B;
  B.kmin = range[0]; B.shift = 0;
  while (B.shift < 63 && (span >> B.shift) >= nb_target) ++B.shift;    // (span >> shift) + 1 <= nb_target, no overflow
  B.nb = (uint32_t)((span >> B.shift) + 1);
  const size_t nb = B.nb;
  if (dbg) std::fprintf(stderr, "[merge_fast] n=%zu kmin=%llu kmax=%llu shift=%u nb=%zu\n", n,
                        (unsigned long long)range[0], (unsigned long long)range[1], B.shift, nb);

  KMD_HIP(kmd::scratch_alloc(&sc.p[1], (nb + 1) * (size_t)S * 4));
  KMD_HIP(kmd::scratch_alloc(&sc.p[2], nb * 4));
  KMD_HIP(kmd::scratch_alloc(&sc.p[3], nb * 4));
  KMD_HIP(kmd::scratch_alloc(&sc.p[4], 4));
  uint32_t* start = static_cast<uint32_t*>(sc.p[1]);
  uint32_t* distinct = static_cast<uint32_t*>(sc.p[2]);
  uint32_t* row_base = static_cast<uint32_t*>(sc.p[3]);
  uint32_t* overflow = static_cast<uint32_t*>(sc.p[4]);
  KMD_HIP(hipMemsetAsync(overflow, 0, 4, st));
  {
    size_t longest = 1;
    for (int s = 0; s < S; ++s) if (offsets[s + 1] - offsets[s] > longest) longest = offsets[s + 1] - offsets[s];
    if (nb + 1 > longest) longest = nb + 1;
    size_t gx = (longest + 255) / 256;
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(k_bucket_starts, dim3((unsigned)gx, (unsigned)S), dim3(256), 0, st, d_kmers, d_offs, (uint32_t)S, B, start);
  }
  KMD_HIP(hipGetLastError());
  KMD_DBG("starts");
  size_t grid = (size_t)n_cu * 16;
  if (grid > (nb + kWavesPerBlock - 1) / kWavesPerBlock) grid = (nb + kWavesPerBlock - 1) / kWavesPerBlock;
  hipLaunchKernelGGL((k_bucket_merge<false, CT>), dim3((unsigned)grid), dim3(64 * kWavesPerBlock), 0, st, d_kmers, d_counts,
                     start, (uint32_t)S, (uint32_t)nb, distinct, (const uint32_t*)nullptr, layout, ld,
                     (CT*)nullptr, (uint64_t*)nullptr, overflow);
  KMD_HIP(hipGetLastError());
  KMD_DBG("pass2");
  size_t tmp = 0;
  KMD_HIP(rocprim::exclusive_scan(nullptr, tmp, distinct, row_base, 0u, nb, rocprim::plus<uint32_t>(), st));
  KMD_HIP(kmd::scratch_alloc(&sc.p[5], tmp ? tmp : 1));
  KMD_HIP(rocprim::exclusive_scan(sc.p[5], tmp, distinct, row_base, 0u, nb, rocprim::plus<uint32_t>(), st));
  uint32_t h[3] = { 0, 0, 0 };
  KMD_HIP(hipMemcpyAsync(&h[0], overflow, 4, hipMemcpyDeviceToHost, st));
  KMD_HIP(hipMemcpyAsync(&h[1], row_base + (nb - 1), 4, hipMemcpyDeviceToHost, st));
  KMD_HIP(hipMemcpyAsync(&h[2], distinct + (nb - 1), 4, hipMemcpyDeviceToHost, st));
  KMD_HIP(hipStreamSynchronize(st));
  if (h[0]) return KMD_OK;                                       // a bucket overflowed: not used
  const size_t n_rows = (size_t)h[1] + h[2];
  *used = true;
  *n_rows_out = n_rows;
  if (n_rows > row_capacity) { kmd::set_error("kmd_merge_partition: row capacity exceeded"); return KMD_E_OVERFLOW; }
  size_t n_el;
  if (layout == KMD_LAYOUT_SOA) { KMD_REQUIRE(ld >= n_rows, "kmd_merge_partition: SoA ld < rows"); n_el = ld * (size_t)S; }
  else if (layout == KMD_LAYOUT_ROWS) { KMD_REQUIRE(ld >= (size_t)S, "kmd_merge_partition: ld < samples"); n_el = ld * n_rows; }
  else n_el = (n_rows + ld - 1) / ld * ld * (size_t)S;
  (void)n_el;   // every cell of the n_rows rows is written by its bucket's workgroup: no zero-fill pass
  hipLaunchKernelGGL((k_bucket_merge<true, CT>), dim3((unsigned)grid), dim3(64 * kWavesPerBlock), 0, st, d_kmers, d_counts,
                     start, (uint32_t)S, (uint32_t)nb, distinct, (const uint32_t*)row_base, layout, ld, d_matrix,
                     d_kmer_out, overflow);
  KMD_HIP(hipGetLastError());
  KMD_HIP(hipStreamSynchronize(st));
  return KMD_OK;
}

} // namespace

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
  if (n == 0) return KMD_OK;
  KMD_REQUIRE(d_kmers && d_counts && d_matrix, "kmd_merge_partition: NULL device buffers");
  hipStream_t st = static_cast<hipStream_t>(stream);

  // bucketed LDS merge when it applies (one limb, enough records); else / on overflow: sort
  const char* force = std::getenv("KMD_MERGE_PATH");            // "sort" | "fast" (tests, benchmarks)
  const bool want_fast = !d_kmers_hi && (uint32_t)n_samples <= kMaxFastSamples &&
                         (force ? std::strcmp(force, "sort") != 0 : n >= (1u << 16));
  if (want_fast)
  {
    int dev = 0, n_cu = 256;
    KMD_HIP(hipGetDevice(&dev));
    KMD_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    bool used = false;
    int rc;
    switch (count_bytes)
    {
      case 1: rc = merge_fast<uint8_t>(n_samples, d_kmers, d_counts, offsets, layout, ld, row_capacity, static_cast<uint8_t*>(d_matrix), d_kmer_out, n_rows_out, n_cu, st, &used); break;
      case 2: rc = merge_fast<uint16_t>(n_samples, d_kmers, d_counts, offsets, layout, ld, row_capacity, static_cast<uint16_t*>(d_matrix), d_kmer_out, n_rows_out, n_cu, st, &used); break;
      default: rc = merge_fast<uint32_t>(n_samples, d_kmers, d_counts, offsets, layout, ld, row_capacity, static_cast<uint32_t*>(d_matrix), d_kmer_out, n_rows_out, n_cu, st, &used); break;
    }
    if (rc != KMD_OK || used) return rc;
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
