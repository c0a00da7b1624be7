// kmd_merge.hip -- K2: the k-way merge of one partition's per-sample k-mer streams into the
// merged count matrix, on the device.
//
// Replaces km::KmerMerger<KSIZE,CMAX>::merge as kmdiff drives it (include/kmdiff/merge.hpp:
// 265-289: paths of one partition, abundance minima all 1, recurrence minimum 1, save_if 0 =>
// every distinct k-mer is emitted, ascending, with the count of each sample or 0).  kmtricks
// itself is not part of the reference tree (empty submodule); the contract restated here is
// the one SURVEY.md 8a R1 derives from the call site and the fixture bytes.
//
// Two device implementations behind kmd_merge_partition (bottom of the file):
//   * the bucketed LDS merge (<= 256 samples, one or two 64-bit limbs): uses that the inputs
//     are sorted -- key-range buckets, one wave per bucket, one pass (second half of the file);
//   * the sort-based merge (tiny inputs, mostly-clustered keys, > 256 samples): records
//     tagged with their sample id, radix-sorted together (rocPRIM), run heads flagged and
//     scanned into row numbers, a scatter kernel writes the matrix (first half of the file).
// Both write the layout K1 wants and the sorted k-mer column.
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
// Bucketed LDS merge (enough records).
//
//   k_splitter_coarse / buckets = runs between splitters taken from the data (every r-th key of the
//   k_splitter_fine     longest stream): where each stream enters each bucket, a table of start
//                       offsets (stream-major).  KMD_MERGE_SLICES=1: k_bucket_starts instead --
//                       buckets = equal slices of the key range, one streaming pass, no searching
//   k_transpose_starts  -> [bucket][sample]: the S offsets of a bucket in one contiguous span
//   k_bucket_split /    buckets holding more records than a wave takes (random keys: Poisson
//   k_refine_starts     sizes; real partitions cluster) are cut again, level by level, on the
//                       start table alone
//   k_bucket_merge      one WAVE per bucket, one pass: records -> registers, LDS hash set ->
//                       distinct count (published), rank by counting; first row by a two-level
//                       decoupled look-back; the bucket's d x S block assembled in LDS and
//                       written whole.  Stage A of the next bucket runs before stage B of the
//                       current one (software pipeline, see the kernel).
// A table that grows 2.5-fold under the cuts, or a bucket still over capacity after kMaxLevels,
// hands over to the sort-based path.
constexpr uint64_t kEmpty = ~0ull;

// bucket(key) = floor((key - kmin) * nb / (span + 1)) as a 64x64 -> high-64 multiply: equal
// slices of the key range for ANY bucket count (a power-of-two slice width would make the
// average bucket anything between 1x and 2x the target)
struct bucket_map { uint64_t kmin; uint64_t mult; uint32_t nb; };

__device__ __forceinline__ uint32_t bucket_of(const bucket_map& B, uint64_t key)
{
  return (uint32_t)__umul64hi(key - B.kmin, B.mult);
}

// start[s * (nb + 1) + j] = index of the first record of stream s whose bucket is >= j
// (j in [0, nb]; stream-major: the writes of a stream are consecutive);
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
      start[(size_t)s * (B.nb + 1) + j] = (uint32_t)begin;
    return;
  }
  for (size_t i = begin + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < end; i += (size_t)gridDim.x * blockDim.x)
  {
    const uint32_t bi = bucket_of(B, keys[i]);
    const int64_t bprev = (i == begin) ? -1 : (int64_t)bucket_of(B, keys[i - 1]);
    for (int64_t j = bprev + 1; j <= (int64_t)bi; ++j) start[(size_t)s * (B.nb + 1) + (size_t)j] = (uint32_t)i;
    if (i == end - 1)
      for (uint32_t j = bi + 1; j <= B.nb; ++j) start[(size_t)s * (B.nb + 1) + j] = (uint32_t)end;
  }
}

// Data-adaptive buckets: boundary j = every r-th key of the LONGEST stream (b_0 = -inf, b_nb = +inf),
// so buckets are narrow where k-mers are dense and wide where they are sparse -- equal slices of
// the key range fit evenly spread keys only.  start[s][j] = first record of stream s with key >=
// b_j, in two steps: every 64th boundary by binary search over the stream, the ones in between
// inside the short run of records those enclose.
constexpr uint32_t kSplitChunk = 64;                      // boundaries per wave

// every kSplitChunk-th boundary by binary search over the whole stream: coarse[s][c]
__global__ void __launch_bounds__(256) k_splitter_coarse(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ offs,
                                                         uint32_t S, uint32_t L, uint32_t r, uint32_t nb, uint32_t n_chunks,
                                                         uint32_t* __restrict__ coarse)
{
  const uint32_t s = blockIdx.y;
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c > n_chunks) return;
  const size_t begin = offs[s], end = offs[s + 1], j = c * kSplitChunk;
  size_t pos = begin;
  if (j >= nb) pos = end;
  else if (j > 0)
  {
    const uint64_t b = keys[offs[L] + j * r];
    size_t lo = begin, hi = end;                         // first index in [begin, end) with key >= b
    while (lo < hi) { const size_t mid = lo + ((hi - lo) >> 1); if (keys[mid] < b) lo = mid + 1; else hi = mid; }
    pos = lo;
  }
  coarse[(size_t)s * (n_chunks + 1) + c] = (uint32_t)pos;
}

// the boundaries in between: one wave per (chunk, stream) loads the chunk's records -- a short
// contiguous run of the stream, coalesced -- into LDS, and every lane places its boundary in it
__global__ void __launch_bounds__(256) k_splitter_fine(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ offs,
                                                       uint32_t S, uint32_t L, uint32_t r, uint32_t nb, uint32_t n_chunks,
                                                       const uint32_t* __restrict__ coarse, uint32_t* __restrict__ start)
{
  __shared__ unsigned long long s_win_all[4][256];
  const uint32_t s = blockIdx.y, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t c = (size_t)blockIdx.x * 4 + w;
  if (c >= n_chunks) return;
  unsigned long long* win = s_win_all[w];
  const size_t j = c * kSplitChunk + lane;
  const size_t p0 = coarse[(size_t)s * (n_chunks + 1) + c], p1 = coarse[(size_t)s * (n_chunks + 1) + c + 1];
  const bool inner = j > 0 && j < nb;
  const uint64_t b = inner ? keys[offs[L] + j * r] : 0ull;
  size_t below = 0;                                      // records of [p0, p1) below this lane's boundary
  for (size_t w0 = p0; w0 < p1; w0 += 256)               // (wave-uniform)
  {
    const uint32_t m = (uint32_t)((p1 - w0) < 256 ? (p1 - w0) : 256);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    for (uint32_t t = lane; t < m; t += 64) win[t] = keys[w0 + t];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint32_t lo = 0, hi = m;                             // first index of the window with key >= b
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (win[mid] < b) lo = mid + 1; else hi = mid; }
    below += lo;
  }
  if (j <= nb) start[(size_t)s * (nb + 1) + j] = (uint32_t)(j == 0 ? offs[s] : j == nb ? offs[s + 1] : p0 + below);
}

// [S][nb + 1] (what k_bucket_starts writes, coalesced) -> [nb + 1][S] (what a bucket reads: its S
// start offsets in one contiguous span), 64 x 64 tiles through LDS
__global__ void __launch_bounds__(256) k_transpose_starts(const uint32_t* __restrict__ sm, uint32_t S, uint32_t nb1,
                                                          uint32_t* __restrict__ start)
{
  __shared__ uint32_t tile[64][65];
  const uint32_t tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const size_t j0 = (size_t)blockIdx.x * 64;
  for (uint32_t s0 = 0; s0 < S; s0 += 64)
  {
    for (uint32_t ss = ty; ss < 64; ss += 4)
      if (s0 + ss < S && j0 + tx < nb1) tile[ss][tx] = sm[(size_t)(s0 + ss) * nb1 + j0 + tx];
    __syncthreads();
    for (uint32_t jj = ty; jj < 64; jj += 4)
      if (s0 + tx < S && j0 + jj < nb1) start[(j0 + jj) * S + s0 + tx] = tile[tx][jj];
    __syncthreads();
  }
}

// Keys are not spread evenly over their range (k-mers of a partition cluster), so some of the
// equal key slices hold more records than a wave can take.  Those buckets are cut again, on the
// start table alone, into equal slices of the key range the bucket's records REALLY span (a dense
// cluster inside a wide slice gets a fine grid of its own); repeated by the host until every
// bucket fits.  split[j] = slices bucket j becomes (1 = kept), klo/kstep[j] = first key and slice
// width of a cut bucket, counters[0] += buckets over capacity.
// The table is read as start[j * js + s * ss]: (S, 1) for the bucket-major table, (1, nb + 1)
// for the stream-major one k_bucket_starts wrote -- one thread per bucket, so the stream-major
// form (level 0, every bucket) is the coalesced one.
__global__ void __launch_bounds__(256) k_bucket_split(const uint64_t* __restrict__ keys,
                                                      const uint32_t* __restrict__ start, size_t js, size_t ss,
                                                      uint32_t S, uint32_t nb, uint32_t cap,
                                                      uint32_t* __restrict__ split, uint64_t* __restrict__ klo,
                                                      uint64_t* __restrict__ kstep, uint32_t* __restrict__ counters,
                                                      int arith, bucket_map B, uint64_t wq, uint64_t wr)
{
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nb) return;
  uint32_t n = 0;
  for (uint32_t s = 0; s < S; ++s) n += start[(j + 1) * js + s * ss] - start[j * js + s * ss];
  uint32_t m = 1;
  if (n > cap && arith && n <= 8 * cap)
  {
    // level 0, a bucket moderately over (the Poisson tail of evenly spread keys): the bucket IS an
    // equal slice of the key range -- cut that slice (no key reads).  A bucket far over capacity
    // holds a dense cluster: that one is cut by the range its records span (below) right away.
    // bucket_of puts the keys from kmin + j 2^64 / mult on into bucket j; 2^64 = wq mult + wr.
    m = (n + cap / 4 - 1) / (cap / 4);
    klo[j] = B.kmin + (uint64_t)j * wq + ((uint64_t)j * wr) / B.mult;
    kstep[j] = wq / m + 1;
    atomicAdd(counters, 1u);
  }
  else if (n > cap)
  {
    uint64_t lo = ~0ull, hi = 0;
    for (uint32_t s = 0; s < S; ++s)
    {
      const uint32_t b = start[j * js + s * ss], e = start[(j + 1) * js + s * ss];
      if (e > b)
      {
        const uint64_t kb = keys[b], ke = keys[e - 1];
        lo = kb < lo ? kb : lo;
        hi = ke > hi ? ke : hi;
      }
    }
    m = (n + cap / 4 - 1) / (cap / 4);
    klo[j] = lo;
    kstep[j] = (hi - lo) / m + 1;                       // m slices of this width cover [lo, hi]
    atomicAdd(counters, 1u);
  }
  split[j] = m;
}

// The refined start table: row first[j] + t = slice t of old bucket j (first = exclusive prefix
// of split); a kept bucket copies its row, a cut one searches its short segments for the slice
// boundaries klo + t * kstep (saturating: the same monotone rule for every stream is all it takes)
// and adds its share to the record counts of the slices (child_n, zeroed by the caller)
__global__ void __launch_bounds__(256) k_refine_starts(const uint64_t* __restrict__ keys,
                                                       const uint32_t* __restrict__ start, uint32_t S, uint32_t nb,
                                                       const uint32_t* __restrict__ split,
                                                       const uint64_t* __restrict__ klo,
                                                       const uint64_t* __restrict__ kstep,
                                                       const uint32_t* __restrict__ first, uint32_t nb_new,
                                                       uint32_t* __restrict__ out, uint32_t* __restrict__ child_n)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ((size_t)nb + 1) * S) return;
  const size_t j = i / S;
  const uint32_t s = (uint32_t)(i % S);
  if (j == nb) { out[(size_t)nb_new * S + s] = start[i]; return; }
  const uint32_t m = split[j], beg = start[i];
  const size_t row = first[j];
  out[row * S + s] = beg;
  if (m == 1) return;
  const uint32_t end = start[i + S];
  const uint64_t k0 = klo[j], step = kstep[j];
  uint32_t lo = beg, prev = beg;
  for (uint32_t t = 1; t < m; ++t)
  {
    uint64_t bound = k0 + (uint64_t)t * step;
    if (__umul64hi((uint64_t)t, step) != 0 || bound < k0) bound = ~0ull;
    uint32_t hi = end;                                   // first record in [lo, end) with key >= bound
    while (lo < hi)
    {
      const uint32_t mid = lo + ((hi - lo) >> 1);
      if (keys[mid] < bound) lo = mid + 1; else hi = mid;
    }
    out[(row + t) * S + s] = lo;
    if (lo > prev) atomicAdd(child_n + row + t - 1, lo - prev);     // records of this stream in slice t - 1
    prev = lo;
  }
  if (end > prev) atomicAdd(child_n + row + m - 1, end - prev);
}

// pieces of cut buckets that are still over capacity (child_n is zero for kept buckets): the check
// that would otherwise be another pass over the whole start table
__global__ void __launch_bounds__(256) k_count_over(const uint32_t* __restrict__ child_n, uint32_t nb, uint32_t cap,
                                                    uint32_t* __restrict__ counter)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nb && child_n[i] > cap) atomicAdd(counter, 1u);
}

// keys of one bucket differ in their low ~40 bits: fold them to 32 and take the TOP bits of a
// multiplicative hash (one 32-bit multiply; the 64-bit finaliser cost 3x the instructions)
__device__ __forceinline__ uint32_t hash_slot(uint64_t k)
{
  return ((uint32_t)k ^ (uint32_t)(k >> 29)) * 0x9E3779B1u;
}

// One WAVE per bucket: a bucket is small (~128 records in S short segments), so a workgroup
// per bucket spends its time in barriers and dependent-load latency.  A wave needs no
// workgroup barrier (its LDS operations execute in order), and 8 KB of LDS per wave keeps
// ~20 buckets in flight per CU: the kernel is bound by dependent-load latency per bucket.
constexpr uint32_t kMaxFastSamples = 256;    // segment tables of one bucket live in LDS
// CAP = records a bucket may hold (hash slots = 2 CAP; the average bucket is CAP / 2, see
// merge_fast), WPB = waves per workgroup: template parameters of the kernel, picked by sample count

__device__ __forceinline__ void wave_sync()
{
  // LDS traffic of one wave is in order; this only stops the compiler from moving or caching
  // LDS accesses across the point where other lanes' values are consumed
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Single-pass row numbering, two-level decoupled look-back.  A persistent grid keeps thousands of
// buckets in flight in near lock-step, so a flat look-back would have to add up thousands of
// "own count" words per bucket; instead buckets are grouped 64 at a time:
//   status[j]        bit 63 = published, bits 31..0 = distinct keys of bucket j
//   group[g].agg     bits 63..56 = buckets of group g that have published, bits 55..0 = their sum
//                    (one atomicAdd per bucket: count and sum can never be seen out of step)
//   group[g].base    bit 63 = published, bits 62..0 = rows in groups 0..g-1
// The wave that owns a group's first bucket walks back over the earlier groups (64 per load) to the
// nearest published base and publishes its own group's base; the other 63 buckets of the group
// only read that one word plus the own counts of the group's earlier buckets.  One group per
// 128-byte line, so the polling of the few dozen groups in flight spreads over the L2 channels.
constexpr unsigned long long kStFlag = 1ull << 63, kStMask = kStFlag - 1;
constexpr int kGroupShift = 56;
constexpr unsigned long long kGroupSumMask = (1ull << kGroupShift) - 1;
struct alignas(128) merge_group { unsigned long long agg, base, pad[14]; };

__device__ __forceinline__ unsigned long long wave_sum64(unsigned long long x)
{
  for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
  return __shfl(x, 0, 64);
}

__device__ __forceinline__ unsigned long long rows_before(const unsigned long long* status, merge_group* group,
                                                          uint32_t j, uint32_t lane)
{
#ifdef KMD_MERGE_FAKE_LOOKBACK   // dev experiment: what the kernel costs without its look-back (rows are wrong)
  return (unsigned long long)j * 5ull;
#endif
  const uint32_t g = j >> 6, r = j & 63;
  unsigned long long base = 0;
  if (r == 0)
  {
    // ---- group leader: earlier groups, nearest first
    uint32_t pos = g;                                 // groups not yet accounted for: [0, pos)
    while (pos > 0)
    {
      const bool valid = lane < pos;
      unsigned long long bs = kStFlag, agg = 0;       // before group 0: base 0
      if (valid)
      {
        // base of group h+1 = rows in groups 0..h: look it up one slot to the right
        const uint32_t h = pos - 1 - lane;
        bs = h + 1 < g ? __hip_atomic_load(&group[h + 1].base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        agg = __hip_atomic_load(&group[h].agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      const bool is_incl = (bs & kStFlag) != 0;       // rows in groups 0..h known
      const bool is_full = (agg >> kGroupShift) == 64;
      const unsigned long long m_none = __ballot(!is_incl && !is_full);
      const unsigned long long m_incl = __ballot(is_incl);
      const int first_none = m_none ? (__ffsll((long long)m_none) - 1) : 64;
      const int first_incl = m_incl ? (__ffsll((long long)m_incl) - 1) : 64;
      if (first_none < first_incl) { __builtin_amdgcn_s_sleep(2); continue; }   // a nearer group is not complete yet
      unsigned long long x = 0;
      if ((int)lane < first_incl) x = agg & kGroupSumMask;
      else if ((int)lane == first_incl) x = bs & kStMask;
      base += wave_sum64(x);
      if (first_incl < 64) break;
      pos = pos > 64 ? pos - 64 : 0;
    }
    if (lane == 0) __hip_atomic_store(&group[g].base, kStFlag | base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return base;
  }
  // ---- the group's base and the earlier buckets of the group
  unsigned long long v = lane < r ? 0ull : kStFlag, bs = 0;
  for (;;)
  {
    if ((v & kStFlag) == 0)
      v = __hip_atomic_load(&status[(size_t)g * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((bs & kStFlag) == 0) bs = __hip_atomic_load(&group[g].base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__ballot((v & kStFlag) == 0) == 0 && (bs & kStFlag) != 0) break;
    __builtin_amdgcn_s_sleep(2);
  }
  uint32_t own = lane < r ? (uint32_t)v : 0u;
  for (int o = 32; o > 0; o >>= 1) own += __shfl_down(own, o, 64);
  return (bs & kStMask) + (unsigned long long)__shfl(own, 0, 64);
}

// t = q * d + r for t < 2^24, d >= 1, with the wave-uniform rcp = 1.0f / d: one multiply and a
// one-step fix-up instead of an integer division (or a subtract-until-it-fits loop) per element
__device__ __forceinline__ void divmod_rcp(uint32_t t, uint32_t d, float rcp, uint32_t& q, uint32_t& r)
{
  q = (uint32_t)((float)t * rcp);
  int rr = (int)t - (int)(q * d);
  if (rr < 0) { --q; rr += (int)d; }
  else if (rr >= (int)d) { ++q; rr -= (int)d; }
  r = (uint32_t)rr;
}

// What stage A (count) of a bucket hands to its stage B (emit): the records (count, sample, row
// within the bucket) stay in registers, the sorted distinct keys in one of the wave's two LDS
// key buffers.
template <int PER_LANE>
struct bucket_state
{
  uint32_t j, n, d;
  bool work;
  uint32_t cnt_r[PER_LANE], sr_r[PER_LANE];   // count; sample (low 16 bits) | row within the bucket << 16
};

// One pass: every wave owns a bucket -- hash set, distinct count, sort (stage A); row number by
// look-back, LDS block, write-out (stage B).  The two stages are software-pipelined: a wave runs
// stage A of its NEXT bucket -- which publishes that bucket's distinct count -- before stage B of
// the current one, so that by the time a bucket looks back, its predecessors' counts have been
// out for a whole iteration and nobody waits for the slowest wave of the sweep.
// The grid must be fully resident (persistent): a wave waits on the status words of
// lower-numbered buckets, which are always being worked on by resident waves.
#ifndef KMD_MERGE_SLOT_MULT
#define KMD_MERGE_SLOT_MULT 1        // hash slots per record of capacity (2: ~9 % slower, one wave less per SIMD)
#endif
#ifndef KMD_MERGE_WAVES_PER_EU
#define KMD_MERGE_WAVES_PER_EU 1     // occupancy the register allocator must leave room for
#endif
// kTwo: k-mers of two 64-bit limbs (32 < k <= 64).  A 128-bit key cannot be claimed with one LDS
// compare-and-swap, so the hash set then holds record indices (a 32-bit CAS claims a slot for
// the first record that reaches it) and keys are compared through the bucket's records, parked
// in LDS; the buckets themselves are cut on the top 64 bits of the keys (merge_fast).
template <typename CT, uint32_t kWaveCap, int kWavesPerBlock, bool kTwo>
__global__ void __launch_bounds__(64 * kWavesPerBlock) __attribute__((amdgpu_waves_per_eu(KMD_MERGE_WAVES_PER_EU)))
k_bucket_merge(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ keys_hi,
                                                              const uint32_t* __restrict__ counts,
                                                              const uint32_t* __restrict__ start, uint32_t S,
                                                              uint32_t nb, unsigned long long* __restrict__ status,
                                                              merge_group* __restrict__ group,
                                                              int layout, size_t ld, size_t row_capacity,
                                                              CT* __restrict__ matrix,
                                                              uint64_t* __restrict__ kmer_out,
                                                              uint64_t* __restrict__ kmer_hi_out,
                                                              uint32_t* __restrict__ overflow)
{
  constexpr uint32_t kWaveSlots = KMD_MERGE_SLOT_MULT * kWaveCap;
  constexpr uint32_t kL = kTwo ? 2 : 1;                 // limbs: [low limbs | high limbs] in every key array
  constexpr uint32_t kEmpty32 = 0xFFFFFFFFu;
  __shared__ unsigned long long s_hash_all[kWavesPerBlock][kWaveSlots];
  __shared__ unsigned long long s_keys_all[kWavesPerBlock][2][kL * kWaveCap];
  // segment tables while the records are loaded; afterwards the same memory holds the unsorted
  // distinct keys and then the slot -> row table
  __shared__ unsigned long long s_seg_all[kWavesPerBlock][kL * kWaveCap];
  __shared__ unsigned long long s_rk_all[kWavesPerBlock][kTwo ? 2 * kWaveCap : 1];   // kTwo: the records' keys
  constexpr uint32_t kMaxS = kWaveCap / 4;              // samples this instantiation serves (fast_bucket_cap)
  constexpr int kSPL = kMaxS / 64;                      // samples per lane in the segment phase
  static_assert(sizeof(unsigned long long) * (kWaveCap / 2) >= sizeof(uint32_t) * (2 * kMaxS + 1), "segment tables (lower half)");
  static_assert(sizeof(unsigned long long) * (kWaveCap / 2) >= sizeof(uint16_t) * kWaveCap, "record -> sample table (upper half)");
  static_assert(sizeof(unsigned long long) * kWaveCap >= sizeof(uint16_t) * kWaveSlots, "slot -> row table");
  constexpr uint32_t cmax = sizeof(CT) == 1 ? 0xFFu : sizeof(CT) == 2 ? 0xFFFFu : 0xFFFFFFFFu;
  constexpr int kPerLane = kWaveCap / 64;               // records of a bucket held by one lane
  using state_t = bucket_state<kPerLane>;
  const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned long long* s_hash = s_hash_all[w];
  uint32_t* s_beg = reinterpret_cast<uint32_t*>(s_seg_all[w]);
  uint32_t* pref = s_beg + kMaxS;
  uint16_t* smp_of = reinterpret_cast<uint16_t*>(s_seg_all[w] + kWaveCap / 2);    // [kWaveCap], upper half of the region
  unsigned long long* s_tmp = s_seg_all[w];                          // unsorted distinct keys
  unsigned long long* s_rk = s_rk_all[w];
  uint32_t* s_own = reinterpret_cast<uint32_t*>(s_hash_all[w]);      // kTwo: slot -> record that claimed it
  uint16_t* s_rank = reinterpret_cast<uint16_t*>(s_seg_all[w]);      // hash slot -> row within the bucket
  const uint32_t n_waves = gridDim.x * kWavesPerBlock;
  const uint32_t j_first = blockIdx.x * kWavesPerBlock + w;
  if (j_first >= nb) return;

  // segment bounds of the wave's first bucket; those of the next bucket are fetched while the
  // current one is processed (one dependent global round trip less per bucket)
  uint32_t nb_beg[kSPL], nb_end[kSPL];
#pragma unroll
  for (int q = 0; q < kSPL; ++q)
  {
    const uint32_t s = lane * kSPL + q;
    nb_beg[q] = 0; nb_end[q] = 0;
    if (s < S) { nb_beg[q] = start[(size_t)j_first * S + s]; nb_end[q] = start[(size_t)(j_first + 1) * S + s]; }
  }

#ifdef KMD_MERGE_TIMING   // dev only: per-phase cycles of one wave, printed with KMD_DEBUG=1
  unsigned long long T[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, tprev = __builtin_readcyclecounter();
#define TICK(i) do { unsigned long long tn = __builtin_readcyclecounter(); T[i] += tn - tprev; tprev = tn; } while (0)
#else
#define TICK(i) do { } while (0)
#endif
  // ---------------- stage A: records -> registers, hash set, distinct count (published), sorted keys
  auto stage_a = [&](uint32_t j, state_t& st, unsigned long long* s_keys)
  {
    // the S segments of this bucket; exclusive prefix of their lengths (kSPL samples per lane)
    uint32_t len[kSPL], lsum = 0;
#pragma unroll
    for (int q = 0; q < kSPL; ++q)
    {
      const uint32_t s = lane * kSPL + q;
      len[q] = 0;
      if (s < S)
      {
        s_beg[s] = nb_beg[q];
        len[q] = nb_end[q] - nb_beg[q];
      }
      lsum += len[q];
    }
    {
      const uint32_t jn = j + n_waves;                  // prefetch the next bucket of this wave
      if (jn < nb)
      {
#pragma unroll
        for (int q = 0; q < kSPL; ++q)
        {
          const uint32_t s = lane * kSPL + q;
          if (s < S) { nb_beg[q] = start[(size_t)jn * S + s]; nb_end[q] = start[(size_t)(jn + 1) * S + s]; }
        }
      }
    }
    uint32_t incl = lsum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1)
    {
      const uint32_t up = __shfl_up(incl, o, 64);
      if ((int)lane >= o) incl += up;
    }
    uint32_t run = incl - lsum;                         // records before this lane's first sample
    const uint32_t n = __builtin_amdgcn_readfirstlane(__shfl(incl, 63, 64));   // wave-uniform: slots past it are skipped by scalar branches
    const bool too_big = n > kWaveCap;
    if (too_big && lane == 0) atomicAdd(overflow, 1u);  // the caller falls back to the sort path
#pragma unroll
    for (int q = 0; q < kSPL; ++q)
    {
      const uint32_t s = lane * kSPL + q;
      if (s < S)
      {
        pref[s] = run;
        // record -> sample table: every record of the bucket looks its stream up with one LDS
        // read (segments are ~3 records long; a binary search over pref is 6 dependent reads)
        if (!too_big)
          for (uint32_t t = 0; t < len[q]; ++t) smp_of[run + t] = (uint16_t)s;
      }
      run += len[q];
    }

    TICK(0);
    // hash set sized to the bucket; every lane keeps its records in registers
    uint32_t d = 0, filled = 0;                         // distinct keys (with / without the empty-marker key)
    bool any_max = false;
    uint32_t slots = 64;
    uint16_t* s_tslot = reinterpret_cast<uint16_t*>(s_keys);        // slot of each compacted key (s_keys is written last)
    uint64_t key_r[kPerLane];
    uint64_t keyh_r[kTwo ? kPerLane : 1];
    uint32_t slot_r[kPerLane];                          // hash slot of the record's key (kNoSlot: the empty-marker key)
    constexpr uint32_t kNoSlot = 0xFFFFFFFFu;
    if (n > 0 && !too_big)
    {
      while (slots < 2 * n && slots < kWaveSlots) slots <<= 1;   // distinct keys are ~n / (samples present per row)
      const uint32_t mask = slots - 1;
      if constexpr (kTwo) { for (uint32_t t = lane; t < slots; t += 64) s_own[t] = kEmpty32; }
      else { for (uint32_t t = lane; t < slots; t += 64) s_hash[t] = kEmpty; }
      wave_sync();
      bool has_max_key = false;                         // the key equal to the empty marker, if present
      // all loads of the bucket first (independent: one memory round trip), then the inserts
#pragma unroll
      for (int r = 0; r < kPerLane; ++r)
      {
        const uint32_t f = (uint32_t)r * 64 + lane;
        key_r[r] = 0; st.cnt_r[r] = 0; st.sr_r[r] = 0; slot_r[r] = kNoSlot;
        if ((uint32_t)r * 64 < n && f < n)
        {
          const uint32_t lo = smp_of[f];                // the stream this record comes from
          const uint32_t i = s_beg[lo] + (f - pref[lo]);
          key_r[r] = keys[i]; st.cnt_r[r] = counts[i]; st.sr_r[r] = lo;
          if constexpr (kTwo) keyh_r[r] = keys_hi[i];
        }
      }
      if constexpr (kTwo)
      {
        // the bucket's keys where every lane can compare against them
#pragma unroll
        for (int r = 0; r < kPerLane; ++r)
        {
          const uint32_t f = (uint32_t)r * 64 + lane;
          if ((uint32_t)r * 64 < n && f < n) { s_rk[f] = key_r[r]; s_rk[kWaveCap + f] = keyh_r[r]; }
        }
        wave_sync();
      }
#ifdef KMD_MERGE_TIMING
      if (key_r[0] == 12345 && st.cnt_r[0] == 77) T[7]++;   // forces the loads to complete here
#endif
      TICK(1);
      // one-limb keys: the first probe of every record slot goes out before any answer is looked at
      // (an LDS compare-and-swap that returns is ~200 cycles; slot after slot they add up)
      unsigned long long first_old[kTwo ? 1 : kPerLane];
      if constexpr (!kTwo)
      {
#pragma unroll
        for (int r = 0; r < kPerLane; ++r)
        {
          first_old[r] = 0;
          const uint32_t f = (uint32_t)r * 64 + lane;
          if ((uint32_t)r * 64 < n && f < n && key_r[r] != kEmpty)
          {
            const uint32_t h = (hash_slot(key_r[r]) >> 16) & mask;   // slots <= 2048: bits 16.. of the product
            slot_r[r] = h;
            first_old[r] = atomicCAS(&s_hash[h], kEmpty, (unsigned long long)key_r[r]);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < kPerLane; ++r)
      {
        if ((uint32_t)r * 64 >= n) continue;              // wave-uniform: nothing in this slot
        const uint32_t f = (uint32_t)r * 64 + lane;
        bool fresh = false;
        if constexpr (kTwo)
        {
          if (f < n)
          {
            const uint64_t k = key_r[r], kh = keyh_r[r];
            uint32_t h = (hash_slot(k ^ (kh * 0x9E3779B97F4A7C15ull)) >> 16) & mask;
            for (;;)
            {
              const uint32_t old = atomicCAS(&s_own[h], kEmpty32, f);
              if (old == kEmpty32) { fresh = true; break; }
              if (s_rk[old] == k && s_rk[kWaveCap + old] == kh) break;
              h = (h + 1) & mask;
            }
            slot_r[r] = h;
          }
        }
        else if (f < n)
        {
          const uint64_t k = key_r[r];
          if (k == kEmpty) has_max_key = true;
          else
          {
            uint32_t h = slot_r[r];                            // first probe: issued above, for all slots at once
            unsigned long long old = first_old[r];
            for (;;)
            {
              if (old == kEmpty) { fresh = true; break; }
              if (old == k) break;
              h = (h + 1) & mask;
              old = atomicCAS(&s_hash[h], kEmpty, (unsigned long long)k);
            }
            slot_r[r] = h;
          }
        }
        // the lanes that claimed a slot hold the bucket's distinct keys: compacted here (ballot
        // prefix), with their slots, for the ranking below -- no scan of the hash table for them
        const unsigned long long fm = __ballot(fresh);
        if (fresh)
        {
          const uint32_t e = d + (uint32_t)__popcll(fm & ((1ull << lane) - 1ull));
          s_tmp[e] = key_r[r];
          if constexpr (kTwo) s_tmp[kWaveCap + e] = keyh_r[r];
          s_tslot[e] = (uint16_t)slot_r[r];
        }
        d += (uint32_t)__popcll(fm);
      }
      filled = d;
      any_max = __ballot(has_max_key) != 0;
      d += any_max ? 1u : 0u;
    }
    // publish the bucket's own count (all that successors need of this bucket)
    if (lane == 0)
    {
      __hip_atomic_store(&status[j], kStFlag | (unsigned long long)d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&group[j >> 6].agg, (1ull << kGroupShift) | (unsigned long long)d, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
    }
    const bool work = n > 0 && !too_big;
    st.j = j; st.n = n; st.d = d; st.work = work;

    TICK(2);
    // rank the distinct keys (compacted at insertion): every lane counts how many are
    // smaller than its own (broadcast LDS reads, no dependent chain -- a bitonic sort of ~100
    // keys is ~30 dependent LDS round trips); rank = row within the bucket
    wave_sync();
    if (work)
    {
      wave_sync();
      unsigned long long mk[kPerLane], mkh[kTwo ? kPerLane : 1];
      uint32_t ms[kPerLane], below[kPerLane];
#pragma unroll
      for (int q = 0; q < kPerLane; ++q)
      {
        const uint32_t e = (uint32_t)q * 64 + lane;
        mk[q] = e < filled ? s_tmp[e] : 0ull;
        if constexpr (kTwo) mkh[q] = e < filled ? s_tmp[kWaveCap + e] : 0ull;
        ms[q] = e < filled ? (uint32_t)s_tslot[e] : 0u;
        below[q] = 0;
      }
      const uint32_t nq = (filled + 63) >> 6;           // key registers in use (wave-uniform)
#pragma unroll 4
      for (uint32_t e = 0; e < filled; ++e)
      {
        const unsigned long long v = s_tmp[e];
        if constexpr (kTwo)
        {
          const unsigned long long vh = s_tmp[kWaveCap + e];
#pragma unroll
          for (int q = 0; q < kPerLane; ++q)
            if ((uint32_t)q < nq) below[q] += (vh < mkh[q] || (vh == mkh[q] && v < mk[q])) ? 1u : 0u;
        }
        else
        {
#pragma unroll
          for (int q = 0; q < kPerLane; ++q)
            if ((uint32_t)q < nq) below[q] += v < mk[q] ? 1u : 0u;
        }
      }
      wave_sync();
#pragma unroll
      for (int q = 0; q < kPerLane; ++q)
      {
        const uint32_t e = (uint32_t)q * 64 + lane;
        if (e < filled)
        {
          s_keys[below[q]] = mk[q];
          if constexpr (kTwo) s_keys[kWaveCap + below[q]] = mkh[q];
          s_rank[ms[q]] = (uint16_t)below[q];
        }
      }
      if (any_max && lane == 0) s_keys[d - 1] = kEmpty;               // the largest key there is
      wave_sync();
#pragma unroll
      for (int r = 0; r < kPerLane; ++r)
      {
        if ((uint32_t)r * 64 < n && (uint32_t)r * 64 + lane < n)
          st.sr_r[r] |= (slot_r[r] == kNoSlot ? d - 1 : (uint32_t)s_rank[slot_r[r]]) << 16;
      }
      wave_sync();                                      // the segment tables of the next bucket go here
    }
  };

  // ---------------- stage B: row number by look-back, LDS block, write-out
  auto stage_b = [&](const state_t& st, const unsigned long long* s_keys, bool have_rb, unsigned long long early_rb)
  {
    const uint32_t j = st.j, n = __builtin_amdgcn_readfirstlane(st.n), d = __builtin_amdgcn_readfirstlane(st.d);
    TICK(3);
    const unsigned long long rb64 = have_rb ? early_rb : rows_before(status, group, j, lane);
    TICK(4);
    if (lane == 0 && j == nb - 1) group[(j >> 6) + 1].base = rb64 + d;          // the partition's row count
    if (!st.work) return;
    if (rb64 + d > row_capacity) { if (lane == 0) atomicAdd(overflow + 1, 1u); return; }   // counted, not written
    const size_t rb = (size_t)rb64;
    if (kmer_out)
      for (uint32_t t = lane; t < d; t += 64) kmer_out[rb + t] = s_keys[t];
    if constexpr (kTwo)
      if (kmer_hi_out)
        for (uint32_t t = lane; t < d; t += 64) kmer_hi_out[rb + t] = s_keys[kWaveCap + t];
    // the bucket's d x S block of the matrix is assembled in LDS (the hash set's memory, free
    // between two stage A's) and written out whole: no zero-fill pass over the matrix, no 4-byte
    // scatter; a block too large for LDS is zero-filled and scattered in place
    const uint32_t cells = d * S;
    const float rcp_d = 1.0f / (float)d;
    const bool in_lds = (size_t)cells * sizeof(CT) <= sizeof(unsigned long long) * kWaveSlots;
    CT* tile = reinterpret_cast<CT*>(s_hash);                        // [sample][row in bucket]
    if (in_lds)
    {
      const uint32_t vecs = (uint32_t)((cells * sizeof(CT) + 15) / 16);
      for (uint32_t t = lane; t < vecs; t += 64) reinterpret_cast<uint4*>(s_hash)[t] = make_uint4(0, 0, 0, 0);
    }
    else
    {
      for (uint32_t t = lane; t < cells; t += 64)
      {
        uint32_t smp, row;
        divmod_rcp(t, d, rcp_d, smp, row);
        matrix[kmd::count_index(layout, ld, (int)S, rb + row, (int)smp)] = (CT)0;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");           // zero-fill before the scatter (global case)
    wave_sync();
#pragma unroll
    for (int r = 0; r < kPerLane; ++r)
    {
      const uint32_t f = (uint32_t)r * 64 + lane;
      if ((uint32_t)r * 64 < n && f < n)
      {
        uint32_t c = st.cnt_r[r];
        if (c > cmax) c = cmax;
        const uint32_t smp = st.sr_r[r] & 0xFFFFu, row = st.sr_r[r] >> 16;
        if (in_lds) tile[smp * d + row] = (CT)c;
        else matrix[kmd::count_index(layout, ld, (int)S, rb + row, (int)smp)] = (CT)c;
      }
    }
    wave_sync();
    if (in_lds)
    {
      if (layout == KMD_LAYOUT_ROWS)            // consecutive lanes -> consecutive samples of a row: one contiguous span
      {
        const float rcp_s = 1.0f / (float)S;
        for (uint32_t t = lane; t < cells; t += 64)
        {
          uint32_t row, smp;
          divmod_rcp(t, S, rcp_s, row, smp);
          matrix[(rb + row) * ld + smp] = tile[smp * d + row];
        }
      }
      else                                      // consecutive lanes -> consecutive rows of a sample
      {
        for (uint32_t t = lane; t < cells; t += 64)
        {
          uint32_t smp, row;
          divmod_rcp(t, d, rcp_d, smp, row);
          matrix[kmd::count_index(layout, ld, (int)S, rb + row, (int)smp)] = tile[t];
        }
      }
    }
    wave_sync();
  };

  state_t cur;
  uint32_t buf = 0;
  stage_a(j_first, cur, s_keys_all[w][0]);
  for (;;)
  {
    const uint32_t jn = cur.j + n_waves;
    const bool have_next = jn < nb;
    state_t nxt;
    // A group's first bucket resolves the group's base BEFORE its wave turns to the next
    // bucket: everything it needs (the earlier groups' sums) was published a stage ago, and
    // the 63 other buckets of the group, which wait for that one word, find it there.
    const bool leader = (cur.j & 63u) == 0;
    unsigned long long early_rb = 0;
    if (leader) early_rb = rows_before(status, group, cur.j, lane);
    if (have_next) stage_a(jn, nxt, s_keys_all[w][buf ^ 1]);
    stage_b(cur, s_keys_all[w][buf], leader, early_rb);
    if (!have_next) break;
    cur = nxt;
    buf ^= 1;
    TICK(5);
  }
#ifdef KMD_MERGE_TIMING
  if (blockIdx.x == 7 && threadIdx.x == 0) for (int i = 0; i < 8; ++i) reinterpret_cast<unsigned long long*>(overflow)[2 + i] = T[i];
#endif
#undef TICK
}

// min of the first keys / max of the last keys of the non-empty streams
// top[i] = the 64 most significant bits of the two-limb key (hi[i], lo[i]) when the high limbs use
// `bits` bits: what the buckets of two-limb k-mers are cut on (monotone in the full key)
__global__ void __launch_bounds__(256) k_top64(const uint64_t* __restrict__ lo, const uint64_t* __restrict__ hi, size_t n,
                                               int bits, uint64_t* __restrict__ top)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  top[i] = bits >= 64 ? hi[i] : bits == 0 ? lo[i] : ((hi[i] << (64 - bits)) | (lo[i] >> bits));
}

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

// A bucket must hold a few whole rows (a row has up to S records: cap >= 4 S, the kernel's tables),
// and the larger the buckets the smaller the [bucket][sample] start table that five passes read and
// write -- against 2 and 1 waves per SIMD for the 512- and 1024-record kernels.  Measured
// (tools/cap_sweep.sh, 50 M records): S=20 256: 1.16 / 512: 1.20 ms; S=32 1.27 / 1.27; S=40 1.37 / 1.27;
// S=64 2.32 / 1.32; S=100 512: 1.61 / 1024: 1.80; S=128 2.25 / 1.81.
// Two-limb keys (their 512- and 1024-record kernels hold one wave per workgroup and more state):
// the smallest capacity that serves S stays best (S=40 256: 3.9 / 512: 4.7 ms; S=100 512: 5.6 / 1024: 6.6).
inline uint32_t fast_bucket_cap(int S, bool two_limbs)
{
  uint32_t cap = two_limbs ? (S <= 64 ? 256u : S <= 128 ? 512u : 1024u) : (S <= 32 ? 256u : S <= 104 ? 512u : 1024u);
  if (const char* e = std::getenv("KMD_MERGE_CAP"))       // dev: another capacity that still serves S (A/B)
  {
    const uint32_t c = (uint32_t)std::atoi(e);
    if ((c == 256 || c == 512 || c == 1024) && c >= 4u * (uint32_t)S) cap = c;
  }
  return cap;
}

// The bucketed LDS merge.  *used = false (and nothing written) when the input does not suit
// it (clustered keys overflow a bucket): the caller then takes the sort-based path.
template <typename CT>
int merge_fast(int S, const uint64_t* d_kmers_lo, const uint64_t* d_kmers_hi, const uint32_t* d_counts,
               const uint64_t* offsets, int layout, size_t ld, size_t row_capacity, CT* d_matrix,
               uint64_t* d_kmer_out, uint64_t* d_kmer_hi_out, uint64_t* n_rows_out, int n_cu, hipStream_t st,
               bool* used)
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
  // d_kmers: the 64-bit keys the buckets are cut on -- the k-mers themselves, or for two-limb
  // k-mers their top 64 bits (the width of the high limbs is taken from the largest one)
  const uint64_t* d_kmers = d_kmers_lo;
  const bool two = d_kmers_hi != nullptr;
  if (two)
  {
    uint64_t hr[2];
    hipLaunchKernelGGL(k_key_range, dim3(1), dim3(64), 0, st, d_kmers_hi, d_offs, (uint32_t)S, d_range);
    KMD_HIP(hipMemcpyAsync(hr, d_range, 16, hipMemcpyDeviceToHost, st));
    KMD_HIP(hipStreamSynchronize(st));
    int bits = 0;
    while (bits < 64 && (hr[1] >> bits) != 0) ++bits;
    void* p_top = nullptr;
    KMD_HIP(sc.take(&p_top, n * 8));
    hipLaunchKernelGGL(k_top64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_kmers_lo, d_kmers_hi, n, bits,
                       static_cast<uint64_t*>(p_top));
    KMD_HIP(hipGetLastError());
    d_kmers = static_cast<const uint64_t*>(p_top);
    if (dbg) std::fprintf(stderr, "[merge_fast] two-limb keys, high limbs of %d bits\n", bits);
  }
  // buckets by splitters (default) or by equal slices of the key range (KMD_MERGE_SLICES=1)
  const bool splitters = std::getenv("KMD_MERGE_SLICES") == nullptr;
  uint64_t range[2] = { 0, ~0ull };
  if (!splitters)                                               // only the slices need the key range
  {
    hipLaunchKernelGGL(k_key_range, dim3(1), dim3(64), 0, st, d_kmers, d_offs, (uint32_t)S, d_range);
    KMD_HIP(hipMemcpyAsync(range, d_range, 16, hipMemcpyDeviceToHost, st));
    KMD_HIP(hipStreamSynchronize(st));
  }
  const uint64_t span = range[1] - range[0];                    // kmax - kmin
  // a bucket must hold a few whole rows, and a row has up to S records: capacity by sample count;
  // half the capacity per bucket on average (one wave each); the start table is capped at 1 GiB
  const uint32_t cap = fast_bucket_cap(S, two);
  uint64_t nb_target = n / (cap / 2) + 1;
  const uint64_t table_cap = (1ull << 30) / (4ull * (uint64_t)S);
  if (nb_target > table_cap) nb_target = table_cap;
  if (nb_target > 0xFFFFFFF0ull) nb_target = 0xFFFFFFF0ull;
  bucket_map B;
  B.kmin = range[0];
  B.nb = (uint32_t)nb_target;
  // mult = floor(nb * 2^64 / (span + 1)); (span * mult) >> 64 < nb for every key in range
  if (span == ~0ull) B.mult = nb_target;
  else B.mult = (uint64_t)((((unsigned __int128)nb_target) << 64) / ((unsigned __int128)span + 1));
  if (span + 1 != 0 && nb_target > span + 1) { B.nb = (uint32_t)(span + 1); B.mult = (uint64_t)((((unsigned __int128)B.nb) << 64) / ((unsigned __int128)span + 1)); }
  uint32_t L = 0, r_split = 1;
  if (splitters)
  {
    for (int s2 = 1; s2 < S; ++s2) if (offsets[s2 + 1] - offsets[s2] > offsets[L + 1] - offsets[L]) L = (uint32_t)s2;
    const uint64_t n_l = offsets[L + 1] - offsets[L];
    uint64_t fill = cap / 2;                                               // records per bucket aimed at
    if (const char* e = std::getenv("KMD_MERGE_FILL")) fill = (uint64_t)cap * (uint64_t)std::atoi(e) / 100;   // dev: % of CAP
    if (fill < 16) fill = 16;
    uint64_t r = (uint64_t)((unsigned __int128)n_l * fill / n);            // every r-th key of the longest stream
    if (r < 1) r = 1;
    uint64_t nbs = (n_l + r - 1) / r;
    if (nbs > nb_target) { r = (n_l + nb_target - 1) / nb_target; nbs = (n_l + r - 1) / r; }
    r_split = (uint32_t)r;
    B.nb = (uint32_t)nbs;
  }
  const size_t nb0 = B.nb;                                    // first buckets; nb = buckets after cutting the heavy ones
  if (dbg) std::fprintf(stderr, "[merge_fast] n=%zu kmin=%llu kmax=%llu mult=%llu nb0=%zu\n", n,
                        (unsigned long long)range[0], (unsigned long long)range[1], (unsigned long long)B.mult, nb0);

  KMD_HIP(kmd::scratch_alloc(&sc.p[1], (nb0 + 1) * (size_t)S * 4));
  KMD_HIP(kmd::scratch_alloc(&sc.p[5], (nb0 + 1) * (size_t)S * 4));
  KMD_HIP(kmd::scratch_alloc(&sc.p[4], 128));
  uint32_t* start = static_cast<uint32_t*>(sc.p[1]);          // [bucket][stream]
  uint32_t* start_sm = static_cast<uint32_t*>(sc.p[5]);       // [stream][bucket]
  uint32_t* overflow = static_cast<uint32_t*>(sc.p[4]);      // [0] bucket too large, [1] row capacity exceeded, [2] buckets cut
  KMD_HIP(hipMemsetAsync(overflow, 0, 16, st));
  {
    size_t longest = 1;
    for (int s = 0; s < S; ++s) if (offsets[s + 1] - offsets[s] > longest) longest = offsets[s + 1] - offsets[s];
    if (nb0 + 1 > longest) longest = nb0 + 1;
    size_t gx = (longest + 255) / 256;
    if (gx > 4096) gx = 4096;
    if (splitters)
    {
      const size_t n_chunks = nb0 / kSplitChunk + 1;                           // chunk c: boundaries [64 c, 64 c + 64)
      void* p_coarse = nullptr;
      KMD_HIP(sc.take(&p_coarse, (n_chunks + 1) * (size_t)S * 4));
      hipLaunchKernelGGL(k_splitter_coarse, dim3((unsigned)((n_chunks + 1 + 255) / 256), (unsigned)S), dim3(256), 0, st, d_kmers,
                         d_offs, (uint32_t)S, L, r_split, (uint32_t)nb0, (uint32_t)n_chunks, static_cast<uint32_t*>(p_coarse));
      hipLaunchKernelGGL(k_splitter_fine, dim3((unsigned)((n_chunks + 3) / 4), (unsigned)S), dim3(256), 0, st, d_kmers, d_offs,
                         (uint32_t)S, L, r_split, (uint32_t)nb0, (uint32_t)n_chunks, static_cast<const uint32_t*>(p_coarse), start_sm);
    }
    else
      hipLaunchKernelGGL(k_bucket_starts, dim3((unsigned)gx, (unsigned)S), dim3(256), 0, st, d_kmers, d_offs, (uint32_t)S, B, start_sm);
    hipLaunchKernelGGL(k_transpose_starts, dim3((unsigned)((nb0 + 1 + 63) / 64)), dim3(256), 0, st, start_sm, (uint32_t)S,
                       (uint32_t)(nb0 + 1), start);
  }
  KMD_HIP(hipGetLastError());
  KMD_DBG("starts");
  // Buckets over capacity are cut into finer slices on the start table.  Level 0 cuts the
  // bucket's own slice of the key range (no key reads: random keys put a few % of the buckets
  // over, Poisson tails); what is still over after that (clusters, the odd tail of a tail) is cut
  // by the key range its records REALLY span.  A table that grows 2.5-fold (dense clusters in
  // an otherwise empty range), or buckets over capacity after kMaxLevels, go to the sort path.
  size_t nb = nb0;
  constexpr int kMaxLevels = 6;
  const uint64_t wq = (uint64_t)((((unsigned __int128)1) << 64) / B.mult);          // 2^64 = wq mult + wr
  const uint64_t wr = (uint64_t)((((unsigned __int128)1) << 64) % B.mult);
  // returns buckets over capacity before the cut (0: table unchanged), -1: give up (sort path), -2: error;
  // *pieces_over = pieces of the cut buckets that are still over capacity
  auto refine_level = [&](int level, bool arith, int* n_over_out, uint32_t* pieces_over) -> int
  {
    *pieces_over = 0;
    uint32_t n_over = 0;
    void *p_split = nullptr, *p_klo = nullptr, *p_kstep = nullptr;
    KMD_HIP(sc.take(&p_split, (nb + 1) * 4));
    KMD_HIP(sc.take(&p_klo, nb * 8));
    KMD_HIP(sc.take(&p_kstep, nb * 8));
    uint32_t* split = static_cast<uint32_t*>(p_split);
    KMD_HIP(hipMemsetAsync(overflow + 2, 0, 4, st));
    const bool sm_form = level == 0;                          // the stream-major table still describes level 0
    hipLaunchKernelGGL(k_bucket_split, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, d_kmers,
                       sm_form ? start_sm : start, sm_form ? (size_t)1 : (size_t)S, sm_form ? nb + 1 : (size_t)1, (uint32_t)S,
                       (uint32_t)nb, cap, split, static_cast<uint64_t*>(p_klo), static_cast<uint64_t*>(p_kstep),
                       overflow + 2, arith ? 1 : 0, B, wq, wr);
    KMD_HIP(hipMemcpyAsync(&n_over, overflow + 2, 4, hipMemcpyDeviceToHost, st));
    KMD_HIP(hipStreamSynchronize(st));
    if (dbg) std::fprintf(stderr, "[merge_fast] level %d: %u of %zu buckets over capacity\n", level, n_over, nb);
    *n_over_out = (int)n_over;
    if (n_over == 0) return KMD_OK;
    void *p_first = nullptr, *p_tmp = nullptr, *p_refined = nullptr;
    KMD_HIP(sc.take(&p_first, (nb + 1) * 4));
    uint32_t* first = static_cast<uint32_t*>(p_first);
    size_t tmp = 0;
    KMD_HIP(rocprim::exclusive_scan(nullptr, tmp, split, first, 0u, nb + 1, rocprim::plus<uint32_t>(), st));
    KMD_HIP(sc.take(&p_tmp, tmp ? tmp : 1));
    KMD_HIP(rocprim::exclusive_scan(p_tmp, tmp, split, first, 0u, nb + 1, rocprim::plus<uint32_t>(), st));
    uint32_t nb_new = 0;
    KMD_HIP(hipMemcpyAsync(&nb_new, first + nb, 4, hipMemcpyDeviceToHost, st));
    KMD_HIP(hipStreamSynchronize(st));
    // a table that grows beyond 2.5 x its first size is mostly empty buckets around a few dense
    // clusters: every one of them still costs a wave its fixed work, and the sort path is then the
    // faster tool (2000 clusters of 2000 consecutive k-mers: 22.7 ms here against 8.9 ms sorted)
    if ((uint64_t)nb_new > table_cap || 2 * (uint64_t)nb_new > 5 * (uint64_t)nb0) { *n_over_out = -1; return KMD_OK; }
    KMD_HIP(sc.take(&p_refined, ((size_t)nb_new + 1) * (size_t)S * 4));
    void* p_child = nullptr;
    KMD_HIP(sc.take(&p_child, (size_t)nb_new * 4));
    KMD_HIP(hipMemsetAsync(p_child, 0, (size_t)nb_new * 4, st));
    KMD_HIP(hipMemsetAsync(overflow + 2, 0, 4, st));
    const size_t cells = (nb + 1) * (size_t)S;
    hipLaunchKernelGGL(k_refine_starts, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, d_kmers, start,
                       (uint32_t)S, (uint32_t)nb, split, static_cast<const uint64_t*>(p_klo),
                       static_cast<const uint64_t*>(p_kstep), first, nb_new, static_cast<uint32_t*>(p_refined),
                       static_cast<uint32_t*>(p_child));
    hipLaunchKernelGGL(k_count_over, dim3((unsigned)(((size_t)nb_new + 255) / 256)), dim3(256), 0, st,
                       static_cast<const uint32_t*>(p_child), nb_new, cap, overflow + 2);
    KMD_HIP(hipGetLastError());
    KMD_HIP(hipMemcpyAsync(pieces_over, overflow + 2, 4, hipMemcpyDeviceToHost, st));
    KMD_HIP(hipStreamSynchronize(st));
    if (dbg) std::fprintf(stderr, "[merge_fast] level %d: %u pieces still over capacity\n", level, *pieces_over);
    start = static_cast<uint32_t*>(p_refined);
    nb = nb_new;
    return KMD_OK;
  };

  uint32_t h_over[2] = { 0, 0 };
  unsigned long long h_last = 0;
  size_t ng = 0;
  // one run of the merge kernel over the current table; h_over / h_last are its verdict
  auto run_merge = [&]() -> int
  {
    ng = (nb + 63) / 64;                                     // look-back groups
    const size_t status_bytes = ((nb * 8 + 127) / 128) * 128;
    void* p_status = nullptr;
    KMD_HIP(sc.take(&p_status, status_bytes + (ng + 1) * sizeof(merge_group)));
    unsigned long long* status = static_cast<unsigned long long*>(p_status);
    merge_group* group = reinterpret_cast<merge_group*>(reinterpret_cast<char*>(p_status) + status_bytes);
    KMD_HIP(hipMemsetAsync(status, 0, status_bytes + (ng + 1) * sizeof(merge_group), st));
    KMD_HIP(hipMemsetAsync(overflow, 0, 8, st));
    // persistent grid: every wave must be resident (look-back waits on lower-numbered buckets)
    auto launch = [&](auto kernel, int wpb) -> int
    {
      int per_cu = 0;
      KMD_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 64 * wpb, 0));
      if (per_cu < 1) per_cu = 1;
      if (const char* e = std::getenv("KMD_MERGE_BLOCKS_PER_CU")) per_cu = std::max(1, std::atoi(e));   // dev: only honoured by the cooperative launch
      const size_t want = (nb + wpb - 1) / wpb;
      uint32_t S32 = (uint32_t)S, nb32 = (uint32_t)nb;
      int lay = layout;
      size_t ld_ = ld, cap_ = row_capacity;
      const uint64_t* a_keys = d_kmers_lo; const uint64_t* a_keys_hi = d_kmers_hi;
      const uint32_t* a_counts = d_counts; const uint32_t* a_start = start;
      unsigned long long* a_status = status; merge_group* a_group = group;
      CT* a_matrix = d_matrix; uint64_t* a_kmer_out = d_kmer_out; uint64_t* a_kmer_hi_out = d_kmer_hi_out;
      uint32_t* a_overflow = overflow;
      void* args[] = { &a_keys, &a_keys_hi, &a_counts, &a_start, &S32, &nb32, &a_status, &a_group, &lay, &ld_, &cap_, &a_matrix,
                       &a_kmer_out, &a_kmer_hi_out, &a_overflow };
      // A cooperative launch is the runtime's own guarantee that the whole grid is resident: the
      // full occupancy can be used.  If it is refused, launch one workgroup per CU less (the
      // occupancy query may over-report by one).
      size_t grid = std::min((size_t)n_cu * (size_t)per_cu, want);
      hipError_t e = std::getenv("KMD_MERGE_NO_COOP") ? hipErrorNotSupported
                   : hipLaunchCooperativeKernel(reinterpret_cast<const void*>(kernel), dim3((unsigned)grid), dim3(64 * wpb), args, 0, st);
      if (e != hipSuccess)
      {
        (void)hipGetLastError();
        if (std::getenv("KMD_MERGE_BLOCKS_PER_CU")) KMD_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 64 * wpb, 0));
        grid = std::min((size_t)n_cu * (size_t)std::max(per_cu - 1, 1), want);
        hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(64 * wpb), 0, st, d_kmers_lo, d_kmers_hi, d_counts, start,
                           (uint32_t)S, (uint32_t)nb, status, group, layout, ld, row_capacity, d_matrix, d_kmer_out,
                           d_kmer_hi_out, overflow);
        KMD_HIP(hipGetLastError());
      }
      if (dbg) std::fprintf(stderr, "[merge_fast] grid %zu x %d threads (%s)\n", grid, 64 * wpb, e == hipSuccess ? "cooperative" : "plain");
      return KMD_OK;
    };
    int rc_launch;
    if (!two)
    {
      if (cap == 256) rc_launch = launch(k_bucket_merge<CT, 256, 2, false>, 2);
      else if (cap == 512) rc_launch = launch(k_bucket_merge<CT, 512, 2, false>, 2);
      else rc_launch = launch(k_bucket_merge<CT, 1024, 1, false>, 1);
    }
    else
    {
      if (cap == 256) rc_launch = launch(k_bucket_merge<CT, 256, 2, true>, 2);
      else if (cap == 512) rc_launch = launch(k_bucket_merge<CT, 512, 1, true>, 1);
      else rc_launch = launch(k_bucket_merge<CT, 1024, 1, true>, 1);
    }
    if (rc_launch != KMD_OK) return rc_launch;
    KMD_HIP(hipMemcpyAsync(h_over, overflow, 8, hipMemcpyDeviceToHost, st));
    KMD_HIP(hipMemcpyAsync(&h_last, &group[ng].base, 8, hipMemcpyDeviceToHost, st));
    KMD_HIP(hipStreamSynchronize(st));
    return KMD_OK;
  };

  // cut until every bucket fits, then merge.  Whether the pieces of a level's cut buckets fit is
  // known from their record counts, added up while the refined table is written: another level --
  // another pass over the whole table -- follows only if one of them does not.
  for (int level = 0;; ++level)
  {
    int n_over = 0;
    uint32_t pieces_over = 0;
    const int rc_l = refine_level(level, level == 0 && !splitters, &n_over, &pieces_over);
    if (rc_l != KMD_OK) return rc_l;
    if (n_over == 0) break;
    if (n_over < 0 || level == kMaxLevels) return KMD_OK;    // mostly clusters / cannot be cut: sort path
    if (pieces_over == 0) break;
  }
  {
    const int rc_m = run_merge();
    if (rc_m != KMD_OK) return rc_m;
  }
#ifdef KMD_MERGE_TIMING
  if (dbg)
  {
    unsigned long long tt[8];
    (void)hipMemcpy(tt, reinterpret_cast<char*>(overflow) + 16, 64, hipMemcpyDeviceToHost);
    std::fprintf(stderr, "[merge_fast] cycles: segments %llu load %llu insert %llu rank %llu look-back %llu emit %llu\n",
                 tt[0], tt[1], tt[2], tt[3], tt[4], tt[5]);
  }
#endif
  KMD_DBG("merge");
  if (h_over[0]) return KMD_OK;                              // still a bucket over capacity: not used, caller sorts
  const size_t n_rows = (size_t)(h_last & kStMask);
  *used = true;
  *n_rows_out = n_rows;
  if (h_over[1] || n_rows > row_capacity) { kmd::set_error("kmd_merge_partition: row capacity exceeded"); return KMD_E_OVERFLOW; }
  if (layout == KMD_LAYOUT_SOA) KMD_REQUIRE(ld >= n_rows, "kmd_merge_partition: SoA ld < rows");
  if (layout == KMD_LAYOUT_ROWS) KMD_REQUIRE(ld >= (size_t)S, "kmd_merge_partition: ld < samples");
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

  // bucketed LDS merge when it applies (one limb, enough records); else / on overflow: sort
  const char* force = std::getenv("KMD_MERGE_PATH");            // "sort" | "fast" | "fast-only" (tests, benchmarks)
  const bool want_fast = (uint32_t)n_samples <= kMaxFastSamples &&
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
      case 1: rc = merge_fast<uint8_t>(n_samples, d_kmers, d_kmers_hi, d_counts, offsets, layout, ld, row_capacity, static_cast<uint8_t*>(d_matrix), d_kmer_out, d_kmer_hi_out, n_rows_out, n_cu, st, &used); break;
      case 2: rc = merge_fast<uint16_t>(n_samples, d_kmers, d_kmers_hi, d_counts, offsets, layout, ld, row_capacity, static_cast<uint16_t*>(d_matrix), d_kmer_out, d_kmer_hi_out, n_rows_out, n_cu, st, &used); break;
      default: rc = merge_fast<uint32_t>(n_samples, d_kmers, d_kmers_hi, d_counts, offsets, layout, ld, row_capacity, static_cast<uint32_t*>(d_matrix), d_kmer_out, d_kmer_hi_out, n_rows_out, n_cu, st, &used); break;
    }
    if (rc != KMD_OK || used) return rc;
    // "fast-only" (tests): report instead of quietly sorting
    KMD_REQUIRE(!(force && std::strcmp(force, "fast-only") == 0), "kmd_merge_partition: bucketed path not applicable to this input");
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
