// kmd_tilemerge.hip -- K2t: the k-way merge of one partition's per-sample k-mer streams, fused with
// the Poisson test: streams in, survivors out, no matrix and no intermediate rows in HBM.
//
// Replaces km::KmerMerger<KSIZE,CMAX>::merge(diff_observer) as kmdiff drives it
// (include/kmdiff/merge.hpp:265-289 with the observer of :68-103): every distinct k-mer of the S
// sorted streams is one row; all the observer's model reads of a row are the sum of its control
// counts and the sum of its case counts (include/kmdiff/model.hpp:144-145).
//
// Shape of the work (HBM-bound: 12 bytes per record are read once, nothing else is large):
//   * k_tile_index: every stream's every 4096th key -- the index all boundary searches start from (lower_bound_indexed);
//   * k_tile_probe: how many records make a row here?  512 records drawn uniformly, the number m of
//     streams holding each one's k-mer; mean(1/m) = distinct k-mers / records (unbiased).  The plan --
//     records per tile such that a tile's distinct k-mers fill half of the hash table, the splitter
//     stride, lanes per run, WHICH of the two table shapes (2048 slots / 512 threads, or 4096 / 1024 where runs are
//     short) -- stays on the device: no host round trip before the main kernel.  The same launch finds the COARSE
//     boundaries (where every stream meets every R0-th key of the longest one: they do not depend on the plan);
//   * k_tile_bounds: the key range of the partition is cut into TILES by splitters taken from the data (every
//     r-th key of the longest stream); where each stream enters each tile: a search between the two coarse
//     boundaries around it, from an interpolated guess (lower_bound_near);
//   * k_tile_sums: one WORKGROUP per tile on a persistent grid.  A tile is S contiguous runs of records,
//     one per stream; its records are split evenly among the waves, a wave streams its share 64 records a
//     round (one per lane) through a buffer descriptor of the run, 4 rounds in flight, the loads issued and
//     waited for by hand (kWide; for runs of a handful of records: sub-groups of G lanes per run, the older
//     code).  Every record goes into a workgroup-wide LDS hash set keyed by the k-mer -- two home buckets of
//     two slots, both read in the first step; a compare-and-swap only to claim an empty slot; a small second
//     table for the k-mers that find all four taken -- and adds its count to that k-mer's control or case sum;
//   * after one barrier the table IS the tile's rows: every thread walks a few slots, a live slot goes
//     through the pre-filter (chi-square bound, then the likelihood ratio in single precision for the rows
//     that pass it: row_may_pass_kl); the ~1 % that pass leave as (k-mer, control sum, case sum) for
//     a list in HBM that is handed out in chunks (no global atomic per tile; holes are marked).
//     k_cand_eval / _scan / _emit (kmd_filter.hip; kmd_eval.h, the code K1 runs) evaluate the list exactly --
//     likelihood ratio, tail function, compaction into the survivor sink.  It is enqueued right behind this
//     kernel, gated on the device (below);
//   * a tile whose k-mers do not fit the table (fewer records per row there than the plan assumed, or
//     keys clustered where the longest stream has none) notices by a probe sequence that does not end,
//     gives up and is listed; the host cuts the listed tiles into equal slices of the key range their
//     records really span and runs the kernel again on those -- repeated until nothing is listed (every
//     level divides a tile's key span);
//   * kmd_merge_filter_batch: several partitions in flight on streams of the library's own, everything of a
//     partition enqueued without a host round trip (tile_merge, async).
// Two-limb k-mers (32 < k <= 64): the same kernel keyed by the low limb, see k_tile_sums.
#include "kmd_internal.h"
#include "kmd_math.h"
#include "kmd_eval.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <atomic>
#include <mutex>
#include <type_traits>
#include <vector>

using namespace kmd::eval;

#ifndef KMD_TILE_U
#define KMD_TILE_U 2
#endif
#ifndef KMD_TILE_DEPTH
#define KMD_TILE_DEPTH 4
#endif
#ifndef KMD_TILE_ABLATE
#define KMD_TILE_ABLATE 0
#endif
#ifndef KMD_TILE_RPL
#define KMD_TILE_RPL 1               // whole-wave path: records per lane and round (1: 64-record rounds, 2: 128: spills at 64 registers)
#endif
#ifndef KMD_TILE_ASM
#define KMD_TILE_ASM 1               // whole-wave path: stages 1 and 2 of a round in the middle of a run as hand-written gfx950 code
#endif
#ifndef KMD_TILE_TIMING
#define KMD_TILE_TIMING 0            // dev: 1 = cycles per LDS round trip inside the hand-written round + the phases of one tile (printf); 2 = the phases only
#endif
#ifndef KMD_TILE_HINT
#define KMD_TILE_HINT 0              // cache policy of the whole-wave path's record loads: 0 default, 1 nt (streaming), 2 sc1, 3 sc0 sc1
#endif
#if KMD_TILE_HINT == 0
#define KMD_TILE_LOAD_HINT ""
#elif KMD_TILE_HINT == 1
#define KMD_TILE_LOAD_HINT " nt"
#elif KMD_TILE_HINT == 2
#define KMD_TILE_LOAD_HINT " sc1"
#else
#define KMD_TILE_LOAD_HINT " sc0 sc1"
#endif
#ifndef KMD_TILE_ALIGN
#define KMD_TILE_ALIGN 0             // whole-wave path: a run's first round starts on a 128-byte line of both arrays (its leading lanes hold the records before the run: switched off)
#endif
#ifndef KMD_TILE_RING
#define KMD_TILE_RING 4              // rounds of loads in flight per wave (8, 12, 16 measured: no faster, more registers)
#endif
// waves per SIMD the register allocator is held to: 16 bytes per slot (32-bit sums) let four workgroups of 8 waves
// share a CU's LDS -- if a wave keeps to 64 VGPRs
#ifndef KMD_TILE_WAVES
#define KMD_TILE_WAVES(sum32, two, wide) ((sum32) && !(two) && (wide) ? 8 : 1)
#endif
#ifndef KMD_TILE_WAVES_BIG
#ifndef KMD_TILE_BIG_WPE
#define KMD_TILE_BIG_WPE (KMD_TILE_BIG_THREADS == 1024 ? 8 : 4)     // waves per SIMD the 4096-slot shape's registers are held to
#endif
#define KMD_TILE_WAVES_BIG(sum32, two, wide) ((sum32) && !(two) && (wide) ? KMD_TILE_BIG_WPE : 1)
#endif
#ifndef KMD_TILE_BIG_THREADS
#define KMD_TILE_BIG_THREADS 1024    // threads of the workgroup that takes the 4096-slot table (dev: 512 = eight waves with twice the registers each)
#endif
#ifndef KMD_TILE_RING_BIG
#define KMD_TILE_RING_BIG KMD_TILE_RING   // rounds in flight per wave under the 4096-slot shape
#endif
#ifndef KMD_TILE_ABORT_EVERY
#define KMD_TILE_ABORT_EVERY 4       // rounds between two looks at the tile's give-up flag (a power of two <= KMD_TILE_RING)
#endif

namespace {

constexpr uint64_t kEmptyKey = ~0ull;
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u64x2 lds_u64x2;        // a bucket of two slots, read with one ds_read_b128
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kRsrcFlags = 0x00020000;           // buffer descriptor, dword 3: raw 32-bit data, no swizzle (gfx9 family)
// the lanes' predicate as a mask, straight from the compare (HIP's __ballot takes an int: a select and a second compare)
__device__ __forceinline__ unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
constexpr uint32_t kMaxStreams = 1024;           // segment tables of a tile live in LDS (16 KB at 1024 streams)
#ifndef KMD_TILE_BIG_SLOTS
#define KMD_TILE_BIG_SLOTS 4096      // dev: 8192 = one workgroup per CU (one-limb k-mers, 32-bit sums, few samples only: the others do not fit the LDS)
#endif
constexpr uint32_t kSmallSlots = 2048, kBigSlots = KMD_TILE_BIG_SLOTS;   // the two table shapes built (512 / 1024 threads); make_plan picks one per partition
constexpr uint32_t kProbes = 512;                // records sampled for the records-per-row estimate (+-5 % at worst; 2048 cost 39 us, 4x this)
// ... fewer with many samples -- a probe is a search in every stream, 100 000 of them at 200 samples (0.1 ms), and
// the multiplicities it averages scatter less there: <= 32 768 searches, never under 128 probes
__host__ __device__ inline uint32_t probes_for(uint32_t S) { const uint32_t p = 32768u / (S ? S : 1u); return p > kProbes ? kProbes : p < 128u ? 128u : p; }
constexpr uint32_t kAbortBit = 0x80000000u;      // over list: the tile gave up on distinct k-mers, not on records
constexpr uint32_t kBigBit = 0x40000000u;        // over list: the tile holds a count too large for 32-bit sums
// record positions and run extents are 32-bit: a run's byte extent (8 x its records) must fit the buffer descriptor's
// 32-bit range and a lane's position may step up to two sub-groups past the last record before it is checked
constexpr uint64_t kMaxRecords = 0xFFFFFFFFull - 128ull;
constexpr uint64_t kMaxRun = 1ull << 29;
constexpr uint32_t kBigCount = 1u << 22;
constexpr uint64_t kBigFromRecords = 200000000ull;   // partitions of this many records take the 4096-slot table whatever their runs (make_plan)
// candidates mode: rows of a tile parked in LDS on their way to the list (rows of 3 records: ~90 of a 4096-slot tile's
// ~1900 leave; configs[2]'s rows of 26: one or two)
constexpr uint32_t stage_rows(uint32_t slots) { return slots >= 4096u ? 128u : 64u; }
constexpr uint32_t kGroup = 63;                  // runs per group of a tile's segment table (one lane each; lane 63 holds none)
constexpr uint32_t kMaxGroups = 17;              // (kMaxStreams + kGroup - 1) / kGroup
constexpr uint32_t kQueue = 192;                 // candidates mode: live slots a wave has queued for its next passes (<= 63 + 128)
constexpr uint32_t kOutChunk = 256;              // candidates mode: entries of the list a workgroup takes at a time
constexpr unsigned long long kHole = ~0ull;      // sum_c of an entry that holds no row (no sum of 32-bit counts reaches it)         // 1024 samples of counts below this cannot overflow a 32-bit sum

// what k_tile_plan decides, on the device
struct tile_plan
{
  uint32_t r;                                    // boundary j = key j * r of the longest stream
  uint32_t nb;                                   // tiles
  uint32_t g_shift;                              // lanes per run of records = 1 << g_shift
  uint32_t fill;                                 // records per tile aimed at
  float rho;                                     // records per row, estimated
  uint32_t slots;                                // the table the tiles are sized for: the instantiation of the merge kernel that takes them
  uint32_t pad[2];
};

struct tile_job
{
  const uint64_t* keys;
  const uint64_t* keys_hi;                       // two-limb k-mers, else NULL
  const uint32_t* counts;
  const uint32_t* start;                         // [rows][S]: row r = where every stream enters tile r; tile r ends at row r + 1
  const uint8_t* todo;                           // NULL: every tile; else only tiles with todo[r] != 0
  const tile_plan* plan;                         // n_tiles / g_shift when n_tiles == 0 (level 0: decided on the device)
  uint32_t S, nc, n_tiles, g_shift, xcd_order;
  uint32_t force_wide;                           // level 0: whole waves per run whatever the plan's lanes per run (few samples: runs of >= 16 records)
  // rows as (k-mer, control sum, case sum) triples instead of the test (kmd_merge_sums)
  uint64_t* kmer_out;
  uint64_t* kmer_hi_out;
  unsigned long long* sum_c_out;
  unsigned long long* sum_k_out;
  unsigned long long row_capacity;
  unsigned long long* n_rows;                    // entries written: rows (kmd_merge_sums) or candidate rows (kmd_merge_filter)
  unsigned long long* row_total;                 // candidates mode: [0] distinct k-mers, [1] rows beyond the log-factorial table
  // candidates mode: what the chi-square pre-filter needs of the model (kmd_eval.h, row_may_pass)
  double dTc, dTk, dTcTk, pf_cut, pf_rhs;        // (pf_rhs = pf_cut x dTcTk)
  uint32_t lf_n;
  // ... and its second stage (row_may_pass_kl): control / case shares of the total, the candidate cut, the count
  // sums it is applied below (0: off)
  float kl_qc, kl_qk, kl_cut;
  uint32_t kl_max;
  // candidates mode: the list is handed out in chunks of kOutChunk entries.  Workgroup b of a launch owns chunk
  // first_base / kOutChunk + b from the start (no atomic); *n_entries starts at first_base + n_regions x kOutChunk,
  // further chunks come from it.  Entries of a chunk that stay unused are marked as holes (sum_c = kHole).
  unsigned long long first_base;
  uint32_t n_regions;
  uint32_t* ran;                                 // set by the instantiation that takes the plan's way (a launch that did not is known by it)
  uint32_t* over_n;                              // tiles listed
  uint32_t* over;                                // [i] tile, [over_stride + i] its records (| kAbortBit)
  uint32_t over_stride;
};

// The pre-filter's second stage, for the rows its chi-square bound lets through.  That bound (kmd_eval.h,
// row_may_pass: LR <= (sc Tk - sk Tc)^2 / (n Tc Tk), passed from HALF the candidate cut) is tight where a row's
// split is near the totals' and loose where it is one-sided: a k-mer seen in cases only passes from n = 7 counts
// where the cut (12.6 at p = 5e-7) takes 18.  On data where k-mers are rare and sample-specific -- 3 records per
// row, 36 M rows -- that is most of the list: 6.25 M entries for 1.67 M candidates, each costing the list's
// evaluation two logarithms in double precision.  Here the likelihood ratio itself,
//     LR = sc ln(sc / (n qc)) + sk ln(sk / (n qk)),  qc = Tc / T, qk = Tk / T
// (the lf[k] and -lambda terms of model.hpp:152-156 cancel), in SINGLE precision with the hardware's reciprocal and
// base-2 logarithm (1 ulp each): for count sums below 2^16 its error is below 5e-7 (sc + sk) + 2e-7 (|t_c| + |t_k|)
// <= 0.07, ten times that is allowed for, and a row is dropped only if it stays below the candidate cut with it --
// rows that pass are evaluated exactly as before (k_cand_eval, kmd_filter.hip), rows dropped were no candidates: every
// exposed number is unchanged.  Sums at or beyond the log-factorial table (or 2^16) pass on the first stage alone.
struct kl_consts { float qc, qk, cut; uint32_t max; };
__device__ __forceinline__ bool row_may_pass_kl(const kl_consts& J, unsigned long long sum_c, unsigned long long sum_k)
{
  if (sum_c >= J.max || sum_k >= J.max) return true;                // (also: stage off, max = 0)
  const float sc = (float)(uint32_t)sum_c, sk = (float)(uint32_t)sum_k, n = sc + sk;
  const float lc = __builtin_amdgcn_logf(sc * __builtin_amdgcn_rcpf(n * J.qc));
  const float lk = __builtin_amdgcn_logf(sk * __builtin_amdgcn_rcpf(n * J.qk));
  const float tc = sum_c ? sc * lc : 0.0f, tk = sum_k ? sk * lk : 0.0f;        // (base 2)
  const float lr = 0.69314718f * (tc + tk);
  const float slack = 5e-6f * n + 2e-6f * (__builtin_fabsf(tc) + __builtin_fabsf(tk)) + 1e-3f;
  return !(lr + slack < J.cut);
}

__host__ __device__ inline uint64_t mix64(uint64_t x)
{
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// (Multi-way searches were measured, twice -- round 2: 7 pivots per step, a third of the dependent round trips:
// k_tile_probe 39 -> 61 us, k_tile_fine 38 -> 62 us; 3 pivots per step: 23 -> 29 us, 12 -> 15 us; round 3, 8-ary over
// the index and the 4096-record window: k_tile_bounds 28 -> 61 us, 224 -> 273 us on 36 M rows.  These searches are
// bound by the number of distinct lines and pages they touch in ~1 GB of keys, not by the length of the chain.)
// first index in [lo, hi) whose key is >= (b, bh), hi if none, looked for AROUND a guess g: doubling steps away from it
// until the answer is bracketed, then a bisection of that bracket -- a handful of loads on one or two lines when the
// guess is good (k_tile_bounds interpolates it between two boundaries it knows), 2 log2(hi - lo) when it is not
__device__ __forceinline__ size_t lower_bound_near(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ keys_hi,
                                                   size_t lo, size_t hi, size_t g, uint64_t b, uint64_t bh)
{
  if (lo >= hi) return lo;
  auto less = [&](size_t i) -> bool { return keys_hi ? (keys_hi[i] < bh || (keys_hi[i] == bh && keys[i] < b)) : keys[i] < b; };
  if (g < lo) g = lo;
  if (g >= hi) g = hi - 1;
  size_t a, e;                                                 // the answer is in [a, e]
  if (less(g))
  {
    a = g + 1; e = hi;
    for (size_t step = 1; a + step - 1 < hi; step <<= 1)
    {
      if (less(a + step - 1)) a += step; else { e = a + step - 1; break; }
    }
    if (a > e) a = e;
  }
  else
  {
    a = lo; e = g;
    for (size_t step = 1; e >= lo + step; step <<= 1)
    {
      if (!less(e - step)) e -= step; else { a = e - step + 1; break; }
    }
  }
  while (a < e) { const size_t mid = a + ((e - a) >> 1); if (less(mid)) a = mid + 1; else e = mid; }
  return a;
}

// The boundary searches were bound by their dependent round trips to HBM: 21 steps of bisection over a 2.6 M-record
// stream, each a miss on a line (and often a page) of its own -- 23 + 15 + 30 us of a 0.5 ms call in round 2's
// k_tile_probe / k_tile_coarse / k_tile_fine.  Now:
//   * a sampled INDEX of every stream (its every 4096th key and its last one; ~200 KB for a 20v20 partition:
//     L2-resident) is searched first: ~10 steps that hit the cache, and the answer is confined to 4096 records;
//   * inside that window the key's place is guessed by INTERPOLATION between the two index keys that bound it (a
//     partition's k-mers are spread roughly evenly: they are hashed into partitions by their minimizers), the guess
//     is bracketed by galloping away from it (16, 64, 256 ... records: a couple of steps, on the guess's own cache
//     lines) and the bracket bisected.  Whatever the keys look like -- dense clusters, gaps -- the search stays within
//     the window: at most ~30 steps, ~8 for evenly spread keys, two or three of them misses.
constexpr uint32_t kIndexShift = 12;             // every 4096th key of a stream is in its index

// samples of a stream of n records: positions 0, 4096, ... and n - 1
__host__ __device__ inline uint32_t index_samples(uint64_t n) { return n ? (uint32_t)((n - 1) >> kIndexShift) + 2u : 0u; }
__device__ __forceinline__ size_t index_pos(size_t begin, uint64_t n, uint32_t i) { const uint64_t p = (uint64_t)i << kIndexShift; return begin + (size_t)(p < n - 1 ? p : n - 1); }

// one thread per sample: idx[ioff[s] + i] = key at the i-th sample position of stream s (ioff on the host: prefix of index_samples)
__global__ void __launch_bounds__(256) k_tile_index(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ keys_hi, const uint64_t* __restrict__ offs,
                                                    const uint32_t* __restrict__ ioff, uint32_t S, uint64_t* __restrict__ idx, uint64_t* __restrict__ idx_hi,
                                                    uint32_t* __restrict__ zero, uint32_t zero_words)
{
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < zero_words) zero[t] = 0u;                            // (the call's small block of counters: no memset of its own)
  if (t >= ioff[S]) return;
  uint32_t lo = 0, hi = S;                                     // stream of sample t: the last s with ioff[s] <= t
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (ioff[mid] <= t) lo = mid; else hi = mid; }
  const size_t at = index_pos(offs[lo], offs[lo + 1] - offs[lo], t - ioff[lo]);
  idx[t] = keys[at];
  if (keys_hi) idx_hi[t] = keys_hi[at];
}

struct stream_index { const uint64_t* idx; const uint64_t* idx_hi; const uint32_t* ioff; };

// first position in stream s (records [begin, end)) whose key is >= (b, bh)
__device__ __forceinline__ size_t lower_bound_indexed(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ keys_hi, const stream_index X,
                                                      uint32_t s, size_t begin, size_t end, uint64_t b, uint64_t bh)
{
  if (begin >= end) return begin;
  const bool two = keys_hi != nullptr;
  const uint64_t n = end - begin;
  const uint64_t* ix = X.idx + X.ioff[s];
  const uint64_t* ixh = two ? X.idx_hi + X.ioff[s] : nullptr;
  const uint32_t m = X.ioff[s + 1] - X.ioff[s];
  auto iless = [&](uint32_t i) -> bool { return two ? (ixh[i] < bh || (ixh[i] == bh && ix[i] < b)) : ix[i] < b; };
  // first sample whose key is >= the key
  uint32_t ilo = 0, ihi = m;
  while (ilo < ihi) { const uint32_t mid = (ilo + ihi) >> 1; if (iless(mid)) ilo = mid + 1; else ihi = mid; }
  if (ilo == 0) return begin;                                  // key <= the stream's first
  if (ilo == m) return end;                                    // key > its last
  // keys[lo] < key <= keys[hi1]
  size_t lo = index_pos(begin, n, ilo - 1), hi1 = index_pos(begin, n, ilo);
  auto val = [&](size_t i) -> uint64_t { return two ? keys_hi[i] : keys[i]; };     // (interpolated on; ties fall to the halving steps)
  auto less = [&](size_t i) -> bool { return two ? (keys_hi[i] < bh || (keys_hi[i] == bh && keys[i] < b)) : keys[i] < b; };
  const uint64_t target = two ? bh : b;
  uint64_t vlo = two ? ixh[ilo - 1] : ix[ilo - 1], vhi = two ? ixh[ilo] : ix[ilo];
  for (int pass = 0; hi1 - lo > 1; ++pass)
  {
    const size_t width = hi1 - lo;
    if (width <= 8 || pass >= 2 || vhi <= vlo || target <= vlo)
    {
      const size_t mid = lo + (width >> 1);
      if (less(mid)) lo = mid; else hi1 = mid;
      continue;
    }
    size_t g = lo + (size_t)((double)(target - vlo) / (double)(vhi - vlo) * (double)width);
    if (g <= lo) g = lo + 1;
    if (g >= hi1) g = hi1 - 1;
    if (less(g))
    {
      lo = g;
      size_t step = 16;
      while (lo + step < hi1 && less(lo + step)) { lo += step; step <<= 2; }
      if (lo + step < hi1) hi1 = lo + step;
    }
    else
    {
      hi1 = g;
      size_t step = 16;
      while (hi1 > lo + step && !less(hi1 - step)) { hi1 -= step; step <<= 2; }
      if (hi1 > lo + step) lo = hi1 - step;
    }
    vlo = val(lo); vhi = val(hi1);
  }
  return hi1;
}

// mult[p] = number of streams that hold the k-mer of probe record p (records drawn uniformly from all
// n, so mean(1 / mult) estimates rows / records without bias); one thread per (probe, stream).
// In the same launch, behind the probe's workgroups: the COARSE boundaries -- where every stream meets every R0-th
// key of the longest stream L (C of them; coarse[c * S + s], c = 0: the stream's begin, c = C: its end).  They do not
// depend on the plan, which the probe is there to make, so the two chains of memory latencies run side by side; a
// tile boundary is then looked for between the two coarse ones around it (k_tile_bounds): a short search, on lines
// its neighbours touch too.
__global__ void __launch_bounds__(256) k_tile_probe(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ keys_hi,
                                                    const uint64_t* __restrict__ offs, const stream_index X, uint32_t S, uint64_t n,
                                                    uint32_t* __restrict__ mult, uint32_t L, uint32_t R0, uint32_t C, uint32_t* __restrict__ coarse)
{
  const uint32_t probe_blocks = (probes_for(S) * S + 255u) / 256u;
  if (blockIdx.x >= probe_blocks)
  {
    const size_t t = (size_t)(blockIdx.x - probe_blocks) * 256 + threadIdx.x;
    if (t >= ((size_t)C + 1) * S) return;
    // (consecutive threads: consecutive boundaries of ONE stream -- their answers lie on the same pages)
    const uint32_t s = (uint32_t)(t / ((size_t)C + 1)), c = (uint32_t)(t - (size_t)s * ((size_t)C + 1));
    const size_t begin = offs[s], end = offs[s + 1];
    size_t pos;
    if (c == 0) pos = begin;
    else if (c >= C) pos = end;
    else
    {
      const size_t at = offs[L] + (size_t)c * R0;
      pos = s == L ? at : lower_bound_indexed(keys, keys_hi, X, s, begin, end, keys[at], keys_hi ? keys_hi[at] : 0ull);
    }
    coarse[(size_t)c * S + s] = (uint32_t)pos;
    return;
  }
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= probes_for(S) * S) return;
  const uint32_t p = t / S, s = t - p * S;
  const size_t i = (size_t)__umul64hi(mix64(p), n);
  const uint64_t k = keys[i], kh = keys_hi ? keys_hi[i] : 0ull;
  const size_t begin = offs[s], end = offs[s + 1];
  const size_t at = lower_bound_indexed(keys, keys_hi, X, s, begin, end, k, kh);
  if (at < end && keys[at] == k && (!keys_hi || keys_hi[at] == kh)) atomicAdd(&mult[p], 1u);
}

// the plan: records per tile such that its distinct k-mers fill `load` of the table.  Every workgroup of
// k_tile_coarse works it out for itself from the probe's 512 counts (the first one writes it down for the
// kernels behind): no launch of its own.
__device__ __forceinline__ tile_plan make_plan(const uint32_t* __restrict__ mult, uint64_t n, uint64_t n_l, uint32_t S,
                                               uint32_t slots_fixed, float load, uint32_t fill_fixed, uint32_t g_fixed, uint32_t grid_hint_small,
                                               uint32_t grid_hint_big, double* s_part)
{
  double acc = 0;
  for (uint32_t p = threadIdx.x; p < probes_for(S); p += blockDim.x) { const uint32_t m = mult[p]; acc += 1.0 / (double)(m ? m : 1u); }
  // (a fixed order of the sum: every workgroup must arrive at the same plan)
  s_part[threadIdx.x] = acc;
  __syncthreads();
  for (uint32_t o = blockDim.x >> 1; o > 0; o >>= 1) { if (threadIdx.x < o) s_part[threadIdx.x] += s_part[threadIdx.x + o]; __syncthreads(); }
  const double rho = (double)probes_for(S) / s_part[0];                    // records per row
  // The table: 2048 slots for a workgroup of 512 threads, or 4096 for 1024 (kBigSlots) -- half the tiles, runs twice
  // as long.  Where rows have few records a tile's run of a sample is short -- 36 M rows of 3 records from 40
  // samples: 73 records, a full round of a wave and a round for the 9 left over -- and what a tile costs whatever it
  // holds (barriers, its segment table, the walk's fixed part) is most of it: the larger shape wins from ~130
  // records per run down (measured, whole call, small / large shape: runs of 665 records 0.42 / 0.46 ms, 400
  // 0.46 / 0.50, 200 0.59 / 0.58, 133 0.72 / 0.67, 73 1.06 / 0.94; 8 or 200 samples with runs of 665: 0.52 / 0.55,
  // 0.59 / 0.61).
  // Round 5: ... and from ~2 x 10^8 records up whatever the runs' length.  The measurements above were taken on 4 M-row
  // partitions; at the size of the job the larger shape wins (whole call, small / large, same box: 20v20 2 M rows 0.305 /
  // 0.315 ms, 4 M 0.396 / 0.425, 8 M 0.657 / 0.652, 16 M 1.157 / 1.142, 39 M -- one configs[2] partition -- 2.59 / 2.42;
  // the MIXED partition 2.15 / 1.97; 100v100 1.6 M rows 0.797 / 0.772, 7.8 M 3.10 / 2.66; 4v4 16 M rows 0.418 / 0.447,
  // 78 M 1.54 / 1.47: profiles/r05_shape_size.txt): half the tiles -- half the boundary searches, half the per-tile
  // barriers and walks' fixed parts -- once there are enough of them to keep every workgroup of either grid busy to the end.
  const bool many_records = n >= kBigFromRecords;
  const uint32_t slots = slots_fixed ? slots_fixed : ((many_records || rho * (double)load * (double)kSmallSlots / (double)S < 160.0) ? kBigSlots : kSmallSlots);
  const uint32_t grid_hint = slots == kBigSlots ? grid_hint_big : grid_hint_small;
  double fill = rho * (double)load * (double)slots;
  const double fill_max = 24.0 * (double)slots;                      // ~0.6 MB of records per tile at most
  if (fill > fill_max) fill = fill_max;
  if (fill < 64.0) fill = 64.0;
  if (fill_fixed) fill = (double)fill_fixed;
  uint64_t r = (uint64_t)((double)n_l * fill / (double)n);           // every r-th key of the longest stream
  if (r < 1) r = 1;
  uint64_t nb = (n_l + r - 1) / r;
  if (nb < 1) nb = 1;
  // a whole number of tiles per workgroup of the persistent grid (3963 tiles on 1024 workgroups: one in eight
  // workgroups idles through the last quarter of the kernel): somewhat smaller tiles, at most the next multiple
  // Lanes per run are chosen from the fill BEFORE that rounding: the host skips the launch of the sub-group
  // instantiation when every plan it can predict takes whole waves (wide_for_sure in tile_merge: fill >= load x
  // slots), and the rounding -- which may shrink a tile to a little over half -- must not take the plan to the
  // instantiation that was not launched (the partition's rows would be lost without an error).
  const double fill_planned = fill;
  if (grid_hint && !fill_fixed && nb > grid_hint && r > 1)
  {
    const uint64_t target = (nb + grid_hint - 1) / grid_hint * grid_hint;
    // (only where it costs at most a quarter of a tile: 1100 tiles on 1024 workgroups stay 1100, not 2048 half-filled ones)
    if (target * 4 <= nb * 5)
    {
      r = (n_l + target - 1) / target;
      if (r < 1) r = 1;
      nb = (n_l + r - 1) / r;
      fill = (double)r * (double)n / (double)n_l;
    }
  }
  // lanes per run: the power of two nearest the average run (one round of a sub-group takes most of it)
  const double run = fill_planned / (double)S;
  uint32_t g = 3;
  while (g < 6 && (double)(1u << g) < run * 0.75) ++g;
  // (whole waves from runs of 14 records up: the whole-wave path -- hand-issued loads, the hand-written round -- beats
  // the sub-group one, the older code, wherever it was tried against it: 200 samples, runs of 35 / 26 records: 1.56
  // against 2.11 ms, 2.16 against 2.65; 600 samples, runs of 38 / 16: 1.19 against 1.72, 2.31 against 2.54;
  // tools/sweep_way.sh)
  if (run >= 14.0) g = 6;
  if (g_fixed) g = g_fixed;
  tile_plan pl;
  pl.r = (uint32_t)r; pl.nb = (uint32_t)nb; pl.g_shift = g; pl.fill = (uint32_t)fill; pl.rho = (float)rho;
  pl.slots = slots; pl.pad[0] = pl.pad[1] = 0;
  return pl;
}

// where stream s enters tile j: boundary j = key j * r of the longest stream L (b_0 = -inf, b_nb = +inf); one thread
// per (boundary, stream), each a search of its stream (lower_bound_indexed).  Every workgroup works the plan out for
// itself from the probe's counts (the first one writes it down for the kernels behind): no launch of its own.
__global__ void __launch_bounds__(256) k_tile_bounds(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ keys_hi,
                                                     const uint64_t* __restrict__ offs, uint32_t S, uint32_t L,
                                                     const uint32_t* __restrict__ mult, uint64_t n, uint64_t n_l, uint32_t slots, float load,
                                                     uint32_t fill_fixed, uint32_t g_fixed, uint32_t grid_hint, uint32_t grid_hint_big,
                                                     tile_plan* __restrict__ plan, uint32_t* __restrict__ start,
                                                     unsigned long long* __restrict__ list_len, unsigned long long list_len0,
                                                     uint32_t R0, uint32_t C, const uint32_t* __restrict__ coarse)
{
  __shared__ double s_part[256];
  const tile_plan pl = make_plan(mult, n, n_l, S, slots, load, fill_fixed, g_fixed, grid_hint, grid_hint_big, s_part);
  if (blockIdx.x == 0 && threadIdx.x == 0) { *plan = pl; *list_len = list_len0; }     // (candidates mode: the workgroups' first chunks are spoken for)
  const uint32_t nb = pl.nb, r = pl.r;
  // (consecutive threads: consecutive boundaries of ONE stream -- their answers lie a tile's run apart, on the same
  // pages: the searches are bound by the address translations they miss)
  // (a grid of a fixed modest size strides over the cells: how many there are is known here, not on the host, whose
  // upper bound -- every record a row of its own -- is two orders of magnitude too many with 200 samples)
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ((size_t)nb + 1) * S; i += (size_t)gridDim.x * blockDim.x)
  {
    const uint32_t s = (uint32_t)(i / ((size_t)nb + 1));
    const size_t j = i - (size_t)s * ((size_t)nb + 1);
    const size_t begin = offs[s], end = offs[s + 1];
    size_t pos;
    if (j == 0) pos = begin;
    else if (j >= nb) pos = end;
    else
    {
      const size_t at = offs[L] + j * r;
      if (s == L) pos = at;
      else
      {
        // (between the coarse boundaries around it: key c R0 <= this key <= key (c + 1) R0 of L, and so are their positions here)
        // (... and about as far along between them as the key is between theirs, if k-mers are spread evenly there:
        // the guess the search starts from -- a bisection of the bracket touched six lines nobody else wanted)
        uint32_t c = (uint32_t)((j * r) / R0);
        if (c >= C) c = C - 1u;
        const size_t lo = coarse[(size_t)c * S + s], hi = coarse[(size_t)(c + 1u) * S + s];
        const size_t at_c = offs[L] + (size_t)c * R0, at_n = offs[L] + (c + 1u < C ? (size_t)(c + 1u) * R0 : (size_t)n_l - 1);
        const bool on_hi = keys_hi && keys_hi[at_c] != keys_hi[at_n];
        const uint64_t kc = on_hi ? keys_hi[at_c] : keys[at_c], kn = on_hi ? keys_hi[at_n] : keys[at_n], kj = on_hi ? keys_hi[at] : keys[at];
        size_t g = lo;
        if (kn > kc && kj >= kc) g = lo + (size_t)((double)(kj - kc) / (double)(kn - kc) * (double)(hi - lo));
        pos = lower_bound_near(keys, keys_hi, lo, hi, g, keys[at], keys_hi ? keys_hi[at] : 0ull);
      }
    }
    start[j * S + s] = (uint32_t)pos;
  }
}

// A listed tile becomes m equal slices of the key range its records really span: rows first ..
// first + m of the next table (the last one only closes slice m - 1).  One workgroup per listed tile.
// Two-limb keys are cut on 64 bits chosen by `shift` from the 128 (the tile's keys agree above them).
__device__ __forceinline__ uint64_t cut_bits(uint64_t lo, uint64_t hi, int shift)
{
  // bits [shift, shift + 64) of the 128-bit key
  return shift == 0 ? lo : shift >= 64 ? (hi >> (shift - 64)) : ((hi << (64 - shift)) | (lo >> shift));
}

__global__ void __launch_bounds__(256) k_tile_refine(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ keys_hi,
                                                     const uint32_t* __restrict__ table, uint32_t S,
                                                     const uint32_t* __restrict__ tile_of, const uint32_t* __restrict__ slices,
                                                     const uint32_t* __restrict__ first, uint32_t* __restrict__ out,
                                                     uint8_t* __restrict__ todo)
{
  __shared__ unsigned long long s_lo, s_hi, s_lo_h, s_hi_h;
  __shared__ int s_shift;
  const uint32_t i = blockIdx.x, s = threadIdx.x;
  const uint32_t tile = tile_of[i], m = slices[i], row0 = first[i];
  if (s == 0) { s_lo = ~0ull; s_hi = 0; s_lo_h = ~0ull; s_hi_h = 0; s_shift = 0; }
  __syncthreads();
  if (keys_hi)
  {
    // 128-bit span: first the high limbs' extremes, then the cut window
    for (uint32_t q = s; q < S; q += blockDim.x)
    {
      const uint32_t b = table[(size_t)tile * S + q], e = table[((size_t)tile + 1) * S + q];
      if (e > b) { atomicMin(&s_lo_h, (unsigned long long)keys_hi[b]); atomicMax(&s_hi_h, (unsigned long long)keys_hi[e - 1]); }
    }
    __syncthreads();
    if (s == 0)
    {
      // the tile's keys agree on the bits above the highest bit in which its extreme HIGH limbs differ;
      // cut on the 64 bits from there down (all of the low limb when the high limbs are equal)
      const unsigned long long x = s_lo_h ^ s_hi_h;
      s_shift = x ? 64 - __builtin_clzll(x) : 0;
    }
    __syncthreads();
  }
  const int shift = s_shift;
  for (uint32_t q = s; q < S; q += blockDim.x)
  {
    const uint32_t b = table[(size_t)tile * S + q], e = table[((size_t)tile + 1) * S + q];
    if (e > b)
    {
      const uint64_t kb = keys_hi ? cut_bits(keys[b], keys_hi[b], shift) : keys[b];
      const uint64_t ke = keys_hi ? cut_bits(keys[e - 1], keys_hi[e - 1], shift) : keys[e - 1];
      atomicMin(&s_lo, (unsigned long long)kb);
      atomicMax(&s_hi, (unsigned long long)ke);
    }
  }
  __syncthreads();
  for (uint32_t t = s; t <= m; t += blockDim.x) todo[row0 + t] = t < m ? 1 : 0;
  const uint64_t k0 = s_lo, step = (s_hi - s_lo) / m + 1;           // m slices of this width cover [lo, hi]
  for (uint32_t q = s; q < S; q += blockDim.x)
  {
    const uint32_t b = table[(size_t)tile * S + q], e = table[((size_t)tile + 1) * S + q];
    out[(size_t)row0 * S + q] = b;
    uint32_t lo = b;
    for (uint32_t t = 1; t < m; ++t)
    {
      uint64_t bound = k0 + (uint64_t)t * step;
      if (__umul64hi((uint64_t)t, step) != 0 || bound < k0) bound = ~0ull;     // saturate: past every key
      uint32_t hi = e;                                                          // first record in [lo, e) with key >= bound
      while (lo < hi)
      {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        const uint64_t k = keys_hi ? cut_bits(keys[mid], keys_hi[mid], shift) : keys[mid];
        if (k < bound) lo = mid + 1; else hi = mid;
      }
      out[((size_t)row0 + t) * S + q] = lo;
    }
    out[((size_t)row0 + m) * S + q] = e;
  }
}

constexpr int ilog2_c(uint32_t v) { return v <= 1 ? 0 : 1 + ilog2_c(v >> 1); }

// inclusive prefix sum over the wave's 64 lanes, six DPP adds: shifts by 1, 2, 4, 8 within rows of 16 lanes, then a row's
// last lane into the next row (row_bcast:15, rows 1 and 3) and lane 31 into the upper half (row_bcast:31).  A lane
// without a source, or masked out of a step, adds 0.  (Written with shuffles and `lane >= o` it was six LDS-crossbar
// round trips and six lane masks the compiler kept in scalar registers for the life of the kernel.)
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t x)
{
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);      // row_shr:1
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);      // row_shr:2
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);      // row_shr:4
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);      // row_shr:8
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1, 3
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2, 3
  return x;
}

// LDS of one workgroup, carved from the dynamic allocation (the segment tables follow it)
// kSum32: a slot's two sums are 32-bit (16 bytes per slot with the key: four workgroups per CU instead of three,
// half the bytes per atomic add); kmd_tilemerge keeps 64-bit sums for the tiles that need them.
template <uint32_t kSlots, int kWaves, bool kTwo, bool kSum32>
struct tile_lds
{
  // kSlots of the main table, kSlots / 16 of the second table behind it (whole-wave path: where the ~1 % of the k-mers
  // go that find their four home slots taken), and a spare slot behind every array (+ 2 / + 4) -- where the
  // whole-wave path lets lanes that have nothing to add add it: its key is 0, never the empty marker, and nothing
  // reads its sums
  static constexpr uint32_t kAll = kSlots + kSlots / 16;
  static constexpr uint32_t kStage = stage_rows(kSlots);
  using stage_sum_t = typename std::conditional<kSum32, uint32_t, unsigned long long>::type;
  // (the small fields FIRST: a DS instruction's offset field is 16 bits, and the address of a field that lies beyond
  // 64 KB -- behind a 4096-slot table -- is a scalar the compiler computes once, keeps for the life of the kernel and,
  // out of scalar registers, spills into vector lanes: round 5's build read eight such addresses back per walk)
  unsigned long long maxsum[2];
  unsigned long long max_hi[2];                          // kTwo: min / max high limb of the records whose low limb is all ones
  unsigned long long base;
  unsigned long long out_base;                           // candidates mode: the workgroup's current chunk of the list
  uint32_t out_used, out_cap, stage_n, pad0;
  uint32_t n[2], fresh[2], abort[2], big[2];
  uint32_t hasmax, bad;
  uint32_t wcnt[kWaves];
  uint32_t grp[2][kMaxGroups + 1];                       // records of each group of kGroup runs of the current / the next tile
  unsigned long long stage_key[kStage], stage_hi[kTwo ? kStage : 1];     // rows parked for the list (k_tile_sums, the walk)
  stage_sum_t stage_c[kStage], stage_k[kStage];
  uint8_t queue[kWaves][kQueue];                         // candidates mode, the walk: live slots of each wave (one byte each), 64 of them evaluated at a time
  alignas(16) unsigned long long key[kAll + 2];
  uint32_t c32[kSum32 ? kAll + 4 : 4];                   // kSum32: control sum of slot i (arrays of their own: a round adds to ONE
  uint32_t k32[kSum32 ? kAll + 4 : 4];                   // of them -- the run is a control's or a case's -- so its lanes spread over all banks)
  unsigned long long sc[kSum32 ? 2 : kAll + 2];
  unsigned long long sk[kSum32 ? 2 : kAll + 2];
  unsigned long long key_hi[kTwo ? kAll + 2 : 2];
  unsigned long long hi_min[kTwo ? kAll + 2 : 2];        // see k_tile_sums: every record's high limb must agree with its slot's
};

// One workgroup per tile on a persistent grid.  Per tile, candidates mode (kFilter):
//   [inserts; wave 0: the rows the previous tile parked -> the list] barrier [segment table of the NEXT tile |
//   walk of the table = rows of this tile: the rows that leave are parked in LDS, every slot is wiped] barrier
// -- both barriers wait for LDS only -- and rows mode (kmd_merge_sums: every row leaves, compact output):
//   [inserts] barrier [segment table | walk: count the rows] barrier [one workgroup-wide reservation in the output]
//   barrier [write them, wipe the slots] barrier.
// Tile order: each XCD (workgroup b runs on XCD b % 8) takes one contiguous eighth of the tiles and its
// workgroups stride through it, so tiles that share cache lines at the ends of their runs meet in one L2.
//
// The table holds DISTINCT k-mers, and how many a tile has is not known beforehand (the plan sizes tiles
// by an estimate of records per row).  A probe sequence is bounded (kMaxProbe slots): a table that fills
// up shows as a sequence that does not end, the tile gives up -- its slots are wiped, it is listed, the
// host cuts it.  (The sub-group path, !kWide, also counts the slots its waves claim and gives up past
// 3/4 full; the whole-wave path dropped the count: it cost more than the tiles it saved.)
//
// kSum32: a slot's control and case sums are 32-bit (arrays of their own).  1024 samples of counts below
// 2^22 cannot overflow them; a record with a larger count flags the tile, which is listed like one that
// gave up and redone by the 64-bit instantiation.
//
// kTwo (32 < k <= 64): the table is keyed by the LOW limb.  Within a tile the high limbs (62 bits of a
// k = 63 k-mer) almost always agree wherever the low limbs do -- they are the slowly varying part of a
// sorted range -- but nothing guarantees it, so every record also folds its high limb into its slot's
// minimum and maximum (two more LDS atomics); a slot whose minimum and maximum differ holds two k-mers
// that share a low limb: the tile is then listed and cut again, which separates them (their high limbs
// differ, so some slice boundary falls between them; at the latest when a slice is a single value of the
// cut window).  A low limb of all ones (the empty marker) is kept apart the same way.
template <int kThreads, uint32_t kSlots, bool kFilter, bool kTwo, bool kWide, bool kSum32>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(kSlots == kBigSlots ? KMD_TILE_WAVES_BIG(kSum32, kTwo, kWide) : KMD_TILE_WAVES(kSum32, kTwo, kWide))))
k_tile_sums(const tile_job J)
{
  constexpr uint32_t kMask = kSlots - 1;
  constexpr int kShift = 32 - ilog2_c(kSlots);
  constexpr int kWaves = kThreads / 64;
  constexpr uint32_t kSec = kSlots / 16, kAll = kSlots + kSec;   // second table (whole-wave path), all slots
  constexpr int kWalk = (int)((kAll + kThreads - 1) / kThreads);   // table slots per thread in the walk
  constexpr int kU = KMD_TILE_U;                         // records per lane and round
  constexpr int kDepth = KMD_TILE_DEPTH;                 // rounds in flight per wave
  constexpr uint32_t kFullAt = kSlots / 4 * 3;           // distinct k-mers at which a tile gives up
  constexpr uint32_t kMaxProbe = 96;
  static_assert((kSlots & kMask) == 0 && kSlots % kThreads == 0, "shape");
  static_assert(kWalk <= 32, "walk bits");
  using lds_t = tile_lds<kSlots, kWaves, kTwo, kSum32>;
  constexpr uint32_t kStage = lds_t::kStage;
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_raw[];      // (16: a bucket of two keys is one ds_read_b128)
  lds_t& M = *reinterpret_cast<lds_t*>(s_raw);
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  // The job description is 250 bytes of kernel arguments.  Read through `J`, every field is fetched once at the top and
  // kept for the life of the kernel -- with 64 VGPRs per wave (8 waves per SIMD) that is ~80 scalar registers for ~110
  // live values: round 5's build spilled 121 of them into vector lanes and read them back with v_readlane in the middle
  // of the walk (83 reloads per tile and wave).  So only what the insert loop needs lives in `J`'s registers (the
  // streams, S, nc, the boundary table); everything a colder phase needs -- the pre-filter's constants, the list's
  // pointers, the over list -- is read where it is used, with s_load, through a pointer to the kernel arguments the
  // compiler cannot see through (job(): it cannot hoist the loads above the statement that made the pointer).
  typedef __attribute__((address_space(4))) const tile_job job_c;
  auto job = [&]() -> job_c*
  {
    job_c* p = (job_c*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("; job" : "+s"(p));
    return p;
  };
  const uint32_t S = J.S;
  {
    job_c* const J0 = job();
    const uint32_t g_shift0 = J0->n_tiles ? J0->g_shift : J0->force_wide ? 6u : J0->plan->g_shift;
    // two instantiations, both launched at level 0 where the plan is on the device: the one whose way of
    // streaming a tile (kWide: whole waves per run, g_shift 6; else sub-groups of lanes) the plan did
    // not choose returns at once
    if ((g_shift0 == 6) != kWide) return;
    // ... and so does the one of the table shape the plan did not choose
    if (!J0->n_tiles && J0->plan->slots != kSlots) return;
    if (blockIdx.x == 0 && tid == 0) *J0->ran = 1u;
  }
  const uint32_t n_tiles = J.n_tiles ? J.n_tiles : J.plan->nb;
  [[maybe_unused]] const uint32_t g_shift = J.n_tiles ? J.g_shift : J.force_wide ? 6u : J.plan->g_shift;
  // segment tables of the current and the next tile, behind the fixed part: [2][begin[S] | length[S] | prefix[S]]
  uint32_t* const s_seg = reinterpret_cast<uint32_t*>(s_raw + (sizeof(lds_t) + 7) / 8);

  // this workgroup's tiles: tile_first, tile_first + stride, ... below tile_end
  uint32_t tile_first = blockIdx.x, tile_end = n_tiles, stride = gridDim.x;
  if (J.xcd_order && (gridDim.x & 7u) == 0 && n_tiles >= gridDim.x)
  {
    const uint32_t x = blockIdx.x & 7u, per = gridDim.x >> 3, chunk = (n_tiles + 7u) >> 3;
    tile_first = x * chunk + (blockIdx.x >> 3);
    tile_end = (x + 1u) * chunk < n_tiles ? (x + 1u) * chunk : n_tiles;
    stride = per;
  }

  // a slot's sums: two 64-bit words, or (kSum32) the halves of one
  auto wipe_sums = [&](uint32_t i)
  {
    if constexpr (kSum32) { M.c32[i] = 0; M.k32[i] = 0; }
    else { M.sc[i] = 0; M.sk[i] = 0; }
  };
  auto read_sums = [&](uint32_t i, unsigned long long& c, unsigned long long& k)
  {
    if constexpr (kSum32) { c = M.c32[i]; k = M.k32[i]; }
    else { c = M.sc[i]; k = M.sk[i]; }
  };
  for (uint32_t i = tid; i < kAll; i += kThreads)
  {
    M.key[i] = kEmptyKey; wipe_sums(i);
    if constexpr (kTwo) { M.key_hi[i] = 0; M.hi_min[i] = ~0ull; }
  }
  if (tid == 0)
  {
    M.key[kAll] = 0; M.key[kAll + 1] = 0;                  // the spare slot: never the empty marker
    M.out_base = job()->first_base + (unsigned long long)blockIdx.x * kOutChunk; M.out_used = 0; M.out_cap = kOutChunk; M.stage_n = 0;
    M.n[0] = 0; M.n[1] = 0; M.fresh[0] = 0; M.fresh[1] = 0; M.abort[0] = 0; M.abort[1] = 0; M.big[0] = 0; M.big[1] = 0;
    M.hasmax = 0; M.bad = 0; M.maxsum[0] = 0; M.maxsum[1] = 0;
    M.max_hi[0] = ~0ull; M.max_hi[1] = 0;
  }
  __syncthreads();
  // The segment table of a tile: where each stream's run begins, how long it is, and -- ONE scan per tile, by the
  // wave that fetches the run (round 5: every wave of the workgroup scanned the tile's runs for itself at the top of
  // its inserts, 16 x 6 shuffle steps per tile) -- how many records the runs up to and including it hold, counted
  // within groups of kGroup runs (one wave, one lane per run; grp[] = a group's records, n = the tile's).
  auto load_segments = [&](uint32_t tile, uint32_t buf)
  {
    if (tile >= tile_end) return;
#if KMD_TILE_ABLATE & 32   // dev: every workgroup streams the same 32 tiles again and again (cache-resident: the inserts alone; results wrong)
    tile &= 31u;
#endif
    uint32_t* const beg = s_seg + (size_t)buf * 3 * S;
    const uint32_t n_grp = (S + kGroup - 1u) / kGroup;
    for (uint32_t g = wave; g < n_grp; g += (uint32_t)kWaves)
    {
      const uint32_t s = g * kGroup + lane;
      const bool ok = lane < kGroup && s < S;
      uint32_t b = 0, e = 0;
      if (ok) { b = J.start[(size_t)tile * S + s]; e = J.start[((size_t)tile + 1) * S + s]; }
      const uint32_t incl = wave_scan_incl(e - b);
      if (ok) { beg[s] = b; beg[S + s] = e - b; beg[2 * S + s] = incl; }
      if (lane == 63) { M.grp[buf][g] = incl; if (incl) atomicAdd(&M.n[buf], incl); }
    }
  };
  load_segments(tile_first, 0);
  __syncthreads();
  // a workgroup barrier that waits for this wave's LDS operations only (__syncthreads also waits for its global
  // stores to be acknowledged -- microseconds, with the whole workgroup standing by)
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  // candidates mode: the rows a tile's walk parked in LDS go to the list (wave 0; the next entries of the workgroup's
  // chunk and, should they not all fit, the first ones of a fresh chunk)
  auto flush_stage = [&]()
  {
    if constexpr (kFilter)
    {
      if (wave != 0) return;
      const uint32_t n_st = M.stage_n < kStage ? M.stage_n : kStage;
      if (n_st == 0) return;
      job_c* const Jf = job();
      unsigned long long base = M.out_base;
      uint32_t used = M.out_used < M.out_cap ? M.out_used : M.out_cap;
      const uint32_t cap = M.out_cap;
      const unsigned long long row_cap = Jf->row_capacity;
      unsigned long long* const sum_c_out = Jf->sum_c_out;
      // (the rows fill what is left of the workgroup's chunk and go on at the head of a fresh one: no holes but at the
      // kernel's end -- with up to kStage rows per tile, abandoning a chunk's tail made a fifth of the list holes, which the
      // candidates' kernels then walk: +11 us of k_cand_eval on rows of 3 records)
      const uint32_t room = cap - used;                             // entries left in the current chunk
      unsigned long long fresh_base = 0;
      if (n_st > room)
      {
        unsigned long long fresh = 0;
        if (lane == 0) fresh = atomicAdd(Jf->n_rows, (unsigned long long)kOutChunk);
        fresh_base = __shfl(fresh, 0, 64);
      }
      static_assert(kStage <= kOutChunk, "a tile's parked rows fit one chunk");
      for (uint32_t i = lane; i < n_st; i += 64)
      {
        const unsigned long long e = i < room ? base + used + i : fresh_base + (i - room);
        if (e < row_cap)
        {
          Jf->kmer_out[e] = M.stage_key[i]; sum_c_out[e] = M.stage_c[i]; Jf->sum_k_out[e] = M.stage_k[i];
          if constexpr (kTwo) Jf->kmer_hi_out[e] = M.stage_hi[i];
        }
      }
      if (n_st > room)
      {
        if (lane == 0) { M.out_base = fresh_base; M.out_cap = kOutChunk; M.out_used = n_st - room; M.stage_n = 0; }
      }
      else
      if (lane == 0) { M.out_used = used + n_st; M.stage_n = 0; }
    }
  };

  const uint32_t G = 1u << g_shift, sub = tid & (G - 1u), q0 = tid >> g_shift, Q = (uint32_t)kThreads >> g_shift;
  [[maybe_unused]] uint32_t beyond_wave = 0, rows_wave = 0;           // candidates mode: this WAVE's rows / rows beyond the log-factorial table (scalar)
  struct batch
  {
    uint64_t k[kU], kh[kTwo ? kU : 1];
    uint32_t c[kU];
    uint32_t valid, ctl;                                 // bit u: record u is there / is a control sample's
  };

  uint32_t it = 0;
  for (uint32_t tile = tile_first; tile < tile_end; tile += stride, ++it)
  {
#if KMD_TILE_TIMING
    const unsigned long long tp_tile = __builtin_readcyclecounter();
#endif
    const uint32_t buf = it & 1u;
    const uint32_t n = M.n[buf];
    const bool wanted = J.todo == nullptr || J.todo[tile] != 0;
    const bool process = wanted && n > 0;
    if (tid == 0) { M.n[buf ^ 1u] = 0; M.fresh[buf ^ 1u] = 0; M.abort[buf ^ 1u] = 0; M.big[buf ^ 1u] = 0; }

    // ---- inserts, kWide: runs of a wave's worth of records and more (the usual case: a tile is sized to
    // hold ~700 rows).  A wave takes whole runs, one after the other -- runs wave, wave + kWaves, ... --
    // 64 records at a time, one per lane.  Everything about WHERE is wave-uniform and lives in scalar
    // registers: the position in the run, how many of the 64 lanes hold a record, whether the run is a
    // control sample's; a lane's address is a scalar base + its lane number.  kRing rounds are in flight.
    if constexpr (kWide)
    {
      if (process)
      {
        constexpr int kRing = kSlots == kBigSlots ? KMD_TILE_RING_BIG : KMD_TILE_RING;
        constexpr int kR = KMD_TILE_RPL;                                              // records per lane and round (lane l: records l, 64 + l, ...)
        constexpr uint32_t kStep = 64u * kR;                                          // records per round
        constexpr uint32_t kBatch = kGroup;                                           // runs of a wave whose description its lanes hold at a time
        const uint32_t* beg = s_seg + (size_t)buf * 3 * S;
        const uint32_t* len = beg + S;
        const uint32_t* upto = beg + 2 * S;                                           // records of the group's runs up to and including this one (load_segments)
        // What the scalar unit does per round decides this kernel (PMC, round 2: 73 scalar instructions per round of
        // 64 records, most of them lane-mask algebra and the iterator's register shuffling -- one scalar unit serves
        // the CU's four SIMDs).  So:
        //  * the wave's runs are described in its LANES (lane j: first record and length of the wave's j-th run with
        //    records in this tile; read from the segment table once per tile, empty runs squeezed out): a round reads
        //    its run with two v_readlane, builds the buffer descriptors from them and steps (run, offset) with
        //    selects -- no branch, no LDS read, nothing carried round to round but two counters;
        //  * a round's records are read through buffer descriptors of the run: base = the run's first record, extent
        //    = the run, scalar offset = bytes done, a lane's offset = 8 x its number; the hardware's range check (on
        //    gfx950 it covers the scalar offset) returns 0 past the end of the run and reads nothing once the runs
        //    are exhausted (extent 0).  Which lanes hold a record is decided from the scalar count of records left,
        //    never from what was loaded;
        //  * a lane's state in the table is ONE number (insert_w).
        uint32_t cmax = 0;                                                            // largest count this lane met (kSum32: one look per tile)
        bool gave_up = false;
#if KMD_TILE_TIMING == 1   // dev: cycles a wave sits out per LDS round trip (bucket reads, swap), printed by one wave
        uint32_t tm_read = 0, tm_nread = 0, tm_cas = 0, tm_ncas = 0;
        const unsigned long long tm_t0 = __builtin_readcyclecounter();
#endif
        const uint32_t lane_k = lane * 8u, lane_c = lane * 4u;
        // The tile's records -- its S runs one behind the other -- are cut into kWaves equal shares, whatever the runs'
        // lengths: wave w takes records [w n / kWaves, (w + 1) n / kWaves) of that sequence, i.e. a few whole runs and
        // a piece of the run at either end.  (Whole runs dealt out in turn left the waves of a workgroup up to 15 %
        // apart at the tile's barrier -- samples differ in depth -- and a sample ten times deeper than the rest would
        // have cost its wave ten times the others' time.)
        // (the loads below are counted by hand: nothing else -- a row on its way to the list -- may be in flight)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t w_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
        const uint32_t share_lo = (uint32_t)(((uint64_t)n * w_u) / (uint32_t)kWaves), share_hi = (uint32_t)(((uint64_t)n * (w_u + 1u)) / (uint32_t)kWaves);
        uint32_t seq0 = 0;                                                            // records of the tile before stream `first`
        for (uint32_t first = 0, grp = 0; first < S && seq0 < share_hi && !gave_up; first += kBatch, ++grp)
        {
          // lane j < 63: the piece of run `first + j` that lies in this wave's share
          const uint32_t my_s = first + lane;
          const bool mine_ok = lane < kBatch && my_s < S;
          const uint32_t my_len = mine_ok ? len[my_s] : 0u;
          const uint32_t incl = mine_ok ? upto[my_s] : 0u;                            // (a lane without a run: an empty piece whatever this is)
          const uint32_t run_lo = seq0 + incl - my_len, run_hi = seq0 + incl;
          seq0 += (uint32_t)__builtin_amdgcn_readfirstlane((int)M.grp[buf][grp]);
          if (seq0 <= share_lo) continue;                                             // (none of these runs reaches the share)
          const uint32_t p_lo = run_lo > share_lo ? run_lo : share_lo, p_hi = run_hi < share_hi ? run_hi : share_hi;
          uint32_t v_rl = p_hi > p_lo ? p_hi - p_lo : 0u;                             // (< 2^29: bit 31 marks a control sample's run)
          uint32_t v_rb = mine_ok ? beg[my_s] + (p_lo - run_lo) : 0u;
          v_rl |= (v_rl != 0u && my_s < J.nc) ? 0x80000000u : 0u;
          {
            // runs with records first, in order (a forward permute: lane i sends to lane dst(i))
            const unsigned long long have = ballot(v_rl != 0u);
            const unsigned long long below = (1ull << lane) - 1ull;
            const uint32_t dst = v_rl != 0u ? (uint32_t)__popcll(have & below) : (uint32_t)__popcll(have) + (uint32_t)__popcll(~have & below);
            v_rb = (uint32_t)__builtin_amdgcn_ds_permute((int)(dst * 4u), (int)v_rb);
            v_rl = (uint32_t)__builtin_amdgcn_ds_permute((int)(dst * 4u), (int)v_rl);
          }
          uint32_t j = 0, off = 0;                                                    // next round: run j of the batch, records done in it
          uint32_t rl = 0, ctl_now = 0;                                               // the run's length, whether it is a control sample's
          u32x4 dk = { 0u, 0u, 0u, (uint32_t)kRsrcFlags }, dc = dk, dh = dk;          // the run's buffer descriptors
          uint64_t rk[kRing][kR], rkh[kTwo ? kRing : 1][kR];
          uint32_t rcnt[kRing][kR], rrem[kRing], rctl[kRing];                         // rrem / rctl: scalar
          [[maybe_unused]] uint32_t rlead[kRing], lead_now = 0;                        // KMD_TILE_ALIGN: lanes at the head of a run's first round that hold no record of it
          auto fetch_w = [&](const int d)
          {
            // The loads are issued and waited for BY HAND (vm_wait below): hipcc's own wait counts let a round's loads
            // stay in flight in the first pass of the ring only -- where the paths into a ring stage meet it settles for
            // the smallest count any of them allows, vmcnt(1) / (0) for stage 0, (3) / (2) for stage 1 ... -- so every
            // round sat out the latency of loads issued one or two rounds before, and four rounds in flight or sixteen
            // made no difference (round 2's observation).  Written as inline assembly the compiler does not see them.
            if (off == 0)
            {
              // a new run: its description out of the lanes, its descriptors (base = its first record, extent = the run)
              const uint32_t jj = j < kBatch ? j : kBatch;                            // (lane 63 never holds a run: length 0)
              uint32_t rb = (uint32_t)__builtin_amdgcn_readlane((int)v_rb, (int)jj);
              const uint32_t rle = (uint32_t)__builtin_amdgcn_readlane((int)v_rl, (int)jj);
              rl = rle & 0x7FFFFFFFu;
              ctl_now = rle >> 31;
#if KMD_TILE_ALIGN
              // Every round of a run reads 512 contiguous bytes of keys and 256 of counts.  Started wherever the run
              // (or this wave's piece of it) starts, each round's last line is also the next round's first -- fetched
              // twice when it has left the cache in between (PMC at configs[2] size: 1.10 x the algorithmic bytes; the
              // runs' two ends alone account for 1.03).  So the first round starts at the record index rounded down to
              // a multiple of 32 -- a 128-byte line of both arrays -- its first `lead` lanes switched off, and all
              // rounds behind it are line-aligned.
              lead_now = rl ? (rb & 31u) : 0u;
              rb -= lead_now; rl += lead_now;
#endif
              const uint64_t ak = (uint64_t)(uintptr_t)(J.keys + rb), ac = (uint64_t)(uintptr_t)(J.counts + rb);
              dk.x = (uint32_t)ak; dk.y = (uint32_t)(ak >> 32) & 0xFFFFu; dk.z = rl * 8u;
              dc.x = (uint32_t)ac; dc.y = (uint32_t)(ac >> 32) & 0xFFFFu; dc.z = rl * 4u;
              if constexpr (kTwo)
              {
                const uint64_t ah = (uint64_t)(uintptr_t)(J.keys_hi + rb);
                dh.x = (uint32_t)ah; dh.y = (uint32_t)(ah >> 32) & 0xFFFFu; dh.z = rl * 8u;
              }
            }
#pragma unroll
            for (int u = 0; u < kR; ++u)
            {
              asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" KMD_TILE_LOAD_HINT " ; ring" : "=v"(rk[d][u]) : "v"(lane_k), "s"(dk), "s"((off + 64u * (uint32_t)u) * 8u));
              asm volatile("buffer_load_dword %0, %1, %2, %3 offen" KMD_TILE_LOAD_HINT " ; ring" : "=v"(rcnt[d][u]) : "v"(lane_c), "s"(dc), "s"((off + 64u * (uint32_t)u) * 4u));
              if constexpr (kTwo) asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" KMD_TILE_LOAD_HINT " ; ring" : "=v"(rkh[d][u]) : "v"(lane_k), "s"(dh), "s"((off + 64u * (uint32_t)u) * 8u));
            }
            rrem[d] = rl - off;                                                       // (0 when the batch is exhausted: rl = 0, off = 0)
            rctl[d] = ctl_now;
#if KMD_TILE_ALIGN
            rlead[d] = lead_now; lead_now = 0;
#endif
            off += kStep;
            if (off >= rl) { off = 0; ++j; }
          };
          // (all slot numbers below are LDS byte addresses: of a key in M.key; >> 1 or as they are, plus a constant, of its sums)
          typedef __attribute__((address_space(3))) unsigned long long lds_u64;
          typedef __attribute__((address_space(3))) uint32_t lds_u32;
          const uint32_t key_lds = (uint32_t)(uintptr_t)(lds_u64*)M.key;             // (16-byte aligned: s_raw is)
          const uint32_t kNone = key_lds + kAll * 8u;
          constexpr uint32_t kHashMul = 0x9E3779B1u;
          constexpr int kBucketBits = ilog2_c(kSlots) - 1;                          // two slots per bucket
          auto key_at = [&](uint32_t at) -> unsigned long long* { return (unsigned long long*)(lds_u64*)(uintptr_t)at; };
          // where a k-mer's probe sequence is at position `pos` (0, 1: bucket 0; 2, 3: bucket 1; then the second table)
          auto seq_at = [&](uint64_t k, uint32_t pos) -> uint32_t
          {
            const uint32_t pr = ((uint32_t)k ^ (uint32_t)(k >> 29)) * kHashMul;       // ONE multiply: its top bits are bucket 0, the next
            const uint32_t c0 = key_lds + ((pr >> (32 - kBucketBits)) << 4);          // ones bucket 1 (two slots = 16 bytes), bits 4 .. the
            const uint32_t c1 = key_lds + (((pr >> (32 - 2 * kBucketBits)) & ((1u << kBucketBits) - 1u)) << 4);     // second table's slot
            return pos < 2u ? c0 + 8u * pos : pos < 4u ? c1 + 8u * (pos - 2u) : key_lds + (kSlots + (((pr >> 4) + pos - 4u) & (kSec - 1u))) * 8u;
          };
          // stage 3 for one record: the whole sequence, one slot per step; kNone (and the tile gives up) if it does not end
          auto walk_seq = [&](uint64_t k) -> uint32_t
          {
            for (uint32_t pos = 0; pos < 4u + kMaxProbe; ++pos)
            {
              const uint32_t at = seq_at(k, pos);
              unsigned long long seen = __hip_atomic_load(key_at(at), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              if (seen == kEmptyKey)
              {
                seen = atomicCAS(key_at(at), (unsigned long long)kEmptyKey, (unsigned long long)k);
                if (seen == kEmptyKey) seen = k;
              }
              if (seen == k) return at;
            }
            M.abort[buf] = 1;
            return kNone;
          };
          // the count of a record into its slot's sum (a lane without a place adds to the spare slot)
          auto add_count = [&](const int d, uint32_t slot, uint32_t c, uint64_t kh, bool is_real)
          {
            if constexpr (kSum32)
            {
              cmax = c > cmax ? c : cmax;                                             // looked at once per tile (below)
              // (scalar choice of the array; a key's address / 2 + a constant = its sum's)
              const uint32_t sums_lds = (uint32_t)(uintptr_t)(lds_u32*)(rctl[d] ? M.c32 : M.k32) - (key_lds >> 1);
              uint32_t* sums = (uint32_t*)(lds_u32*)(uintptr_t)(sums_lds + (slot >> 1));
#if KMD_TILE_ABLATE & 2   // dev: a plain store instead of the atomic add (results wrong)
              *sums = c;
#elif KMD_TILE_ABLATE & 64   // dev: no sums at all
#else
              atomicAdd(sums, c);
#endif
            }
            else
            {
              const uint32_t sums_lds = (uint32_t)(uintptr_t)(lds_u64*)(rctl[d] ? M.sc : M.sk) - key_lds;
              atomicAdd((unsigned long long*)(lds_u64*)(uintptr_t)(sums_lds + slot), (unsigned long long)c);
            }
            if constexpr (kTwo)
            {
              // (a lane without a record folds its zero into the spare slot's pair)
              const uint32_t at = (is_real ? slot : kNone) - key_lds;
              atomicMax(reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(M.key_hi) + at), (unsigned long long)kh);
              atomicMin(reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(M.hi_min) + at), (unsigned long long)kh);
            }
          };
          // ONE record per lane, all stages, as the compiler writes them: the rounds at the end of a run (kThere: some
          // lanes hold no record; kMarker: the all-ones k-mer may be among them), two-limb k-mers, 64-bit sums
          auto insert_rec = [&](const int d, const int u, auto there_tag, auto marker_tag)
          {
            constexpr bool kThere = decltype(there_tag)::value, kMarker = decltype(marker_tag)::value;
            const uint64_t k = rk[d][u];
            uint32_t cnt = rcnt[d][u];
            bool real = true;
            if constexpr (kThere) real = real & (lane + 64u * (uint32_t)u < rrem[d]);
#if KMD_TILE_ALIGN
            if constexpr (kThere) real = real & (lane + 64u * (uint32_t)u >= rlead[d]);
#endif
            if constexpr (kMarker)
            {
              if (real & (k == kEmptyKey))
              {
                // an all-ones (low) limb is the table's empty marker: such a k-mer has its own pair of sums
                atomicAdd(&M.maxsum[rctl[d] ? 0 : 1], (unsigned long long)cnt);
                if constexpr (kTwo) { atomicMin(&M.max_hi[0], (unsigned long long)rkh[d][u]); atomicMax(&M.max_hi[1], (unsigned long long)rkh[d][u]); }
                M.hasmax = 1;
              }
              real = real & (k != kEmptyKey);
            }
            // no record, or that one: nothing to add, nothing to claim (stage 1 may well "find" an empty slot for the
            // marker, or the k-mer 0 for a lane without a record: adding 0 there changes nothing)
            if constexpr (kThere || kMarker) cnt = real ? cnt : 0u;
            const uint32_t c0 = seq_at(k, 0), c1 = seq_at(k, 2);
            const u64x2 q0 = *(const lds_u64x2*)(uintptr_t)c0, q1 = *(const lds_u64x2*)(uintptr_t)c1;
            // stage 1 (a k-mer sits in one slot at most: the order of the four is free)
            uint32_t sl = kNone;
            sl = (q1.y == k) ? c1 + 8u : sl;
            sl = (q1.x == k) ? c1 : sl;
            sl = (q0.y == k) ? c0 + 8u : sl;
            sl = (q0.x == k) ? c0 : sl;
            // stage 2: the first empty one of the four, else the second table's slot
            if ((sl == kNone) & real)
            {
              uint32_t e = seq_at(k, 4);
              e = (q1.y == kEmptyKey) ? c1 + 8u : e;
              e = (q1.x == kEmptyKey) ? c1 : e;
              e = (q0.y == kEmptyKey) ? c0 + 8u : e;
              e = (q0.x == kEmptyKey) ? c0 : e;
#if KMD_TILE_ABLATE & 1   // dev: a plain store instead of the compare-and-swap (results wrong)
              *key_at(e) = k; sl = e;
#else
              const unsigned long long old = atomicCAS(key_at(e), (unsigned long long)kEmptyKey, (unsigned long long)k);
              // claimed, or another record of the k-mer was faster; else another k-mer took it meanwhile (stage 3)
              sl = ((old == kEmptyKey) | (old == k)) ? e : kNone;
#endif
            }
#if !(KMD_TILE_ABLATE & 128)   // dev: no stage 3 (results wrong)
            if (ballot((sl == kNone) & real)) { if ((sl == kNone) & real) sl = walk_seq(k); }
#endif
            add_count(d, sl, cnt, kTwo ? rkh[d][u] : 0ull, real);
          };
          // One round into the table.  A k-mer's probe sequence: the two slots of its home bucket 0, the two of its
          // home bucket 1, then a slot of the small second table behind the first and on from there; slots are never
          // released within a tile, so a k-mer is never behind an empty slot and the first slot of the sequence that
          // holds it OR is empty is its place.  Three stages, cheapest first:
          //   1. both buckets are read (two 16-byte LDS reads per record, four candidates).  A row has rho records
          //      and only the first claims a slot: for most records one of the four IS the k-mer -- four compares,
          //      four selects, done;
          //   2. the lanes left (a few per round: first records of their rows, and the records of the ~1 % of the k-mers
          //      that found their four home slots taken) swap on the first empty candidate -- or, if none is empty, on
          //      their slot of the second table, which those few k-mers barely fill: the swap claims the slot or finds
          //      the k-mer there;
          //   3. what is left -- a claim lost to another k-mer, two k-mers on one slot of the second table -- walks the
          //      whole sequence slot by slot (once in a hundred rounds).  (Round 2 sent every record of a k-mer beyond
          //      its home buckets down such a walk in the main table: three to four steps, in every second round -- a
          //      quarter of the kernel's instructions.)
          // A lane's state is ONE number, the LDS address of its slot's key (kNone: not placed yet).  kNone is the
          // address of a spare slot behind the tables: a lane that has nothing to add (no record, a tile that gave up)
          // adds its count there -- no lane mask around the adds, no masks kept across the stages.
          // In the middle of a run every lane holds a record and none is the all-ones k-mer (a run ascends: only its
          // last record can be): stages 1 and 2 of such a round are written by hand below -- the compiler's version of
          // them spent a third of its instructions on lane-mask algebra and wait states -- for one or two records per lane.
          auto insert_mid = [&](const int d)
          {
#if KMD_TILE_ABLATE & 16   // dev: the loads alone (results wrong)
#pragma unroll
            for (int u = 0; u < kR; ++u) if ((rk[d][u] + rcnt[d][u]) == 0x123456789ull) M.hasmax = 1;
            return;
#endif
#if KMD_TILE_ASM && !KMD_TILE_ABLATE
            if constexpr (!kTwo && kSum32 && kR == 2)
            {
              // two records per lane (A: v[40:47], B: v[48:55]): four bucket reads behind one wait, both swaps behind
              // one wait -- half the LDS round trips a wave sits out per record, half the scalar work
              const uint64_t kA = rk[d][0], kB = rk[d][1];
              uint32_t a0, a1, b0, b1, t0, t1, u0, u1, sea, seb, sla, slb;
              uint64_t m0, m1, m2, nA, nB, sv;
              const uint64_t empty = kEmptyKey;
              asm volatile(
                  "v_alignbit_b32 %[a0], %[kAhi], %[kAlo], 29\n\t"
                  "v_alignbit_b32 %[b0], %[kBhi], %[kBlo], 29\n\t"
                  "v_xor_b32 %[a0], %[a0], %[kAlo]\n\t"
                  "v_xor_b32 %[b0], %[b0], %[kBlo]\n\t"
                  "v_mul_lo_u32 %[sea], %[a0], %[c1]\n\t"
                  "v_mul_lo_u32 %[seb], %[b0], %[c1]\n\t"
                  "v_mov_b32 %[sla], %[none]\n\t"
                  "v_mov_b32 %[slb], %[none]\n\t"
                  "v_lshrrev_b32 %[a0], %[sh0], %[sea]\n\t"
                  "v_bfe_u32 %[a1], %[sea], %[sh1], %[nb]\n\t"
                  "v_lshrrev_b32 %[b0], %[sh0], %[seb]\n\t"
                  "v_bfe_u32 %[b1], %[seb], %[sh1], %[nb]\n\t"
                  "v_lshl_add_u32 %[a0], %[a0], 4, %[kb]\n\t"
                  "v_lshl_add_u32 %[a1], %[a1], 4, %[kb]\n\t"
                  "v_lshl_add_u32 %[b0], %[b0], 4, %[kb]\n\t"
                  "v_lshl_add_u32 %[b1], %[b1], 4, %[kb]\n\t"
                  "ds_read_b128 v[40:43], %[a0]\n\t"
                  "ds_read_b128 v[44:47], %[a1]\n\t"
                  "ds_read_b128 v[48:51], %[b0]\n\t"
                  "ds_read_b128 v[52:55], %[b1]\n\t"
                  "v_bfe_u32 %[sea], %[sea], 4, %[nsec]\n\t"
                  "v_bfe_u32 %[seb], %[seb], 4, %[nsec]\n\t"
                  "v_add_u32 %[t0], 8, %[a0]\n\t"
                  "v_add_u32 %[t1], 8, %[a1]\n\t"
                  "v_add_u32 %[u0], 8, %[b0]\n\t"
                  "v_add_u32 %[u1], 8, %[b1]\n\t"
                  "v_lshl_add_u32 %[sea], %[sea], 3, %[sb]\n\t"
                  "v_lshl_add_u32 %[seb], %[seb], 3, %[sb]\n\t"
                  "s_waitcnt lgkmcnt(0)\n\t"
                  "v_cmp_eq_u64 %[m0], v[46:47], %[kA]\n\t"
                  "v_cmp_eq_u64 %[m1], v[44:45], %[kA]\n\t"
                  "v_cmp_eq_u64 %[m2], v[42:43], %[kA]\n\t"
                  "v_cmp_eq_u64 vcc, v[40:41], %[kA]\n\t"
                  "v_cndmask_b32 %[sla], %[sla], %[t1], %[m0]\n\t"
                  "v_cndmask_b32 %[sla], %[sla], %[a1], %[m1]\n\t"
                  "v_cndmask_b32 %[sla], %[sla], %[t0], %[m2]\n\t"
                  "v_cndmask_b32 %[sla], %[sla], %[a0], vcc\n\t"
                  "v_cmp_eq_u64 %[m0], v[54:55], %[kB]\n\t"
                  "v_cmp_eq_u64 %[m1], v[52:53], %[kB]\n\t"
                  "v_cmp_eq_u64 %[m2], v[50:51], %[kB]\n\t"
                  "v_cmp_eq_u64 vcc, v[48:49], %[kB]\n\t"
                  "v_cndmask_b32 %[slb], %[slb], %[u1], %[m0]\n\t"
                  "v_cndmask_b32 %[slb], %[slb], %[b1], %[m1]\n\t"
                  "v_cndmask_b32 %[slb], %[slb], %[u0], %[m2]\n\t"
                  "v_cndmask_b32 %[slb], %[slb], %[b0], vcc\n\t"
                  "v_cmp_eq_u32 %[nA], %[sla], %[none]\n\t"
                  "v_cmp_eq_u32 %[nB], %[slb], %[none]\n\t"
                  "s_or_b64 vcc, %[nA], %[nB]\n\t"
                  "s_and_saveexec_b64 %[sv], vcc\n\t"
                  "s_cbranch_execz 1f\n\t"
                  "v_cmp_eq_u64 %[m0], -1, v[46:47]\n\t"
                  "v_cmp_eq_u64 %[m1], -1, v[44:45]\n\t"
                  "v_cmp_eq_u64 %[m2], -1, v[42:43]\n\t"
                  "v_cmp_eq_u64 vcc, -1, v[40:41]\n\t"
                  "v_cndmask_b32 %[sea], %[sea], %[t1], %[m0]\n\t"
                  "v_cndmask_b32 %[sea], %[sea], %[a1], %[m1]\n\t"
                  "v_cndmask_b32 %[sea], %[sea], %[t0], %[m2]\n\t"
                  "v_cndmask_b32 %[sea], %[sea], %[a0], vcc\n\t"
                  "v_cmp_eq_u64 %[m0], -1, v[54:55]\n\t"
                  "v_cmp_eq_u64 %[m1], -1, v[52:53]\n\t"
                  "v_cmp_eq_u64 %[m2], -1, v[50:51]\n\t"
                  "v_cmp_eq_u64 vcc, -1, v[48:49]\n\t"
                  "v_cndmask_b32 %[seb], %[seb], %[u1], %[m0]\n\t"
                  "v_cndmask_b32 %[seb], %[seb], %[b1], %[m1]\n\t"
                  "v_cndmask_b32 %[seb], %[seb], %[u0], %[m2]\n\t"
                  "v_cndmask_b32 %[seb], %[seb], %[b0], vcc\n\t"
                  "v_cndmask_b32 %[sea], %[none], %[sea], %[nA]\n\t"            // (a record placed already swaps on the spare slot)
                  "v_cndmask_b32 %[seb], %[none], %[seb], %[nB]\n\t"
                  "ds_cmpst_rtn_b64 v[40:41], %[sea], %[empty], %[kA]\n\t"      // (the buckets are done with: their registers take what the swaps return)
                  "ds_cmpst_rtn_b64 v[48:49], %[seb], %[empty], %[kB]\n\t"
                  "s_waitcnt lgkmcnt(0)\n\t"
                  "v_cmp_eq_u64 %[m0], -1, v[40:41]\n\t"
                  "v_cmp_eq_u64 %[m1], v[40:41], %[kA]\n\t"
                  "v_cmp_eq_u64 %[m2], -1, v[48:49]\n\t"
                  "v_cmp_eq_u64 vcc, v[48:49], %[kB]\n\t"
                  "s_or_b64 %[m0], %[m0], %[m1]\n\t"
                  "s_or_b64 vcc, vcc, %[m2]\n\t"
                  "v_cndmask_b32 %[t0], %[none], %[sea], %[m0]\n\t"
                  "v_cndmask_b32 %[u0], %[none], %[seb], vcc\n\t"
                  "v_cndmask_b32 %[sla], %[sla], %[t0], %[nA]\n\t"
                  "v_cndmask_b32 %[slb], %[slb], %[u0], %[nB]\n\t"
                  "1:\n\t"
                  "s_or_b64 exec, exec, %[sv]"
                  : [sla] "=&v"(sla), [slb] "=&v"(slb), [a0] "=&v"(a0), [a1] "=&v"(a1), [b0] "=&v"(b0), [b1] "=&v"(b1), [t0] "=&v"(t0), [t1] "=&v"(t1),
                    [u0] "=&v"(u0), [u1] "=&v"(u1), [sea] "=&v"(sea), [seb] "=&v"(seb),
                    [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [nA] "=&s"(nA), [nB] "=&s"(nB), [sv] "=&s"(sv)
                  : [kA] "v"(kA), [kAlo] "v"((uint32_t)kA), [kAhi] "v"((uint32_t)(kA >> 32)), [kB] "v"(kB), [kBlo] "v"((uint32_t)kB), [kBhi] "v"((uint32_t)(kB >> 32)),
                    [none] "v"(kNone), [empty] "v"(empty), [kb] "s"(key_lds), [sb] "s"(key_lds + kSlots * 8u), [c1] "s"(kHashMul),
                    [sh0] "n"(32 - kBucketBits), [sh1] "n"(32 - 2 * kBucketBits), [nb] "n"(kBucketBits), [nsec] "n"(ilog2_c(kSec))
                  : "vcc", "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
              if (ballot((sla == kNone) | (slb == kNone)))
              {
                if (sla == kNone) sla = walk_seq(kA);
                if (slb == kNone) slb = walk_seq(kB);
              }
              add_count(d, sla, rcnt[d][0], 0ull, true);
              add_count(d, slb, rcnt[d][1], 0ull, true);
              return;
            }
            else if constexpr (!kTwo && kSum32 && kR == 1)
            {
              // 8 vector instructions of hashing, two 16-byte LDS reads, four compares + four selects (stage 1); for
              // the lanes left four compares + four selects, the swap, three more (stage 2) -- no wait states (a
              // compare's mask is used three instructions later at the earliest), three scalar instructions in all.
              // v[40:47] hold the two buckets.
              const uint64_t k = rk[d][0];
              const uint32_t cnt = rcnt[d][0];
              // (scalar choice of the array; a key's address / 2 + a constant = its sum's)
              const uint32_t sums_lds = (uint32_t)(uintptr_t)(lds_u32*)(rctl[d] ? M.c32 : M.k32) - (key_lds >> 1);
              uint32_t a0, a1, t0, t1, se, sl;
              uint64_t m0, m1, m2, sv;
              const uint64_t empty = kEmptyKey;
              asm volatile(
                  "v_alignbit_b32 %[a0], %[khi], %[klo], 29\n\t"
                  "v_xor_b32 %[a0], %[a0], %[klo]\n\t"
                  "v_mul_lo_u32 %[se], %[a0], %[c1]\n\t"                        // p
                  "v_mov_b32 %[sl], %[none]\n\t"
                  "v_lshrrev_b32 %[a0], %[sh0], %[se]\n\t"                      // bucket 0: the top bits of p
                  "v_bfe_u32 %[a1], %[se], %[sh1], %[nb]\n\t"                   // bucket 1: the bits below them
                  "v_lshl_add_u32 %[a0], %[a0], 4, %[kb]\n\t"
                  "v_lshl_add_u32 %[a1], %[a1], 4, %[kb]\n\t"
#if KMD_TILE_TIMING == 1
                  "s_memtime s[90:91]\n\t"
                  "s_waitcnt lgkmcnt(0)\n\t"
#endif
                  "ds_read_b128 v[40:43], %[a0]\n\t"
                  "ds_read_b128 v[44:47], %[a1]\n\t"
                  "v_bfe_u32 %[se], %[se], 4, %[nsec]\n\t"                      // the second table's slot
                  "v_add_u32 %[t0], 8, %[a0]\n\t"
                  "v_add_u32 %[t1], 8, %[a1]\n\t"
                  "v_lshl_add_u32 %[se], %[se], 3, %[sb]\n\t"
                  "s_waitcnt lgkmcnt(0)\n\t"
#if KMD_TILE_TIMING == 1
                  "s_memtime s[92:93]\n\t"
                  "s_waitcnt lgkmcnt(0)\n\t"
                  "s_sub_u32 s92, s92, s90\n\t"
                  "s_add_u32 %[tr], %[tr], s92\n\t"
                  "s_add_u32 %[nr], %[nr], 1\n\t"
#endif
                  "v_cmp_eq_u64 %[m0], v[46:47], %[k]\n\t"
                  "v_cmp_eq_u64 %[m1], v[44:45], %[k]\n\t"
                  "v_cmp_eq_u64 %[m2], v[42:43], %[k]\n\t"
                  "v_cmp_eq_u64 vcc, v[40:41], %[k]\n\t"
                  "v_cndmask_b32 %[sl], %[sl], %[t1], %[m0]\n\t"
                  "v_cndmask_b32 %[sl], %[sl], %[a1], %[m1]\n\t"
                  "v_cndmask_b32 %[sl], %[sl], %[t0], %[m2]\n\t"
                  "v_cndmask_b32 %[sl], %[sl], %[a0], vcc\n\t"
                  "v_cmp_eq_u32 vcc, %[sl], %[none]\n\t"
                  "s_and_saveexec_b64 %[sv], vcc\n\t"
                  "s_cbranch_execz 1f\n\t"
                  "v_cmp_eq_u64 %[m0], -1, v[46:47]\n\t"
                  "v_cmp_eq_u64 %[m1], -1, v[44:45]\n\t"
                  "v_cmp_eq_u64 %[m2], -1, v[42:43]\n\t"
                  "v_cmp_eq_u64 vcc, -1, v[40:41]\n\t"
                  "v_cndmask_b32 %[se], %[se], %[t1], %[m0]\n\t"
                  "v_cndmask_b32 %[se], %[se], %[a1], %[m1]\n\t"
                  "v_cndmask_b32 %[se], %[se], %[t0], %[m2]\n\t"
                  "v_cndmask_b32 %[se], %[se], %[a0], vcc\n\t"
#if KMD_TILE_TIMING == 1
                  "s_memtime s[90:91]\n\t"
                  "s_waitcnt lgkmcnt(0)\n\t"
#endif
                  "ds_cmpst_rtn_b64 v[40:41], %[se], %[empty], %[k]\n\t"
                  "s_waitcnt lgkmcnt(0)\n\t"
#if KMD_TILE_TIMING == 1
                  "s_memtime s[92:93]\n\t"
                  "s_waitcnt lgkmcnt(0)\n\t"
                  "s_sub_u32 s92, s92, s90\n\t"
                  "s_add_u32 %[tc], %[tc], s92\n\t"
                  "s_add_u32 %[nc_], %[nc_], 1\n\t"
#endif
                  "v_cmp_eq_u64 %[m0], -1, v[40:41]\n\t"
                  "v_cmp_eq_u64 vcc, v[40:41], %[k]\n\t"
                  "s_or_b64 vcc, vcc, %[m0]\n\t"
                  "v_cndmask_b32 %[sl], %[none], %[se], vcc\n\t"
                  "1:\n\t"
                  "s_or_b64 exec, exec, %[sv]\n\t"
                  // the count goes to the slot's sum (a lane without a slot yet: to the spare slot's, which nobody reads)
                  // and the mask of those lanes leaves in m0 -- written here, as the compiler wrote the test and the
                  // add behind the block they were ten and nine instructions, most of them scalar
                  "v_cmp_eq_u32 %[m0], %[sl], %[none]\n\t"
                  "v_lshrrev_b32 %[t0], 1, %[sl]\n\t"
                  "v_add_u32 %[t0], %[sbase], %[t0]\n\t"
                  "ds_add_u32 %[t0], %[cnt]"
                  : [sl] "=&v"(sl), [a0] "=&v"(a0), [a1] "=&v"(a1), [se] "=&v"(se), [t0] "=&v"(t0), [t1] "=&v"(t1),
                    [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [sv] "=&s"(sv)
#if KMD_TILE_TIMING == 1
                    , [tr] "+s"(tm_read), [nr] "+s"(tm_nread), [tc] "+s"(tm_cas), [nc_] "+s"(tm_ncas)
#endif
                  : [k] "v"(k), [klo] "v"((uint32_t)k), [khi] "v"((uint32_t)(k >> 32)), [none] "v"(kNone), [empty] "v"(empty), [cnt] "v"(cnt), [sbase] "s"(sums_lds),
                    [kb] "s"(key_lds), [sb] "s"(key_lds + kSlots * 8u), [c1] "s"(kHashMul), [sh0] "n"(32 - kBucketBits), [sh1] "n"(32 - 2 * kBucketBits),
                    [nb] "n"(kBucketBits), [nsec] "n"(ilog2_c(kSec))
                  : "vcc", "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47"
#if KMD_TILE_TIMING == 1
                    , "s90", "s91", "s92", "s93"
#endif
                  );
              cmax = cnt > cmax ? cnt : cmax;                                        // looked at once per tile (below)
              if (m0) { if (sl == kNone) add_count(d, walk_seq(k), cnt, 0ull, true); }   // (stage 3: a handful of rounds per tile)
              return;
            }
#endif
            // (two-limb k-mers: the table's key is the LOW limb, which does not ascend along a run -- the marker is looked for)
#pragma unroll
            for (int u = 0; u < kR; ++u)
              if constexpr (kTwo) insert_rec(d, u, std::false_type(), std::true_type()); else insert_rec(d, u, std::false_type(), std::false_type());
          };
          // the oldest round in flight has arrived (the kRing - 1 younger ones may still be under way); the round's
          // registers pass through the statement, so nothing that reads them can be moved above it
          auto vm_wait = [&](const int d)
          {
            constexpr int kLoads = kR * (kTwo ? 3 : 2);                               // load instructions per round
#pragma unroll
            for (int u = 0; u < kR; ++u)
            {
              if constexpr (kTwo) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(rk[d][u]), "+v"(rcnt[d][u]), "+v"(rkh[d][u]) : "n"(kLoads * (kRing - 1)));
              else asm volatile("s_waitcnt vmcnt(%2)" : "+v"(rk[d][u]), "+v"(rcnt[d][u]) : "n"(kLoads * (kRing - 1)));
            }
          };
#pragma unroll
          for (int d = 0; d < kRing; ++d) fetch_w(d);
          for (bool more = true; more;)
          {
#pragma unroll
            for (int d = 0; d < kRing; ++d)
            {
              if (!more) break;
              if (rrem[d] == 0) { more = false; break; }                              // the batch's runs are exhausted in order
              vm_wait(d);
              // A run's LAST round may hold the all-ones k-mer (also when it is a full one) and lanes without a record.
              // Without that k-mer -- almost always -- it goes the fast way too, with its empty lanes switched off
              // around it (one record per lane: the lanes that hold one are the first `rem`).
#if KMD_TILE_ALIGN
              const bool whole_round = rrem[d] > kStep && rlead[d] == 0;
#else
              const bool whole_round = rrem[d] > kStep;
#endif
              if (whole_round) insert_mid(d);
              else
              {
                bool fast = false;
                if constexpr (kR == 1 && !kTwo && kSum32 && KMD_TILE_ASM && !KMD_TILE_ABLATE && KMD_TILE_TIMING != 1)
                {
#if KMD_TILE_ALIGN
                  const bool there = (lane < rrem[d]) & (lane >= rlead[d]);
#else
                  const bool there = lane < rrem[d];
#endif
                  if (!ballot(there & (rk[d][0] == kEmptyKey)))
                  {
                    fast = true;
                    if (there) insert_mid(d);
                  }
                }
                if (!fast)
                {
#pragma unroll
                  for (int u = 0; u < kR; ++u) insert_rec(d, u, std::true_type(), std::true_type());
                }
              }
              fetch_w(d);
              if ((d & (KMD_TILE_ABORT_EVERY - 1)) == KMD_TILE_ABORT_EVERY - 1 && M.abort[buf]) { more = false; gave_up = true; break; }   // (LDS read, same for the whole wave)
            }
          }
          // (the loads still in flight -- of nothing: the batch is exhausted, or of a tile that gave up -- must have
          // landed before their registers serve anything else)
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#if KMD_TILE_TIMING == 1
        if (blockIdx.x == 77 && tid == 64 && tm_nread)
          printf("[tile timing] tile %u: %u rounds, bucket reads %.0f cycles each, %u swaps %.0f cycles each, whole insert loop %llu cycles (%.0f per round)\n",
                 tile, tm_nread, (double)tm_read / tm_nread, tm_ncas, tm_ncas ? (double)tm_cas / tm_ncas : 0.0,
                 __builtin_readcyclecounter() - tm_t0, (double)(__builtin_readcyclecounter() - tm_t0) / tm_nread);
#endif
        // a count too large for 32-bit sums: the tile is redone with 64-bit ones (never, in practice)
        if constexpr (kSum32) if (ballot(cmax >= kBigCount) && lane == 0) M.big[buf] = 1;
      }
    }
    // ---- inserts, general: sub-group q0 (G lanes) streams the runs q0, q0 + Q, ... of this tile
    if (!kWide && process)
    {
      const uint32_t* beg = s_seg + (size_t)buf * 3 * S;
      const uint32_t* len = beg + S;
      uint32_t s = q0, pos = 0, end = 0;
      // the lane's next record: `pos` in run `s`; a lane that has none left parks on record 0 (s >= S)
      auto seek = [&]()
      {
        while (s < S)
        {
          const uint32_t b = beg[s];
          pos = b + sub; end = b + len[s];
          if (pos < end) return;
          s += Q;
        }
        pos = 0; end = 0;
      };
      // the loads are unconditional (a lane without a record left reads record 0, its `valid` bit stays
      // clear): a round is a fixed number of load instructions, so the ones of later rounds can stay in
      // flight while an earlier round is inserted (the wait counts are static)
      auto fetch = [&](batch& B)
      {
        B.valid = 0; B.ctl = 0;
#pragma unroll
        for (int u = 0; u < kU; ++u)
        {
          B.k[u] = __builtin_nontemporal_load(J.keys + pos);
          B.c[u] = __builtin_nontemporal_load(J.counts + pos);
          if constexpr (kTwo) B.kh[u] = __builtin_nontemporal_load(J.keys_hi + pos);
          if (s < S)
          {
            B.valid |= 1u << u;
            if (s < J.nc) B.ctl |= 1u << u;
            pos += G;
            if (pos >= end) { s += Q; seek(); }
          }
        }
      };
      // One round into the table; false: the tile gave up (table too full).
      // What a wave pays for is the LONGEST probe sequence among its 64 x kU records, so the sequences
      // must be short for all of them.  A k-mer has two home slots (two hashes); the sequence is
      // home 0, home 1, home 1 + 1, home 1 + 2, ...  The first step looks at BOTH homes at once (plain
      // 64-bit LDS reads -- most records find their k-mer already there: a row has rho records, only the
      // first one claims a slot, with a compare-and-swap): at 1/3 load that settles ~97 % of the records;
      // the few left walk on in a tail loop.  Everything is wave-uniform with per-lane state moved by
      // selects: a divergent probe loop per record (a returning CAS per step) cost twice the instructions
      // in exec-mask bookkeeping.  Slots are never released within a tile, so whichever record of a k-mer
      // comes first, later ones walk the same sequence to the same slot.
      auto insert = [&](const batch& A) -> bool
      {
        uint32_t slot[kU], nxt[kU];                      // where the k-mer is / next slot to look at
        bool pend[kU];
        uint32_t fresh_bits = 0;
        bool special = false;
        unsigned long long s0[kU], s1[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u)
        {
          const bool v = (A.valid >> u) & 1u;
          const uint32_t x = (uint32_t)A.k[u] ^ (uint32_t)(A.k[u] >> 29);
          slot[u] = (x * 0x9E3779B1u) >> kShift;
          nxt[u] = ((x ^ (x >> 15)) * 0x85EBCA6Bu) >> kShift;
          pend[u] = v && A.k[u] != kEmptyKey;
          special |= v && A.k[u] == kEmptyKey;
          s0[u] = __hip_atomic_load(&M.key[slot[u]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          s1[u] = __hip_atomic_load(&M.key[nxt[u]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (ballot(special))
        {
          // an all-ones (low) limb is the table's empty marker: such a k-mer has its own pair of sums
#pragma unroll
          for (int u = 0; u < kU; ++u)
            if (((A.valid >> u) & 1u) && A.k[u] == kEmptyKey)
            {
              atomicAdd(&M.maxsum[((A.ctl >> u) & 1u) ? 0 : 1], (unsigned long long)A.c[u]);
              if constexpr (kTwo) { atomicMin(&M.max_hi[0], (unsigned long long)A.kh[u]); atomicMax(&M.max_hi[1], (unsigned long long)A.kh[u]); }
              M.hasmax = 1;
            }
        }
        // step 1: both homes
        bool want_cas = false;
        bool claim[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u)
        {
          const bool hit0 = s0[u] == A.k[u], hit1 = s1[u] == A.k[u];
          const bool e0 = s0[u] == kEmptyKey, e1 = s1[u] == kEmptyKey;
          const uint32_t home0 = slot[u], home1 = nxt[u];
          const bool found = hit0 || hit1;
          claim[u] = pend[u] && !found && (e0 || e1);        // the first empty home (home 1 only if home 0 is taken)
          // found: there.  claiming: the slot to claim.  neither: on to home 1 + 1
          slot[u] = (hit0 || (!hit1 && e0)) ? home0 : home1;
          nxt[u] = (claim[u] && e0) ? home1 : ((home1 + 1u) & kMask);
          pend[u] = pend[u] && !found;
          want_cas |= claim[u];
        }
#if !(KMD_TILE_ABLATE & 1)
        if (ballot(want_cas))
        {
#pragma unroll
          for (int u = 0; u < kU; ++u)
            if (claim[u])
            {
              const unsigned long long old = atomicCAS(&M.key[slot[u]], (unsigned long long)kEmptyKey, (unsigned long long)A.k[u]);
              if (old == kEmptyKey) { fresh_bits |= 1u << u; pend[u] = false; }
              else if (old == A.k[u]) pend[u] = false;       // (another record of this k-mer was faster)
            }                                                // else: lost the slot to another k-mer, walk on at nxt
        }
#else
#pragma unroll
        for (int u = 0; u < kU; ++u) if (claim[u]) { M.key[slot[u]] = A.k[u]; pend[u] = false; }
#endif
        // the tail: one slot per step from nxt on
        bool gave_up = false;
        for (uint32_t step = 0;; ++step)
        {
          bool any = false;
#pragma unroll
          for (int u = 0; u < kU; ++u) any |= pend[u];
          if (!ballot(any)) break;
          if (step >= kMaxProbe) { gave_up = any; break; }
          unsigned long long seen[kU];
#pragma unroll
          for (int u = 0; u < kU; ++u) seen[u] = __hip_atomic_load(&M.key[nxt[u]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          bool want = false;
#pragma unroll
          for (int u = 0; u < kU; ++u) want |= pend[u] && seen[u] == kEmptyKey;
          if (ballot(want))
          {
#pragma unroll
            for (int u = 0; u < kU; ++u)
              if (pend[u] && seen[u] == kEmptyKey)
              {
                seen[u] = atomicCAS(&M.key[nxt[u]], (unsigned long long)kEmptyKey, (unsigned long long)A.k[u]);
                if (seen[u] == kEmptyKey) { seen[u] = A.k[u]; fresh_bits |= 1u << u; }
              }
          }
#pragma unroll
          for (int u = 0; u < kU; ++u)
          {
            const bool hit = pend[u] && seen[u] == A.k[u];
            slot[u] = hit ? nxt[u] : slot[u];
            pend[u] = pend[u] && !hit;
            nxt[u] = (nxt[u] + 1u) & kMask;
          }
        }
        uint32_t claimed = 0;
#pragma unroll
        for (int u = 0; u < kU; ++u)
        {
          const bool v = (A.valid >> u) & 1u;
          if (v && A.k[u] != kEmptyKey && !pend[u])
          {
            const bool ctl = (A.ctl >> u) & 1u;
            if constexpr (kSum32)
            {
              if (A.c[u] >= kBigCount) M.big[buf] = 1;
              atomicAdd(&(ctl ? M.c32 : M.k32)[slot[u]], A.c[u]);
            }
            else atomicAdd(ctl ? &M.sc[slot[u]] : &M.sk[slot[u]], (unsigned long long)A.c[u]);
            if constexpr (kTwo)
            {
              atomicMax(&M.key_hi[slot[u]], (unsigned long long)A.kh[u]);
              atomicMin(&M.hi_min[slot[u]], (unsigned long long)A.kh[u]);
            }
          }
          claimed += (uint32_t)__popcll(ballot(((fresh_bits >> u) & 1u) != 0));
        }
        // slots this wave claimed in the round -> the tile's count; a table 3/4 full gives up
        if (lane == 0 && claimed)
        {
          const uint32_t before = atomicAdd(&M.fresh[buf], claimed);
          if (before + claimed > kFullAt) M.abort[buf] = 1;
        }
        if (gave_up) M.abort[buf] = 1;
        return M.abort[buf] == 0;                        // (LDS read, same for the whole wave)
      };
      // a ring of kDepth rounds: round i + kDepth is requested as soon as round i has been inserted
      auto run_ring = [&](auto&& fetch)
      {
        batch R[kDepth];
#pragma unroll
        for (int d = 0; d < kDepth; ++d) fetch(R[d]);
        for (bool more = true; more;)
        {
#pragma unroll
          for (int d = 0; d < kDepth; ++d)
          {
            if (!more) break;
            if (!ballot(R[d].valid != 0)) { more = false; break; }     // the runs are exhausted in order
            if (!insert(R[d])) { more = false; break; }
            fetch(R[d]);
          }
        }
      };
      if constexpr (!kWide)
      {
        seek();
        run_ring(fetch);
      }
    }
#if KMD_TILE_TIMING
    const unsigned long long tp_ins = __builtin_readcyclecounter();
#endif
    // (candidates mode: the rows the previous tile parked go to the list now -- their stores land while this tile's
    // table is walked -- and the barriers wait for LDS only)
    if constexpr (kFilter) { flush_stage(); lds_barrier(); } else __syncthreads();
#if KMD_TILE_TIMING
    const unsigned long long tp_bar1 = __builtin_readcyclecounter();
#endif

    // ---- the next tile's segment table; the table walk: this tile's rows
    load_segments(tile + stride, buf ^ 1u);
#if KMD_TILE_TIMING
    const unsigned long long tp_seg = __builtin_readcyclecounter();
    unsigned long long tp_walk1 = 0, tp_resv = 0;
#endif
    if (process)
    {
      const bool big = kSum32 && M.big[buf] != 0;
      const bool aborted = M.abort[buf] != 0 || big;
      if constexpr (kTwo)
      {
        // two k-mers in one slot?  then this tile is cut again instead of emitted
        bool bad = false;
        if (!aborted)
          for (uint32_t i = tid; i < kAll; i += kThreads)
            bad |= M.key[i] != kEmptyKey && M.key_hi[i] != M.hi_min[i];
        if (tid == 0 && !aborted && M.hasmax && M.max_hi[0] != M.max_hi[1]) bad = true;
        if (ballot(bad) && lane == 0) M.bad = 1;
        __syncthreads();
      }
      // (wave-uniform, and said so: what comes out of LDS is a vector value to the compiler, and a loop or a counter that
      // depends on one is kept in vector registers under lane masks)
      const bool bad_tile = __builtin_amdgcn_readfirstlane((int)(aborted || (kTwo && M.bad != 0))) != 0;
      if constexpr (kTwo) { __syncthreads(); if (tid == 0) M.bad = 0; }     // everyone has read it
      if (bad_tile && tid == 0)
      {
        job_c* const Jo = job();
        const uint32_t at = atomicAdd(Jo->over_n, 1u);
        Jo->over[at] = tile; Jo->over[Jo->over_stride + at] = (n < kBigBit ? n : kBigBit - 1u) | kAbortBit | (big ? kBigBit : 0u);
      }
      if constexpr (kFilter)
      {
        // The walk, candidates mode.  What a tile costs here does not depend on how many records it had -- a 4096-slot
        // table holds ~2000 rows whether they were 5 000 records or 50 000 -- so on rows of few records this half of
        // the kernel is as long as the inserts.  Round 5's walk gave every thread five slots and ran the pre-filter,
        // in double precision, on all of them, dead or alive (half the lanes idle), its second stage and the emission
        // in a loop that ran two or three times a tile with a third of the lanes busy: 382 vector + 222 scalar
        // instructions per wave and tile.  Now:
        //  * COMPACTION FIRST.  A thread reads the keys of two adjacent slots (one ds_read_b128: the whole table in two
        //    steps of the workgroup, the second table by its last waves); a ballot and a prefix count put the LIVE
        //    slots into the wave's queue in LDS (kQueue bytes) -- five vector instructions per 64 slots;
        //  * whenever 64 slots are queued (and once more at the end, for the rest) the wave evaluates them, every lane a
        //    live row: both sums (gathered), the chi-square bound; the second stage and the emission only if a lane
        //    needs them; the key is read only for the rows that leave;
        //  * only live slots are wiped (an empty slot is empty: its sums are 0 -- nothing adds to a slot it did not
        //    find its key in, but for the spare slot behind the tables), in the pass that evaluates them;
        //  * rows and rows beyond the table are counted per WAVE, in scalar registers, from the ballots;
        //  * the rows that leave are parked in LDS with one atomic per wave and pass (kStage of them; the workgroup's
        //    chunk of the list takes what does not fit, at once, and should THAT run out the wave takes a chunk of its
        //    own -- no second walk, no barriers for the rare case) and written to the list by wave 0 when the NEXT
        //    tile's inserts are done: nothing waits for the stores -- the barriers of the tile loop wait for LDS only.
        //  * the pre-filter's constants come from the kernel arguments here, where they are used (job()).
        // (Measured in earlier rounds and still true: the wait at the barrier below is other workgroups' turn on the CU.)
        job_c* const Jw = job();
        const double w_dTc = Jw->dTc, w_dTk = Jw->dTk, w_rhs = Jw->pf_rhs;
        const uint32_t w_lf_n = Jw->lf_n;
        const kl_consts w_kl { Jw->kl_qc, Jw->kl_qk, Jw->kl_cut, Jw->kl_max };
        typedef __attribute__((address_space(3))) uint8_t lds_u8;
        typedef __attribute__((address_space(3))) uint32_t lds_u32s;
        uint8_t* const q = M.queue[wave];
        uint32_t qn = 0;                                                  // slots queued (wave-uniform)
        // (row_may_pass, kmd_eval.h, on sums that are exact in one conversion; pf_rhs = pf_cut Tc Tk from the host)
        auto may_pass = [&](unsigned long long c, unsigned long long k) -> bool
        {
          const double dsc = (double)c, dsk = (double)k;
          const double a = dsc * w_dTk - dsk * w_dTc;
          return !(a * a < (dsc + dsk) * w_rhs);
        };
        // the rows of the lanes that call it (a subset of the wave, together) leave for the list
        auto emit = [&](unsigned long long key, unsigned long long key_hi, unsigned long long sum_c, unsigned long long sum_k)
        {
          const unsigned long long m = ballot(true);
          const uint32_t rank = (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
          // (one lane adds for all of them -- written as the instruction: the compiler's own aggregation of a uniform
          // atomic would wrap a second ballot, prefix count and broadcast around this one)
          uint32_t first = 0;
          if (rank == 0)
            asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(first) : "v"((uint32_t)(uintptr_t)(lds_u32s*)&M.stage_n), "v"((uint32_t)__popcll(m)) : "memory");
          const uint32_t at = (uint32_t)__builtin_amdgcn_readfirstlane((int)first) + rank;
          if (at < kStage)
          {
            M.stage_key[at] = key; M.stage_c[at] = (typename lds_t::stage_sum_t)sum_c; M.stage_k[at] = (typename lds_t::stage_sum_t)sum_k;
            if constexpr (kTwo) M.stage_hi[at] = key_hi;
            return;
          }
          // (more rows than the tile may park: straight into the workgroup's chunk of the list)
          job_c* const Je = job();
          const uint32_t pos = atomicAdd(&M.out_used, 1u);
          unsigned long long e = M.out_base + pos;
          if (pos >= M.out_cap)
          {
            // ... and that is full, too: a chunk of this wave's own, the rows at its head, the rest holes
            const unsigned long long lm = ballot(true);
            const uint32_t lrank = (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(lm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lm, 0u));
            const uint32_t ltot = (uint32_t)__popcll(lm);
            unsigned long long fresh = 0;
            if (lrank == 0) fresh = atomicAdd(Je->n_rows, (unsigned long long)kOutChunk);
            fresh = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(fresh >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)fresh);
            e = fresh + lrank;
            for (unsigned long long h = fresh + ltot + lrank; h < fresh + kOutChunk; h += ltot)
              if (h < Je->row_capacity) Je->sum_c_out[h] = kHole;
          }
          if (e < Je->row_capacity)
          {
            Je->kmer_out[e] = key; Je->sum_c_out[e] = sum_c; Je->sum_k_out[e] = sum_k;
            if constexpr (kTwo) Je->kmer_hi_out[e] = key_hi;
          }
        };
        // The queue holds a slot in ONE byte: a wave walks the pairs tid, tid + kThreads, ... -- its slots are
        // step x 2 kThreads + 128 wave + (0 .. 127) -- so an entry is [step : 1][2 lane + half : 7] and the slot comes back
        // with a shift and two adds (the second table's slots, walked by the workgroup's last waves behind their share of
        // the main table, when the queue has been drained: step 0 again, another base).
        // The last `cnt` (<= 64) slots of the queue: their rows through the pre-filter, the slots wiped.  kFull: 64 of them,
        // every lane a row -- no lane mask around the pass.
        auto evaluate = [&](const uint32_t cnt, const uint32_t slot_base, auto full_tag)
        {
          constexpr bool kFull = decltype(full_tag)::value;
          qn -= cnt;
          if (!bad_tile) rows_wave += cnt;
          if (kFull || lane < cnt)
          {
            const uint32_t e = *(const lds_u8*)(uintptr_t)(q + qn + lane);
            const uint32_t i = slot_base + (e >> 7) * (2u * (uint32_t)kThreads) + (e & 127u);
            unsigned long long c, k;
            read_sums(i, c, k);
            if (!bad_tile)
            {
              beyond_wave += (uint32_t)__popcll(ballot((c >= w_lf_n) | (k >= w_lf_n)));
              bool leaves = may_pass(c, k);
#if KMD_TILE_ABLATE & 4   // dev: no pre-filter evaluation, nothing leaves (results wrong)
              leaves = false;
#endif
              if (ballot(leaves)) { if (leaves) leaves = row_may_pass_kl(w_kl, c, k); }
              if (ballot(leaves)) { if (leaves) emit(M.key[i], kTwo ? M.key_hi[i] : 0ull, c, k); }
            }
            M.key[i] = kEmptyKey; wipe_sums(i);
            if constexpr (kTwo) { M.key_hi[i] = 0; M.hi_min[i] = ~0ull; }
          }
        };
        // a pair's two slots into the queue (step: the entry's top bit)
        auto enqueue_pair = [&](const u64x2 two_keys, const uint32_t step)
        {
          const bool live0 = two_keys.x != kEmptyKey, live1 = two_keys.y != kEmptyKey;
          const unsigned long long m0 = ballot(live0), m1 = ballot(live1);
          const uint32_t n0 = (uint32_t)__popcll(m0);
          const uint32_t p0 = qn + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u));
          const uint32_t p1 = qn + n0 + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u));
          const uint32_t e0 = (step << 7) | (lane << 1);
          if (live0) *(lds_u8*)(uintptr_t)(q + p0) = (uint8_t)e0;
          if (live1) *(lds_u8*)(uintptr_t)(q + p1) = (uint8_t)(e0 | 1u);
          qn += n0 + (uint32_t)__popcll(m1);
        };
        static_assert(kSlots % (2 * kThreads) == 0 && (kSec / 2) % 64 == 0 && kSec / 2 <= (uint32_t)kThreads, "the walk's steps");
        constexpr uint32_t kMainSteps = kSlots / (2u * kThreads);
        static_assert(kMainSteps <= 2, "one bit of a queue entry says which step");
        static_assert(kQueue >= 63u + 128u, "a step's 128 slots behind what the last pass left");
        const uint32_t wave_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
        // (the second table's pairs: the workgroup's last waves -- wave 0 has the next tile's segments to fetch)
        constexpr uint32_t kSecWaves = kSec / 128u;
        const uint32_t n_phases = wave_u >= (uint32_t)kWaves - kSecWaves ? 2u : 1u;
#pragma nounroll
        for (uint32_t phase = 0; phase < n_phases; ++phase)
        {
          // phase 0: this wave's share of the main table; phase 1: of the second table
          const uint32_t slot_base = phase == 0u ? wave_u * 128u : kSlots + (wave_u - ((uint32_t)kWaves - kSecWaves)) * 128u;
          const uint32_t n_steps = phase == 0u ? kMainSteps : 1u;
#pragma nounroll
          for (uint32_t st = 0; st < n_steps; ++st)
          {
            const uint32_t first_slot = slot_base + st * 2u * (uint32_t)kThreads + 2u * lane;
            enqueue_pair(*(const lds_u64x2*)(uintptr_t)(M.key + first_slot), st);
#pragma nounroll
            while (qn >= 64u) evaluate(64u, slot_base, std::true_type());
          }
          if (qn) evaluate(qn, slot_base, std::false_type());
        }
        if (__builtin_amdgcn_readfirstlane((int)(wave == 0 && M.hasmax != 0)))       // the all-ones k-mer, if this tile had it
        {
          if (!bad_tile)
          {
            const unsigned long long mc = M.maxsum[0], mk = M.maxsum[1];
            rows_wave += 1u;
            beyond_wave += (mc >= w_lf_n || mk >= w_lf_n) ? 1u : 0u;
            if (lane == 0 && may_pass(mc, mk) && row_may_pass_kl(w_kl, mc, mk)) emit(kEmptyKey, M.max_hi[1], mc, mk);
          }
          if (lane == 0) { M.hasmax = 0; M.maxsum[0] = 0; M.maxsum[1] = 0; M.max_hi[0] = ~0ull; M.max_hi[1] = 0; }
        }
#if KMD_TILE_TIMING
        tp_walk1 = __builtin_readcyclecounter();
#endif
      }
      else
      {
      // the walk, rows mode (kmd_merge_sums: every row leaves, the output is compact): every thread owns kWalk
      // slots.  Pass 1 counts the rows; the tile takes that many consecutive entries of the output with ONE
      // global atomic; pass 2 writes them and wipes the slots.
      uint32_t out_bits = 0, mine = 0;
      bool special_out = false;
      if (!bad_tile)
      {
#pragma unroll
        for (int j = 0; j < kWalk; ++j)
        {
          const uint32_t i = tid + (uint32_t)j * kThreads;
          const bool live = i < kAll && M.key[i] != kEmptyKey;           // (the last step covers the end of the second table)
          bool leaves = live;
#if KMD_TILE_ABLATE & 4   // dev: nothing leaves (results wrong)
          leaves = false;
#endif
          out_bits |= leaves ? 1u << j : 0u;
          mine += leaves ? 1u : 0u;
        }
        if (tid == 0 && M.hasmax)                          // the all-ones k-mer, if this tile had it
        {
          special_out = true;
          mine += special_out ? 1u : 0u;
        }
      }
#if KMD_TILE_TIMING
      tp_walk1 = __builtin_readcyclecounter();
#endif
#if !(KMD_TILE_ABLATE & 8)   // dev: no reservation (and its two barriers)
      for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
      if (lane == 0) M.wcnt[wave] = mine;
      __syncthreads();
      if (tid == 0)
      {
        uint32_t total = 0;
        for (int w = 0; w < kWaves; ++w) total += M.wcnt[w];
        M.base = total ? atomicAdd(J.n_rows, (unsigned long long)total) : 0ull;
      }
      __syncthreads();
#endif
#if KMD_TILE_TIMING
      tp_resv = __builtin_readcyclecounter();
#endif
      unsigned long long out_at = M.base;
      for (uint32_t w = 0; w < wave; ++w) out_at += M.wcnt[w];
      if (special_out)                                     // (thread 0: the first entry of the tile)
      {
        if (out_at < J.row_capacity)
        {
          J.kmer_out[out_at] = kEmptyKey; J.sum_c_out[out_at] = M.maxsum[0]; J.sum_k_out[out_at] = M.maxsum[1];
          if constexpr (kTwo) J.kmer_hi_out[out_at] = M.max_hi[1];
        }
      }
      if (wave == 0) out_at += __shfl((unsigned long long)(special_out ? 1u : 0u), 0, 64);
#pragma unroll
      for (int j = 0; j < kWalk; ++j)
      {
        const uint32_t i = tid + (uint32_t)j * kThreads;
        const uint64_t key = i < kAll ? M.key[i] : kEmptyKey;
        const bool live = key != kEmptyKey;
        const bool leaves = (out_bits >> j) & 1u;
        const unsigned long long m = ballot(leaves);
        if (live)
        {
          if (leaves)
          {
            const unsigned long long e = out_at + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
            if (e < J.row_capacity)
            {
              unsigned long long sum_c, sum_k;
              read_sums(i, sum_c, sum_k);
              J.kmer_out[e] = key; J.sum_c_out[e] = sum_c; J.sum_k_out[e] = sum_k;
              if constexpr (kTwo) J.kmer_hi_out[e] = M.key_hi[i];
            }
          }
          M.key[i] = kEmptyKey; wipe_sums(i);
          if constexpr (kTwo) { M.key_hi[i] = 0; M.hi_min[i] = ~0ull; }
        }
        out_at += (unsigned long long)__popcll(m);
      }
      if (tid == 0 && M.hasmax) { M.hasmax = 0; M.maxsum[0] = 0; M.maxsum[1] = 0; M.max_hi[0] = ~0ull; M.max_hi[1] = 0; }
      }
    }
#if KMD_TILE_TIMING
    const unsigned long long tp_walk2 = __builtin_readcyclecounter();
#endif
    if constexpr (kFilter) lds_barrier(); else __syncthreads();
#if KMD_TILE_TIMING
    if (blockIdx.x == 77 && tid == 64)
      printf("[tile phases] tile %u (%u records): inserts %llu, inserts -> barrier %llu, next segments %llu, walk %llu (rows mode: pass 1 %llu, reservation %llu), last barrier %llu\n",
             tile, n, tp_ins - tp_tile, tp_bar1 - tp_ins, tp_seg - tp_bar1, tp_walk2 - tp_seg, tp_walk1 ? tp_walk1 - tp_seg : 0ull, tp_resv ? tp_resv - tp_walk1 : 0ull,
             __builtin_readcyclecounter() - tp_walk2);
#endif
  }

  if constexpr (kFilter)
  {
    // the last tile's parked rows; then what is left of this workgroup's chunk, and the first chunks of workgroups
    // this launch does not have (the list was laid out for n_regions of them): holes
    flush_stage();
    __syncthreads();
    job_c* const Jz = job();
    const unsigned long long row_cap = Jz->row_capacity, first_base = Jz->first_base;
    unsigned long long* const sum_c_out = Jz->sum_c_out;
    for (unsigned long long e = M.out_base + (M.out_used < M.out_cap ? M.out_used : M.out_cap) + tid; e < M.out_base + M.out_cap; e += kThreads)
      if (e < row_cap) sum_c_out[e] = kHole;
    for (uint32_t r = blockIdx.x + gridDim.x; r < Jz->n_regions; r += gridDim.x)
      for (unsigned long long e = first_base + (unsigned long long)r * kOutChunk + tid; e < first_base + (unsigned long long)(r + 1u) * kOutChunk; e += kThreads)
        if (e < row_cap) sum_c_out[e] = kHole;
    // this workgroup's rows and rows beyond the log-factorial table (merge.hpp:76; kmd_filter.hip counts
    // the same two for rows of a matrix)
    if (lane == 0 && rows_wave) atomicAdd(&Jz->row_total[0], (unsigned long long)rows_wave);
    if (lane == 0 && beyond_wave) atomicAdd(&Jz->row_total[1], (unsigned long long)beyond_wave);
  }
}

// kernel attribute (LDS beyond 64 KB) once per kernel and device
template <typename K> int allow_lds(K kernel, size_t lds_bytes)
{
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, size_t> allowed;
  if (lds_bytes <= 64 * 1024) return KMD_OK;
  const void* fn = reinterpret_cast<const void*>(kernel);
  int dev = 0;
  KMD_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  size_t& have = allowed[{ fn, dev }];
  if (lds_bytes > have)
  {
    KMD_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    have = lds_bytes;
  }
  return KMD_OK;
}

// The blocks go back to the process-wide cache when the set dies -- where another call (another partition in
// flight on another host thread) may take them at once.  On the normal way out the caller has synchronised the
// stream; on an error return kernels enqueued on it may still be running: the set then drains the stream first.
struct scratch_set
{
  std::vector<void*> blocks;
  hipStream_t stream = nullptr;
  bool drained = false;                                    // set by the owner once nothing enqueued reads the blocks
  explicit scratch_set(hipStream_t st) : stream(st) {}
  hipError_t take(void** out, size_t bytes)
  {
    const hipError_t e = kmd::scratch_alloc(out, bytes ? bytes : 1);
    if (e == hipSuccess) blocks.push_back(*out);
    return e;
  }
  ~scratch_set()
  {
    if (!drained && !blocks.empty()) (void)hipStreamSynchronize(stream);
    for (void* b : blocks) kmd::scratch_free(b);
  }
};

struct tile_shape { int threads; uint32_t slots; };
template <int kT, uint32_t kS> struct shape_tag { static constexpr int threads = kT; static constexpr uint32_t slots = kS; };

inline tile_shape pick_shape()
{
  tile_shape sh { 0, 0 };                                 // (0: the plan picks, make_plan)
  if (const char* e = std::getenv("KMD_TILE_SHAPE"))     // dev: "threads x slots" (A/B)
  {
    int t = 0; unsigned s = 0;
    if (std::sscanf(e, "%dx%u", &t, &s) == 2) { sh.threads = t; sh.slots = s; }
  }
  return sh;
}

inline uint32_t env_u32(const char* name, uint32_t dflt)
{
  const char* e = std::getenv(name);
  return e ? (uint32_t)std::strtoul(e, nullptr, 10) : dflt;
}

// streams -> entries (k-mer, control sum, case sum): every row (pf == NULL, kmd_merge_sums) or the rows
// the chi-square pre-filter of *pf lets through (kmd_merge_filter).  *n_entries = entries the run
// produced (the first row_capacity of them written), totals[0] = distinct k-mers, totals[1] = rows with a
// count sum beyond the log-factorial table (pre-filter mode).
// Synchronous: the tiles that gave up are known only when the kernel has run.
// `behind_level0` (may be empty): work the caller wants enqueued right behind the first pass, BEFORE the host
// learns how that pass went -- it is given the device addresses of [entries, distinct k-mers, rows beyond the
// table] and of the count of unfinished tiles, and must gate itself on them (k_cand_eval / _scan / _emit do).
// *clean = the first pass finished every tile (the gated work was live if the list did not overflow either).
using level0_hook = std::function<int(const uint64_t* d_live, const uint32_t* d_over_n)>;
// `async` (kmd_merge_filter_batch): the first pass and what the hook puts behind it are enqueued, the 64 bytes that
// say how it went are copied to async->h_small (page-locked), and the function returns WITHOUT waiting: the caller
// synchronises the stream later and reads them (tiles that gave up, a list that overflowed: it runs the partition
// again the synchronous way).  The scratch blocks and the upload staging then belong to *async.
struct merge_async
{
  scratch_set sc;
  char* h_small = nullptr;                         // 64 page-locked bytes: the read-back
  uint64_t* h_up = nullptr;                        // page-locked staging of the upload (offsets, index offsets), kUpWords words
  int way = -1;                                    // -1: launch both instantiations (the plan picks one on the device); 1 / 0: only
                                                   // the whole-wave / sub-group one -- a guess (the batch's earlier partitions): the
                                                   // read-back says whether it was taken (bytes 60..63)
  uint32_t shape = 0;                              // likewise the table shape (slots; 0: launch both)
  explicit merge_async(hipStream_t st) : sc(st) {}
};
constexpr size_t kUpWords = (size_t)kMaxStreams + 1 + ((size_t)kMaxStreams + 2) / 2;
// the table shape the last plan on each device took (a job's partitions are alike): what level 0 of the next call -- and
// the first partitions of the next batch -- launch alone; the read-back says whether that was right
std::atomic<uint32_t> g_last_shape[64];
int tile_merge(int S, int nc, const uint64_t* d_keys, const uint64_t* d_keys_hi, const uint32_t* d_counts,
               const uint64_t* offsets, const filter_params* pf, uint64_t* d_kmer_out, uint64_t* d_kmer_hi_out,
               uint64_t* d_sum_c, uint64_t* d_sum_k, size_t row_capacity, uint64_t* n_entries, uint64_t totals[2], hipStream_t st,
               const level0_hook& behind_level0 = level0_hook(), bool* clean = nullptr, struct merge_async* async = nullptr)
{
  const bool fused = pf != nullptr;
  const size_t n = (size_t)offsets[S];
  const bool two = d_keys_hi != nullptr;
  const bool dbg = std::getenv("KMD_DEBUG") != nullptr;
  int dev = 0, n_cu = 256;
  KMD_HIP(hipGetDevice(&dev));
  KMD_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  const tile_shape sh = pick_shape();
  const float load = (float)env_u32("KMD_TILE_LOAD_PCT", 50) / 100.0f;        // distinct k-mers per slot aimed at
  // two limbs under the 4096-slot table: 32 bytes per slot -- the segment tables of ~650 samples still fit beside it
  const bool two_limb_big_fits = sizeof(tile_lds<kBigSlots, KMD_TILE_BIG_THREADS / 64, true, true>) + 8 + 6 * (size_t)S * 4 <= 160 * 1024;

  uint32_t L = 0;
  for (int s = 1; s < S; ++s) if (offsets[s + 1] - offsets[s] > offsets[L + 1] - offsets[L]) L = (uint32_t)s;
  const uint64_t n_l = offsets[L + 1] - offsets[L];
  // the most tiles the plan can ask for: every record its own row (rho = 1), or the dev override
  const uint32_t slots_min = sh.slots ? sh.slots : kSmallSlots;
  uint32_t fill_min = (uint32_t)std::max(64.0f, load * (float)slots_min);
  if (const uint32_t f = env_u32("KMD_TILE_FILL", 0)) fill_min = std::min(fill_min, std::max(64u, f));
  const uint64_t r_min = std::max<uint64_t>(1, (uint64_t)((double)n_l * (double)fill_min / (double)n));
  const uint32_t grid_hint = env_u32("KMD_TILE_GRID_HINT", (uint32_t)n_cu * 4u), grid_hint_big = env_u32("KMD_TILE_GRID_HINT", (uint32_t)n_cu * 2u);   // workgroups of the level-0 launch (512 / 1024 threads)
  // (+ grid_hint: the plan may round the number of tiles up to a multiple of it)
  const uint32_t nb_max = (uint32_t)std::max<uint64_t>(1, (n_l + r_min - 1) / r_min) + std::max(grid_hint, grid_hint_big);
  // candidates mode: the most workgroups a launch of the merge kernel can have -- each owns a first chunk of the list
  const uint32_t regions_max = (uint32_t)n_cu * std::max<uint32_t>(env_u32("KMD_TILE_BLOCKS_PER_CU", 0), 2048u / (uint32_t)(sh.threads ? sh.threads : 512));

  scratch_set sc_own(st);
  scratch_set& sc = async ? async->sc : sc_own;
  void *p_offs = nullptr, *p_table = nullptr, *p_over = nullptr, *p_small = nullptr;
  KMD_HIP(sc.take(&p_offs, ((size_t)S + 1 + ((size_t)S + 2) / 2) * 8));
  KMD_HIP(sc.take(&p_table, ((size_t)nb_max + 1) * (size_t)S * 4));
  KMD_HIP(sc.take(&p_small, 64 + (size_t)kProbes * 4));              // [plan | entries, rows, rows beyond the table | probe multiplicities]
  // coarse boundaries: every R0-th key of the longest stream, at most 4096 of them (fewer with many samples: <= 2^18
  // searches, 2^16 beyond 64 samples -- as many as hide behind the probe's)
  // (measured, 20v20 / 36 M rows of 3 records / 100v100: probe + boundaries 49 / 238 / 160 us before, 46 / 167 / 143 now)
  const uint32_t c_max = std::max<uint32_t>(1u, std::min<uint32_t>(env_u32("KMD_TILE_COARSE", 4096u), (1u << env_u32("KMD_TILE_COARSE_CELLS", S <= 64 ? 18 : 16)) / (uint32_t)S));
  const uint32_t R0 = (uint32_t)std::max<uint64_t>(1, (n_l + c_max - 1) / c_max);
  const uint32_t C = (uint32_t)std::max<uint64_t>(1, (n_l + R0 - 1) / R0);
  void* p_coarse = nullptr;
  KMD_HIP(sc.take(&p_coarse, ((size_t)C + 1) * (size_t)S * 4));
  tile_plan* d_plan = static_cast<tile_plan*>(p_small);
  unsigned long long* d_rows = reinterpret_cast<unsigned long long*>(static_cast<char*>(p_small) + 32);
  uint32_t* d_over_n = reinterpret_cast<uint32_t*>(static_cast<char*>(p_small) + 56);
  uint32_t* d_ran = reinterpret_cast<uint32_t*>(static_cast<char*>(p_small) + 60);          // (= d_over_n + 1: the gate of the candidates' kernels reads both)
  uint32_t* d_mult = reinterpret_cast<uint32_t*>(static_cast<char*>(p_small) + 64);
  // [offsets (S + 1) x u64 | index offsets (S + 1) x u32]: one upload
  std::vector<uint64_t> h_up_own;
  const size_t up_words = (size_t)S + 1 + ((size_t)S + 2) / 2;
  if (!async) h_up_own.assign(up_words, 0);
  uint64_t* const h_up = async ? async->h_up : h_up_own.data();
  std::memcpy(h_up, offsets, ((size_t)S + 1) * 8);
  uint32_t* h_ioff = reinterpret_cast<uint32_t*>(h_up + S + 1);
  h_ioff[0] = 0;
  for (int s = 0; s < S; ++s) h_ioff[s + 1] = h_ioff[s] + index_samples(offsets[s + 1] - offsets[s]);
  const uint32_t n_index = h_ioff[S];
  void *p_idx = nullptr, *p_idx_hi = nullptr;
  KMD_HIP(sc.take(&p_idx, (size_t)n_index * 8));
  if (two) KMD_HIP(sc.take(&p_idx_hi, (size_t)n_index * 8));
  KMD_HIP(hipMemcpyAsync(p_offs, h_up, up_words * 8, hipMemcpyHostToDevice, st));
  const uint64_t* d_offs = static_cast<const uint64_t*>(p_offs);
  const uint32_t* d_ioff = reinterpret_cast<const uint32_t*>(d_offs + S + 1);
  const stream_index X { static_cast<const uint64_t*>(p_idx), static_cast<const uint64_t*>(p_idx_hi), d_ioff };
  {
    constexpr uint32_t kZeroWords = 16 + kProbes;                     // p_small: 64 bytes + the probe's counts
    hipLaunchKernelGGL(k_tile_index, dim3((std::max(n_index, kZeroWords) + 255) / 256), dim3(256), 0, st, d_keys, d_keys_hi, d_offs, d_ioff, (uint32_t)S,
                       static_cast<uint64_t*>(p_idx), static_cast<uint64_t*>(p_idx_hi), static_cast<uint32_t*>(p_small), kZeroWords);
    const size_t coarse_cells = ((size_t)C + 1) * S;
    hipLaunchKernelGGL(k_tile_probe, dim3((unsigned)((probes_for((uint32_t)S) * (size_t)S + 255) / 256 + (coarse_cells + 255) / 256)), dim3(256), 0, st, d_keys, d_keys_hi, d_offs, X, (uint32_t)S,
                       (uint64_t)n, d_mult, L, R0, C, static_cast<uint32_t*>(p_coarse));
    const size_t cells = ((size_t)nb_max + 1) * S;                      // (threads beyond the plan's tiles leave at once)
    hipLaunchKernelGGL(k_tile_bounds, dim3((unsigned)std::min<size_t>((cells + 255) / 256, (size_t)n_cu * 8)), dim3(256), 0, st, d_keys, d_keys_hi, d_offs, (uint32_t)S, L,
                       d_mult, (uint64_t)n, n_l, sh.slots ? sh.slots : (two && (std::getenv("KMD_TILE_SUM64") || !two_limb_big_fits) ? kSmallSlots : 0u), load, env_u32("KMD_TILE_FILL", 0), env_u32("KMD_TILE_G", 0),
                       grid_hint, grid_hint_big, d_plan, static_cast<uint32_t*>(p_table), d_rows, (unsigned long long)(fused ? (size_t)regions_max * kOutChunk : 0),
                       R0, C, static_cast<const uint32_t*>(p_coarse));
    KMD_HIP(hipGetLastError());
  }

  tile_job J;
  std::memset(&J, 0, sizeof J);
  J.keys = d_keys; J.keys_hi = d_keys_hi; J.counts = d_counts;
  J.S = (uint32_t)S; J.nc = (uint32_t)nc; J.plan = d_plan;
  J.xcd_order = env_u32("KMD_TILE_XCD", 1);
  J.kmer_out = d_kmer_out; J.kmer_hi_out = d_kmer_hi_out;
  J.sum_c_out = reinterpret_cast<unsigned long long*>(d_sum_c); J.sum_k_out = reinterpret_cast<unsigned long long*>(d_sum_k);
  J.row_capacity = row_capacity;
  J.n_rows = d_rows;
  J.row_total = d_rows + 1;
  J.first_base = 0; J.n_regions = regions_max;
  J.ran = d_ran;
  J.force_wide = 0;
  J.kl_max = 0; J.kl_qc = J.kl_qk = J.kl_cut = 0.0f;
  if (pf)
  {
    J.dTc = pf->dTc; J.dTk = pf->dTk; J.dTcTk = pf->dTcTk; J.pf_cut = pf->pf_cut; J.pf_rhs = pf->pf_cut * pf->dTcTk; J.lf_n = pf->lf_n;
    if (pf->pf_cut > -INFINITY && !std::getenv("KMD_PREFILTER_KL_OFF"))
    {
      J.kl_qc = (float)(pf->dTc / pf->dT); J.kl_qk = (float)(pf->dTk / pf->dT);
      J.kl_cut = std::nextafterf((float)pf->lr_cut, -INFINITY);                 // (never above the cut)
      J.kl_max = std::min<uint32_t>(pf->lf_n, 1u << 16);
    }
  }

  auto launch = [&](auto kernel, int threads, size_t lds_fixed, uint32_t tiles_at_most) -> int
  {
    const size_t lds = (lds_fixed + 7) / 8 * 8 + 6 * (size_t)S * 4;        // + [2][begin | length | prefix] of S streams
    int rc = allow_lds(kernel, lds);
    if (rc != KMD_OK) return rc;
    int per_cu = 0;
    KMD_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds));
    if (per_cu < 1) per_cu = 1;
    if (const uint32_t e = env_u32("KMD_TILE_BLOCKS_PER_CU", 0)) per_cu = (int)e;
    size_t grid = (size_t)n_cu * (size_t)per_cu;
    if (grid > tiles_at_most) grid = tiles_at_most;
    if (grid > J.n_regions) grid = J.n_regions;
    if (dbg) std::fprintf(stderr, "[tile_merge] <= %u tiles, grid %zu x %d (%d per CU), lds %zu\n", tiles_at_most, grid, threads, per_cu, lds);
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(threads), lds, st, J);
    KMD_HIP(hipGetLastError());
    return KMD_OK;
  };
  // level 0: which way of streaming a tile the plan takes is decided on the device -- except that with few
  // samples every plan takes whole waves per run (a tile holds >= load x slots records, a run >= that / S)
  // (up to 64 samples a run holds >= 16 records: the whole-wave path, which takes runs of any length, is the faster
  // one there whatever the plan would say -- and the launch of an instantiation that only leaves again is saved)
  // (the smaller shape's tiles are the smaller ones: what holds for them holds for both)
  const bool wide_for_sure = !env_u32("KMD_TILE_FILL", 0) && !env_u32("KMD_TILE_G", 0) &&
                             (double)std::max(64.0f, load * (float)slots_min) / (double)S >= 16.0;
  J.force_wide = wide_for_sure ? 1u : 0u;
  // Which TABLE SHAPE the plan takes is decided on the device too (make_plan).  A launch of the other one leaves at
  // once, but not for free (its turn in the stream: ~5 us): level 0 launches the shape the last plan on this device
  // took -- a job's partitions are alike -- and the read-back says whether that was right (bytes 60..63: did a merge
  // kernel run); if not, level 0 is launched again with the shape the plan did take.  The batch keeps its own guess.
  uint32_t shape_level0 = sh.slots ? sh.slots : async ? async->shape : g_last_shape[dev & 63].load(std::memory_order_relaxed);   // 0: both
  uint32_t shape_known = sh.slots;                               // levels > 0: the plan's, read back
  bool sum32 = std::getenv("KMD_TILE_SUM64") == nullptr;     // 32-bit sums until a tile reports a count too large for them
  // (what the instantiations that can be launched take of the 160 KB of LDS, 1024 samples' segment tables included; the
  // 4096-slot table with two limbs fits up to ~650 samples: two_limb_big_fits, below)
#if KMD_TILE_BIG_SLOTS == 4096
  static_assert(sizeof(tile_lds<kBigSlots, KMD_TILE_BIG_THREADS / 64, false, false>) + 8 + 6 * kMaxStreams * 4 <= 160 * 1024, "4096 slots, one limb, 64-bit sums");
#endif
  static_assert(sizeof(tile_lds<kSmallSlots, 8, true, false>) + 8 + 6 * kMaxStreams * 4 <= 160 * 1024, "2048 slots, two limbs, 64-bit sums");
  auto run = [&](uint32_t tiles_at_most) -> int
  {
    // one instantiation per (shape, fused, two limbs, whole-wave runs, 32-bit sums)
    auto pick = [&](auto shape_tag) -> int
    {
      constexpr int T = decltype(shape_tag)::threads;
      constexpr uint32_t SL = decltype(shape_tag)::slots;
      int rc_ = KMD_OK;
      for (int wide_ = 1; wide_ >= 0 && rc_ == KMD_OK; --wide_)
      {
        if (J.n_tiles && (J.g_shift == 6) != (wide_ == 1)) continue;          // way known on the host: launch that one only
        if (!J.n_tiles && wide_ == 0 && wide_for_sure) continue;
        if (!J.n_tiles && !wide_for_sure && async && async->way >= 0 && async->way != wide_) continue;          // (a guess: see merge_async)
        const unsigned sel = (fused ? 8u : 0u) | (two ? 4u : 0u) | (wide_ ? 2u : 0u) | (sum32 ? 1u : 0u);
        switch (sel)
        {
#define KMD_TILE_SEL(F, W2, WD, S32) \
          case ((F) ? 8u : 0u) | ((W2) ? 4u : 0u) | ((WD) ? 2u : 0u) | ((S32) ? 1u : 0u): \
            rc_ = launch(k_tile_sums<T, SL, F, W2, WD, S32>, T, sizeof(tile_lds<SL, T / 64, W2, S32>), tiles_at_most); break;
          KMD_TILE_SEL(false, false, false, false) KMD_TILE_SEL(false, false, false, true)
          KMD_TILE_SEL(false, false, true, false)  KMD_TILE_SEL(false, false, true, true)
          KMD_TILE_SEL(false, true, false, false)  KMD_TILE_SEL(false, true, false, true)
          KMD_TILE_SEL(false, true, true, false)   KMD_TILE_SEL(false, true, true, true)
          KMD_TILE_SEL(true, false, false, false)  KMD_TILE_SEL(true, false, false, true)
          KMD_TILE_SEL(true, false, true, false)   KMD_TILE_SEL(true, false, true, true)
          KMD_TILE_SEL(true, true, false, false)   KMD_TILE_SEL(true, true, false, true)
          KMD_TILE_SEL(true, true, true, false)    KMD_TILE_SEL(true, true, true, true)
#undef KMD_TILE_SEL
        }
      }
      return rc_;
    };
    const uint32_t which = J.n_tiles ? shape_known : shape_level0;     // 0: both (level 0 only)
    if (which != 0 && which != kSmallSlots && which != kBigSlots) { kmd::set_error("kmd: KMD_TILE_SHAPE not built"); return KMD_E_INVALID; }
    if (sh.threads && !((sh.threads == 512 && sh.slots == kSmallSlots) || (sh.threads == KMD_TILE_BIG_THREADS && sh.slots == kBigSlots)))
    {
      kmd::set_error("kmd: KMD_TILE_SHAPE not built");
      return KMD_E_INVALID;
    }
    int rc_ = KMD_OK;
    if (which == 0 || which == kSmallSlots) rc_ = pick(shape_tag<512, kSmallSlots>());
    if (rc_ == KMD_OK && (which == 0 || which == kBigSlots)) rc_ = pick(shape_tag<KMD_TILE_BIG_THREADS, kBigSlots>());
    return rc_;
  };

  // level 0: the planned table; further levels: the slices of the tiles that gave up
  uint32_t n_tiles = 0;                                         // 0: the plan's (on the device)
  uint32_t list_cap = nb_max;
  const uint32_t* table = static_cast<const uint32_t*>(p_table);
  const uint8_t* todo = nullptr;
  tile_plan h_plan;
  std::memset(&h_plan, 0, sizeof h_plan);
  alignas(8) char h_small[64];
  unsigned long long h_len0 = 0;                                  // (source of an asynchronous copy: lives as long as the function)
  std::vector<uint32_t> h_over;
  for (int level = 0;; ++level)
  {
    KMD_HIP(sc.take(&p_over, 2 * (size_t)list_cap * 4));
    if (level > 0) KMD_HIP(hipMemsetAsync(d_over_n, 0, 4, st));
    if (level > 0 && fused)
    {
      // the first chunks of this launch's workgroups lie behind what the list holds so far
      unsigned long long so_far = 0;
      std::memcpy(&so_far, h_small + 32, 8);
      J.first_base = so_far;
      J.n_regions = std::min<uint32_t>(regions_max, n_tiles);                 // (the launch has at most a workgroup per tile)
      h_len0 = so_far + (unsigned long long)J.n_regions * kOutChunk;
      KMD_HIP(hipMemcpyAsync(d_rows, &h_len0, 8, hipMemcpyHostToDevice, st));
    }
    J.start = table; J.todo = todo; J.n_tiles = n_tiles;
    J.over_n = d_over_n; J.over = static_cast<uint32_t*>(p_over); J.over_stride = list_cap;
    int rc = run(list_cap);
    if (rc != KMD_OK) return rc;
    if (level == 0 && behind_level0)
    {
      rc = behind_level0(reinterpret_cast<const uint64_t*>(d_rows), d_over_n);
      if (rc != KMD_OK) return rc;
    }
    // one read-back per level: [plan | entries, rows, rows beyond the table | tiles listed]
    if (async)
    {
      KMD_HIP(hipMemcpyAsync(async->h_small, p_small, 64, hipMemcpyDeviceToHost, st));
      return KMD_OK;                                              // (the caller waits, reads, and owns the scratch)
    }
    KMD_HIP(hipMemcpyAsync(h_small, p_small, 64, hipMemcpyDeviceToHost, st));
    KMD_HIP(hipStreamSynchronize(st));
    std::memcpy(&h_plan, h_small, sizeof h_plan);
    const uint32_t n_over = *reinterpret_cast<const uint32_t*>(h_small + 56);
    if (level == 0)
    {
      const uint32_t h_ran = *reinterpret_cast<const uint32_t*>(h_small + 60);
      shape_known = h_plan.slots;
      if (!sh.slots) g_last_shape[dev & 63].store(h_plan.slots, std::memory_order_relaxed);
      if (!h_ran)
      {
        // the guessed shape was not the plan's: nothing ran (the candidates' kernels behind it included, which are
        // gated on that) -- once more, with the plan's
        KMD_REQUIRE(shape_level0 != 0 && shape_level0 != h_plan.slots, "kmd: the merge kernel did not run");
        if (dbg) std::fprintf(stderr, "[tile_merge] level 0 again: the plan took the %u-slot table, the launch was the %u-slot one\n", h_plan.slots, shape_level0);
        shape_level0 = h_plan.slots;
        --level;
        continue;
      }
    }
    if (dbg)
      std::fprintf(stderr, "[tile_merge] level %d: %u of %u tiles gave up (plan: %.2f records per row, %u records per tile, r %u, G %u)\n",
                   level, n_over, level ? n_tiles : h_plan.nb, h_plan.rho, h_plan.fill, h_plan.r, 1u << h_plan.g_shift);
    if (level == 0 && clean) *clean = n_over == 0;
    if (n_over == 0) break;
    KMD_REQUIRE(level < 80, "kmd: tile refinement did not converge");
    // cut the listed tiles by the key range they span: a tile that ran out of table is cut as if every
    // record were a row of its own (its records per row are unknown but not what the plan assumed)
    h_over.resize(2 * (size_t)n_over);
    KMD_HIP(hipMemcpy(h_over.data(), static_cast<uint32_t*>(p_over), (size_t)n_over * 4, hipMemcpyDeviceToHost));
    KMD_HIP(hipMemcpy(h_over.data() + n_over, static_cast<uint32_t*>(p_over) + list_cap, (size_t)n_over * 4, hipMemcpyDeviceToHost));
    std::vector<uint32_t> h_ref(3 * (size_t)n_over);              // tile, slices, first row
    for (uint32_t i = 0; i < n_over; ++i)
      if (h_over[n_over + i] & kBigBit) sum32 = false;            // a count >= 2^22: the tiles listed from here on are redone with 64-bit sums
    // (two limbs AND 64-bit sums: the 4096-slot table does not fit the LDS -- the slices are cut for the smaller one)
    if (two && !sum32) shape_known = kSmallSlots;
    const uint64_t per_slice = std::max<uint64_t>(64, (uint64_t)(load * (float)shape_known));
    uint64_t rows = 0;
    for (uint32_t i = 0; i < n_over; ++i)
    {
      const uint64_t cnt = h_over[n_over + i] & ~(kAbortBit | kBigBit);
      uint64_t m = 2 * ((cnt + per_slice - 1) / per_slice);
      if (m < 2) m = 2;
      if (m > (1u << 20)) m = 1u << 20;
      h_ref[i] = h_over[i]; h_ref[n_over + i] = (uint32_t)m; h_ref[2 * (size_t)n_over + i] = (uint32_t)rows;
      rows += m + 1;
    }
    KMD_REQUIRE(rows < 0x7FFFFFFFull, "kmd: tile refinement table too large");
    void *p_ref = nullptr, *p_sub = nullptr, *p_todo = nullptr;
    KMD_HIP(sc.take(&p_ref, h_ref.size() * 4));
    KMD_HIP(sc.take(&p_sub, (size_t)rows * (size_t)S * 4));
    KMD_HIP(sc.take(&p_todo, (size_t)rows));
    KMD_HIP(hipMemcpyAsync(p_ref, h_ref.data(), h_ref.size() * 4, hipMemcpyHostToDevice, st));
    const uint32_t* d_ref = static_cast<const uint32_t*>(p_ref);
    hipLaunchKernelGGL(k_tile_refine, dim3(n_over), dim3(256), 0, st, d_keys, d_keys_hi, table, (uint32_t)S, d_ref, d_ref + n_over,
                       d_ref + 2 * (size_t)n_over, static_cast<uint32_t*>(p_sub), static_cast<uint8_t*>(p_todo));
    KMD_HIP(hipGetLastError());
    KMD_HIP(hipStreamSynchronize(st));                            // h_ref is read by the copy
    table = static_cast<const uint32_t*>(p_sub);
    todo = static_cast<const uint8_t*>(p_todo);
    n_tiles = (uint32_t)(rows - 1);
    list_cap = n_tiles;
    // lanes per run for the slices: they hold ~per_slice / 2 records
    uint32_t g = 3;
    while (g < 6 && (double)(1u << g) < (double)per_slice * 0.5 / (double)S * 0.75) ++g;
    J.g_shift = g;
  }
  unsigned long long h_rows[3];
  std::memcpy(h_rows, h_small + 32, sizeof h_rows);               // (read back with the last level)
  if (n_entries) *n_entries = (uint64_t)h_rows[0];
  if (totals) { totals[0] = fused ? (uint64_t)h_rows[1] : (uint64_t)h_rows[0]; totals[1] = (uint64_t)h_rows[2]; }
  sc.drained = true;                                            // (the last level's read-back waited for the stream)
  return KMD_OK;
}

} // namespace

// ---- C-ABI -------------------------------------------------------------------------------------

extern "C" int kmd_merge_filter(const kmd_model* m, int n_samples, const uint64_t* d_kmers, const uint64_t* d_kmers_hi,
                                const uint32_t* d_counts, const uint64_t* offsets, double threshold,
                                const kmd_survivors* out, uint64_t* d_counters, uint64_t* n_rows_out, void* stream)
{
  KMD_REQUIRE(m && offsets && d_counters, "kmd_merge_filter: NULL model, offsets or counters");
  KMD_REQUIRE(n_samples == m->nc + m->nk, "kmd_merge_filter: n_samples != controls + cases of the model");
  KMD_REQUIRE((uint32_t)n_samples <= kMaxStreams, "kmd_merge_filter: more than 1024 samples");
  const size_t n = (size_t)offsets[n_samples];
  KMD_REQUIRE(n < kMaxRecords, "kmd_merge_filter: more than 2^32-129 records in one partition");
  for (int s = 0; s < n_samples; ++s)
  {
    KMD_REQUIRE(offsets[s] <= offsets[s + 1], "kmd_merge_filter: offsets must be ascending");
    KMD_REQUIRE(offsets[s + 1] - offsets[s] < kMaxRun, "kmd_merge_filter: more than 2^29-1 records of one sample in one partition");
  }
  if (n_rows_out) *n_rows_out = 0;
  if (n == 0) return KMD_OK;
  KMD_REQUIRE(d_kmers && d_counts, "kmd_merge_filter: NULL device buffers");
  kmd_tile t { d_counts, 4, KMD_LAYOUT_SOA, n, nullptr, nullptr, n, 0 };      // for the shared checks; never read as a matrix
  filter_params P;
  int rc = kmd::fill_filter_params(P, m, &t, threshold);
  if (rc != KMD_OK) return rc;
  P.counters = reinterpret_cast<unsigned long long*>(d_counters);
  if (out) P.out = *out;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // The merge leaves the rows the pre-filter lets through (~1 % of the rows; all of them when the
  // pre-filter is off) in a scratch list; should the list prove too small, the merge runs again with
  // the size it reported -- nothing of the first run has reached the caller's counters or sink.
  const bool two = d_kmers_hi != nullptr;
  // (the workgroups' first chunks alone are up to 2^18 entries; measured: 0.4 % of the records on configs[2]'s rows of 26
  // records, 1.8 % on rows of 3 -- a sixteenth leaves room, and a list that does overflow is run again below)
  size_t cap = std::max<size_t>((size_t)1 << 20, n / 16);
  if (const uint32_t e = env_u32("KMD_TILE_CAND_CAP", 0)) cap = e;
  uint64_t entries = 0, totals[2] = { 0, 0 };
  scratch_set sc(st);
  void *p_k = nullptr, *p_h = nullptr, *p_c = nullptr, *p_s = nullptr, *p_w = nullptr;
  for (int attempt = 0;; ++attempt)
  {
    KMD_HIP(sc.take(&p_k, cap * 8)); KMD_HIP(sc.take(&p_c, cap * 8)); KMD_HIP(sc.take(&p_s, cap * 8));
    if (two) KMD_HIP(sc.take(&p_h, cap * 8));
    KMD_HIP(sc.take(&p_w, kmd::filter_candidates_work_bytes(cap, m)));          // (the exact evaluation's: p-values per entry, offsets per wave)
    // The exact evaluation of the list is enqueued right behind the first pass of the merge, gated on the device
    // by "every tile finished and the list did not overflow" (almost always): one host round trip per call.
    bool clean = false;
    const size_t cap_now = cap;
    auto speculate = [&](const uint64_t* d_live, const uint32_t* d_over_n) -> int
    {
      return kmd::launch_filter_candidates(P, m, static_cast<const uint64_t*>(p_k), static_cast<const uint64_t*>(p_h),
                                           static_cast<const uint64_t*>(p_c), static_cast<const uint64_t*>(p_s), 0, 0, 0, p_w, cap_now, st,
                                           d_live, d_over_n, cap_now);
    };
    rc = tile_merge(n_samples, m->nc, d_kmers, d_kmers_hi, d_counts, offsets, &P, static_cast<uint64_t*>(p_k), static_cast<uint64_t*>(p_h),
                    static_cast<uint64_t*>(p_c), static_cast<uint64_t*>(p_s), cap, &entries, totals, st, speculate, &clean);
    if (rc != KMD_OK) return rc;
    if (std::getenv("KMD_DEBUG")) std::fprintf(stderr, "[kmd_merge_filter] list: %llu entries of %zu (holes included), %llu rows\n", (unsigned long long)entries, cap, (unsigned long long)totals[0]);
    if (entries <= cap)
    {
      if (n_rows_out) *n_rows_out = totals[0];
      if (clean) { sc.drained = true; return KMD_OK; }        // the gated launch did the work; tile_merge has synchronised behind it
      break;
    }
    // (which tiles give up and are cut again can depend on the order their k-mers arrived in: a second run may need a
    // chunk or two more than the first reported)
    KMD_REQUIRE(attempt < 4, "kmd_merge_filter: candidate list kept overflowing");
    cap = (size_t)entries + (size_t)entries / 4 + 4 * kOutChunk;
  }
  // the long way (tiles were cut again after the first pass): the list is complete only now
  rc = kmd::launch_filter_candidates(P, m, static_cast<const uint64_t*>(p_k), static_cast<const uint64_t*>(p_h),
                                     static_cast<const uint64_t*>(p_c), static_cast<const uint64_t*>(p_s), (size_t)entries, totals[0], totals[1], p_w, cap, st);
  if (rc != KMD_OK) return rc;
  KMD_HIP(hipStreamSynchronize(st));                            // the scratch list goes back to the cache
  sc.drained = true;
  return KMD_OK;
}

// ---- a batch of partitions ----------------------------------------------------------------------
// A job has hundreds of partitions and a quarter of a kmd_merge_filter call is not the merge kernel (index, probe,
// boundary searches, candidate evaluation, launches, the read-back and its wake-up).  Here the calls of up to
// kBatchStreams partitions are in flight on streams of the library's own -- everything a partition needs is enqueued
// without a host round trip (tile_merge, async), so the small kernels of one run beside the merge kernel of another
// -- and the host waits once per partition, in turn, when its stream slot is needed again (two or three slots are used: below).  A partition whose first
// pass did not finish every tile, or whose candidate list overflowed (both rare), is run again the synchronous way:
// nothing of its first run has reached the caller's counters or sink.
namespace {
#ifndef KMD_BATCH_STREAMS
#define KMD_BATCH_STREAMS 6          // partitions of a batch in flight (dev: A/B)
#endif
constexpr int kBatchStreams = KMD_BATCH_STREAMS;
constexpr size_t kBatchSlotBytes = 64 + kUpWords * 8;      // page-locked, per stream: [read-back | upload staging]
struct batch_streams
{
  hipStream_t st[kBatchStreams] = {};
  char* h_small = nullptr;                         // kBatchStreams x kBatchSlotBytes page-locked bytes
  hipEvent_t ev = nullptr;
  hipEvent_t cand_done[kBatchStreams] = {};        // behind the candidate evaluation (and its near-threshold pass) of the slot's latest partition
};
std::mutex g_batch_mu;
std::map<int, batch_streams> g_batch;              // per device

int batch_get(batch_streams** out)
{
  int dev = 0;
  KMD_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_batch_mu);
  batch_streams& B = g_batch[dev];
  if (!B.h_small)
  {
    for (int i = 0; i < kBatchStreams; ++i) KMD_HIP(hipStreamCreateWithFlags(&B.st[i], hipStreamNonBlocking));
    void* p = nullptr;
    KMD_HIP(hipHostMalloc(&p, (size_t)kBatchStreams * kBatchSlotBytes, hipHostMallocDefault));
    B.h_small = static_cast<char*>(p);
    KMD_HIP(hipEventCreateWithFlags(&B.ev, hipEventDisableTiming));
    for (int i = 0; i < kBatchStreams; ++i) KMD_HIP(hipEventCreateWithFlags(&B.cand_done[i], hipEventDisableTiming));
  }
  *out = &B;
  return KMD_OK;
}
} // namespace

extern "C" int kmd_merge_filter_batch(const kmd_model* m, int n_partitions, int n_samples, const uint64_t* const* d_kmers,
                                      const uint64_t* const* d_kmers_hi, const uint32_t* const* d_counts,
                                      const uint64_t* const* offsets, double threshold, const kmd_survivors* out,
                                      uint64_t* const* d_counters, uint64_t* n_rows_out, void* stream)
{
  KMD_REQUIRE(m && n_partitions >= 0 && d_kmers && d_counts && offsets && d_counters, "kmd_merge_filter_batch: NULL arguments");
  KMD_REQUIRE(n_samples == m->nc + m->nk, "kmd_merge_filter_batch: n_samples != controls + cases of the model");
  KMD_REQUIRE((uint32_t)n_samples <= kMaxStreams, "kmd_merge_filter_batch: more than 1024 samples");
  if (n_partitions == 0) return KMD_OK;
  batch_streams* B = nullptr;
  int rc = batch_get(&B);
  if (rc != KMD_OK) return rc;
  // one batch at a time per device (the streams and the read-back slots are the library's); what the caller's stream
  // holds so far comes first
  static std::mutex run_mu;
  std::lock_guard<std::mutex> run_lock(run_mu);
  hipStream_t user = static_cast<hipStream_t>(stream);
  KMD_HIP(hipEventRecord(B->ev, user));
  for (int i = 0; i < kBatchStreams; ++i) KMD_HIP(hipStreamWaitEvent(B->st[i], B->ev, 0));

  // (every partition's pointers are looked at before anything is enqueued: an error here leaves nothing half done)
  for (int p = 0; p < n_partitions; ++p)
    KMD_REQUIRE(offsets[p] && d_counters[p], "kmd_merge_filter_batch: NULL offsets or counters of a partition");
  // Partitions that share a survivor sink or counters.  The candidate evaluation of a partition ends with the pass over
  // its near-threshold rows (k_resolve_near, kmd_filter.hip), which may strike a record and then compacts the WHOLE sink
  // [0, counters[KMD_CNT_SIG]) in place -- assuming nothing else appends to it meanwhile.  On one stream that holds; two
  // partitions of a batch that name the same sink run on different streams.  So: prev_alias[p] = the latest earlier
  // partition that shares p's counters or any of its sink's arrays; p's candidate kernels wait for that partition's
  // (an event behind them on its stream).  The merge kernels -- 90 % of a partition's time -- still overlap.
  std::vector<int> prev_alias((size_t)n_partitions, -1);
  std::vector<char> shares((size_t)n_partitions, 0);
  std::vector<char> queued((size_t)n_partitions, 0);          // 1: in flight with an event behind its candidate kernels
  {
    auto same = [](const void* a, const void* b) { return a && a == b; };
    for (int p = 1; p < n_partitions; ++p)
      for (int q = p - 1; q >= 0; --q)
      {
        bool al = d_counters[p] == d_counters[q];
        if (out && !al)
        {
          const kmd_survivors &x = out[p], &y = out[q];
          al = same(x.d_row, y.d_row) || same(x.d_kmer_lo, y.d_kmer_lo) || same(x.d_kmer_hi, y.d_kmer_hi) || same(x.d_pvalue, y.d_pvalue) ||
               same(x.d_sign, y.d_sign) || same(x.d_mean_control, y.d_mean_control) || same(x.d_mean_case, y.d_mean_case);
        }
        if (al) { prev_alias[(size_t)p] = q; shares[(size_t)p] = shares[(size_t)q] = 1; break; }
      }
  }
  struct in_flight { int part = -1; std::unique_ptr<merge_async> A; size_t cap = 0; void *p_k = nullptr, *p_h = nullptr, *p_c = nullptr, *p_s = nullptr, *p_w = nullptr; };
  in_flight F[kBatchStreams];
  int first_error = KMD_OK;
  // Which instantiation of the merge kernel a partition's plan takes -- whole waves per run, or sub-groups of lanes --
  // is decided on the device, so a single call launches both and one leaves at once.  Among kernels of other
  // partitions that empty launch still has to wait for its turn on the CUs (60 us in the kernel trace of a batch)
  // with the rest of its stream behind it: once a partition of the batch has come back, the others launch only the
  // way its plan took -- the read-back says whether that was right, and the synchronous way stands behind it.
  int known_way = -1;
  // (the shape the last plan on this device took -- round 5: a batch used to start from "both", and each launch of the
  // instantiation that only leaves again waited ~120 us for its turn among the other partitions' kernels, with the rest of
  // its stream behind it: 19 such launches in a trace of 36 partitions)
  int dev_b = 0;
  KMD_HIP(hipGetDevice(&dev_b));
  uint32_t known_shape = g_last_shape[dev_b & 63].load(std::memory_order_relaxed);
  // the synchronous way, for the partitions the fast way could not finish
  auto redo = [&](int p) -> int
  {
    // (a partition that shares its sink: nothing of the batch may be appending to it while this call's
    // near-threshold pass compacts it -- the rare way, so simply: everything in flight first)
    if (shares[(size_t)p]) for (int i = 0; i < kBatchStreams; ++i) KMD_HIP(hipStreamSynchronize(B->st[i]));
    queued[(size_t)p] = 2;
    return kmd_merge_filter(m, n_samples, d_kmers[p], d_kmers_hi ? d_kmers_hi[p] : nullptr, d_counts[p], offsets[p], threshold,
                            out ? &out[p] : nullptr, d_counters[p], n_rows_out ? &n_rows_out[p] : nullptr, B->st[0]);
  };
  auto finish = [&](int slot) -> int
  {
    in_flight& f = F[slot];
    if (f.part < 0) return KMD_OK;
    const int p = f.part;
    f.part = -1;
    KMD_HIP(hipStreamSynchronize(B->st[slot]));
    f.A->sc.drained = true;
    const char* h = B->h_small + (size_t)slot * kBatchSlotBytes;
    unsigned long long rows3[3];
    std::memcpy(rows3, h + 32, sizeof rows3);
    const uint32_t n_over = *reinterpret_cast<const uint32_t*>(h + 56), ran = *reinterpret_cast<const uint32_t*>(h + 60);
    tile_plan pl;
    std::memcpy(&pl, h, sizeof pl);
    // the way and the table shape the plan of a partition of this job took -- also when the guess was wrong and no merge
    // kernel ran (k_tile_bounds wrote the plan all the same): the partitions behind it then launch what this one needed
    if (pl.slots == kSmallSlots || pl.slots == kBigSlots)
    {
      known_way = pl.g_shift == 6 ? 1 : 0; known_shape = pl.slots;
      if (!pick_shape().slots) g_last_shape[dev_b & 63].store(pl.slots, std::memory_order_relaxed);
    }
    const bool ok = ran != 0 && n_over == 0 && rows3[0] <= f.cap;
    f.A.reset();                                                  // the scratch goes back to the cache
    if (ok) { if (n_rows_out) n_rows_out[p] = rows3[1]; return KMD_OK; }
    return redo(p);
  };
  // How many partitions are in flight (round 5; kBatchStreams is the most): three -- two when partitions are of the job's
  // size.  A merge kernel fills the chip by itself; what a second partition in flight adds is its small kernels and its
  // first tiles in the other's tail, and more than that only has the persistent grids of several partitions take turns
  // on the same CUs.  Measured, per partition, in flight 6 / 3 / 2 (profiles/r05_ab_k2t.txt): 39 M rows 2.33-2.35 /
  // 2.20-2.21 / 2.15-2.17 ms, 16 M 0.98 / 0.96 / 0.95, 8 M 0.524 / 0.490 / 0.536, 4 M 0.30 / 0.29 / 0.31.
  int n_fly = 3;
  for (int p = 0; p < n_partitions; ++p)
    if (offsets[p][n_samples] != 0) { n_fly = offsets[p][n_samples] >= 500000000ull ? 2 : 3; break; }
  if (const uint32_t e = env_u32("KMD_BATCH_IN_FLIGHT", 0)) n_fly = (int)std::min<uint32_t>(e, kBatchStreams);      // dev: A/B
  for (int p = 0; p < n_partitions; ++p)
  {
    const int slot = p % n_fly;
    rc = finish(slot);
    if (rc != KMD_OK && first_error == KMD_OK) first_error = rc;
    if (n_rows_out) n_rows_out[p] = 0;
    const uint64_t* offs = offsets[p];
    const size_t n = (size_t)offs[n_samples];
    if (n == 0) continue;
    bool sane = n < kMaxRecords && d_kmers[p] && d_counts[p];
    for (int s = 0; s < n_samples && sane; ++s) sane = offs[s] <= offs[s + 1] && offs[s + 1] - offs[s] < kMaxRun;
    if (!sane)
    {
      rc = redo(p);                                               // (says what is wrong with it)
      if (rc != KMD_OK && first_error == KMD_OK) first_error = rc;
      continue;
    }
    hipStream_t st = B->st[slot];
    in_flight& f = F[slot];
    f.A.reset(new merge_async(st));
    f.A->h_small = B->h_small + (size_t)slot * kBatchSlotBytes;
    f.A->h_up = reinterpret_cast<uint64_t*>(f.A->h_small + 64);
    f.A->way = known_way;
    f.A->shape = known_shape;
    kmd_tile t { d_counts[p], 4, KMD_LAYOUT_SOA, n, nullptr, nullptr, n, 0 };
    filter_params P;
    rc = kmd::fill_filter_params(P, m, &t, threshold);
    if (rc != KMD_OK) { f.A.reset(); if (first_error == KMD_OK) first_error = rc; continue; }
    P.counters = reinterpret_cast<unsigned long long*>(d_counters[p]);
    if (out) P.out = out[p];
    const bool two = d_kmers_hi && d_kmers_hi[p];
    f.cap = std::max<size_t>((size_t)1 << 20, n / 16);
    if (const uint32_t e = env_u32("KMD_TILE_CAND_CAP", 0)) f.cap = e;
    hipError_t he = f.A->sc.take(&f.p_k, f.cap * 8);
    if (he == hipSuccess) he = f.A->sc.take(&f.p_c, f.cap * 8);
    if (he == hipSuccess) he = f.A->sc.take(&f.p_s, f.cap * 8);
    if (he == hipSuccess && two) he = f.A->sc.take(&f.p_h, f.cap * 8);
    if (he == hipSuccess) he = f.A->sc.take(&f.p_w, kmd::filter_candidates_work_bytes(f.cap, m));
    if (he != hipSuccess) { (void)hipGetLastError(); f.A.reset(); rc = redo(p); if (rc != KMD_OK && first_error == KMD_OK) first_error = rc; continue; }
    const size_t cap_now = f.cap;
    auto speculate = [&](const uint64_t* d_live, const uint32_t* d_over_n) -> int
    {
      // (the nearest earlier sharer that went the asynchronous way; one that was empty is stepped over, one that was
      // run the synchronous way has drained everything before it)
      int q = prev_alias[(size_t)p];
      while (q >= 0 && queued[(size_t)q] == 0) q = prev_alias[(size_t)q];
      if (q >= 0 && queued[(size_t)q] == 1 && q % n_fly != slot) KMD_HIP(hipStreamWaitEvent(st, B->cand_done[q % n_fly], 0));
      const int rc_ = kmd::launch_filter_candidates(P, m, static_cast<const uint64_t*>(f.p_k), static_cast<const uint64_t*>(f.p_h),
                                                    static_cast<const uint64_t*>(f.p_c), static_cast<const uint64_t*>(f.p_s), 0, 0, 0, f.p_w, cap_now, st,
                                                    d_live, d_over_n, cap_now);
      if (rc_ == KMD_OK && shares[(size_t)p]) { KMD_HIP(hipEventRecord(B->cand_done[slot], st)); queued[(size_t)p] = 1; }
      return rc_;
    };
    uint64_t entries = 0, totals[2] = { 0, 0 };
    rc = tile_merge(n_samples, m->nc, d_kmers[p], two ? d_kmers_hi[p] : nullptr, d_counts[p], offs, &P, static_cast<uint64_t*>(f.p_k),
                    static_cast<uint64_t*>(f.p_h), static_cast<uint64_t*>(f.p_c), static_cast<uint64_t*>(f.p_s), f.cap, &entries, totals, st,
                    speculate, nullptr, f.A.get());
    if (rc != KMD_OK) { (void)hipStreamSynchronize(st); f.A.reset(); if (first_error == KMD_OK) first_error = rc; continue; }
    f.part = p;
  }
  for (int slot = 0; slot < kBatchStreams; ++slot)
  {
    rc = finish(slot);
    if (rc != KMD_OK && first_error == KMD_OK) first_error = rc;
  }
  return first_error;
}

// The same merge for a consumer that wants the rows themselves: every distinct k-mer leaves as
// (k-mer, sum of its control counts, sum of its case counts), compact, in no particular order.
extern "C" int kmd_merge_sums(int n_samples, int nb_controls, const uint64_t* d_kmers, const uint64_t* d_kmers_hi,
                              const uint32_t* d_counts, const uint64_t* offsets, size_t row_capacity, uint64_t* d_kmer_out,
                              uint64_t* d_kmer_hi_out, uint64_t* d_sum_control, uint64_t* d_sum_case, uint64_t* n_rows_out,
                              void* stream)
{
  KMD_REQUIRE(n_samples > 0 && nb_controls >= 0 && nb_controls <= n_samples && offsets && n_rows_out, "kmd_merge_sums: arguments");
  KMD_REQUIRE((uint32_t)n_samples <= kMaxStreams, "kmd_merge_sums: more than 1024 samples");
  const size_t n = (size_t)offsets[n_samples];
  KMD_REQUIRE(n < kMaxRecords, "kmd_merge_sums: more than 2^32-129 records in one partition");
  for (int s = 0; s < n_samples; ++s)
  {
    KMD_REQUIRE(offsets[s] <= offsets[s + 1], "kmd_merge_sums: offsets must be ascending");
    KMD_REQUIRE(offsets[s + 1] - offsets[s] < kMaxRun, "kmd_merge_sums: more than 2^29-1 records of one sample in one partition");
  }
  *n_rows_out = 0;
  if (n == 0) return KMD_OK;
  KMD_REQUIRE(d_kmers && d_counts && d_kmer_out && d_sum_control && d_sum_case, "kmd_merge_sums: NULL device buffers");
  KMD_REQUIRE(!d_kmers_hi || d_kmer_hi_out, "kmd_merge_sums: two-limb k-mers need d_kmer_hi_out");
  const int rc = tile_merge(n_samples, nb_controls, d_kmers, d_kmers_hi, d_counts, offsets, nullptr, d_kmer_out, d_kmer_hi_out,
                            d_sum_control, d_sum_case, row_capacity, n_rows_out, nullptr, static_cast<hipStream_t>(stream));
  if (rc != KMD_OK) return rc;
  if (*n_rows_out > row_capacity) { kmd::set_error("kmd_merge_sums: row capacity exceeded"); return KMD_E_OVERFLOW; }
  return KMD_OK;
}
