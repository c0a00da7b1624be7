// kmd_filter.hip -- K1: merge observer + Poisson likelihood-ratio test + threshold +
// survivor compaction for one partition tile, hand-written for gfx950 (CDNA4, wave64).
//
// Replaces, per row, diff_observer::process (include/kmdiff/merge.hpp:68-103) calling
// PoissonLikelihood::process (include/kmdiff/model.hpp:142-176).
//
// Shape of the work (HBM-bound, no MFMA: nothing here is a contraction):
//   * one wavefront lane per k-mer row; with the SoA layout each lane owns RPL consecutive
//     rows so every column read is one 16-byte load per lane (1 KiB per wave instruction);
//   * persistent workgroups (grid = CUs x blocks/CU) stage the head of the log-factorial
//     table in LDS once and then grid-stride over row tiles;
//   * the four log() of the likelihood ratio run for every row, the chi-square tail
//     function only for rows whose LR can reach the threshold (igamc is monotone in LR; the
//     reference never exposes p for rows with p > threshold, merge.hpp:78);
//   * survivors are compacted with a wave ballot + one atomic per wave;
//   * count sums beyond the table (LogFactorialTable::operator[] fallback,
//     log_factorial_table.hpp:14-18) are evaluated in O(1) by the Stirling series (the
//     reference's O(k) loop would stall a wave for milliseconds on one high-count row).
#include "kmd_internal.h"
#include "kmd_math.h"
#include "kmd_eval.h"

#include <cstdlib>
#include <map>
#include <mutex>
#include <unordered_map>

using namespace kmd::eval;

namespace {

#ifndef KMD_BLOCK
#define KMD_BLOCK 1024
#endif
#ifndef KMD_BLOCKS_PER_CU
#define KMD_BLOCKS_PER_CU 1
#endif
#ifndef KMD_MINWAVES
#define KMD_MINWAVES 4
#endif
#ifndef KMD_BATCH
#define KMD_BATCH 4
#endif
#ifndef KMD_RPL_U32
#define KMD_RPL_U32 2
#endif
#ifndef KMD_RPL_U16
#define KMD_RPL_U16 4
#endif
#ifndef KMD_U8_DOT4
#define KMD_U8_DOT4 1
#endif
#ifndef KMD_RPL_U8
#define KMD_RPL_U8 8
#endif
#ifndef KMD_XPREFETCH
#define KMD_XPREFETCH 1
#endif
constexpr int kBlock = KMD_BLOCK;


template <typename T, int N> struct vec_of;
template <> struct vec_of<uint32_t, 1> { using type = uint32_t; };
template <> struct vec_of<uint32_t, 2> { using type = uint2; };
template <> struct vec_of<uint32_t, 4> { using type = uint4; };
template <> struct vec_of<uint16_t, 1> { using type = uint16_t; };
template <> struct vec_of<uint16_t, 2> { using type = uint32_t; };
template <> struct vec_of<uint16_t, 4> { using type = uint2; };
template <> struct vec_of<uint16_t, 8> { using type = uint4; };
template <> struct vec_of<uint8_t, 1> { using type = uint8_t; };
template <> struct vec_of<uint8_t, 4> { using type = uint32_t; };
template <> struct vec_of<uint8_t, 8> { using type = uint2; };
template <> struct vec_of<uint8_t, 16> { using type = uint4; };

// accumulator wide enough for nc+nk <= 65535 samples
template <typename CT> struct acc_of { using type = uint32_t; };
template <> struct acc_of<uint32_t> { using type = uint64_t; };

// the 4 / 2 counts packed in a dword added to acc in one instruction (v_sad_u8 / v_sad_u16 against
// zero): for the row-major kernels, where a lane's dword holds counts of ONE row
template <typename CT>
__device__ __forceinline__ uint32_t add_packed(uint32_t w, uint32_t acc)
{
  if constexpr (sizeof(CT) == 1) return __builtin_amdgcn_sad_u8(w, 0u, acc);
  else return __builtin_amdgcn_sad_u16(w, 0u, acc);
}



// One row from its two count sums to the survivor sink, at once.  Must be called by all 64 lanes
// of the wave together.
__device__ __forceinline__ void finish_row(const filter_params& P, const double2* s_tab,
                                           const row_state& st, uint32_t& n_beyond)
{
#ifdef KMD_ABLATE_MATH   // dev only: memory-side ceiling of the load loop (results are wrong)
  if (st.valid && (st.sum_c ^ st.sum_k) == 0x7fffffffffffull) P.counters[KMD_CNT_NEAR_UNRESOLVED] = st.row;
  return;
#endif
  row_state e = st;
  e.valid = row_may_pass(P, st, n_beyond);
  if (!__ballot(e.valid)) return;
  evaluate_row(P, s_tab, e);
}

// Deferred evaluation.  About 1 % of the rows pass the pre-filter, so in half of the 64-row
// steps of a wave SOME lane does -- and evaluating at once runs the two logarithms, the division
// and the compaction logic with one or two live lanes out of 64.  Instead the rows that pass are
// parked in a wave-private LDS queue and evaluated 64 at a time with every lane busy: ~40 x
// fewer passes through the expensive code.  The survivor order is unspecified either way
// (kmd_survivors_sort_by_row restores the reference's); every exposed number is unchanged.
#ifndef KMD_DEFER
#define KMD_DEFER 1
#endif
constexpr uint32_t kQueueCap = 128;                      // >= 63 parked + 64 pushed by one step
constexpr size_t kQueueBytesPerWave = kQueueCap * 3 * sizeof(unsigned long long);
struct wave_queue
{
  unsigned long long *sc, *sk, *row;                     // [kQueueCap] each, LDS
  uint32_t n;                                            // parked rows (wave-uniform)
};

__device__ __forceinline__ void queue_fence()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int kRowMode = 0>
__device__ __forceinline__ void defer_row(const filter_params& P, const double2* s_tab, const row_state& st,
                                          uint32_t& n_beyond, wave_queue& Q)
{
#ifdef KMD_ABLATE_MATH
  if (st.valid && (st.sum_c ^ st.sum_k) == 0x7fffffffffffull) P.counters[KMD_CNT_NEAR_UNRESOLVED] = st.row;
  return;
#endif
  const bool maybe = row_may_pass(P, st, n_beyond);
  const unsigned long long m = __ballot(maybe);
  if (!m) return;
  const uint32_t lane = (uint32_t)__lane_id();
  if (maybe)
  {
    const uint32_t at = Q.n + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    Q.sc[at] = st.sum_c; Q.sk[at] = st.sum_k; Q.row[at] = st.row;
  }
  Q.n += (uint32_t)__popcll(m);
  if (Q.n >= 64)
  {
    queue_fence();
    Q.n -= 64;                                           // the 64 most recent: nothing to move
    row_state e;
    e.sum_c = Q.sc[Q.n + lane]; e.sum_k = Q.sk[Q.n + lane]; e.row = Q.row[Q.n + lane];
    e.valid = true;
    queue_fence();                                       // read before the next step overwrites
    evaluate_row<kRowMode>(P, s_tab, e);
  }
}

template <int kRowMode = 0>
__device__ __forceinline__ void drain_queue(const filter_params& P, const double2* s_tab, wave_queue& Q)
{
  if (Q.n == 0) return;
  queue_fence();
  const uint32_t lane = (uint32_t)__lane_id();
  row_state e;
  e.valid = lane < Q.n;
  e.sum_c = e.valid ? Q.sc[lane] : 0; e.sum_k = e.valid ? Q.sk[lane] : 0; e.row = e.valid ? Q.row[lane] : 0;
  Q.n = 0;
  evaluate_row<kRowMode>(P, s_tab, e);
}


__device__ __forceinline__ void stage_table(const filter_params& P, double2* s_tab)
{
  for (uint32_t i = threadIdx.x; i < P.lds_n; i += blockDim.x) s_tab[i] = P.tab[i];
  __syncthreads();
}

// ---- SoA layout: counts[sample][row] ------------------------------------------------------
// Column loads are issued in batches of kBatch (one 16-byte load per lane per column when
// RPL > 1) into two register sets: while one batch is being added up the next one is already
// in flight, and the first batch of the NEXT tile is issued before the FP64 phase of the
// current one, so a wave always has kBatch..2*kBatch KiB-sized loads outstanding.
constexpr int kBatch = KMD_BATCH;

// Every count is read exactly once: with KMD_NT_LOADS the column loads carry the non-temporal
// hint (no reuse to protect in L2 / MALL).
#ifndef KMD_NT_LOADS
#define KMD_NT_LOADS 1
#endif
template <typename V> __device__ __forceinline__ V stream_load(const V* p)
{
#if KMD_NT_LOADS
  if constexpr (sizeof(V) == 16)
  {
    typedef uint32_t n4 __attribute__((ext_vector_type(4)));
    const n4 t = __builtin_nontemporal_load(reinterpret_cast<const n4*>(p));
    V r; r.x = t.x; r.y = t.y; r.z = t.z; r.w = t.w; return r;
  }
  else if constexpr (sizeof(V) == 8)
  {
    typedef uint32_t n2 __attribute__((ext_vector_type(2)));
    const n2 t = __builtin_nontemporal_load(reinterpret_cast<const n2*>(p));
    V r; r.x = t.x; r.y = t.y; return r;
  }
  else return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}

template <typename CT, int RPL>
struct soa_batch
{
  using V = typename vec_of<CT, RPL>::type;
  V v[kBatch];

  // columns s0 .. s0+kBatch-1 of the lane's RPL rows; indices past the last column re-read
  // column S-1 (served by L1/TA, never added)
  __device__ __forceinline__ void load(const CT* __restrict__ rows0, size_t ld, int s0, int S)
  {
#pragma unroll
    for (int j = 0; j < kBatch; ++j)
    {
      const int s = (s0 + j < S) ? (s0 + j) : (S - 1);
      v[j] = stream_load(reinterpret_cast<const V*>(rows0 + (size_t)s * ld));
    }
  }

  template <typename ACC>
  __device__ __forceinline__ void add_one(int j, ACC (&acc)[RPL]) const
  {
    if constexpr (RPL == 1)
    {
      acc[0] += (ACC)v[j];
    }
    else if constexpr (sizeof(CT) == 4 && RPL == 2)
    {
      acc[0] += v[j].x; acc[1] += v[j].y;
    }
    else if constexpr (sizeof(CT) == 4)
    {
      acc[0] += v[j].x; acc[1] += v[j].y; acc[2] += v[j].z; acc[3] += v[j].w;
    }
    else
    {
      constexpr int per = 4 / sizeof(CT);
      constexpr int ndw = RPL / per;                     // dwords in the vector
      constexpr uint32_t mask = sizeof(CT) == 1 ? 0xFFu : 0xFFFFu;
      uint32_t w[ndw];
      unpack_dwords(v[j], w);
#pragma unroll
      for (int d = 0; d < ndw; ++d)
#pragma unroll
        for (int e = 0; e < per; ++e)
        {
#if KMD_U8_DOT4
          // byte e of the dword added in one instruction: v_dot4_u32_u8 against a one-hot selector
          if constexpr (sizeof(CT) == 1 && sizeof(ACC) == 4)
            acc[d * per + e] = __builtin_amdgcn_udot4(w[d], 1u << (8 * e), acc[d * per + e], false);
          else
#endif
          acc[d * per + e] += (w[d] >> (8 * sizeof(CT) * e)) & mask;
        }
    }
  }

  static __device__ __forceinline__ void unpack_dwords(const uint32_t& x, uint32_t (&w)[1]) { w[0] = x; }
  static __device__ __forceinline__ void unpack_dwords(const uint2& x, uint32_t (&w)[2]) { w[0] = x.x; w[1] = x.y; }
  static __device__ __forceinline__ void unpack_dwords(const uint4& x, uint32_t (&w)[4])
  { w[0] = x.x; w[1] = x.y; w[2] = x.z; w[3] = x.w; }

  template <typename ACC>
  __device__ __forceinline__ void add_masked(int j, uint32_t m, ACC (&acc)[RPL]) const
  {
    if constexpr (RPL == 1)
    {
      acc[0] += (ACC)((uint32_t)v[j] & m);
    }
    else if constexpr (sizeof(CT) == 4 && RPL == 2)
    {
      acc[0] += v[j].x & m; acc[1] += v[j].y & m;
    }
    else if constexpr (sizeof(CT) == 4)
    {
      acc[0] += v[j].x & m; acc[1] += v[j].y & m; acc[2] += v[j].z & m; acc[3] += v[j].w & m;
    }
    else
    {
      constexpr int per = 4 / sizeof(CT);
      constexpr int ndw = RPL / per;
      constexpr uint32_t mask = sizeof(CT) == 1 ? 0xFFu : 0xFFFFu;
      uint32_t w[ndw];
      unpack_dwords(v[j], w);
#pragma unroll
      for (int d = 0; d < ndw; ++d)
#pragma unroll
        for (int e = 0; e < per; ++e)
          acc[d * per + e] += ((w[d] & m) >> (8 * sizeof(CT) * e)) & mask;
    }
  }

  // all conditions are wave-uniform (scalar branches)
  template <typename ACC>
  __device__ __forceinline__ void consume(int s0, int nc, int S, ACC (&sc)[RPL], ACC (&sk)[RPL]) const
  {
    if (s0 + kBatch <= nc)
    {
#pragma unroll
      for (int j = 0; j < kBatch; ++j) add_one(j, sc);
    }
    else if (s0 >= nc && s0 + kBatch <= S)
    {
#pragma unroll
      for (int j = 0; j < kBatch; ++j) add_one(j, sk);
    }
    else
    {
      // straddling / tail batch: branch-free select with wave-uniform masks (a branchy
      // form makes the compiler address sc/sk through scratch memory)
#pragma unroll
      for (int j = 0; j < kBatch; ++j)
      {
        const int s = s0 + j;
        const uint32_t mc = (s < nc) ? 0xFFFFFFFFu : 0u;
        const uint32_t mk = (s >= nc && s < S) ? 0xFFFFFFFFu : 0u;
        add_masked(j, mc, sc);
        add_masked(j, mk, sk);
      }
    }
  }
};

// first element (sample 0) of a kernel tile: plain SoA tiles are tile_rows apart in every
// column; tiled-layout tiles live inside blocks of T rows that are S*T elements apart
template <typename CT>
__device__ __forceinline__ const CT* tile_base(const filter_params& P, const CT* base, size_t tile,
                                               size_t tile_rows)
{
  if (P.tiles_per_blk == 0) return base + tile * tile_rows;
  const size_t blk = tile / P.tiles_per_blk, sub = tile - blk * P.tiles_per_blk;
  return base + blk * P.blk_stride + sub * tile_rows;
}

template <typename CT, int RPL>
__global__ void __launch_bounds__(kBlock, KMD_MINWAVES) k_filter_soa(const filter_params P)
{
  extern __shared__ double2 s_lf[];
  stage_table(P, s_lf);
  uint32_t n_beyond = 0;       // rows of this lane with a count sum beyond the table
  wave_queue Q;                // rows that passed the pre-filter, evaluated 64 at a time
  {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(s_lf + P.lds_n) + (size_t)(threadIdx.x >> 6) * kQueueCap * 3;
    Q.sc = q; Q.sk = q + kQueueCap; Q.row = q + 2 * kQueueCap; Q.n = 0;
  }

  using ACC = typename acc_of<CT>::type;
  const CT* __restrict__ base = static_cast<const CT*>(P.counts);
  const size_t tile_rows = (size_t)blockDim.x * RPL;
  const size_t n_tiles = (P.n_rows + tile_rows - 1) / tile_rows;
  // tiles whose vector loads are in bounds: all of them for the tiled layout (the buffer is
  // whole blocks), else the ones with every row in range
  const size_t n_full_tiles = P.tiles_per_blk ? n_tiles : P.n_rows / tile_rows;
  const int S = P.nc + P.nk;
  const int n_batches = (S + kBatch - 1) / kBatch;

  if (blockIdx.x == 0 && threadIdx.x == 0)
    atomicAdd(&P.counters[KMD_CNT_TOTAL], (unsigned long long)P.n_rows);    // merge.hpp:76

#ifdef KMD_TIMING
  unsigned long long t_load = 0, t_math = 0, n_tiles_done = 0;
  const unsigned long long t_begin = __builtin_readcyclecounter(), w_begin = wall_clock64();
#endif
  soa_batch<CT, RPL> A, B;
  size_t tile = blockIdx.x;
#if KMD_XPREFETCH
  if (tile < n_full_tiles)
    A.load(tile_base(P, base, tile, tile_rows) + (size_t)threadIdx.x * RPL, P.ld, 0, S);
#endif

  for (; tile < n_tiles; tile += gridDim.x)
  {
    const size_t r0 = tile * tile_rows + (size_t)threadIdx.x * RPL;
    ACC sc[RPL], sk[RPL];
#pragma unroll
    for (int j = 0; j < RPL; ++j) { sc[j] = 0; sk[j] = 0; }

#ifdef KMD_TIMING
    const unsigned long long t0 = __builtin_readcyclecounter();
#endif
    if (tile < n_full_tiles)
    {
      const CT* __restrict__ rows0 = tile_base(P, base, tile, tile_rows) + (size_t)threadIdx.x * RPL;
#if !KMD_XPREFETCH
      A.load(rows0, P.ld, 0, S);
#endif
      for (int b = 0; b < n_batches; b += 2)
      {
        if (b + 1 < n_batches) B.load(rows0, P.ld, (b + 1) * kBatch, S);
        A.consume(b * kBatch, P.nc, S, sc, sk);
        if (b + 2 < n_batches) A.load(rows0, P.ld, (b + 2) * kBatch, S);
        if (b + 1 < n_batches) B.consume((b + 1) * kBatch, P.nc, S, sc, sk);
      }
#if KMD_XPREFETCH
      const size_t next = tile + gridDim.x;
      if (next < n_full_tiles)
        A.load(tile_base(P, base, next, tile_rows) + (size_t)threadIdx.x * RPL, P.ld, 0, S);
#endif
    }
    else
    {
      // the ragged last tile: scalar loads with a bound check per row
#pragma unroll
      for (int j = 0; j < RPL; ++j)
      {
        if (r0 + j < P.n_rows)       // (no break: keeps sc/sk in registers)
        {
          const CT* __restrict__ col = base + r0 + j;
          ACC a = 0, b = 0;
          for (int s = 0; s < P.nc; ++s, col += P.ld) a += (ACC)(*col);
          for (int s = 0; s < P.nk; ++s, col += P.ld) b += (ACC)(*col);
          sc[j] = a; sk[j] = b;
        }
      }
    }

#ifdef KMD_TIMING
    const unsigned long long t1 = __builtin_readcyclecounter();
#endif
    // FP64 phase, one row at a time: a rolled loop that rotates the sums through slot 0
    // keeps one copy of finish_row's code and registers instead of RPL of them
#pragma unroll 1
    for (int j = 0; j < RPL; ++j)
    {
      row_state st;
      st.sum_c = sc[0]; st.sum_k = sk[0];
      st.row = r0 + j;
      st.valid = (r0 + j) < P.n_rows;
#if KMD_DEFER
      defer_row(P, s_lf, st, n_beyond, Q);
#else
      finish_row(P, s_lf, st, n_beyond);
#endif
#pragma unroll
      for (int i = 0; i + 1 < RPL; ++i) { sc[i] = sc[i + 1]; sk[i] = sk[i + 1]; }
    }
#ifdef KMD_TIMING
    const unsigned long long t2 = __builtin_readcyclecounter();
    t_load += t1 - t0; t_math += t2 - t1; ++n_tiles_done;
#endif
  }
  drain_queue(P, s_lf, Q);
  flush_beyond(P, n_beyond);
#ifdef KMD_TIMING
  if (blockIdx.x == 0 && threadIdx.x == 0)
  {
    P.counters[8] = t_load; P.counters[9] = t_math; P.counters[10] = n_tiles_done;
    P.counters[11] = __builtin_readcyclecounter() - t_begin; P.counters[12] = wall_clock64() - w_begin;
  }
#endif
}

// ---- row-major layout: counts[row][sample] -------------------------------------------------
// What km::MatrixReader / KmerMerger hand the observer (merge.hpp:68,194-203): k_filter_rows_wave / _wide (rows whose
// pitch and base are 16-byte aligned), k_filter_rows_flat (any other pitch up to 8 KB) and k_filter_rows_direct (the
// rest: every lane walks its own row) below.  (Rounds 1-2 also kept a workgroup-tile kernel with barriers and dword
// loads, k_filter_rows<CT, VEC>, for what the flat kernel does not take -- 1.1-1.4 TB/s; those cases, a base pointer
// that is not 16-byte aligned or an unaligned pitch over 8 KB, now go to the direct kernel.)
constexpr int kRowsBlock = 256;

// ---- rows given by their two sums (the fused merge, kmd_merge_sums) ---------------------------
// One lane per row: 16 bytes in, straight into the pre-filter and the deferred-evaluation queue.
__global__ void __launch_bounds__(kBlock) k_filter_sums(const filter_params P, const unsigned long long* __restrict__ sum_c,
                                                        const unsigned long long* __restrict__ sum_k)
{
  extern __shared__ double2 s_lf[];
  stage_table(P, s_lf);
  uint32_t n_beyond = 0;
  wave_queue Q;
  {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(s_lf + P.lds_n) + (size_t)(threadIdx.x >> 6) * kQueueCap * 3;
    Q.sc = q; Q.sk = q + kQueueCap; Q.row = q + 2 * kQueueCap; Q.n = 0;
  }
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t n_round = (P.n_rows + stride - 1) / stride * stride;          // whole waves take every step together
  uint32_t n_valid = 0;                                                      // entries that are rows, not holes
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride)
  {
    row_state st;
    st.row = i;
    st.valid = i < P.n_rows;
    st.sum_c = st.valid ? __builtin_nontemporal_load(sum_c + i) : 0ull;
    st.sum_k = st.valid ? __builtin_nontemporal_load(sum_k + i) : 0ull;
    if (st.sum_c == ~0ull) st.valid = false;                                 // a hole kmd_merge_sums left
    n_valid += st.valid ? 1u : 0u;
    defer_row(P, s_lf, st, n_beyond, Q);
  }
  drain_queue(P, s_lf, Q);
  flush_beyond(P, n_beyond);
  for (int o = 32; o > 0; o >>= 1) n_valid += __shfl_down(n_valid, o, 64);
  if ((threadIdx.x & 63) == 0 && n_valid) atomicAdd(&P.counters[KMD_CNT_TOTAL], (unsigned long long)n_valid);   // merge.hpp:76
}

// ---- candidate rows of the fused merge (kmd_tilemerge.hip): a device-resident list of rows that passed the
// pre-filter.  Unlike the rows K1 sees, a good part of these are survivors -- on data where k-mers are rare and
// sample-specific, millions per partition -- and the sink's cursor is ONE address: a returning atomic on it costs
// ~22 ns, one after the other, from wherever on the chip it comes (measured: a wave that parked ~200 survivors in
// LDS per atomic still spent 0.44 ms on 1.7 M survivors -- 17 000 atomics and nothing else; its arithmetic, a
// continued fraction in double precision per row, would fit in 0.15).  So there is no atomic per anything here:
//   k_cand_eval  evaluates the list -- wave g of W takes entries [64 (r W + g), +64), r = 0, 1, ... -- and leaves
//                p_bits[i] per entry (all ones: nothing to emit; else the p-value, its sign bit set if the row is
//                within 1e-8 of the threshold, kmd_eval.h) and four counts per wave;
//   k_cand_scan  (one workgroup) turns the waves' survivor counts into sink offsets -- ONE fetch-add on the sink's
//                cursor per partition -- and adds the totals to the counters;
//   k_cand_emit  walks the list the same way and writes the survivors' records: wave g's land at its offset, in
//                list order.
// `gate` (may be NULL): the launches were enqueued BEHIND the merge that fills the list, before the host knew how
// the merge went -- gate[0..2] = entries, distinct k-mers, rows beyond the table as the merge left them on the
// device, gate_over[0] = tiles it could not finish, gate_over[1] = 1 if the merge ran at all.  If any tile is
// unfinished, the list overflowed or the merge did not run, the kernels do nothing at all (the host then goes the long way and launches them again without a gate).
constexpr int kCandBlock = 256;
constexpr unsigned long long kCandNone = ~0ull;
struct cand_counts { uint32_t surv, cand, near, ctrl; };

__device__ __forceinline__ bool cand_gate(const unsigned long long* __restrict__ gate, const uint32_t* __restrict__ gate_over,
                                          unsigned long long gate_cap, size_t& n_rows)
{
  if (!gate) return true;
  // (gate_over[1]: set by the merge kernel that took the plan's way -- the host may have launched only the one it
  // guessed, and a merge that never ran leaves a list of whatever the scratch held)
  if (gate_over[0] != 0 || gate_over[1] == 0 || gate[0] > gate_cap) return false;
  n_rows = (size_t)gate[0];
  return true;
}

__global__ void __launch_bounds__(kCandBlock) k_cand_eval(const filter_params P, const unsigned long long* __restrict__ sum_c,
                                                          const unsigned long long* __restrict__ sum_k,
                                                          const unsigned long long* __restrict__ gate, const uint32_t* __restrict__ gate_over,
                                                          unsigned long long gate_cap, unsigned long long* __restrict__ p_bits,
                                                          cand_counts* __restrict__ wave_counts)
{
  size_t n_rows = P.n_rows;
  if (!cand_gate(gate, gate_over, gate_cap, n_rows)) return;
  const uint32_t lane = threadIdx.x & 63, g = blockIdx.x * (kCandBlock / 64) + (threadIdx.x >> 6);
  const size_t stride = (size_t)gridDim.x * kCandBlock;
  uint32_t n_surv = 0, n_cand = 0, n_near = 0, n_ctrl = 0;                                // wave-uniform
  // (the next step's sums are on their way while this step's rows are evaluated)
  size_t i = (size_t)g * 64 + lane;
  unsigned long long nx_c = i < n_rows ? sum_c[i] : kCandNone, nx_k = i < n_rows ? sum_k[i] : 0ull;
  for (size_t i0 = (size_t)g * 64; i0 < n_rows; i0 += stride, i += stride)
  {
    row_state st;
    st.row = i;
    st.valid = nx_c != kCandNone;                         // (a hole of the list: the fused merge hands its list out in chunks, kmd_tilemerge.hip; or beyond its end)
    st.sum_c = st.valid ? nx_c : 0ull;
    st.sum_k = st.valid ? nx_k : 0ull;
    {
      const size_t j = i + stride;
      nx_c = j < n_rows ? sum_c[j] : kCandNone; nx_k = j < n_rows ? sum_k[j] : 0ull;
    }
    const row_result R = evaluate_core(P, nullptr, st);
    unsigned long long bits = kCandNone;
    if (R.cand && (R.surv || R.near)) bits = (unsigned long long)__double_as_longlong(R.p) | (R.near ? 1ull << 63 : 0ull);
    if (i < n_rows) p_bits[i] = bits;
    const unsigned long long cm = __ballot(R.cand);
    if (!cm) continue;
    n_cand += (uint32_t)__popcll(cm);
    n_near += (uint32_t)__popcll(__ballot(R.near));
    n_surv += (uint32_t)__popcll(__ballot(R.surv));                                        // merge.hpp:78
    n_ctrl += (uint32_t)__popcll(__ballot(R.surv && R.sign == KMD_SIGN_CONTROL));          // merge.hpp:95-98
  }
  if (lane == 0) { cand_counts c; c.surv = n_surv; c.cand = n_cand; c.near = n_near; c.ctrl = n_ctrl; wave_counts[g] = c; }
}

// one workgroup: wave_off[g] = the sink slot of wave g's first survivor
__global__ void __launch_bounds__(1024) k_cand_scan(const filter_params P, unsigned long long rows_total, unsigned long long rows_beyond,
                                                    const unsigned long long* __restrict__ gate, const uint32_t* __restrict__ gate_over,
                                                    unsigned long long gate_cap, const cand_counts* __restrict__ wave_counts, uint32_t n_waves,
                                                    unsigned long long* __restrict__ wave_off)
{
  size_t n_rows = P.n_rows;
  if (!cand_gate(gate, gate_over, gate_cap, n_rows)) return;
  if (gate) { rows_total = gate[1]; rows_beyond = gate[2]; }
  __shared__ unsigned long long s_wave[16];              // the scan's waves: survivors (then their exclusive prefix)
  __shared__ unsigned long long s_tot[3];
  __shared__ unsigned long long s_base;
  const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t per = (n_waves + 1023u) / 1024u, first = threadIdx.x * per;
  unsigned long long surv = 0, cand = 0, near = 0, ctrl = 0;
  for (uint32_t g = first; g < first + per && g < n_waves; ++g)
  {
    const cand_counts c = wave_counts[g];
    surv += c.surv; cand += c.cand; near += c.near; ctrl += c.ctrl;
  }
  if (threadIdx.x < 3) s_tot[threadIdx.x] = 0;
  unsigned long long incl = surv;                        // inclusive scan within the wave
  for (int o = 1; o < 64; o <<= 1) { const unsigned long long v = __shfl_up(incl, o, 64); if ((int)lane >= o) incl += v; }
  for (int o = 32; o > 0; o >>= 1) { cand += __shfl_down(cand, o, 64); near += __shfl_down(near, o, 64); ctrl += __shfl_down(ctrl, o, 64); }
  if (lane == 63) s_wave[w] = incl;
  __syncthreads();
  if (lane == 0)
  {
    if (cand) atomicAdd(&s_tot[0], cand);
    if (near) atomicAdd(&s_tot[1], near);
    if (ctrl) atomicAdd(&s_tot[2], ctrl);
  }
  if (threadIdx.x == 0)
  {
    unsigned long long total = 0;
    for (int k = 0; k < 16; ++k) { const unsigned long long v = s_wave[k]; s_wave[k] = total; total += v; }
    s_base = total ? atomicAdd(&P.counters[KMD_CNT_SIG], total) : 0ull;                   // merge.hpp:101
    if (rows_total) atomicAdd(&P.counters[KMD_CNT_TOTAL], rows_total);                    // merge.hpp:76
    if (rows_beyond) atomicAdd(&P.counters[KMD_CNT_DEFERRED], rows_beyond);
  }
  __syncthreads();
  unsigned long long at = s_base + s_wave[w] + incl - surv;
  for (uint32_t g = first; g < first + per && g < n_waves; ++g) { wave_off[g] = at; at += wave_counts[g].surv; }
  if (threadIdx.x == 1023)
  {
    const unsigned long long total = s_wave[15] + incl;  // the last thread's inclusive sum closes the scan
    if (s_tot[0]) atomicAdd(&P.counters[KMD_CNT_CANDIDATES], s_tot[0]);
    if (s_tot[1]) atomicAdd(&P.counters[KMD_CNT_NEAR_THRESHOLD], s_tot[1]);
    if (s_tot[2]) atomicAdd(&P.counters[KMD_CNT_SIG_CONTROL], s_tot[2]);
    if (total - s_tot[2]) atomicAdd(&P.counters[KMD_CNT_SIG_CASE], total - s_tot[2]);     // merge.hpp:95-98
  }
}

__global__ void __launch_bounds__(kCandBlock) k_cand_emit(const filter_params P, const unsigned long long* __restrict__ sum_c,
                                                          const unsigned long long* __restrict__ sum_k,
                                                          const unsigned long long* __restrict__ gate, const uint32_t* __restrict__ gate_over,
                                                          unsigned long long gate_cap, const unsigned long long* __restrict__ p_bits,
                                                          const unsigned long long* __restrict__ wave_off)
{
  size_t n_rows = P.n_rows;
  if (!cand_gate(gate, gate_over, gate_cap, n_rows)) return;
  // (the sink's arrays and the list do not overlap -- said here, with __restrict__, because the compiler cannot know)
  const uint64_t* __restrict__ const in_lo = P.kmer_lo;
  const uint64_t* __restrict__ const in_hi = P.kmer_hi;
  uint64_t* __restrict__ const o_row = P.out.d_row;
  uint64_t* __restrict__ const o_lo = P.out.d_kmer_lo;
  uint64_t* __restrict__ const o_hi = P.kmer_hi ? P.out.d_kmer_hi : nullptr;
  double* __restrict__ const o_p = P.out.d_pvalue;
  int32_t* __restrict__ const o_sign = P.out.d_sign;
  double* __restrict__ const o_mc = P.out.d_mean_control;
  double* __restrict__ const o_mk = P.out.d_mean_case;
  const unsigned long long o_cap = P.out.capacity;
  const uint32_t lane = threadIdx.x & 63, g = blockIdx.x * (kCandBlock / 64) + (threadIdx.x >> 6);
  const size_t stride = (size_t)gridDim.x * kCandBlock;
  unsigned long long at = wave_off[g];                                                    // wave-uniform
  size_t i = (size_t)g * 64 + lane;
  unsigned long long nx = i < n_rows ? p_bits[i] : kCandNone;
  for (size_t i0 = (size_t)g * 64; i0 < n_rows; i0 += stride, i += stride)
  {
    const unsigned long long bits = nx;
    { const size_t j = i + stride; nx = j < n_rows ? p_bits[j] : kCandNone; }
    const bool has = bits != kCandNone;
    if (!__ballot(has)) continue;
    const bool near = has && (bits >> 63) != 0;
    const double p = __longlong_as_double((long long)(bits & ~(1ull << 63)));
    const bool surv = has && p <= P.threshold;                                            // merge.hpp:78
    const unsigned long long sm = __ballot(surv);
    const unsigned long long slot = at + (unsigned long long)__popcll(sm & ((1ull << lane) - 1ull));
    at += (unsigned long long)__popcll(sm);
    if (has)
    {
      const unsigned long long sc = sum_c[i], sk = sum_k[i];
      if (surv && slot < o_cap)
      {
        const uint64_t klo = in_lo[i], khi = o_hi ? in_hi[i] : 0ull;
        double mean_control; int sign;
        kmd::sign_of(sc, sk, P.dTc, P.dTk, mean_control, sign);
        if (o_row) o_row[slot] = klo;
        if (o_lo) o_lo[slot] = klo;
        if (o_hi) o_hi[slot] = khi;
        if (o_p) o_p[slot] = p;
        if (o_sign) o_sign[slot] = sign;
        if (o_mc) o_mc[slot] = mean_control;
        if (o_mk) o_mk[slot] = (double)sk;
      }
      if (near)
      {
        row_state st; st.row = i; st.valid = true; st.sum_c = sc; st.sum_k = sk;
        note_near_row(P, st, surv ? (long long)slot : -1ll);
      }
    }
  }
}

// ---- row-major rows, 16-byte aligned pitch: wave-private staging -------------------------
// What a host that hands over kmtricks rows (matrix_proxy, merge.hpp:194-203) delivers.  A wave
// owns 64 consecutive rows at a time.  Their bytes are fetched with 16-byte loads that are
// contiguous across the lanes (lane L takes vector k*64+L of the [64 rows x cv vectors] block),
// parked in a wave-private LDS tile whose row pitch is an ODD number of vectors (16 lanes x
// 16 B of a row-walk then hit 16 different bank groups), and each lane walks its own row.  No
// workgroup barrier; the loads of the next block are issued before the current one is summed.
// CV = vectors of a row handled per pass (bounds the registers of the in-flight block), BLOCK =
// threads per workgroup (its waves share one copy of the table head); picked by row width in
// launch_rows: a single pass per row and as many waves as registers and LDS allow.
template <typename CT, int kRowsCV, int kRowsWaveBlock, bool kDefer>
__global__ void __launch_bounds__(kRowsWaveBlock) k_filter_rows_wave(const filter_params P, const uint32_t row_vecs,
                                                                     const uint32_t n_chunks, const uint32_t cvb)
{
  extern __shared__ double2 s_all[];
  double2* s_lf = s_all;
  stage_table(P, s_lf);
  uint32_t n_beyond = 0;
  typedef uint32_t n4 __attribute__((ext_vector_type(4)));
  constexpr uint32_t per = 4 / sizeof(CT);              // counts per dword
  constexpr uint32_t epv = 4 * per;                     // counts per 16-byte vector
  constexpr uint32_t emask = sizeof(CT) == 1 ? 0xFFu : sizeof(CT) == 2 ? 0xFFFFu : 0xFFFFFFFFu;
  using ACC = typename acc_of<CT>::type;
  const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t S = (uint32_t)(P.nc + P.nk), nc = (uint32_t)P.nc;
  const uint32_t pitch_max = cvb | 1u;                  // cvb = vectors per pass (<= kRowsCV)
  // LDS: table head | (kDefer) one deferred-evaluation queue per wave | one tile per wave
  wave_queue Q;
  {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(s_all + P.lds_n) + (size_t)w * kQueueCap * 3;
    Q.sc = q; Q.sk = q + kQueueCap; Q.row = q + 2 * kQueueCap; Q.n = 0;
  }
  n4* tile = reinterpret_cast<n4*>(reinterpret_cast<char*>(s_all + P.lds_n) +
                                   (kDefer ? (size_t)(kRowsWaveBlock / 64) * kQueueBytesPerWave : 0)) + (size_t)w * 64 * pitch_max;
  const size_t ld_vecs = P.ld * sizeof(CT) / 16;        // row pitch in vectors (exact)
  const n4* __restrict__ base = static_cast<const n4*>(P.counts);
  const size_t n_tiles = (P.n_rows + 63) / 64;
  const size_t n_waves = (size_t)gridDim.x * (kRowsWaveBlock / 64);

  if (blockIdx.x == 0 && threadIdx.x == 0)
    atomicAdd(&P.counters[KMD_CNT_TOTAL], (unsigned long long)P.n_rows);

  n4 buf[kRowsCV];
  // loads of chunk `ch` of the 64 rows starting at row0 (rows past the end re-read the last row)
  auto issue = [&](size_t row0, uint32_t ch)
  {
    const uint32_t c0 = ch * cvb;
    const uint32_t cv = (row_vecs - c0) < cvb ? (row_vecs - c0) : cvb;
    const uint32_t q = 64u / cv, rem = 64u % cv;        // (row, c) of vector v+64 from those of v
    uint32_t r = lane / cv, c = lane - r * cv;
    // The passes of 16 vectors (rows of 11 .. 15 vectors: 24v24 four-byte counts) issue every load every time -- a k past
    // the chunk's vectors re-reads the block's last vector, the same cache line as its neighbours' --: a load under a
    // condition makes the number in flight unknown to hipcc where the paths meet, and it waits for all of them and
    // branches around the loads (round 4, found on k_filter_rows_flat: +28 % there; here 24v24 5.18 -> 5.56 TB/s).  The
    // narrower instantiations, whose rows fill their passes exactly, keep the conditions: unconditional they measured
    // 20v20 6.05 -> 5.75, 4v4 6.06 -> 5.97 (the selects cost, nothing was being skipped).
    constexpr bool kUncond = kRowsCV >= 16;
#pragma unroll
    for (int k = 0; k < kRowsCV; ++k)
    {
      const bool in = (uint32_t)k < cv;                 // wave-uniform
      if (kUncond || in)
      {
        size_t row = row0 + ((!kUncond || in) ? r : 63u);
        if (row >= P.n_rows) row = P.n_rows - 1;
        buf[k] = __builtin_nontemporal_load(base + row * ld_vecs + c0 + ((!kUncond || in) ? c : cv - 1u));
        r += q; c += rem;
        if (c >= cv) { c -= cv; ++r; }
      }
    }
  };

  size_t t = (size_t)blockIdx.x * (kRowsWaveBlock / 64) + w;
  if (t < n_tiles) issue(t * 64, 0);
  for (; t < n_tiles; t += n_waves)
  {
    const size_t row0 = t * 64;
    ACC sc = 0, sk = 0;
    for (uint32_t ch = 0; ch < n_chunks; ++ch)
    {
      const uint32_t c0 = ch * cvb;
      const uint32_t cv = (row_vecs - c0) < cvb ? (row_vecs - c0) : cvb;
      const uint32_t pitch = cv | 1u;
      {
        const uint32_t q = 64u / cv, rem = 64u % cv;
        uint32_t r = lane / cv, c = lane - r * cv;
#pragma unroll
        for (int k = 0; k < kRowsCV; ++k)
          if ((uint32_t)k < cv)
          {
            tile[r * pitch + c] = buf[k];
            r += q; c += rem;
            if (c >= cv) { c -= cv; ++r; }
          }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // next block of loads in flight while this one is added up (the wave's last block: itself once more, unused)
      if constexpr (kRowsCV >= 16)
      {
        const bool more_chunks = ch + 1 < n_chunks, more_tiles = t + n_waves < n_tiles;
        issue(more_chunks ? row0 : more_tiles ? (t + n_waves) * 64 : row0, more_chunks ? ch + 1 : 0u);
      }
      else
      {
        if (ch + 1 < n_chunks) issue(row0, ch + 1);
        else if (t + n_waves < n_tiles) issue((t + n_waves) * 64, 0);
      }
      const n4* __restrict__ mine = tile + lane * pitch;
      for (uint32_t c = 0; c < cv; ++c)
      {
        const n4 v = mine[c];
        const uint32_t e0 = (c0 + c) * epv;             // first count of this vector (wave-uniform)
        if constexpr (per == 1)
        {
          if (e0 + 4 <= nc) sc += (ACC)v.x + v.y + v.z + v.w;
          else if (e0 >= nc && e0 + 4 <= S) sk += (ACC)v.x + v.y + v.z + v.w;
          else
          {
            const uint32_t d[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (uint32_t e = 0; e < 4; ++e)
              if (e0 + e < nc) sc += d[e]; else if (e0 + e < S) sk += d[e];
          }
        }
        else
        {
          const uint32_t d[4] = { v.x, v.y, v.z, v.w };
          const bool all_c = e0 + epv <= nc, all_k = e0 >= nc && e0 + epv <= S;
          if (all_c)
          {
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) sc = add_packed<CT>(d[j], sc);
          }
          else if (all_k)
          {
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) sk = add_packed<CT>(d[j], sk);
          }
          else
          {
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j)
#pragma unroll
              for (uint32_t e = 0; e < per; ++e)
              {
                const uint32_t x = (d[j] >> (8 * sizeof(CT) * e)) & emask;
                const uint32_t el = e0 + j * per + e;
                if (el < nc) sc += x;
                else if (el < S) sk += x;
              }
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // tile consumed before it is overwritten
      __builtin_amdgcn_wave_barrier();
    }
    row_state st;
    st.row = row0 + lane;
    st.valid = st.row < P.n_rows;
    st.sum_c = sc; st.sum_k = sk;
    if constexpr (kDefer) defer_row(P, s_lf, st, n_beyond, Q);
    else finish_row(P, s_lf, st, n_beyond);
  }
  if constexpr (kDefer) drain_queue(P, s_lf, Q);
  flush_beyond(P, n_beyond);
}

// ---- row-major rows, ANY pitch: wave-private flat staging --------------------------------------
// A matrix with 21 + 21 four-byte counts has 168-byte rows: no 16-byte vector of a row is
// aligned, yet R consecutive rows are one contiguous span that starts on a 16-byte boundary
// whenever R x pitch is a multiple of 16 (R = 64 always is; the buffer itself is aligned).  A
// wave copies such a span into its own LDS tile with 16-byte non-temporal loads that are
// contiguous across the lanes -- the bytes land where they were, no re-pitching -- and then
// G = 64 / R lanes share a row: lane (r, g) adds the counts g, g + G, ... of row r with typed
// LDS reads, a butterfly over the G lanes yields the row's two sums, and the R row lanes go
// through the pre-filter and the deferred-evaluation queue.  R is the largest of 64, 32, ... 2
// whose span fits the tile and keeps the alignment.  The loads of the next span are in flight
// while the current one is summed.  The matrix's last vector may be cut by the end of the
// buffer: it is fetched count by count.
// kFlatVecs = 16-byte vectors per lane per span, i.e. the tile is kFlatVecs KB per wave: 4 KB tiles x 16
// waves per CU for ordinary rows (measured: 8 KB x 12 waves 4.3 TB/s, 4 KB x 16 waves 4.6 TB/s at
// 21v21); 8 and 16 KB tiles (12 / 8 waves) only for rows so wide that no aligned group of rows fits
// the smaller tile (501 four-byte counts: 4 rows = 8016 B).
template <typename CT, int kFlatVecs, int kFlatBlock>
__global__ void __launch_bounds__(kFlatBlock) k_filter_rows_flat(const filter_params P, const uint32_t R)
{
  extern __shared__ double2 s_all[];
  double2* s_lf = s_all;
  stage_table(P, s_lf);
  uint32_t n_beyond = 0;
  typedef uint32_t n4 __attribute__((ext_vector_type(4)));
  using ACC = typename acc_of<CT>::type;
  const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t S = (uint32_t)(P.nc + P.nk), nc = (uint32_t)P.nc;
  wave_queue Q;
  {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(s_all + P.lds_n) + (size_t)w * kQueueCap * 3;
    Q.sc = q; Q.sk = q + kQueueCap; Q.row = q + 2 * kQueueCap; Q.n = 0;
  }
  n4* tile = reinterpret_cast<n4*>(reinterpret_cast<char*>(s_all + P.lds_n) + (size_t)(kFlatBlock / 64) * kQueueBytesPerWave) +
             (size_t)w * 64 * kFlatVecs;
  const CT* tile_ct = reinterpret_cast<const CT*>(tile);
  const size_t pitch = P.ld * sizeof(CT);               // bytes
  const size_t total_vecs = (P.n_rows * pitch) / 16;    // whole vectors inside the buffer
  const uint32_t span_vecs = (uint32_t)(R * pitch / 16);
  const n4* __restrict__ base = static_cast<const n4*>(P.counts);
  const CT* __restrict__ base_ct = static_cast<const CT*>(P.counts);
  const size_t n_tiles = (P.n_rows + R - 1) / R;
  const size_t n_waves = (size_t)gridDim.x * (kFlatBlock / 64);
  const uint32_t G = 64u / R, r = lane & (R - 1), g = lane / R;

  if (blockIdx.x == 0 && threadIdx.x == 0)
    atomicAdd(&P.counters[KMD_CNT_TOTAL], (unsigned long long)P.n_rows);

  // Round 4.  This kernel ran at 4.7 TB/s on 21v21 with its waves waiting 70 % of their cycles (tools/pmc_k1.sh).  What
  // it was waiting for: every load sat under a condition (lanes past the span, spans past the buffer), so hipcc could not
  // count the loads in flight where the paths meet and settled for vmcnt(0) -- and skipped around the loads with branches.
  // Now EVERY lane loads EVERY time (a lane past the span or the buffer re-reads the last vector inside both: the same
  // cache line as its neighbours'): 21v21 4.74 -> 6.09 TB/s, 3v3 4.8 -> 6.8, u8 20v20 4.1 -> 5.3, 60v61 4.6 -> 5.7
  // (profiles/r04_ab_k1r.txt).  KMD_FLAT_DEPTH register sets hold that many spans per wave in flight: a second set
  // measured 3-6 % SLOWER than one (5.75 / 6.65 / 5.11 / 5.58: the registers cost more than the deeper queue gives; with
  // conditional loads it had bought nothing at all), a third spills (4.4 TB/s).  One set, as before.
#ifndef KMD_FLAT_DEPTH
#define KMD_FLAT_DEPTH 1
#endif
  constexpr int kDepth = KMD_FLAT_DEPTH;
  n4 bufs[kDepth][kFlatVecs];
  auto issue = [&](n4 (&buf)[kFlatVecs], size_t t)
  {
    // (every lane loads, every time: a lane past the span or the buffer re-reads the last vector inside both -- the
    // same cache line as its neighbours'.  A load under a condition makes the number of loads in flight unknown to
    // the compiler where the paths meet, and it then waits for ALL of them -- the other set's just-issued span
    // included: measured, the second set bought nothing until the loads were unconditional)
    const size_t v0 = t * span_vecs;
#pragma unroll
    for (int k = 0; k < kFlatVecs; ++k)
    {
      uint32_t i = (uint32_t)k * 64 + lane;
      i = i < span_vecs ? i : span_vecs - 1u;
      size_t v = v0 + i;
      v = v < total_vecs ? v : total_vecs - 1;
      buf[k] = __builtin_nontemporal_load(base + v);
    }
  };
  auto consume = [&](n4 (&buf)[kFlatVecs], size_t t)
  {
    const size_t v0 = t * span_vecs;
#pragma unroll
    for (int k = 0; k < kFlatVecs; ++k)
    {
      const uint32_t i = (uint32_t)k * 64 + lane;
      if ((uint32_t)k * 64 < span_vecs && i < span_vecs)
      {
        if (v0 + i < total_vecs) tile[i] = buf[k];
        else
        {
          // the vector the end of the buffer cuts (at most one in the matrix): count by count
          const size_t e0 = (v0 + i) * (16 / sizeof(CT)), e_end = P.n_rows * P.ld;
          CT* d = reinterpret_cast<CT*>(tile + i);
          for (uint32_t j = 0; j < 16 / sizeof(CT); ++j) d[j] = e0 + j < e_end ? base_ct[e0 + j] : (CT)0;
        }
      }
    }
    queue_fence();
    issue(buf, t + kDepth * n_waves < n_tiles ? t + kDepth * n_waves : t);      // this set's next span in flight while the tile is added up (past the end: this span again, unused)
    const CT* __restrict__ mine = tile_ct + (size_t)r * P.ld;
    ACC sc = 0, sk = 0;
    // (one- and two-byte counts read as dwords with one v_sad per dword -- a quarter / half of the LDS reads -- were measured
    // in round 4: u8 20v20 +-0 (5.17 TB/s: at 1.08e11 rows/s the per-row pre-filter is the cost, K1 tiled reaches 1.2e11),
    // u16 21v21 5.9 -> 5.1; the count-by-count walk stays)
    // eight LDS reads in flight per lane (the walk is latency-bound otherwise)
    uint32_t e = g;
    for (; e + 7 * G < nc; e += 8 * G)
    {
      CT x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = mine[e + j * G];
#pragma unroll
      for (int j = 0; j < 8; ++j) sc += x[j];
    }
    for (; e < nc; e += G) sc += mine[e];
    // the case lanes start at the first count >= nc that is theirs
    e = nc + ((g + G - nc % G) % G);
    for (; e + 7 * G < S; e += 8 * G)
    {
      CT x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = mine[e + j * G];
#pragma unroll
      for (int j = 0; j < 8; ++j) sk += x[j];
    }
    for (; e < S; e += G) sk += mine[e];
    for (uint32_t o = R; o < 64; o <<= 1)
    {
      sc += (ACC)__shfl_xor((unsigned long long)sc, (int)o, 64);
      sk += (ACC)__shfl_xor((unsigned long long)sk, (int)o, 64);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // tile read before it is overwritten
    __builtin_amdgcn_wave_barrier();
    row_state st;
    st.row = t * R + r;
    st.valid = g == 0 && st.row < P.n_rows;
    st.sum_c = sc; st.sum_k = sk;
    defer_row(P, s_lf, st, n_beyond, Q);
  };

  size_t t = (size_t)blockIdx.x * (kFlatBlock / 64) + w;
  if (t < n_tiles)
  {
#pragma unroll
    for (int d = 0; d < kDepth; ++d) issue(bufs[d], t + d * n_waves < n_tiles ? t + d * n_waves : t);
  }
  for (; t < n_tiles; t += kDepth * n_waves)
  {
#pragma unroll
    for (int d = 0; d < kDepth; ++d)
      if (t + d * n_waves < n_tiles) consume(bufs[d], t + d * n_waves);
  }
  drain_queue(P, s_lf, Q);
  flush_beyond(P, n_beyond);
}

// ---- row-major WIDE rows: G = 8 or 16 lanes per row -------------------------------------------
// A row of hundreds of bytes needs no transposition through LDS: G consecutive lanes read 16 G
// consecutive bytes of ONE row per load (one or two whole cache lines), 64 / G rows per wave
// instruction, and add up what they read; a G-lane butterfly then yields the row's two sums.  G
// such steps fill a 64-entry LDS strip with the sums of 64 rows, which go through the pre-filter and
// the deferred-evaluation queue with every lane busy.  No tile, 16 waves per CU, many independent
// loads in flight per wave (kWideU steps are issued at once).
#ifndef KMD_WIDE_BLOCK
#define KMD_WIDE_BLOCK 1024
#endif
#ifndef KMD_WIDE8_MIN
#define KMD_WIDE8_MIN 16        // rows of at least this many vectors take 8 lanes per row
#endif
#ifndef KMD_WIDE_U
#define KMD_WIDE_U 2
#endif

constexpr int kWideBlock = KMD_WIDE_BLOCK;
constexpr int kWideU = KMD_WIDE_U;            // steps (of 4 rows) whose loads are issued together
constexpr int kWideP = 4;                     // passes (of 16 vectors) of a row per chunk

template <typename CT, int G>
__global__ void __launch_bounds__(kWideBlock) k_filter_rows_wide(const filter_params P, const uint32_t row_vecs)
{
  extern __shared__ double2 s_all[];
  double2* s_lf = s_all;
  stage_table(P, s_lf);
  uint32_t n_beyond = 0;
  typedef uint32_t n4 __attribute__((ext_vector_type(4)));
  constexpr uint32_t per = 4 / sizeof(CT), epv = 4 * per;
  constexpr uint32_t emask = sizeof(CT) == 1 ? 0xFFu : sizeof(CT) == 2 ? 0xFFFFu : 0xFFFFFFFFu;
  using ACC = typename acc_of<CT>::type;
  const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr uint32_t kRPS = 64 / G;                      // rows per step
  const uint32_t g = lane / G, q = lane % G;             // row within a step, vector within a pass
  const uint32_t S = (uint32_t)(P.nc + P.nk), nc = (uint32_t)P.nc;
  wave_queue Q;
  {
    unsigned long long* qb = reinterpret_cast<unsigned long long*>(s_all + P.lds_n) + (size_t)w * kQueueCap * 3;
    Q.sc = qb; Q.sk = qb + kQueueCap; Q.row = qb + 2 * kQueueCap; Q.n = 0;
  }
  // the 64 rows' sums of one wave iteration: [64] x {sum_c, sum_k}
  unsigned long long* strip = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(s_all + P.lds_n) +
                                                                     (size_t)(kWideBlock / 64) * kQueueBytesPerWave) + (size_t)w * 128;
  const size_t ld_vecs = P.ld * sizeof(CT) / 16;
  const n4* __restrict__ base = static_cast<const n4*>(P.counts);
  const size_t n_tiles = (P.n_rows + 63) / 64;
  const size_t n_waves = (size_t)gridDim.x * (kWideBlock / 64);
  const uint32_t n_pass = (row_vecs + G - 1) / G;

  if (blockIdx.x == 0 && threadIdx.x == 0)
    atomicAdd(&P.counters[KMD_CNT_TOTAL], (unsigned long long)P.n_rows);

  for (size_t t = (size_t)blockIdx.x * (kWideBlock / 64) + w; t < n_tiles; t += n_waves)
  {
    const size_t row0 = t * 64;
    for (uint32_t step0 = 0; step0 < (uint32_t)G; step0 += kWideU)
    {
      ACC sc[kWideU], sk[kWideU];
#pragma unroll
      for (int u = 0; u < kWideU; ++u) { sc[u] = 0; sk[u] = 0; }
      for (uint32_t p0 = 0; p0 < n_pass; p0 += kWideP)
      {
        n4 buf[kWideU][kWideP];
#pragma unroll
        for (int u = 0; u < kWideU; ++u)
        {
          size_t row = row0 + (size_t)(step0 + u) * kRPS + g;
          if (row >= P.n_rows) row = P.n_rows - 1;       // re-read the last row (never used)
          const n4* __restrict__ rp = base + row * ld_vecs;
#pragma unroll
          for (int p = 0; p < kWideP; ++p)
          {
            const uint32_t c = (p0 + p) * G + q;
            // (these loads stay under their condition: issued for every lane -- a lane past the row's vectors re-reading
            // its last one -- the kernel lost 8-25 %: 50v50 5.26 -> 4.46 TB/s, 34v34 5.39 -> 4.06, 100v100 6.35 -> 5.8;
            // round 4, the opposite of what the same change did for k_filter_rows_flat)
            buf[u][p] = n4{ 0, 0, 0, 0 };
            if (c < row_vecs) buf[u][p] = __builtin_nontemporal_load(rp + c);
          }
        }
#pragma unroll
        for (int p = 0; p < kWideP; ++p)
        {
          const uint32_t c = (p0 + p) * G + q;
          const uint32_t e0 = c * epv;                   // first count of this lane's vector
          // whole pass on one side of the control / case boundary: wave-uniform fast path
          const uint32_t pe0 = (p0 + p) * G * epv, pe1 = pe0 + G * epv;
          const bool all_c = pe1 <= nc, all_k = pe0 >= nc && pe1 <= S;
#pragma unroll
          for (int u = 0; u < kWideU; ++u)
          {
            const uint32_t d[4] = { buf[u][p].x, buf[u][p].y, buf[u][p].z, buf[u][p].w };
            if constexpr (per == 1)
            {
              if (all_c) sc[u] += (ACC)d[0] + d[1] + d[2] + d[3];
              else if (all_k) sk[u] += (ACC)d[0] + d[1] + d[2] + d[3];
              else
              {
#pragma unroll
                for (uint32_t e = 0; e < 4; ++e)
                {
                  const uint32_t el = e0 + e;
                  sc[u] += el < nc ? d[e] : 0u;
                  sk[u] += (el >= nc && el < S) ? d[e] : 0u;
                }
              }
            }
            else
            {
              if (all_c)
              {
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) sc[u] = add_packed<CT>(d[j], sc[u]);
              }
              else if (all_k)
              {
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) sk[u] = add_packed<CT>(d[j], sk[u]);
              }
              else
              {
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j)
#pragma unroll
                  for (uint32_t e = 0; e < per; ++e)
                  {
                    const uint32_t x = (d[j] >> (8 * sizeof(CT) * e)) & emask;
                    const uint32_t el = e0 + j * per + e;
                    sc[u] += el < nc ? x : 0u; sk[u] += (el >= nc && el < S) ? x : 0u;
                  }
              }
            }
          }
        }
      }
      // G-lane butterflies: every lane of a row's group ends with the row's sums
#pragma unroll
      for (int u = 0; u < kWideU; ++u)
      {
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1)
        {
          sc[u] += (ACC)__shfl_xor((unsigned long long)sc[u], o, 64);
          sk[u] += (ACC)__shfl_xor((unsigned long long)sk[u], o, 64);
        }
        if (q == 0)
        {
          const uint32_t slot = (step0 + (uint32_t)u) * kRPS + g;
          strip[2 * slot] = (unsigned long long)sc[u];
          strip[2 * slot + 1] = (unsigned long long)sk[u];
        }
      }
    }
    queue_fence();
    row_state st;
    st.row = row0 + lane;
    st.valid = st.row < P.n_rows;
    st.sum_c = strip[2 * lane]; st.sum_k = strip[2 * lane + 1];
    queue_fence();                                       // read before the next iteration's writes
    defer_row(P, s_lf, st, n_beyond, Q);
  }
  drain_queue(P, s_lf, Q);
  flush_beyond(P, n_beyond);
}

// Row-major rows whose pitch is not a whole number of dwords: each lane walks its own row
// straight from global memory (correct for any pitch/alignment; not a tuned path).
template <typename CT>
__global__ void __launch_bounds__(kRowsBlock) k_filter_rows_direct(const filter_params P)
{
  extern __shared__ double2 s_lf[];
  stage_table(P, s_lf);
  uint32_t n_beyond = 0;       // rows of this lane with a count sum beyond the table
  const CT* __restrict__ base = static_cast<const CT*>(P.counts);
  const size_t n_tiles = (P.n_rows + kRowsBlock - 1) / kRowsBlock;
  if (blockIdx.x == 0 && threadIdx.x == 0)
    atomicAdd(&P.counters[KMD_CNT_TOTAL], (unsigned long long)P.n_rows);
  for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x)
  {
    row_state st;
    st.row = tile * kRowsBlock + threadIdx.x;
    st.valid = st.row < P.n_rows;
    uint64_t sc = 0, sk = 0;
    if (st.valid)
    {
      const CT* __restrict__ r = base + st.row * P.ld;
      for (int s = 0; s < P.nc; ++s) sc += r[s];
      for (int s = 0; s < P.nk; ++s) sk += r[P.nc + s];
    }
    st.sum_c = sc; st.sum_k = sk;
    finish_row(P, s_lf, st, n_beyond);
  }
  flush_beyond(P, n_beyond);
}

// ---- every row's result, no threshold: IModel::process over a tile -----------------------
template <typename CT>
__global__ void __launch_bounds__(256) k_process_all(const filter_params P, int layout,
                                                     double* __restrict__ o_p, int32_t* __restrict__ o_sign,
                                                     double* __restrict__ o_mc, double* __restrict__ o_mk)
{
  const CT* __restrict__ base = static_cast<const CT*>(P.counts);
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  // whole waves stay in the loop together (the table fallback is wave-cooperative)
  const size_t n_round = (P.n_rows + 63) / 64 * 64;
  for (size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x; row < n_round; row += stride)
  {
    const bool valid = row < P.n_rows;
    uint64_t sc = 0, sk = 0;
    if (valid)
    {
      const int S = P.nc + P.nk;
      for (int s = 0; s < P.nc; ++s) sc += base[kmd::count_index(layout, P.ld, S, row, s)];
      for (int s = P.nc; s < S; ++s) sk += base[kmd::count_index(layout, P.ld, S, row, s)];
    }
    const uint32_t kc = kmd::table_index(sc), kk = kmd::table_index(sk);
    double lf_c = 0, lf_k = 0;
    if (kc < P.lf_n) lf_c = P.lf[kc];
    if (kk < P.lf_n) lf_k = P.lf[kk];
    if (valid && kc >= P.lf_n) lf_c = lf_beyond_table(kc);
    if (valid && kk >= P.lf_n) lf_k = lf_beyond_table(kk);
    if (valid)
    {
      const kmd::lrt_result r = kmd::lrt_from_sums(sc, sk, lf_c, lf_k, P.dT, P.dTc, P.dTk);
      if (o_p) o_p[row] = kmd::igamc_half(r.lr, P.lg_half);
      if (o_sign) o_sign[row] = r.sign;
      if (o_mc) o_mc[row] = r.mean_control;
      if (o_mk) o_mk[row] = (double)sk;
    }
  }
}

} // namespace

int kmd::fill_filter_params(filter_params& P, const kmd_model* m, const kmd_tile* t, double threshold)
{
  KMD_REQUIRE(m && t, "kmd: NULL model or tile");
  KMD_REQUIRE(t->count_bytes == 1 || t->count_bytes == 2 || t->count_bytes == 4,
              "kmd: count_bytes must be 1, 2 or 4");
  KMD_REQUIRE(kmd::layout_ok(t->layout), "kmd: bad layout");
  KMD_REQUIRE(t->n_rows == 0 || t->d_counts, "kmd: NULL count matrix");
  KMD_REQUIRE(m->nc + m->nk <= 65535, "kmd: more than 65535 samples");
  if (t->layout == KMD_LAYOUT_SOA) KMD_REQUIRE(t->ld >= t->n_rows, "kmd: SoA ld < n_rows");
  else if (t->layout == KMD_LAYOUT_ROWS) KMD_REQUIRE(t->ld >= (size_t)(m->nc + m->nk), "kmd: row-major ld < nc+nk");
  else KMD_REQUIRE(t->ld > 0 && t->ld % 4096 == 0, "kmd: tiled layout needs ld (rows per block) % 4096 == 0");
  P.tiles_per_blk = 0; P.blk_stride = 0;
  P.counts = t->d_counts; P.ld = t->ld; P.n_rows = t->n_rows; P.row_base = t->row_base;
  P.kmer_lo = t->d_kmer_lo; P.kmer_hi = t->d_kmer_hi;
  P.nc = m->nc; P.nk = m->nk;
  P.dT = m->dT; P.dTc = m->dTc; P.dTk = m->dTk; P.lg_half = m->lg_half;
  P.threshold = threshold;
  // the cut only depends on the threshold: cache the last one (hot loop calls reuse it)
  {
    std::lock_guard<std::mutex> lock(m->cut_mu);
    if (!(m->cut_valid && m->cut_threshold_bits == kmd::bits_of(threshold)))
    {
      m->cut_value = kmd::lr_cut_for_threshold(threshold, m->lg_half);
      m->cut_threshold_bits = kmd::bits_of(threshold);
      m->cut_valid = true;
    }
    P.lr_cut = m->cut_value;
  }
  P.dTcTk = m->dTc * m->dTk;
  // pre-filter (finish_row): on when the rounding of the bound is far inside its factor-2
  // slack: cut not tiny, totals of comparable size, totals large enough that count sums up to
  // 2^40 keep the bound's absolute error (~2 eps n max(Tc,Tk)/min(Tc,Tk)) below 1e-2 * cut
  {
    const double ratio = m->dTc > m->dTk ? m->dTc / m->dTk : m->dTk / m->dTc;
    const bool on = P.lr_cut > 1e-2 && P.lr_cut < INFINITY && ratio <= 16.0 && m->dTc > 0 && m->dTk > 0;
    P.pf_cut = on ? 0.5 * P.lr_cut : -INFINITY;
    if (const char* e = std::getenv("KMD_PREFILTER")) if (e[0] == '0') P.pf_cut = -INFINITY;
  }
  P.lf = m->d_lf; P.tab = reinterpret_cast<const double2*>(m->d_tab); P.log_int = m->d_log_int; P.lf_n = (uint32_t)m->lf_n;
  P.lds_n = 0;
  P.counters = nullptr;
  P.out = kmd_survivors{};
  P.near = nullptr;
  return KMD_OK;
}

namespace {

template <typename K> int allow_big_lds(K kernel, size_t lds_bytes);

template <typename CT, int RPL>
int launch_soa_rpl(const filter_params& P, const kmd_model* m, size_t lds_bytes, int blocks_per_cu,
                   bool tiled, hipStream_t stream)
{
  const size_t tile_rows = (size_t)kBlock * RPL;
  filter_params Q = P;
  if (tiled)
  {
    Q.tiles_per_blk = P.ld / tile_rows;               // ld = T, a multiple of tile_rows (checked by the caller)
    Q.blk_stride = (size_t)(P.nc + P.nk) * P.ld;
  }
  size_t n_tiles = (P.n_rows + tile_rows - 1) / tile_rows;
  size_t grid = (size_t)m->n_cu * blocks_per_cu;
  if (grid > n_tiles) grid = n_tiles;
  if (grid == 0) grid = 1;
  int rc = allow_big_lds(k_filter_soa<CT, RPL>, lds_bytes);
  if (rc != KMD_OK) return rc;
  hipLaunchKernelGGL((k_filter_soa<CT, RPL>), dim3((unsigned)grid), dim3(kBlock), lds_bytes, stream, Q);
  KMD_HIP(hipGetLastError());
  return KMD_OK;
}

// rows per lane: the widest vector (8 bytes per lane; KMD_RPL_*) the alignment allows and,
// for the tiled layout, that keeps a kernel tile (kBlock * RPL rows) inside one T-row block
template <typename CT>
int launch_soa(const filter_params& P, const kmd_model* m, size_t lds_bytes, int blocks_per_cu,
               bool tiled, hipStream_t stream)
{
  constexpr int vec = sizeof(CT) == 4 ? KMD_RPL_U32 : sizeof(CT) == 2 ? KMD_RPL_U16 : KMD_RPL_U8;
  constexpr int half = vec >= 2 ? vec / 2 : 1;        // 4 bytes per lane
  auto ok = [&](int rpl) {
    const size_t vbytes = (size_t)rpl * sizeof(CT);
    if ((reinterpret_cast<uintptr_t>(P.counts) % vbytes) || ((P.ld * sizeof(CT)) % vbytes)) return false;
    return !tiled || (P.ld % ((size_t)kBlock * rpl) == 0);
  };
  if (ok(vec)) return launch_soa_rpl<CT, vec>(P, m, lds_bytes, blocks_per_cu, tiled, stream);
  if (half != vec && half != 1 && ok(half)) return launch_soa_rpl<CT, half>(P, m, lds_bytes, blocks_per_cu, tiled, stream);
  if (tiled) KMD_REQUIRE(P.ld % (size_t)kBlock == 0, "kmd: tiled block rows must be a multiple of the workgroup size");
  return launch_soa_rpl<CT, 1>(P, m, lds_bytes, blocks_per_cu, tiled, stream);
}

template <typename K>
int allow_big_lds(K kernel, size_t lds_bytes)
{
  // one hipFuncSetAttribute per kernel, device and size class, not per launch (kernels of
  // different instantiations share the function TYPE, so the cache is keyed by address; the
  // attribute belongs to the device the call is made on)
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

template <typename CT>
int launch_rows(filter_params& P, const kmd_model* m, hipStream_t stream)
{
  const uint32_t S = (uint32_t)(P.nc + P.nk);
  const size_t half = m->lds_per_block_max / 2 - 256;
  const size_t n_tiles = (P.n_rows + kRowsBlock - 1) / kRowsBlock;
  // pitch not a multiple of 16 bytes (21v21 four-byte counts: 168 B): the flat wave-private kernel,
  // with the most rows per span (64, 32, ... 2) that fit its tile (4 KB; 8 or 16 KB for very wide rows)
  // and keep spans 16-byte aligned
  if ((((P.ld * sizeof(CT)) % 16 != 0) || std::getenv("KMD_ROWS_FLAT_ALL")) && ((reinterpret_cast<uintptr_t>(P.counts) & 15u) == 0) &&
      P.n_rows * P.ld * sizeof(CT) >= 16 &&              // (its loads are unconditional: there must be one whole vector to read)
      std::getenv("KMD_ROWS_FLAT_OFF") == nullptr)
  {
    const size_t pitch = P.ld * sizeof(CT);
    auto rows_per_span = [&](size_t tile_bytes) -> uint32_t
    {
      uint32_t R = 64;
      while (R >= 2 && (R * pitch > tile_bytes || (R * pitch) % 16 != 0)) R >>= 1;
      return R >= 2 ? R : 0;
    };
    auto launch = [&](auto kernel, int vecs, int block, uint32_t R) -> int
    {
      const size_t wpb = (size_t)block / 64;
      const size_t extra = wpb * kQueueBytesPerWave + wpb * 64 * (size_t)vecs * 16;
      const size_t avail = m->lds_per_block_max - 256 - extra;
      size_t want = (size_t)P.lf_n * sizeof(double2);
      if (want > avail) want = avail / sizeof(double2) * sizeof(double2);
      P.lds_n = (uint32_t)(want / sizeof(double2));
      const size_t n_wtiles = (P.n_rows + R - 1) / R;
      size_t grid = (size_t)m->n_cu;
      if (grid > (n_wtiles + wpb - 1) / wpb) grid = (n_wtiles + wpb - 1) / wpb;
      int rc = allow_big_lds(kernel, want + extra);
      if (rc != KMD_OK) return rc;
      hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(block), want + extra, stream, P, R);
      KMD_HIP(hipGetLastError());
      return KMD_OK;
    };
    const char* force_kb = std::getenv("KMD_FLAT_TILE_KB");             // dev: 8 / 16 take the larger tiles (fewer waves) where 4 KB would do
    const int min_kb = force_kb ? std::atoi(force_kb) : 4;
    if (const uint32_t R = min_kb <= 4 ? rows_per_span(4096) : 0) return launch(k_filter_rows_flat<CT, 4, 1024>, 4, 1024, R);
    if (const uint32_t R = min_kb <= 8 ? rows_per_span(8192) : 0) return launch(k_filter_rows_flat<CT, 8, 768>, 8, 768, R);
    if (const uint32_t R = rows_per_span(16384)) return launch(k_filter_rows_flat<CT, 16, 512>, 16, 512, R);
  }
  // 16-byte aligned rows: the wave-private kernel (a row's last vector may reach into the
  // padding up to the pitch, never past it)
  const bool vec_rows = ((P.ld * sizeof(CT)) % 16 == 0) && ((reinterpret_cast<uintptr_t>(P.counts) & 15u) == 0);
  if (vec_rows)
  {
    const uint32_t row_vecs = (uint32_t)(((size_t)S * sizeof(CT) + 15) / 16);
    auto launch = [&](auto kernel, int cv_max, int block, bool defer) -> int
    {
      const uint32_t n_chunks = (row_vecs + cv_max - 1) / cv_max;
      // full passes of cv_max vectors (256 B: whole cache lines of a row) and a short last one --
      // measured better than balanced passes, whose pieces straddle lines (S=68: 1.24 vs 1.57 ms)
      const uint32_t cvb = (uint32_t)cv_max < row_vecs ? (uint32_t)cv_max : row_vecs;
      const size_t wpb = (size_t)block / 64;
      const size_t tiles_bytes = wpb * 64 * (cvb | 1u) * 16 + (defer ? wpb * kQueueBytesPerWave : 0);    // + the waves' queues
      const size_t avail = m->lds_per_block_max - 256 - tiles_bytes;
      size_t want = (size_t)P.lf_n * sizeof(double2);
      if (want > avail) want = avail / sizeof(double2) * sizeof(double2);
      P.lds_n = (uint32_t)(want / sizeof(double2));
      const size_t lds = want + tiles_bytes;
      const size_t n_wtiles = (P.n_rows + 63) / 64;
      size_t grid = (size_t)m->n_cu;
      if (grid > (n_wtiles + wpb - 1) / wpb) grid = (n_wtiles + wpb - 1) / wpb;
      int rc = allow_big_lds(kernel, lds);
      if (rc != KMD_OK) return rc;
      hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(block), lds, stream, P, row_vecs, n_chunks, cvb);
      KMD_HIP(hipGetLastError());
      return KMD_OK;
    };
    // one pass per row where the registers allow it, and then as many waves as fit
    // measured: narrow rows are instruction-bound and gain from the deferred-evaluation queue
    // (S=8: 1.08 -> 0.67 ms; u16 S=40: 0.90 -> 0.80 ms even with 12 instead of 16 waves); from 10
    // vectors on the queues' LDS would cost resident waves and the kernel is HBM-bound anyway
    if (row_vecs <= 4) return launch(k_filter_rows_wave<CT, 4, 1024, true>, 4, 1024, true);
    if (row_vecs <= 8) return launch(k_filter_rows_wave<CT, 8, 768, true>, 8, 768, true);
    if (row_vecs <= 10) return launch(k_filter_rows_wave<CT, 10, 768, false>, 10, 768, false);
    // G lanes per row: bound by the per-step work at ~9.5e9 rows/s (G = 16) / ~1.9e10 (G = 8); the
    // tile kernel above stays the choice where that is below what it reaches
    const int wide_g = std::getenv("KMD_ROWS_WIDE_OFF") ? 0 : row_vecs >= 32 ? 16 : row_vecs >= KMD_WIDE8_MIN ? 8 : 0;
    if (wide_g == 0) return launch(k_filter_rows_wave<CT, 16, 512, false>, 16, 512, false);
    {
      const size_t wpb = kWideBlock / 64;
      const size_t extra = wpb * kQueueBytesPerWave + wpb * 128 * sizeof(unsigned long long);
      const size_t avail = m->lds_per_block_max - 256 - extra;
      size_t want = (size_t)P.lf_n * sizeof(double2);
      if (want > avail) want = avail / sizeof(double2) * sizeof(double2);
      P.lds_n = (uint32_t)(want / sizeof(double2));
      const size_t n_wtiles = (P.n_rows + 63) / 64;
      size_t grid = (size_t)m->n_cu;
      if (grid > (n_wtiles + wpb - 1) / wpb) grid = (n_wtiles + wpb - 1) / wpb;
      if (wide_g == 16)
      {
        int rc = allow_big_lds(k_filter_rows_wide<CT, 16>, want + extra);
        if (rc != KMD_OK) return rc;
        hipLaunchKernelGGL((k_filter_rows_wide<CT, 16>), dim3((unsigned)grid), dim3(kWideBlock), want + extra, stream, P, row_vecs);
      }
      else
      {
        int rc = allow_big_lds(k_filter_rows_wide<CT, 8>, want + extra);
        if (rc != KMD_OK) return rc;
        hipLaunchKernelGGL((k_filter_rows_wide<CT, 8>), dim3((unsigned)grid), dim3(kWideBlock), want + extra, stream, P, row_vecs);
      }
      KMD_HIP(hipGetLastError());
      return KMD_OK;
    }
  }
  // everything else -- a base pointer that is not 16-byte aligned, a pitch that is not a whole number of dwords or an
  // unaligned one beyond the flat kernel's tiles: each lane walks its own row
  {
    size_t want = (size_t)P.lf_n * sizeof(double2);
    if (want > half) want = half / sizeof(double2) * sizeof(double2);
    P.lds_n = (uint32_t)(want / sizeof(double2));
    size_t grid = (size_t)m->n_cu * 4;
    if (grid > n_tiles) grid = n_tiles;
    int rc = allow_big_lds(k_filter_rows_direct<CT>, want);
    if (rc != KMD_OK) return rc;
    hipLaunchKernelGGL((k_filter_rows_direct<CT>), dim3((unsigned)grid), dim3(kRowsBlock), want, stream, P);
  }
  KMD_HIP(hipGetLastError());
  return KMD_OK;
}

} // namespace

// smallest LR with igamc(1/2, LR) <= threshold (bisection on the host with the same
// Cephes restatement), lowered by a margin that is ~1e6 x the device/host libm difference.
double kmd::lr_cut_for_threshold(double threshold, double lg_half)
{
  if (!(threshold >= 0)) return INFINITY;               // negative or NaN: nothing passes
  if (threshold >= 1) return -INFINITY;                 // p <= 1 always
  double lo = 0, hi = 800;                              // igamc(1/2, 800) == 0
  for (int i = 0; i < 200; ++i)
  {
    const double mid = 0.5 * (lo + hi);
    if (kmd::igamc_half(mid, lg_half) <= threshold) hi = mid; else lo = mid;
  }
  const double margin = hi * 1e-9 > 1e-6 ? hi * 1e-9 : 1e-6;
  return lo - margin;
}

extern "C" int kmd_poisson_filter(const kmd_model* m, const kmd_tile* tile, double threshold,
                                  const kmd_survivors* out, uint64_t* d_counters, void* stream)
{
  filter_params P;
  int rc = kmd::fill_filter_params(P, m, tile, threshold);
  if (rc != KMD_OK) return rc;
  KMD_REQUIRE(d_counters, "kmd_poisson_filter: NULL counters");
  P.counters = reinterpret_cast<unsigned long long*>(d_counters);
  if (out) P.out = *out;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (tile->n_rows == 0) return KMD_OK;
  rc = kmd::near_list_begin(P, st);                     // rows within 1e-8 of the threshold: listed, resolved behind the kernel
  if (rc != KMD_OK) return rc;

  // LDS budget: the table head; two workgroups per CU when it fits in half of the LDS
  const size_t lds_cap = m->lds_per_block_max;          // 160 KiB on gfx950
  if (tile->layout == KMD_LAYOUT_SOA || tile->layout == KMD_LAYOUT_TILED)
  {
    const bool tiled = tile->layout == KMD_LAYOUT_TILED;
    // the whole default table ({lf, log} pairs, 160 000 B) fits the 160 KiB of one CU: one
    // workgroup of kBlock threads per CU holds it; larger tables keep their head in LDS
    size_t want = m->lf_n * sizeof(double2);
    const int blocks_per_cu = KMD_BLOCKS_PER_CU;
    const size_t queues = (size_t)(kBlock / 64) * kQueueBytesPerWave;     // deferred-evaluation queues, one per wave
    const size_t budget = lds_cap / blocks_per_cu - 256 - queues;
    if (want > budget) want = budget / sizeof(double2) * sizeof(double2);
    P.lds_n = (uint32_t)(want / sizeof(double2));
    switch (tile->count_bytes)
    {
      case 1: rc = launch_soa<uint8_t>(P, m, want + queues, blocks_per_cu, tiled, st); break;
      case 2: rc = launch_soa<uint16_t>(P, m, want + queues, blocks_per_cu, tiled, st); break;
      default: rc = launch_soa<uint32_t>(P, m, want + queues, blocks_per_cu, tiled, st); break;
    }
  }
  else
  {
    switch (tile->count_bytes)
    {
      case 1: rc = launch_rows<uint8_t>(P, m, st); break;
      case 2: rc = launch_rows<uint16_t>(P, m, st); break;
      default: rc = launch_rows<uint32_t>(P, m, st); break;
    }
  }
  const int rc_near = kmd::near_list_end(P, 0, st);
  return rc != KMD_OK ? rc : rc_near;
}

// kmd_poisson_filter for rows that come as (k-mer, control sum, case sum): what kmd_merge_sums
// writes.  The survivors' `row` is the index into those arrays.
extern "C" int kmd_poisson_filter_sums(const kmd_model* m, const uint64_t* d_kmer, const uint64_t* d_sum_control,
                                       const uint64_t* d_sum_case, size_t n_rows, double threshold,
                                       const kmd_survivors* out, uint64_t* d_counters, void* stream)
{
  KMD_REQUIRE(m && d_counters, "kmd_poisson_filter_sums: NULL model or counters");
  KMD_REQUIRE(n_rows == 0 || (d_sum_control && d_sum_case), "kmd_poisson_filter_sums: NULL sums");
  kmd_tile t { d_sum_control, 4, KMD_LAYOUT_SOA, n_rows, d_kmer, nullptr, n_rows, 0 };   // for the shared checks; counts are never read
  filter_params P;
  int rc = kmd::fill_filter_params(P, m, &t, threshold);
  if (rc != KMD_OK) return rc;
  P.counters = reinterpret_cast<unsigned long long*>(d_counters);
  if (out) P.out = *out;
  if (n_rows == 0) return KMD_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // a short table head in LDS: this kernel runs for ~0.1 ms, staging the 112 KB head K1 keeps would be
  // most of it; the table is read for candidates only, the rest of it from L2
  size_t want = std::min<size_t>(m->lf_n, 1024) * sizeof(double2);
  const size_t queues = (size_t)(kBlock / 64) * kQueueBytesPerWave;
  P.lds_n = (uint32_t)(want / sizeof(double2));
  size_t grid = (size_t)m->n_cu;
  if (grid > (n_rows + kBlock - 1) / kBlock) grid = (n_rows + kBlock - 1) / kBlock;
  rc = allow_big_lds(k_filter_sums, want + queues);
  if (rc != KMD_OK) return rc;
  rc = kmd::near_list_begin(P, st);
  if (rc != KMD_OK) return rc;
  hipLaunchKernelGGL(k_filter_sums, dim3((unsigned)grid), dim3(kBlock), want + queues, st, P,
                     reinterpret_cast<const unsigned long long*>(d_sum_control), reinterpret_cast<const unsigned long long*>(d_sum_case));
  KMD_HIP(hipGetLastError());
  return kmd::near_list_end(P, 0, st);
}

// LogFactorialTable::log_factorial (log_factorial_table.cpp:13-22), the reference's value for a sum beyond its table:
// res += log(k), k-- down to 2.  By one wave for one k: 64 lanes take 64 correctly rounded logarithms, then the 64
// additions happen in the reference's order.
__device__ __forceinline__ double lf_running_sum_wave(uint64_t k, int lane)        // all 64 lanes, same k
{
  double res = 0;
  for (uint64_t j0 = k; j0 > 1; j0 = j0 > 64 ? j0 - 64 : 0)
  {
    const uint64_t j = j0 > (uint64_t)lane ? j0 - (uint64_t)lane : 0;
    const double l = j > 1 ? kmd::libm_rounded::log((double)j) : 0.0;                // (res + 0.0 == res: the lanes past k = 2)
#pragma unroll
    for (int t = 0; t < 64; ++t) res += __shfl(l, t, 64);
  }
  return res;
}

// The running sum again, fast enough for a sink full of such records (bench C3: 5 417 survivors with sums of ~1.3e5 each --
// 1.4e9 terms; with a correctly rounded logarithm per term and 64 dependent additions per step that was 5.3 ms, twice the
// partition's merge + test).  Two things make it cheap without changing a bit of the result:
//  * the logarithms come from a table (log_int[j] = correctly rounded log(j), j < kChainMax: 8 MB per device, built once);
//  * while the sum stays inside one binade [2^E, 2^(E+1)) every partial sum is a multiple of u = 2^(E-52), and
//    fl(r + l) = r + u RN(l / u) whenever l / u is not exactly half-way between two integers (then the tie goes to the even
//    r + ..., which depends on r): the 64 roundings are independent, their integer sum is exact, and one step of 64 terms
//    is a load, a rounding and a wave reduction.  Steps that may cross a binade or hold a tie take the 64 additions in order.
__device__ __forceinline__ double lf_running_sum_table(const double* __restrict__ log_int, uint64_t k, int lane)       // all 64 lanes, same k < kChainMax
{
  // the 64 terms of the step that starts at g: lane t holds log(g - t) (0.0 past the last term, k = 2: res + 0.0 == res)
  auto terms = [&](uint64_t g, int step) {
    const uint64_t off = (uint64_t)(64 * step + lane);
    const uint64_t j = g > off ? g - off : 0;
    return j > 1 ? log_int[j] : 0.0;
  };
  auto wave_sum = [](double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; };   // (of integers below 2^53: exact)
  constexpr double kTwo53 = 9007199254740992.0;
  double res = 0;
  uint64_t g0 = k;                                              // first term not yet added
  while (g0 > 1)
  {
    double u = 0, inv_u = 0, R = 0;
    if (res >= 32.0)                                            // u >= 2^-47 > ulp(l) / 2: the scaled terms stay below 2^52
    {
      const int E = (int)((unsigned long long)__double_as_longlong(res) >> 52) - 1023;
      u = __longlong_as_double((long long)(E - 52 + 1023) << 52);                        // 2^(E-52)
      inv_u = __longlong_as_double((long long)(52 - E + 1023) << 52);
      R = res * inv_u;                                          // res / u: an integer in [2^52, 2^53), exactly
      // steps that cannot leave the binade: every remaining term is at most log(g0), so n_safe steps add less than
      // 2^53 - R grid units whatever they hold.  Their roundings are summed per lane (integers as doubles: every partial sum
      // is below 2^53, so the additions are exact), the next steps' loads in flight, and reduced across the wave once
      const double q_max = ::ceil(log_int[g0] * inv_u) + 1.0;
      const unsigned long long n_safe = (unsigned long long)((kTwo53 - R) / (64.0 * q_max));        // (rounded down twice)
      unsigned long long done = 0;
      double Q = 0;
      constexpr int kG = 4;                                     // steps per group; the next group's loads are issued before this one is looked at
      double nx[kG];
#pragma unroll
      for (int a = 0; a < kG; ++a) nx[a] = terms(g0, a);
      while (n_safe - done >= kG && g0 > 1)
      {
        double x[kG], q[kG];
        bool tie = false;
#pragma unroll
        for (int a = 0; a < kG; ++a) x[a] = nx[a] * inv_u;                                // exact (a power of two)
        const uint64_t g1 = g0 > 64 * kG ? g0 - 64 * kG : 0;
#pragma unroll
        for (int a = 0; a < kG; ++a) nx[a] = terms(g1, a);
#pragma unroll
        for (int a = 0; a < kG; ++a) { q[a] = ::rint(x[a]); tie |= ::fabs(x[a] - q[a]) == 0.5; }
        if (__ballot(tie)) break;                               // a tie rounds to the even SUM: that step goes the ordered way below
        Q += (q[0] + q[1]) + (q[2] + q[3]);
        static_assert(kG == 4, "the sum above");
        done += kG;
        g0 = g1;
      }
      if (done)
      {
        res = (R + wave_sum(Q)) * u;
        continue;
      }
    }
    // one step on its own: rounded independently if it has no tie and stays in the binade, else the 64 additions in order
    const double l = terms(g0, 0);
    bool fast = false;
    if (res >= 32.0)
    {
      const double x = l * inv_u;
      const double q = ::rint(x);
      const bool tie = ::fabs(x - q) == 0.5;
      const double Q = wave_sum(q);                             // (64 terms below 2^52 / 64 each: exact)
      fast = !__ballot(tie) && R + Q < kTwo53;
      if (fast) res = (R + Q) * u;
    }
    if (!fast)
    {
#pragma unroll
      for (int t = 0; t < 64; ++t) res += __shfl(l, t, 64);
    }
    g0 = g0 > 64 ? g0 - 64 : 0;
  }
  return res;
}

// ---- the rows within 1e-8 of the threshold (kmd_eval.h: evaluate_core flags them, note_near_row lists them).
// One wave behind every filter launch: the listed rows are re-evaluated with correctly rounded log / exp
// (kmd_ddmath.h), and where that decision differs from the one the filter made with the device's libm the
// survivor sink is corrected: a record that should not be there is struck out (and the sink compacted), one
// that is missing is appended; survivors that stay get the rounded p-value.  The list is empty in practice
// (0 rows in 10^10 synthetic ones): the kernel then reads one word and leaves.
template <int kRowMode>
__global__ void __launch_bounds__(64) k_resolve_near(const filter_params P, const int test_flip)
{
  const unsigned long long flagged = P.near[0];
  const unsigned long long listed = flagged < kNearCap ? flagged : kNearCap;
  if (listed == 0) return;
  const int lane = (int)threadIdx.x;
  // rows flagged beyond the list's capacity keep the decision the filter made: said so, not hidden
  if (lane == 0 && flagged > kNearCap) atomicAdd(&P.counters[KMD_CNT_NEAR_UNRESOLVED], flagged - kNearCap);
  unsigned long long struck = 0;
  for (unsigned long long e0 = 0; e0 < listed; e0 += 64)
  {
    const bool have = e0 + lane < listed;
    const unsigned long long* e = P.near + 1 + 4 * (e0 + (have ? lane : 0));
    const uint64_t sum_c = e[0], sum_k = e[1], row = e[2];
    const long long slot = (long long)e[3];
    bool now = false;
    double p = 1.0, mean_control = 0.0;
    int sign = KMD_SIGN_NO;
    // table terms: the model's, or -- for a sum beyond the table (below kChainMax: evaluate_core flags no others) -- the
    // reference's running sum, one row at a time by the whole wave
    double2 tc = make_double2(0.0, 0.0), tk = make_double2(0.0, 0.0);
    if (have && sum_c < P.lf_n) tc = P.tab[sum_c];
    if (have && sum_k < P.lf_n) tk = P.tab[sum_k];
    unsigned long long chain = __ballot(have && (sum_c >= P.lf_n || sum_k >= P.lf_n));
    while (chain)
    {
      const int src = __ffsll((long long)chain) - 1;
      chain &= chain - 1;
      const uint64_t c = (uint64_t)__shfl((unsigned long long)sum_c, src, 64), k = (uint64_t)__shfl((unsigned long long)sum_k, src, 64);
      double2 xc = make_double2(0.0, 0.0), xk = make_double2(0.0, 0.0);
      if (c >= P.lf_n) { xc.x = lf_running_sum_table(P.log_int, c, lane); xc.y = P.log_int[c]; }
      if (k >= P.lf_n) { xk.x = lf_running_sum_table(P.log_int, k, lane); xk.y = P.log_int[k]; }
      if (lane == src) { if (c >= P.lf_n) tc = xc; if (k >= P.lf_n) tk = xk; }
    }
    if (have)
    {
      const double lr = kmd::lr_from_sums<kmd::libm_rounded>(sum_c, sum_k, tc.x, tk.x, tc.y, tk.y, P.dT, P.dTc, P.dTk);
      p = kmd::igamc_half<kmd::libm_rounded>(lr, P.lg_half);
      now = p <= P.threshold;                                                              // merge.hpp:78
      kmd::sign_of(sum_c, sum_k, P.dTc, P.dTk, mean_control, sign);
    }
    const bool was = have && slot >= 0;
    // dev (KMD_TEST_NEAR_FLIP, tests only): every listed row's decision is the opposite of the filter's, so that every
    // one of them is struck out or appended -- the paths a libm's last bit takes once in 10^10 rows
    if (test_flip) now = have && !was;
    // stays: the rounded p-value; goes: struck out (p = -1 marks the record until the sink is compacted)
    if (was && (unsigned long long)slot < P.out.capacity && P.out.d_pvalue) P.out.d_pvalue[slot] = now ? p : -1.0;
    const unsigned long long go_ctrl = __ballot(was && !now && sign == KMD_SIGN_CONTROL), go = __ballot(was && !now);
    const unsigned long long come_ctrl = __ballot(have && !was && now && sign == KMD_SIGN_CONTROL), come = __ballot(have && !was && now);
    struck += (unsigned long long)__popcll(go);
    unsigned long long base = 0;
    if (lane == 0)
    {
      if (go_ctrl) atomicAdd(&P.counters[KMD_CNT_SIG_CONTROL], 0ull - (unsigned long long)__popcll(go_ctrl));
      if (go & ~go_ctrl) atomicAdd(&P.counters[KMD_CNT_SIG_CASE], 0ull - (unsigned long long)__popcll(go & ~go_ctrl));
      if (come_ctrl) atomicAdd(&P.counters[KMD_CNT_SIG_CONTROL], (unsigned long long)__popcll(come_ctrl));
      if (come & ~come_ctrl) atomicAdd(&P.counters[KMD_CNT_SIG_CASE], (unsigned long long)__popcll(come & ~come_ctrl));
      if (come) base = atomicAdd(&P.counters[KMD_CNT_SIG], (unsigned long long)__popcll(come));
    }
    base = __shfl(base, 0, 64);
    if (have && !was && now)
    {
      const unsigned long long at = base + (unsigned long long)__popcll(come & ((1ull << lane) - 1ull));
      if (at < P.out.capacity)
      {
        if (P.out.d_row) P.out.d_row[at] = kRowMode == 1 ? P.kmer_lo[row] : P.row_base + row;
        if (P.out.d_kmer_lo && P.kmer_lo) P.out.d_kmer_lo[at] = P.kmer_lo[row];
        if (P.out.d_kmer_hi && P.kmer_hi) P.out.d_kmer_hi[at] = P.kmer_hi[row];
        if (P.out.d_pvalue) P.out.d_pvalue[at] = p;
        if (P.out.d_sign) P.out.d_sign[at] = sign;
        if (P.out.d_mean_control) P.out.d_mean_control[at] = mean_control;
        if (P.out.d_mean_case) P.out.d_mean_case[at] = (double)sum_k;
      }
    }
  }
  // the list is handed back empty: it belongs to the stream and serves the next filter launch on it as it is
  if (lane == 0) P.near[0] = 0;
  if (struck == 0 || !P.out.d_pvalue) return;
  // compact the sink in place (this wave alone, behind the filter kernel on its stream): records marked
  // p = -1 leave, the others close ranks in order
  __threadfence();
  const unsigned long long n_sig = P.counters[KMD_CNT_SIG];
  const unsigned long long n = n_sig < P.out.capacity ? n_sig : P.out.capacity;
  unsigned long long w = 0;
  for (unsigned long long r0 = 0; r0 < n; r0 += 64)
  {
    const unsigned long long r = r0 + lane;
    const bool keep = r < n && P.out.d_pvalue[r] >= 0.0;
    const unsigned long long km = __ballot(keep);
    const unsigned long long to = w + (unsigned long long)__popcll(km & ((1ull << lane) - 1ull));
    // (reads of this step happen before its writes: every lane loads first)
    unsigned long long v_row = 0, v_lo = 0, v_hi = 0; double v_p = 0, v_mc = 0, v_mk = 0; int v_s = 0;
    if (keep)
    {
      if (P.out.d_row) v_row = P.out.d_row[r];
      if (P.out.d_kmer_lo) v_lo = P.out.d_kmer_lo[r];
      if (P.out.d_kmer_hi) v_hi = P.out.d_kmer_hi[r];
      v_p = P.out.d_pvalue[r];
      if (P.out.d_sign) v_s = P.out.d_sign[r];
      if (P.out.d_mean_control) v_mc = P.out.d_mean_control[r];
      if (P.out.d_mean_case) v_mk = P.out.d_mean_case[r];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if (keep && to != r)
    {
      if (P.out.d_row) P.out.d_row[to] = v_row;
      if (P.out.d_kmer_lo) P.out.d_kmer_lo[to] = v_lo;
      if (P.out.d_kmer_hi) P.out.d_kmer_hi[to] = v_hi;
      P.out.d_pvalue[to] = v_p;
      if (P.out.d_sign) P.out.d_sign[to] = v_s;
      if (P.out.d_mean_control) P.out.d_mean_control[to] = v_mc;
      if (P.out.d_mean_case) P.out.d_mean_case[to] = v_mk;
    }
    w += (unsigned long long)__popcll(km);
    __threadfence();
  }
  // the count stays a count of survivors, also when the sink was too small for them: the host compares it with
  // the capacity and runs the partition again with a larger sink (w alone would hide that)
  if (lane == 0) P.counters[KMD_CNT_SIG] = n_sig <= P.out.capacity ? w : n_sig - struck;
}

// The list of a launch: one per (device, stream), made at the first filter launch on that stream and kept; it is
// empty whenever no filter work is in flight on the stream (k_resolve_near hands it back empty), so a launch costs
// neither an allocation nor a memset.  Launches on one stream are ordered, launches on different streams have
// different lists.
namespace {
std::mutex g_near_mu;
std::map<std::pair<int, hipStream_t>, unsigned long long*> g_near_lists;
constexpr size_t kNearListsMax = 256;                   // streams served with a kept list; beyond: allocated per launch
}

// kmd_stream_destroy / kmd_release_cache: the caller has nothing in flight on the stream(s)
void kmd::near_list_forget(hipStream_t stream)
{
  std::lock_guard<std::mutex> lock(g_near_mu);
  for (auto it = g_near_lists.begin(); it != g_near_lists.end();)
    if (it->first.second == stream) { (void)hipFree(it->second); it = g_near_lists.erase(it); }
    else ++it;
}

void kmd::near_lists_release()
{
  std::lock_guard<std::mutex> lock(g_near_mu);
  for (auto& kv : g_near_lists) (void)hipFree(kv.second);
  g_near_lists.clear();
}

int kmd::near_list_begin(filter_params& P, hipStream_t stream)
{
  P.near = nullptr;
  static const bool off = std::getenv("KMD_NO_GUARD") != nullptr;      // dev: the filter without its guard
  if (off) return KMD_OK;
  int dev = 0;
  KMD_HIP(hipGetDevice(&dev));
  const size_t bytes = (1 + 4 * kNearCap) * sizeof(unsigned long long);
  {
    std::lock_guard<std::mutex> lock(g_near_mu);
    auto it = g_near_lists.find({ dev, stream });
    if (it != g_near_lists.end()) { P.near = it->second; return KMD_OK; }
    if (g_near_lists.size() < kNearListsMax)
    {
      // The list's count is zeroed ON THE STREAM the filter kernels run on.  (Round 4 zeroed it with hipMemset -- the
      // null stream, which the library's and its callers' non-blocking streams are not ordered against, and which
      // returns before the fill has run: a filter launch on a NEW stream could find whatever the fresh allocation held,
      // k_resolve_near then "resolved" up to 4096 entries of garbage -- log_int[garbage] -- and the queue died of a
      // memory aperture violation, taking the process with it: the abort of DESIGN 10, seen when several host threads
      // made their first call on fresh streams right after kmd_release_cache had handed used memory back.)
      void* p = nullptr;
      if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return KMD_OK; }
      static const int init_mode = [] { const char* e = std::getenv("KMD_TEST_NEAR_INIT"); return e ? std::atoi(e) : 0; }();
      hipError_t e = hipSuccess;
      // dev / tests: 1 = the fresh list is filled with ones first (what used memory may look like: an initialisation that
      // is not ordered before the kernels then fails every time, not once in fifteen runs); 2 = that, and round 4's
      // initialisation on the null stream (the regression test's "old tree")
      if (init_mode >= 1) { e = hipMemsetAsync(p, 0xFF, bytes, stream); if (e == hipSuccess) e = hipStreamSynchronize(stream); }     // (the ones are THERE before the count is zeroed)
      if (e == hipSuccess) e = init_mode == 2 ? hipMemset(p, 0, sizeof(unsigned long long)) : hipMemsetAsync(p, 0, sizeof(unsigned long long), stream);
      if (e != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); return KMD_OK; }
      g_near_lists[{ dev, stream }] = static_cast<unsigned long long*>(p);
      P.near = static_cast<unsigned long long*>(p);
      return KMD_OK;
    }
  }
  void* p = nullptr;
  if (hipMallocAsync(&p, bytes, stream) != hipSuccess) { (void)hipGetLastError(); return KMD_OK; }
  P.near = static_cast<unsigned long long*>(p);
  KMD_HIP(hipMemsetAsync(P.near, 0, sizeof(unsigned long long), stream));
  return KMD_OK;
}

int kmd::near_list_end(const filter_params& P, int row_mode, hipStream_t stream)
{
  if (!P.near) return KMD_OK;
  const int test_flip = std::getenv("KMD_TEST_NEAR_FLIP") != nullptr ? 1 : 0;
  if (row_mode == 1) hipLaunchKernelGGL(k_resolve_near<1>, dim3(1), dim3(64), 0, stream, P, test_flip);
  else hipLaunchKernelGGL(k_resolve_near<0>, dim3(1), dim3(64), 0, stream, P, test_flip);
  KMD_HIP(hipGetLastError());
  bool kept = false;
  {
    int dev = 0;
    KMD_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_near_mu);
    auto it = g_near_lists.find({ dev, stream });
    kept = it != g_near_lists.end() && it->second == P.near;
  }
  if (!kept) KMD_HIP(hipFreeAsync(P.near, stream));
  return KMD_OK;
}

// device scratch the three kernels need for a list of up to `cap` entries: [p_bits: cap x 8][per wave: offset 8 + counts 16]
static uint32_t cand_waves(const kmd_model* m) { return (uint32_t)m->n_cu * 4u * (kCandBlock / 64); }
size_t kmd::filter_candidates_work_bytes(size_t cap, const kmd_model* m)
{
  return cap * 8 + (size_t)cand_waves(m) * (8 + sizeof(cand_counts));
}

int kmd::launch_filter_candidates(const filter_params& P_in, const kmd_model* m, const uint64_t* d_kmer, const uint64_t* d_kmer_hi,
                                  const uint64_t* d_sum_c, const uint64_t* d_sum_k, size_t n, uint64_t rows_total, uint64_t rows_beyond,
                                  void* d_work, size_t cap, hipStream_t stream, const uint64_t* d_gate, const uint32_t* d_gate_over, size_t gate_cap)
{
  filter_params P = P_in;
  P.kmer_lo = d_kmer; P.kmer_hi = d_kmer_hi; P.row_base = 0; P.n_rows = n;
  P.lds_n = 0;                                            // the table is read for candidates only: from L2
  if ((d_gate ? gate_cap : n) > cap) { kmd::set_error("launch_filter_candidates: scratch smaller than the list"); return KMD_E_INVALID; }
  // gated: the number of entries is on the device, the grid is the full one (idle waves leave at once)
  const uint32_t waves_max = cand_waves(m);
  size_t grid = d_gate ? (size_t)waves_max / (kCandBlock / 64) : std::min<size_t>((size_t)waves_max / (kCandBlock / 64), (n + kCandBlock - 1) / kCandBlock);
  if (grid < 1) grid = 1;
  const uint32_t n_waves = (uint32_t)grid * (kCandBlock / 64);
  unsigned long long* p_bits = static_cast<unsigned long long*>(d_work);
  unsigned long long* wave_off = p_bits + cap;
  cand_counts* wave_counts = reinterpret_cast<cand_counts*>(wave_off + waves_max);
  const unsigned long long* sc = reinterpret_cast<const unsigned long long*>(d_sum_c);
  const unsigned long long* sk = reinterpret_cast<const unsigned long long*>(d_sum_k);
  const unsigned long long* gate = reinterpret_cast<const unsigned long long*>(d_gate);
  int rc = near_list_begin(P, stream);
  if (rc != KMD_OK) return rc;
  hipLaunchKernelGGL(k_cand_eval, dim3((unsigned)grid), dim3(kCandBlock), 0, stream, P, sc, sk, gate, d_gate_over, (unsigned long long)gate_cap, p_bits, wave_counts);
  hipLaunchKernelGGL(k_cand_scan, dim3(1), dim3(1024), 0, stream, P, (unsigned long long)rows_total, (unsigned long long)rows_beyond, gate, d_gate_over,
                     (unsigned long long)gate_cap, wave_counts, n_waves, wave_off);
  hipLaunchKernelGGL(k_cand_emit, dim3((unsigned)grid), dim3(kCandBlock), 0, stream, P, sc, sk, gate, d_gate_over, (unsigned long long)gate_cap, p_bits, wave_off);
  KMD_HIP(hipGetLastError());
  return near_list_end(P, 1, stream);
}

extern "C" int kmd_poisson_process(const kmd_model* m, const kmd_tile* tile, double* d_pvalue,
                                   int32_t* d_sign, double* d_mean_control, double* d_mean_case,
                                   void* stream)
{
  filter_params P;
  int rc = kmd::fill_filter_params(P, m, tile, 1.0);
  if (rc != KMD_OK) return rc;
  if (tile->n_rows == 0) return KMD_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  size_t grid = (tile->n_rows + 255) / 256;
  if (grid > (size_t)m->n_cu * 8) grid = (size_t)m->n_cu * 8;
  switch (tile->count_bytes)
  {
    case 1: hipLaunchKernelGGL((k_process_all<uint8_t>), dim3((unsigned)grid), dim3(256), 0, st, P, tile->layout, d_pvalue, d_sign, d_mean_control, d_mean_case); break;
    case 2: hipLaunchKernelGGL((k_process_all<uint16_t>), dim3((unsigned)grid), dim3(256), 0, st, P, tile->layout, d_pvalue, d_sign, d_mean_control, d_mean_case); break;
    default: hipLaunchKernelGGL((k_process_all<uint32_t>), dim3((unsigned)grid), dim3(256), 0, st, P, tile->layout, d_pvalue, d_sign, d_mean_control, d_mean_case); break;
  }
  KMD_HIP(hipGetLastError());
  return KMD_OK;
}

// ---- p-values to the reference's last bit (kmd_pvalues_refine).
// The filters evaluate the two null-hypothesis logarithms and Cephes' exp / log with the device's libm, whose last bit may
// differ from glibc's; `k * log(lambda)` (model.hpp:137) multiplies that bit by the count sum, so the p-value of a record
// deviates from a glibc-built reference's by ~1e-16 x sum relative in LR -- 1e-10 absolute on p is reached at sums of
// ~10^4 when p is of order 1 (tests/soak.py found 1.1e-10 at p = 0.92, sums 7155 + 8169).  The decisions are guarded
// separately (k_resolve_near); this pass is for the NUMBER: each record's two sums are recovered from its two means
// (mean_case IS the case sum; mean_control = fl(fl(sc Tk) / Tc) is inverted and checked by re-evaluating it) and the chain
// of model.hpp:147-161 is repeated with correctly rounded log / exp (kmd_ddmath.h), as k_resolve_near does for the rows
// near the threshold.
// A sum beyond the log-factorial table: the filters take Stirling's series there (kmd_eval.h), the reference a k-term
// running sum res += log(k), k-- (log_factorial_table.cpp:13-22), whose rounding is its own (~sqrt(k) ulp away from
// ln k!; it enters alt and null alike and cancels to within an ulp of k: 5e-10 relative on p measured).  Here the running
// sum itself is repeated -- by the whole wave for one record at a time: 64 lanes take 64 logarithms, then the 64
// additions happen in the reference's order -- for sums below kChainMax (2^20: 16 K steps of 64, ~4 ms of one
// wave; the reference spends ~10 ms of a core on such a row); records with a larger sum keep the filter's value.
// kernel 1: one lane per record -- the sums recovered, records inside the table rewritten, the others listed
__global__ void __launch_bounds__(256) k_refine_pvalues(const double2* __restrict__ tab, const unsigned long long lf_n, const double dT, const double dTc,
                                                        const double dTk, const double lg_half, const unsigned long long n,
                                                        const double* __restrict__ mean_control, const double* __restrict__ mean_case,
                                                        double* __restrict__ pvalue, unsigned long long* __restrict__ list,
                                                        const double* __restrict__ log_int)
{
  const int lane = (int)(threadIdx.x & 63);
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i0 = (unsigned long long)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); i0 < n; i0 += stride)      // wave-uniform trip count
  {
    const unsigned long long i = i0 + (unsigned long long)lane;
    uint64_t sum_c = 0, sum_k = 0;
    bool found = false;
    if (i < n)
    {
      const double mc = mean_control[i], mk = mean_case[i];
      if (mk >= 0.0 && mk < 9.0e15 && mc >= 0.0)
      {
        sum_k = (uint64_t)mk;
        const double guess = mc * dTc / dTk;
        if ((double)sum_k == mk && guess < 9.0e15)
        {
          const uint64_t g = (uint64_t)::llrint(guess);
          for (int d = 0; d < 5 && !found; ++d)               // g, g-1, g+1, g-2, g+2
          {
            const long long off = (d & 1) ? -(long long)((d + 1) / 2) : (long long)(d / 2);
            if (off < 0 && g < (uint64_t)(-off)) continue;
            const uint64_t c = g + (uint64_t)off;
            if ((double)c * dTk / dTc == mc) { sum_c = c; found = true; }             // model.hpp:165, as kmd::sign_of evaluates it
          }
        }
      }
    }
    const bool beyond = found && (sum_c >= lf_n || sum_k >= lf_n);
    if (found && !beyond)
    {
      const double2 tc = tab[sum_c], tk = tab[sum_k];
      const double lr = kmd::lr_from_sums<kmd::libm_rounded>(sum_c, sum_k, tc.x, tk.x, tc.y, tk.y, dT, dTc, dTk);
      pvalue[i] = kmd::igamc_half<kmd::libm_rounded>(lr, lg_half);
    }
    // the records beyond the table go on the list { count, then (index, sum_c, sum_k) }: kernel 2 gives each a wave
    const bool listed = beyond && sum_c < kChainMax && sum_k < kChainMax;
    if (beyond && !listed)
    {
      // a sum of 2^20 or more: the table term stays Stirling's (evaluate_core's), the logarithms are rounded correctly --
      // the part of the deviation that grows with the sum (k ulp(log lambda): 5e-9 relative on p at sums of 2.6e6) goes,
      // what stays is the rounding of the reference's own running sum (~ulp(k))
      const uint32_t kc = kmd::table_index(sum_c), kk = kmd::table_index(sum_k);
      double2 tc = make_double2(0.0, 0.0), tk = make_double2(0.0, 0.0);
      if (kc < lf_n) tc = tab[kc]; else tc.x = lf_beyond_table(kc);
      if (kk < lf_n) tk = tab[kk]; else tk.x = lf_beyond_table(kk);
      if (sum_c >= lf_n) tc.y = kmd::libm_rounded::log((double)sum_c);
      if (sum_k >= lf_n) tk.y = kmd::libm_rounded::log((double)sum_k);
      const double lr = kmd::lr_from_sums<kmd::libm_rounded>(sum_c, sum_k, tc.x, tk.x, tc.y, tk.y, dT, dTc, dTk);
      pvalue[i] = kmd::igamc_half<kmd::libm_rounded>(lr, lg_half);
    }
    const unsigned long long mask = __ballot(listed);
    if (mask && !list)
    {
      // a small call (no list was allocated): the wave takes its records beyond the table one at a time, here
      unsigned long long todo = mask;
      while (todo)
      {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const uint64_t c = (uint64_t)__shfl((unsigned long long)sum_c, src, 64), k = (uint64_t)__shfl((unsigned long long)sum_k, src, 64);
        double2 tc, tk;
        if (c < lf_n) tc = tab[c]; else { tc.x = lf_running_sum_table(log_int, c, lane); tc.y = log_int[c]; }
        if (k < lf_n) tk = tab[k]; else { tk.x = lf_running_sum_table(log_int, k, lane); tk.y = log_int[k]; }
        if (lane == src)
        {
          const double lr = kmd::lr_from_sums<kmd::libm_rounded>(c, k, tc.x, tk.x, tc.y, tk.y, dT, dTc, dTk);
          pvalue[i] = kmd::igamc_half<kmd::libm_rounded>(lr, lg_half);
        }
      }
    }
    else if (mask)
    {
      unsigned long long base = 0;
      if (lane == 0) base = atomicAdd(&list[0], (unsigned long long)__popcll(mask));
      base = __shfl(base, 0, 64);
      if (listed)
      {
        unsigned long long* e = list + 1 + 3 * (base + (unsigned long long)__popcll(mask & ((1ull << lane) - 1ull)));
        e[0] = i; e[1] = sum_c; e[2] = sum_k;
      }
    }
  }
}

__global__ void __launch_bounds__(256) k_build_log_int(double* __restrict__ log_int)
{
  const unsigned long long j = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < kChainMax) log_int[j] = j > 0 ? kmd::libm_rounded::log((double)j) : 0.0;
}

// kernel 2: two waves per listed record (one running sum each, then the chain of model.hpp:147-161 on the first lane)
__global__ void __launch_bounds__(128) k_refine_chain(const double2* __restrict__ tab, const unsigned long long lf_n, const double dT, const double dTc,
                                                      const double dTk, const double lg_half, const unsigned long long* __restrict__ list,
                                                      const double* __restrict__ log_int, double* __restrict__ pvalue)
{
  __shared__ double2 s_term[2];
  const int lane = (int)(threadIdx.x & 63), half = (int)(threadIdx.x >> 6);
  const unsigned long long count = list[0];
  for (unsigned long long e = blockIdx.x; e < count; e += gridDim.x)
  {
    const unsigned long long i = list[1 + 3 * e];
    const uint64_t c = list[2 + 3 * e], k = list[3 + 3 * e], mine = half ? k : c;
    double2 t;
    if (mine < lf_n) t = tab[mine]; else { t.x = lf_running_sum_table(log_int, mine, lane); t.y = log_int[mine]; }
    if (lane == 0) s_term[half] = t;
    __syncthreads();
    if (threadIdx.x == 0)
    {
      const double2 tc = s_term[0], tk = s_term[1];
      const double lr = kmd::lr_from_sums<kmd::libm_rounded>(c, k, tc.x, tk.x, tc.y, tk.y, dT, dTc, dTk);
      pvalue[i] = kmd::igamc_half<kmd::libm_rounded>(lr, lg_half);
    }
    __syncthreads();
  }
}

// dev / tests: the running sum of each k[i] both ways (the term-by-term one of k_resolve_near, the table one above)
__global__ void __launch_bounds__(64) k_test_running_sums(const unsigned long long* __restrict__ k, const unsigned long long n, const double* __restrict__ log_int,
                                                          double* __restrict__ plain, double* __restrict__ fast)
{
  const int lane = (int)threadIdx.x;
  for (unsigned long long e = blockIdx.x; e < n; e += gridDim.x)
  {
    const double a = plain ? lf_running_sum_wave(k[e], lane) : 0.0, b = lf_running_sum_table(log_int, k[e], lane);
    if (lane == 0) { if (plain) plain[e] = a; fast[e] = b; }
  }
}

namespace {
std::mutex g_log_int_mu;
std::map<int, double*> g_log_int;          // per device

} // namespace

int kmd::log_int_table(const double** out)
{
  int dev = 0;
  KMD_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_log_int_mu);
  auto it = g_log_int.find(dev);
  if (it == g_log_int.end())
  {
    void* p = nullptr;
    KMD_HIP(hipMalloc(&p, kChainMax * sizeof(double)));
    hipLaunchKernelGGL(k_build_log_int, dim3((unsigned)(kChainMax / 256)), dim3(256), 0, nullptr, static_cast<double*>(p));
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);       // built once, before any stream reads it
    if (e != hipSuccess) { (void)hipFree(p); return kmd::hip_fail(e, "k_build_log_int", __FILE__, __LINE__); }
    it = g_log_int.emplace(dev, static_cast<double*>(p)).first;
  }
  *out = it->second;
  return KMD_OK;
}

extern "C" int kmd_test_running_sums(const uint64_t* d_k, size_t n, double* d_plain, double* d_fast, void* stream)
{
  const double* log_int = nullptr;
  const int rc = kmd::log_int_table(&log_int);
  if (rc != KMD_OK) return rc;
  if (n == 0) return KMD_OK;
  hipLaunchKernelGGL(k_test_running_sums, dim3((unsigned)std::min<size_t>(n, 4096)), dim3(64), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const unsigned long long*>(d_k), (unsigned long long)n, log_int, d_plain, d_fast);
  KMD_HIP(hipGetLastError());
  return KMD_OK;
}

extern "C" int kmd_pvalues_refine(const kmd_model* m, size_t n, const double* d_mean_control, const double* d_mean_case, double* d_pvalue, void* stream)
{
  KMD_REQUIRE(m != nullptr, "kmd_pvalues_refine: no model");
  if (n == 0) return KMD_OK;
  KMD_REQUIRE(d_mean_control && d_mean_case && d_pvalue, "kmd_pvalues_refine: the two means and the p-values are needed");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const double* log_int = m->d_log_int;
  const double2* tab = reinterpret_cast<const double2*>(m->d_tab);
  size_t grid = (n + 255) / 256;
  if (grid > (size_t)m->n_cu * 8) grid = (size_t)m->n_cu * 8;
  if (n <= 2048)
  {
    // a handful of records (the IModel plugin refines one per call): one launch, nothing allocated
    hipLaunchKernelGGL(k_refine_pvalues, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, tab, (unsigned long long)m->lf_n, m->dT, m->dTc, m->dTk, m->lg_half,
                       (unsigned long long)n, d_mean_control, d_mean_case, d_pvalue, static_cast<unsigned long long*>(nullptr), log_int);
    KMD_HIP(hipGetLastError());
    return KMD_OK;
  }
  // (the list from the library's scratch cache: a hipMallocAsync / hipFreeAsync pair per call bypassed it; the block is
  // parked again once the stream has drained -- the two kernels take microseconds on a sink's worth of records)
  void* list = nullptr;
  KMD_HIP(kmd::scratch_alloc(&list, (1 + 3 * n) * sizeof(unsigned long long)));
  {
    const hipError_t e0 = hipMemsetAsync(list, 0, sizeof(unsigned long long), st);
    if (e0 != hipSuccess) { kmd::scratch_free(list); return kmd::hip_fail(e0, "hipMemsetAsync", __FILE__, __LINE__); }
  }
  hipLaunchKernelGGL(k_refine_pvalues, dim3((unsigned)grid), dim3(256), 0, st, tab, (unsigned long long)m->lf_n, m->dT, m->dTc, m->dTk, m->lg_half,
                     (unsigned long long)n, d_mean_control, d_mean_case, d_pvalue, static_cast<unsigned long long*>(list), log_int);
  // as many waves as the chip holds at a comfortable occupancy; those without a record leave at once
  const size_t chain_grid = std::min<size_t>(n, (size_t)m->n_cu * 8);
  hipLaunchKernelGGL(k_refine_chain, dim3((unsigned)chain_grid), dim3(128), 0, st, tab, (unsigned long long)m->lf_n, m->dT, m->dTc, m->dTk, m->lg_half,
                     static_cast<const unsigned long long*>(list), log_int, d_pvalue);
  const hipError_t launched = hipGetLastError();
  const hipError_t drained = hipStreamSynchronize(st);          // nothing reads the list any more: back to the cache
  kmd::scratch_free(list);
  KMD_HIP(launched);
  KMD_HIP(drained);
  return KMD_OK;
}
