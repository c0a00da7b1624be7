// kmd_eval.h -- the per-row evaluation shared by every kernel that ends in the Poisson test
// (kmd_filter.hip: rows of a count matrix; kmd_tilemerge.hip: rows that leave the k-way merge as two
// count sums): diff_observer::process (include/kmdiff/merge.hpp:68-103) calling
// PoissonLikelihood::process (include/kmdiff/model.hpp:142-176), from the two count sums on.
#pragma once
#include "kmd_internal.h"
#include "kmd_math.h"

namespace kmd { namespace eval {

struct filter_params
{
  const void* counts;
  size_t ld;            // SoA: column stride; rows: row stride; tiled: T (= column stride)
  size_t tiles_per_blk; // tiled: kernel tiles per T-row block (T / tile_rows); else 0
  size_t blk_stride;    // tiled: S * T elements between blocks
  size_t n_rows;
  uint64_t row_base;
  const uint64_t* kmer_lo;
  const uint64_t* kmer_hi;
  int nc, nk;
  double dT, dTc, dTk, lg_half, lr_cut, threshold;
  double dTcTk;             // dTc * dTk
  double pf_cut;            // chi-square pre-filter cut on the likelihood ratio (or -inf: off)
  const double* lf;         // lf[k]                        (k_process_all)
  const double2* tab;       // { lf[k], log(double(k)) }    (filter kernels; head staged in LDS)
  const double* log_int;    // correctly rounded log(j), j < kChainMax (k_resolve_near: the running sum beyond the table)
  uint32_t lf_n;
  uint32_t lds_n;           // table entries held in LDS
  kmd_survivors out;
  unsigned long long* counters;
  unsigned long long* near;   // rows within 1e-8 of the threshold: [0] count, then kNearCap x { sum_c, sum_k, row, slot } (or NULL)
};

constexpr unsigned long long kNearCap = 4096;
// sums up to which the reference's log-factorial running sum (log_factorial_table.cpp:13-22) is repeated term for term
// where its bits matter (rows near the threshold, kmd_pvalues_refine): 2^20 terms = 16 K steps of one wave, ~4 ms
constexpr unsigned long long kChainMax = 1ull << 20;

// LogFactorialTable::operator[] for k >= table size (log_factorial_table.hpp:14-18 falls back
// to the O(k) loop log(k) + log(k-1) + ... + log(2), src/log_factorial_table.cpp:13-22).
//   k <  kStirlingMin : the same descending loop, per lane (bounded, reference order);
//   k >= kStirlingMin : ln k! by the Stirling series, O(1):
//        (k + 1/2) ln k - k + ln(2 pi)/2 + 1/(12k) - 1/(360k^3) + 1/(1260k^5)
//     truncation error < 1e-20 for k >= 256; the result is within ~1 ulp of ln k!, whereas
//     the reference's k-term running sum carries its own rounding error of order
//     sqrt(k) ulp.  The table value enters alt and null hypotheses identically
//     (model.hpp:152-156), so this difference cancels in the likelihood ratio down to the
//     rounding of the individual terms (tests/test_gpu_parity.py::test_table_fallback*).
constexpr uint32_t kStirlingMin = 256;

__device__ __forceinline__ double lf_beyond_table(uint32_t k)
{
  if (k < kStirlingMin)
  {
    double res = 0;
    for (uint32_t j = k; j > 1; --j) res += ::log((double)j);
    return res;
  }
  const double x = (double)k;
  const double r = 1.0 / x, r2 = r * r;
  const double corr = r * (8.3333333333333333e-02 - r2 * (2.7777777777777778e-03 - r2 * 7.9365079365079365e-04));
  return ((x + 0.5) * ::log(x) - x) + (0.91893853320467274178 + corr);
}

struct row_state
{
  uint64_t sum_c, sum_k;
  uint64_t row;        // local row index in the tile
  bool valid;
};

// One row from its two count sums to the survivor sink.  Must be called by all 64 lanes of
// the wave together (ballots / cooperative fallback inside).
// Can this row still reach `p <= threshold`?  (Also counts the rows beyond the table.)
template <class PP>
__device__ __forceinline__ bool row_may_pass(const PP& P, const row_state& st, uint32_t& n_beyond)
{
  if (st.valid && (st.sum_c >= P.lf_n || st.sum_k >= P.lf_n))
    ++n_beyond;        // rows beyond the table; flushed once per wave at kernel end (with many
                       // samples most waves see such rows: a global atomic here serialises the chip)

  // Pre-filter.  In exact arithmetic LR = n KL(x || q) with n = sc + sk, x = sc / n,
  // q = Tc / (Tc + Tk) (the lf[k] and -lambda terms of model.hpp:152-156 cancel), and
  // KL(x || q) <= (x - q)^2 / (q (1 - q))  (from ln t <= t - 1), i.e.
  //     LR <= (sc Tk - sk Tc)^2 / (n Tc Tk).
  // A row whose bound is below HALF the candidate cut cannot reach `p <= threshold`; it is
  // dropped here for ~25 flops instead of a division and two logarithms.  The factor 2 and
  // the host-side enabling conditions (fill_params) cover the rounding of the bound itself;
  // rows that pass are evaluated exactly as before, so every exposed number is unchanged.
  const double dsc = (double)st.sum_c, dsk = (double)st.sum_k;
  const double a = dsc * P.dTk - dsk * P.dTc;
  return st.valid && !(a * a < P.pf_cut * ((dsc + dsk) * P.dTcTk));
}

// What the exact evaluation of one row decides (per lane).
struct row_result
{
  bool cand, surv, near;
  double p, mean_control;
  int sign;
};

// The exact evaluation of rows that passed the pre-filter: likelihood ratio, candidate cut, tail
// function, decision.  Must be called by all 64 lanes of the wave together (a ballot skips the tail
// function when no lane needs it); lanes without a row pass valid = false.
__device__ __forceinline__ row_result evaluate_core(const filter_params& P, const double2* s_tab, const row_state& st)
{
  // table entry of each sum: { lf[k], log(k) }.  k = table_index(sum) (model.hpp:152-156);
  // sums beyond the table (or >= 2^31, where k wraps to 0 but lambda does not) take the
  // logarithm on the device.
  const uint32_t kc = kmd::table_index(st.sum_c);
  const uint32_t kk = kmd::table_index(st.sum_k);
  double2 tc = make_double2(0.0, 0.0), tk = make_double2(0.0, 0.0);
  if (kc < P.lds_n) tc = s_tab[kc]; else if (kc < P.lf_n) tc = P.tab[kc];
  if (kk < P.lds_n) tk = s_tab[kk]; else if (kk < P.lf_n) tk = P.tab[kk];
  const bool big_c = st.valid && st.sum_c >= P.lf_n;
  const bool big_k = st.valid && st.sum_k >= P.lf_n;
  if (big_c | big_k)                     // rare with the default 10000-entry table
  {
    if (big_c) { if (kc >= P.lf_n) tc.x = lf_beyond_table(kc); tc.y = ::log((double)st.sum_c); }
    if (big_k) { if (kk >= P.lf_n) tk.x = lf_beyond_table(kk); tk.y = ::log((double)st.sum_k); }
  }

  const double lr = kmd::lr_from_sums(st.sum_c, st.sum_k, tc.x, tk.x, tc.y, tk.y, P.dT, P.dTc, P.dTk);
  row_result R;
  R.cand = st.valid && (lr >= P.lr_cut);
  R.surv = false; R.near = false; R.p = 1.0; R.mean_control = 0.0; R.sign = KMD_SIGN_NO;
  if (__ballot(R.cand) && R.cand)
  {
    R.p = kmd::igamc_half(lr, P.lg_half);                   // model.hpp:161
    // the guard of the decision: a p-value within 1e-8 (relative) of the threshold is ~10 x closer than
    // the device's and a host's libm can move it apart.  Such a row is only FLAGGED here (and listed by the
    // caller, note_near_rows); k_resolve_near, a one-wave kernel behind every filter launch, repeats its
    // four libm calls with correctly rounded log / exp (kmd_ddmath.h) and corrects the sink if the
    // decision changes.  (Inlined here, that arithmetic cost every kernel that evaluates rows registers:
    // K1 spilled and lost 4 % at 20v20, 24 % at 4v4.)  Sums beyond the log-factorial table: the table term
    // used above is Stirling's, the reference's a k-term running sum; k_resolve_near repeats that sum for
    // sums below kChainMax, larger ones are left alone (no bit pattern to match at a bearable cost)
    // (a threshold of 1 or more keeps every row whatever its last bit -- the tail function never exceeds 1 --: nothing to
    // guard, and thousands of rows with p = 1 exactly would fill the list for nothing: tests/soak.py, round 5)
    R.near = P.threshold < 1.0 && fabs(R.p - P.threshold) <= 1e-8 * P.threshold && (!(big_c | big_k) || (st.sum_c < kChainMax && st.sum_k < kChainMax));
    R.surv = (R.p <= P.threshold);                          // merge.hpp:78
    kmd::sign_of(st.sum_c, st.sum_k, P.dTc, P.dTk, R.mean_control, R.sign);
  }
  return R;
}

// A flagged row goes on the launch's near list with the sink slot its (ocml) decision gave it, -1 if none.
__device__ __forceinline__ void note_near_row(const filter_params& P, const row_state& st, long long slot)
{
  if (!P.near) return;
  const unsigned long long at = atomicAdd(&P.near[0], 1ull);
  if (at >= kNearCap) return;                             // (counted by KMD_CNT_NEAR_THRESHOLD all the same)
  unsigned long long* e = P.near + 1 + 4 * at;
  e[0] = st.sum_c; e[1] = st.sum_k; e[2] = st.row; e[3] = (unsigned long long)slot;
}

// kRowMode 0: st.row is the row's index in the tile (survivor `row` = row_base + index).
// kRowMode 1: st.row indexes a list of candidate rows that have no index of their own (rows that come
//             out of the fused merge, kmd_tilemerge.hip): the survivor's `row` is the k-mer's low limb.
// Evaluation + compaction into the survivor sink (a ballot, one atomic per counter and wave).
template <int kRowMode = 0>
__device__ __forceinline__ void evaluate_row(const filter_params& P, const double2* s_tab, const row_state& st)
{
  const row_result R = evaluate_core(P, s_tab, st);
  const unsigned long long cand_mask = __ballot(R.cand);
  if (cand_mask)
  {
    const bool surv = R.surv;
    const unsigned long long near_mask = __ballot(R.near);
    const unsigned long long surv_mask = __ballot(surv);
    const unsigned long long ctrl_mask = __ballot(surv && R.sign == KMD_SIGN_CONTROL);
    const int lane = __lane_id();
    const int leader = __ffsll((long long)cand_mask) - 1;
    unsigned long long base = 0;
    if (lane == leader)
    {
      atomicAdd(&P.counters[KMD_CNT_CANDIDATES], (unsigned long long)__popcll(cand_mask));
      if (near_mask) atomicAdd(&P.counters[KMD_CNT_NEAR_THRESHOLD], (unsigned long long)__popcll(near_mask));
      if (surv_mask)
      {
        const unsigned long long ns = __popcll(surv_mask), nctl = __popcll(ctrl_mask);
        base = atomicAdd(&P.counters[KMD_CNT_SIG], ns);                    // merge.hpp:101
        if (nctl) atomicAdd(&P.counters[KMD_CNT_SIG_CONTROL], nctl);       // merge.hpp:95-96
        if (ns - nctl) atomicAdd(&P.counters[KMD_CNT_SIG_CASE], ns - nctl);// merge.hpp:97-98
      }
    }
    if (surv_mask)
      base = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(base >> 32), leader) << 32) |
             (unsigned)__builtin_amdgcn_readlane((int)base, leader);
    if (near_mask && R.near)
      note_near_row(P, st, surv ? (long long)(base + __popcll(surv_mask & ((1ull << lane) - 1ull))) : -1ll);
    if (surv_mask)
    {
      if (surv)
      {
        const unsigned long long slot =
            base + __popcll(surv_mask & ((1ull << lane) - 1ull));
        if (slot < P.out.capacity)
        {
          if (P.out.d_row) P.out.d_row[slot] = kRowMode == 1 ? P.kmer_lo[st.row] : P.row_base + st.row;
          if (P.out.d_kmer_lo && P.kmer_lo) P.out.d_kmer_lo[slot] = P.kmer_lo[st.row];
          if (P.out.d_kmer_hi && P.kmer_hi) P.out.d_kmer_hi[slot] = P.kmer_hi[st.row];
          if (P.out.d_pvalue) P.out.d_pvalue[slot] = R.p;
          if (P.out.d_sign) P.out.d_sign[slot] = R.sign;
          if (P.out.d_mean_control) P.out.d_mean_control[slot] = R.mean_control;
          if (P.out.d_mean_case) P.out.d_mean_case[slot] = (double)st.sum_k;
        }
      }
    }
  }
}

// one atomic per wave per launch for the beyond-table row count
__device__ __forceinline__ void flush_beyond(const filter_params& P, uint32_t n_beyond)
{
  for (int o = 32; o > 0; o >>= 1) n_beyond += __shfl_down(n_beyond, o, 64);
  if (__lane_id() == 0 && n_beyond)
    atomicAdd(&P.counters[KMD_CNT_DEFERRED], (unsigned long long)n_beyond);
}

} } // namespace kmd::eval

namespace kmd {
// host side: everything of filter_params that depends on the model and the threshold only (the
// candidate cut, the pre-filter switch); `t` carries the shared argument checks (kmd_filter.hip)
int fill_filter_params(eval::filter_params& P, const kmd_model* m, const kmd_tile* t, double threshold);
// The exact evaluation of the candidate rows the fused merge (kmd_tilemerge.hip) left in a device list:
// n entries (k-mer, control sum, case sum).  They passed the pre-filter there; `rows_total` (distinct
// k-mers of the partition) and `rows_beyond` (rows past the log-factorial table) are added to the counters.
// The near list of one filter launch (stream-ordered allocation) and the kernel that resolves it; both no-ops
// when the device has no memory pools.  begin: before the filter kernel (fills P.near); end: right after it.
int near_list_begin(eval::filter_params& P, hipStream_t stream);
int near_list_end(const eval::filter_params& P, int row_mode, hipStream_t stream);
// (d_work: filter_candidates_work_bytes(cap, m) bytes of device scratch, cap >= the list's entries)
size_t filter_candidates_work_bytes(size_t cap, const kmd_model* m);
int launch_filter_candidates(const eval::filter_params& P, const kmd_model* m, const uint64_t* d_kmer, const uint64_t* d_kmer_hi,
                             const uint64_t* d_sum_c, const uint64_t* d_sum_k, size_t n, uint64_t rows_total, uint64_t rows_beyond,
                             void* d_work, size_t cap, hipStream_t stream, const uint64_t* d_gate = nullptr,
                             const uint32_t* d_gate_over = nullptr, size_t gate_cap = 0);
}
