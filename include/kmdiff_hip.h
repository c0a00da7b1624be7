/* kmdiff_hip.h -- C-ABI of libkmdiff_hip.so: the MI355X (gfx950) implementation of the
 * `kmdiff diff` hot path of tlemane/kmdiff v1.1.0.
 *
 * This is the drop-in boundary: plain C, pointers and sizes only, status codes instead of
 * exceptions.  Every entry point names the reference interface it replaces (paths are
 * relative to the reference tree).  INTEGRATION.md shows the binding a kmdiff maintainer
 * would add on the reference side.
 *
 * Conventions
 *   - "d_" arguments are DEVICE pointers (HBM); all others are host pointers.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *   - Functions that take a stream are asynchronous unless stated otherwise.
 *   - Every function returns KMD_OK (0) or a negative kmd_status; kmd_last_error() gives
 *     the message of the last failure on the calling thread.
 *   - Significance values are the reference's enum (include/kmdiff/kmer.hpp:33-38).
 */
#ifndef KMDIFF_HIP_H
#define KMDIFF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3: kmd_transport gained `abort`; kmd_transport_abort; (2 -> 3 also covers round 4's additions: kmd_correct_sharded,
 * kmd_pack_block(_bound), kmd_unpack_streams, kmd_pvalues_refine, kmd_synth_streams, kmd_transport_local_*).  A host
 * checks kmd_abi_version() == KMD_ABI_VERSION before it binds anything else. */
#define KMD_ABI_VERSION 3

typedef enum {
  KMD_OK = 0,
  KMD_E_INVALID = -1,      /* bad argument (NULL, unsupported count width, misaligned ...) */
  KMD_E_HIP = -2,          /* a HIP runtime call failed; see kmd_last_error() */
  KMD_E_NO_DEVICE = -3,    /* no gfx950 device visible */
  KMD_E_OVERFLOW = -4,     /* survivor capacity exceeded: counters are exact, records truncated */
  KMD_E_NOMEM = -5
} kmd_status;

/* Significance, include/kmdiff/kmer.hpp:33-38 */
enum { KMD_SIGN_CONTROL = 0, KMD_SIGN_CASE = 1, KMD_SIGN_NO = 2 };

/* CorrectionType, include/kmdiff/correction.hpp:7-14 */
enum { KMD_CORR_NOTHING = 0, KMD_CORR_BONFERRONI = 1, KMD_CORR_BENJAMINI = 2,
       KMD_CORR_SIDAK = 3, KMD_CORR_HOLM = 4 };

/* Count-matrix layouts accepted by the kernels.
 *   KMD_LAYOUT_SOA : counts[sample][row]   (column stride ld >= n_rows, in elements);
 *                    the device-native layout: one lane per row, fully coalesced.
 *   KMD_LAYOUT_ROWS: counts[row][sample]   (row stride ld >= nc+nk, in elements);
 *                    what km::MatrixReader / KmerMerger hand the observer
 *                    (include/kmdiff/merge.hpp:68,194-203); staged through LDS.
 *   KMD_LAYOUT_TILED: counts[row / T][sample][row % T] with T = ld rows per block
 *                    (T a multiple of 4096; the buffer holds ceil(n_rows / T) whole blocks; with 1-byte
 *                    counts prefer a multiple of 8192: 8-byte loads per lane, +16 %).
 *                    SoA inside a block, so lanes stay coalesced, but one block of rows is
 *                    one contiguous span of S*T counts: better DRAM page locality than S
 *                    far-apart columns (measured +3..10 % over plain SoA, HISTORY.md 4 K1). */
enum { KMD_LAYOUT_ROWS = 0, KMD_LAYOUT_SOA = 1, KMD_LAYOUT_TILED = 2 };

const char* kmd_status_string(int status);
const char* kmd_last_error(void);
int kmd_abi_version(void);
/* (the test hooks kmd_test_* are declared in kmdiff_hip_test.h: not part of the interface a host binds) */

/* ---- device plumbing (so that a C/C++ host needs nothing but this library) ----------- */
int kmd_device_count(int* n);
int kmd_set_device(int device);
int kmd_device_name(char* buf, size_t len);
int kmd_malloc(void** d_ptr, size_t bytes);
int kmd_free(void* d_ptr);
/* page-locked host memory for staging buffers (hipHostMalloc): host-to-device copies from it run
 * at the link's rate instead of through the driver's bounce buffers */
int kmd_malloc_host(void** h_ptr, size_t bytes);
int kmd_free_host(void* h_ptr);
int kmd_memcpy_h2d(void* d_dst, const void* src, size_t bytes, void* stream);
int kmd_memcpy_d2h(void* dst, const void* d_src, size_t bytes, void* stream);
int kmd_memset(void* d_dst, int value, size_t bytes, void* stream);
int kmd_stream_sync(void* stream);
/* A stream of the caller's own (non-blocking: it does not synchronise with the default stream the
 * compute entry points use when `stream` is NULL), and a host-to-device copy that only enqueues:
 * a host can upload the next partition from page-locked memory while the kernels of the current
 * one run.  kmd_stream_sync(stream) waits for the copies. */
int kmd_stream_create(void** stream);
int kmd_stream_destroy(void* stream);
int kmd_memcpy_h2d_async(void* d_dst, const void* src, size_t bytes, void* stream);
/* The library parks its internal scratch buffers (sort keys, flags, tallies) instead of
 * returning them to the driver after every call; this frees the parked ones (and the per-stream
 * near-threshold lists of the filter).  Call it with nothing in flight. */
int kmd_release_cache(void);
/* elapsed milliseconds of `fn`-independent timing helpers: HIP events on `stream` */
int kmd_event_create(void** ev);
int kmd_event_destroy(void* ev);
int kmd_event_record(void* ev, void* stream);
/* work enqueued on `stream` after this call waits for what `ev` recorded (hipStreamWaitEvent): copies on one stream,
 * the kernels that read them on another, no host round trip between */
int kmd_stream_wait_event(void* stream, void* ev);
int kmd_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms); /* syncs on ev_stop */

/* ---- the model ------------------------------------------------------------------------
 * Replaces PoissonLikelihood<MAX_C>'s constructor (include/kmdiff/model.hpp:106-118,
 * 185-191) and LogFactorialTable (include/kmdiff/log_factorial_table.hpp:9-26,
 * src/log_factorial_table.cpp:5-22): per-sample totals are summed into Tc, Tk; the
 * log-factorial table of `log_factorial_size` entries (CLI --log-factorial, default
 * 10000, src/cli.cpp:354-357) is built once with the reference's descending summation
 * and uploaded to HBM.  Sums >= log_factorial_size (LogFactorialTable::operator[]'s O(k)
 * fallback) are evaluated on the device: the same descending loop below 256, the Stirling
 * series of ln k! above (kmd_filter.hip, lf_beyond_table). */
typedef struct kmd_model kmd_model;

int kmd_model_create(kmd_model** out, int nb_controls, int nb_cases,
                     const uint64_t* total_controls, const uint64_t* total_cases,
                     size_t log_factorial_size);
int kmd_model_destroy(kmd_model* m);
/* introspection used by tests: host copy of the table, Tc/Tk */
int kmd_model_info(const kmd_model* m, int* nc, int* nk, uint64_t* tc, uint64_t* tk,
                   size_t* lf_n);
int kmd_model_lf_table(const kmd_model* m, double* out, size_t n);

/* ---- stage 1: merge observer + Poisson LRT + threshold + compaction ---------------------
 * Replaces diff_observer<KSIZE,CMAX>::process (include/kmdiff/merge.hpp:68-103) applied to
 * every row of one partition tile, i.e. the body of the per-partition task of
 * global_merge::merge (merge.hpp:259-307) after the k-way merge: for each row
 *   (p, sign, mean_control, mean_case) = PoissonLikelihood::process(controls, cases)
 *                                               (include/kmdiff/model.hpp:142-176)
 *   total++ ; if (p <= threshold) push KmerSign, ++n_sig, ++n_sig_control | ++n_sig_case.
 */

/* Device-resident survivor sink: replaces IAccumulator<KmerSign<KSIZE>>::push
 * (include/kmdiff/accumulator.hpp:36-54) with SoA device arrays of `capacity` records.
 * Any array pointer may be NULL (field not recorded).  Survivors are appended in
 * unspecified order; `row` (row_base + row index in the tile) restores the reference's
 * ascending-k-mer order (kmd_survivors_sort_by_row).
 * Sharing: a sink and its counters may be shared by calls on ONE stream (they accumulate), and by the partitions of one
 * kmd_merge_filter_batch call (the library orders what must be ordered); calls in flight at the same time on DIFFERENT
 * streams need sinks and counters of their own -- the pass over the rows within 1e-8 of the threshold may strike a
 * record and then compacts the whole sink, which nothing else may be appending to meanwhile. */
typedef struct {
  uint64_t* d_row;
  uint64_t* d_kmer_lo;      /* 2-bit packed k-mer, low 64 bits (k <= 32: the whole k-mer)  */
  uint64_t* d_kmer_hi;      /* high 64 bits for 32 < k <= 64, else NULL                     */
  double*   d_pvalue;
  int32_t*  d_sign;
  double*   d_mean_control; /* KmerSign::m_mean_control = sum_ctrl * Tk / Tc (model.hpp:165) */
  double*   d_mean_case;    /* KmerSign::m_mean_case    = raw case sum                        */
  size_t    capacity;
} kmd_survivors;

/* Device counters (uint64_t[KMD_NCOUNTERS], caller zeroes them once per partition/run;
 * calls accumulate).  [0..3] are diff_observer's m_total, m_sign_kmer_per_part,
 * m_sign_controls, m_sign_cases (merge.hpp:104-131). */
enum { KMD_CNT_TOTAL = 0, KMD_CNT_SIG = 1, KMD_CNT_SIG_CONTROL = 2, KMD_CNT_SIG_CASE = 3,
       KMD_CNT_CANDIDATES = 4,   /* rows whose tail function was evaluated                 */
       KMD_CNT_DEFERRED = 5,     /* rows with a count sum >= log_factorial_size            */
       KMD_CNT_NEAR_THRESHOLD = 6, /* candidates with |p / threshold - 1| <= 1e-8: decided with correctly rounded
                                      log / exp, so that no libm's last bit can move them across (kmd_ddmath.h) */
       KMD_CNT_NEAR_UNRESOLVED = 7, /* near-threshold rows beyond the 4096 a launch can list: they were counted in
                                       [6] but kept the device libm's decision -- a caller that wants the guard's
                                       guarantee for them reruns the partition in smaller pieces (0 in practice) */
       KMD_NCOUNTERS = 8 };

typedef struct {
  const void*     d_counts;   /* count matrix tile                                         */
  int             count_bytes;/* 1, 2 or 4: km::selectC<MAX_C>::type (imodel.hpp:27)       */
  int             layout;     /* KMD_LAYOUT_SOA | KMD_LAYOUT_ROWS | KMD_LAYOUT_TILED       */
  size_t          ld;         /* leading dimension in elements                             */
  const uint64_t* d_kmer_lo;  /* per-row k-mers (may be NULL: survivors carry only `row`)  */
  const uint64_t* d_kmer_hi;
  size_t          n_rows;
  uint64_t        row_base;   /* global index of row 0 within its partition                */
} kmd_tile;

/* TOLERANCE of the p-values this call (and kmd_merge_filter / _batch / kmd_poisson_filter_sums) writes to the sink:
 * the set of survivors, their k-mers, rows, signs, means and the counters are exact at every threshold; the p-values
 * are within 1e-10 (absolute) of the reference's for thresholds < 1.  At a threshold of 1 or more every row is kept, and
 * rows with p of order 1 and count sums of ~10^4 deviate by up to 1.1e-10 (4.7e-10 beyond the log-factorial table):
 * `k * log(lambda)` multiplies the last bit of the device's logarithm by the sum (PARITY.md 1).  A caller that needs
 * the 1e-10 bar there follows the call with kmd_pvalues_refine (below) on the sink, as the CLI and the IModel plugin do. */
int kmd_poisson_filter(const kmd_model* m, const kmd_tile* tile, double threshold,
                       const kmd_survivors* out, uint64_t* d_counters, void* stream);

/* Every row's result (no threshold): IModel<MAX_C>::process (imodel.hpp:36) over a tile.
 * Output arrays are device pointers of n_rows elements; any may be NULL. */
int kmd_poisson_process(const kmd_model* m, const kmd_tile* tile, double* d_pvalue,
                        int32_t* d_sign, double* d_mean_control, double* d_mean_case,
                        void* stream);

/* Sort the first n survivor records by `row` ascending (the order the reference pushes
 * them in, merge.hpp:100).  Synchronous. */
int kmd_survivors_sort_by_row(const kmd_survivors* s, size_t n, void* stream);
/* The same order for survivors that carry their k-mer but no row index (kmd_merge_filter): ascending
 * (d_kmer_hi, d_kmer_lo) -- the order the merge emits rows in.  Synchronous. */
int kmd_survivors_sort_by_kmer(const kmd_survivors* s, size_t n, void* stream);

/* p-values to the reference's last bit.  The filters take the two null-hypothesis logarithms of
 * PoissonLikelihood::process (model.hpp:155-156) and the exp / log inside igamc from the device's libm; where its last
 * bit differs from glibc's, `k * log(lambda)` (model.hpp:137) multiplies the difference by the count sum: the p-value
 * of a record then deviates from a glibc-built reference's by ~1e-16 x sum in LR (1e-10 absolute on p is reached at
 * sums of ~10^4 when p is of order 1).  This pass recomputes the first n p-values from the records' own means
 * (d_mean_case is the case sum; d_mean_control = sum_c Tk / Tc is inverted exactly) with correctly rounded log / exp:
 * they then carry the bits glibc >= 2.28 gives the reference wherever glibc itself returned the rounded value (all but
 * ~1 call in 10^3, where one ulp of a logarithm remains).  A sum >= log_factorial_size: the reference's table term there
 * is a k-term running sum (log_factorial_table.cpp:13-22), the filters' is Stirling's series; the pass repeats the
 * running sum itself, term for term in the reference's order, for sums below 2^20 (same bits; 0.1 us per 64 terms).  Sums
 * of 2^20 and more get the rounded logarithms but keep Stirling's term (within ~ulp(sum) of the reference: 5e-10
 * relative on p at sums of 10^6).  The DECISIONS need no such pass: rows within 1e-8 of the threshold are resolved inside every
 * filter call.  Works on any three arrays of that meaning (a sink's, or kmd_poisson_process's outputs).  Up to 2048
 * records: asynchronous; more: returns when the stream has drained (its work list goes back to the library's cache). */
int kmd_pvalues_refine(const kmd_model* m, size_t n, const double* d_mean_control, const double* d_mean_case,
                       double* d_pvalue, void* stream);

/* Gather the count vectors of survivors as doubles, KmerSign::m_counts_ratio
 * (merge.hpp:91-92): d_out[i*S + s] = (double) counts[row_i - row_base][s]. */
int kmd_survivors_gather_counts(const kmd_tile* tile, int n_samples, const uint64_t* d_rows,
                                size_t n, double* d_out, void* stream);

/* ---- stage 3: significance correction ---------------------------------------------------
 * Replaces make_corrector + aggregator::worker / sorted_aggregator::run
 * (src/corrector.cpp:6-116, include/kmdiff/aggregator.hpp:137-171,240-322): d_keep[i] = 1
 * when survivor i passes `correction` at level `threshold` with N = total_kmers.  BH and
 * Holm walk the survivors in ascending p and stop at the first rejection.  n_kept,
 * n_control, n_case are host outputs (control = sign CONTROL, case = everything else,
 * aggregator.hpp:155-162).  Synchronous. */
int kmd_correct(int correction, double threshold, uint64_t total_kmers,
                const double* d_pvalue, const int32_t* d_sign, size_t n, uint8_t* d_keep,
                uint64_t* n_kept, uint64_t* n_control, uint64_t* n_case, void* stream);

/* Sharded BH / Holm (one process per GPU): the ascending walk over ALL ranks' survivors is
 * reproduced from (1) a 4096-bin log-spaced histogram of the p-values (top 12 magnitude bits of
 * the double; all-gathered / summed over ranks, 32 KB each), (2) the first bin the walk cannot
 * accept wholesale, and (3) the exact walk over the p-values of that bin and the ones after it
 * only, started at the rank the earlier bins already consumed (kmdiff_amd/dist.py). */
int kmd_pvalue_histogram(const double* d_pvalue, size_t n, uint64_t* d_hist /* [4096], accumulates */,
                         void* stream);
int kmd_correct_critical_bin(int correction, double threshold, uint64_t total_kmers,
                             const uint64_t* d_hist, uint32_t* bin, uint64_t* n_before, void* stream);
/* kmd_correct with `rank_offset` applies already made (BH: m_rank starts at 1 + offset,
 * Holm: m_total at N - offset; src/corrector.cpp:27-35,68-71).  Synchronous. */
int kmd_correct_from_rank(int correction, double threshold, uint64_t total_kmers, uint64_t rank_offset,
                          const double* d_pvalue, const int32_t* d_sign, size_t n, uint8_t* d_keep,
                          uint64_t* n_kept, uint64_t* n_control, uint64_t* n_case, void* stream);

/* ---- stage 3 across GPUs: the one exchange step of a sharded run ---------------------
 * The reference sums its partitions' counters with std::accumulate (include/kmdiff/merge.hpp:316, 402-413) and pops ALL
 * survivors from one priority queue in ascending p, BH / Holm stopping at the first rejection (include/kmdiff/
 * aggregator.hpp:286-310, 325-339).  With partition p on rank p % N the survivors live on N devices; the global walk
 * is reproduced without moving them (kmd_shard.hip): all-reduce of the counters; for BH / Holm an all-gather of the
 * ranks' 4096-bin p-value histograms, the first bin the walk cannot accept wholesale, an all-gather of the p-values
 * from that bin on only, and the exact walk over those on every rank.
 *
 * The wire: two collectives over DEVICE buffers, called by every rank in the same order; each returns when the result
 * is in the caller's buffer (they synchronise `stream` themselves).  Two transports come with the library --
 * kmd_transport_local_create (N host threads of one process: `kmdiff-hip diff --devices N`) and, in
 * libkmdiff_hip_rccl.so, kmd_transport_rccl_init (one process per GPU: ncclAllReduce / ncclAllGather over xGMI); a
 * host with a wire of its own fills the struct itself (kmdiff_amd/dist.py does, with torch.distributed). */
typedef struct kmd_transport {
  void* ctx;
  int rank, world;
  /* d_buf[i] = sum over ranks of d_buf[i], i < n, in place */
  int (*allreduce_u64)(void* ctx, uint64_t* d_buf, size_t n, void* stream);
  /* d_recv[r * bytes .. (r + 1) * bytes) = rank r's d_send; `bytes` is the same on every rank */
  int (*allgather)(void* ctx, const void* d_send, void* d_recv, size_t bytes, void* stream);
  /* optional (may be NULL): this rank cannot go on -- the other ranks' pending and later collectives return an error
   * instead of waiting for it (kmd_correct_sharded calls it on every error return once world > 1) */
  void (*abort)(void* ctx);
} kmd_transport;

/* kmd_correct for the survivors of THIS rank in a run of t->world ranks (t == NULL: one rank):
 *   counters_local  : host, this rank's uint64_t[KMD_NCOUNTERS] (diff_observer's counters summed over its partitions)
 *   counters_global : host output (may be NULL), their sum over the ranks; [KMD_CNT_TOTAL] is the N of the correctors
 *   d_keep / n_kept / n_control / n_case : as kmd_correct, for this rank's n survivors
 * Every rank of the run must call it (the collectives inside are matched calls).  The decisions are those of
 * kmd_correct over the concatenation of all ranks' survivors.  Synchronous. */
int kmd_correct_sharded(const kmd_transport* t, int correction, double threshold, const uint64_t* counters_local,
                        uint64_t* counters_global, const double* d_pvalue, const int32_t* d_sign, size_t n,
                        uint8_t* d_keep, uint64_t* n_kept, uint64_t* n_control, uint64_t* n_case, void* stream);
/* The in-process transport: out[0 .. world) are the ranks' handles (shared state behind them); rank r's collectives
 * are called by a host thread of its own that has made rank r's device current (several ranks may share a device). */
int kmd_transport_local_create(int world, kmd_transport* out);
int kmd_transport_local_destroy(int world, kmd_transport* t);
/* A rank that fails OUTSIDE a collective (its device could not be set, an allocation failed before the exchange ...)
 * says so, so that the others do not wait for it for ever: t->abort(t->ctx) if the transport has one. */
int kmd_transport_abort(const kmd_transport* t);

/* ---- stage 0: k-way merge of one partition -------------------------------------------------
 * Replaces km::KmerMerger<KSIZE,CMAX>(paths, ab_mins = 1.., k, r_min = 1, save_if = 0).merge(obs)
 * as kmdiff drives it (include/kmdiff/merge.hpp:265-289; kmtricks is an un-vendored
 * dependency): S per-sample streams of (k-mer, count) records, each sorted by ascending
 * 2-bit-packed k-mer (A=0 C=1 T=2 G=3, first base most significant), are merged into the
 * count matrix: one row per distinct k-mer, ascending; count of each sample, 0 when absent.
 *   d_kmers / d_counts : all streams concatenated in sample order; d_kmers holds the low 64
 *                        bits of each k-mer, d_kmers_hi the high 64 bits (NULL for k <= 32)
 *   offsets            : host array of n_samples+1 record offsets into them
 *   d_matrix           : output, `layout`/`ld` as in kmd_tile, room for row_capacity rows
 *   d_kmer_out / d_kmer_hi_out : output k-mer columns (row_capacity entries), may be NULL
 *   n_rows_out         : host output, number of merged rows
 * Synchronous.  KMD_E_OVERFLOW (with *n_rows_out = rows needed) if row_capacity is too small. */
int kmd_merge_partition(int n_samples, const uint64_t* d_kmers, const uint64_t* d_kmers_hi,
                        const uint32_t* d_counts, const uint64_t* offsets, int count_bytes,
                        int layout, size_t ld, size_t row_capacity, void* d_matrix,
                        uint64_t* d_kmer_out, uint64_t* d_kmer_hi_out, uint64_t* n_rows_out,
                        void* stream);

/* ---- the compact transfer format of a partition's streams (host -> device) ----------------------------------
 * The reference streams a partition's LZ4 files straight into its merge (include/kmdiff/merge.hpp:265-266,
 * cmd/diff.hpp:92-95); here they are decoded on the host and cross PCIe.  A sample's stream is sorted, so it is sent
 * as blocks of KMD_PACK_BLOCK records -- first k-mer, bit-packed deltas of the block's widest delta, one-byte counts
 * with an escape list (kmd_pack.hip: 4-6.5 bytes per record instead of 12) -- and unpacked on the device into the
 * (k-mer, count) arrays kmd_merge_filter takes.  One-limb k-mers (k <= 32) only.
 *   kmd_pack_block : host, any thread: n <= KMD_PACK_BLOCK records of ONE stream -> out (room for
 *                    kmd_pack_block_bound() bytes); returns the bytes written, a multiple of 8 (0: bad arguments).
 *                    A stream is its blocks one after the other (all of KMD_PACK_BLOCK records but the last);
 *                    the caller notes where each begins: block_off8 = byte offset within the stream / 8.
 *   kmd_unpack_streams : d_packed = the streams' packed bytes, stream s at bytes [stream_base[s], stream_base[s + 1])
 *                    (host array of n_samples + 1 multiples of 8); d_block_off8 = the streams' block tables one after the other (stream s has
 *                    ceil(records / KMD_PACK_BLOCK) entries); offsets[n_samples + 1] = the streams' record offsets
 *                    (host), as kmd_merge_filter takes them: d_kmers / d_counts get offsets[n_samples] records.
 *                    Asynchronous on `stream` (put it behind the copies of the packed bytes).  A block whose header does
 *                    not describe exactly the bytes it has (damage in transit) comes out as zero records; nothing
 *                    outside a block's bytes is read. */
#define KMD_PACK_BLOCK 256
size_t kmd_pack_block_bound(void);
size_t kmd_pack_block(const uint64_t* kmers, const uint32_t* counts, uint32_t n, void* out);
/* The same block from records as a kmtricks k-mer file holds them (one-limb k-mers: [k-mer, 8 bytes][count, count_bytes = 1,
 * 2 or 4 bytes] one behind the other -- km::KmerWriter's payload, what an LZ4 decoder leaves of cmd/diff.hpp:92-95's files):
 * the decoder's bytes go in as they are, exactly n x (8 + count_bytes) of them are read.  Returns what kmd_pack_block
 * returns for the same records (0: bad arguments). */
size_t kmd_pack_records(const void* records, uint32_t count_bytes, uint32_t n, void* out);
/* A whole stream, block by block (host, any thread): n records -> out (room for out_capacity bytes; ceil(n / 256) x
 * kmd_pack_block_bound() always suffices), block_off8[ceil(n / 256)] = where each block begins / 8.  Returns the bytes
 * written (a multiple of 8); 0 with n > 0: out too small or bad arguments. */
size_t kmd_pack_stream(const uint64_t* kmers, const uint32_t* counts, size_t n, void* out, size_t out_capacity, uint32_t* block_off8);
int kmd_unpack_streams(int n_samples, const void* d_packed, const uint64_t* stream_base, const uint32_t* d_block_off8,
                       const uint64_t* offsets, uint64_t* d_kmers, uint32_t* d_counts, void* stream);

/* ---- stage 0 + 1 fused: streams in, survivors out ---------------------------------------------
 * km::KmerMerger<KSIZE,CMAX>::merge(diff_observer) as kmdiff runs it per partition (include/kmdiff/
 * merge.hpp:265-289 with the observer of :68-103): the S sorted streams are merged and every distinct
 * k-mer is tested at once.  All PoissonLikelihood::process reads of a row are the sum of its control
 * counts and the sum of its case counts (model.hpp:144-145), so no matrix is built: the streams are
 * read once (12 bytes per record), rows exist only in LDS (kmd_tilemerge.hip).
 *   m          : the model; n_samples must be its controls + cases, samples [0, nc) are the controls
 *                (merge.hpp:70-72); at most 1024 samples, fewer than 2^32-129 records in the partition and
 *                fewer than 2^29 records of any one sample (positions and run extents are 32-bit on the device)
 *   d_kmers_hi : high limbs for 32 < k <= 64, else NULL
 *   out        : survivor sink as in kmd_poisson_filter.  A row has no index here: `row` holds the low
 *                limb of the k-mer; kmd_survivors_sort_by_kmer gives the reference's ascending order
 *   d_counters : as kmd_poisson_filter (KMD_CNT_TOTAL += distinct k-mers of the partition)
 *   n_rows_out : host output (may be NULL), distinct k-mers of this partition
 * Any key distribution is taken (tiles of the key range that hold too many records are cut again on
 * the device; no fallback, no refusal).  Synchronous. */
int kmd_merge_filter(const kmd_model* m, int n_samples, const uint64_t* d_kmers, const uint64_t* d_kmers_hi,
                     const uint32_t* d_counts, const uint64_t* offsets, double threshold,
                     const kmd_survivors* out, uint64_t* d_counters, uint64_t* n_rows_out, void* stream);

/* A batch of partitions through kmd_merge_filter, two or three of them in flight on streams of the library's own: a job
 * (one ThreadPool task per partition, merge.hpp:259-307) has hundreds, and a quarter of a single call is not the merge
 * kernel -- index, probe and boundary searches, candidate evaluation, launches, the read-back.  Enqueued without a
 * host round trip, those run beside the merge kernel of another partition; the host waits once per partition.
 *   d_kmers / d_kmers_hi / d_counts / offsets / d_counters : arrays of n_partitions pointers, each as in
 *                 kmd_merge_filter (d_kmers_hi NULL, or NULL entries, for k <= 32; a partition of 0 records is skipped)
 *   out         : NULL or n_partitions sinks (entries may name the same sink and counters: calls accumulate)
 *   n_rows_out  : NULL or n_partitions host values, distinct k-mers per partition
 *   stream      : what it holds so far is waited for; the call returns when every partition is done
 * Same results as n_partitions single calls (survivor order within a sink aside).  A partition that needs the
 * slow way (tiles cut again, a candidate list that overflowed) is run again synchronously: nothing of its first
 * run has reached the counters or the sink.  Returns the first error met; the other partitions are still done. */
int kmd_merge_filter_batch(const kmd_model* m, int n_partitions, int n_samples, const uint64_t* const* d_kmers,
                           const uint64_t* const* d_kmers_hi, const uint32_t* const* d_counts,
                           const uint64_t* const* offsets, double threshold, const kmd_survivors* out,
                           uint64_t* const* d_counters, uint64_t* n_rows_out, void* stream);

/* The same merge for a consumer that wants the rows themselves: every distinct k-mer leaves as
 * (k-mer, sum of its control counts, sum of its case counts), 24 (two limbs: 32) bytes instead of
 * 8 + 4 S; compact, in no particular order (identify rows by their k-mer).  Samples [0, nb_controls)
 * are the controls.  d_kmers_hi / d_kmer_hi_out: NULL for k <= 32.
 *   *n_rows_out : distinct k-mers.  KMD_E_OVERFLOW (with *n_rows_out = rows needed, the first
 *                 row_capacity of them written) if row_capacity is too small.
 * kmd_poisson_filter_sums tests such rows; its survivors' `row` is the index into these arrays.
 * Synchronous. */
int kmd_merge_sums(int n_samples, int nb_controls, const uint64_t* d_kmers, const uint64_t* d_kmers_hi,
                   const uint32_t* d_counts, const uint64_t* offsets, size_t row_capacity, uint64_t* d_kmer_out,
                   uint64_t* d_kmer_hi_out, uint64_t* d_sum_control, uint64_t* d_sum_case, uint64_t* n_rows_out,
                   void* stream);
int kmd_poisson_filter_sums(const kmd_model* m, const uint64_t* d_kmer, const uint64_t* d_sum_control,
                            const uint64_t* d_sum_case, size_t n_rows, double threshold,
                            const kmd_survivors* out, uint64_t* d_counters, void* stream);
/* KmerSign::m_counts_ratio (merge.hpp:91-92) for survivors of the fused paths, which have no matrix to
 * gather from: d_out[i*S + s] = (double) count of k-mer d_row_kmer[j] (high limb d_row_kmer_hi[j]) in
 * sample s (0 when absent), looked up in the per-sample streams the merge was given; j = d_rows[i], or i
 * when d_rows is NULL (the survivors' own k-mer columns).  d_kmers_hi / d_row_kmer_hi: NULL for k <= 32.
 * Synchronous. */
int kmd_survivors_gather_counts_streams(int n_samples, const uint64_t* d_kmers, const uint64_t* d_kmers_hi,
                                        const uint32_t* d_counts, const uint64_t* offsets,
                                        const uint64_t* d_row_kmer, const uint64_t* d_row_kmer_hi,
                                        const uint64_t* d_rows, size_t n, double* d_out, void* stream);

/* ---- stage 2 (optional): population-stratification re-test ---------------------------------
 * Replaces pop_strat_corrector (include/kmdiff/popstrat.hpp:148-367): constructor
 * (src/popstrat.cpp:136-151), load_Z / load_Y (:153-171), init_global_features +
 * standardize (:270-370, quirks preserved) and the once-fitted null model
 * glm_irls(null features, Y) (:316-324, src/linear_model.cpp:297-410).
 *   Z : n x z_cols row-major principal components as read from pcs.evec (z_cols = 10,
 *       popstrat.hpp:152); the first `npc` columns are used (--n-pc, default 2)
 *   Y : n phenotypes, Case -> 0.0, Control -> 1.0 (popstrat.cpp:168)
 *   max_iter <= 0 selects the reference default 100 (popstrat.hpp:151)
 * Covariates and sex are not representable in the reference either (load_C never terminates
 * with a file, load_ginfo never finds a sex column: SURVEY.md 8a R9), so the design is
 * [1, PC_1..PC_npc, total, kmer_count/total]. */
typedef struct kmd_popstrat kmd_popstrat;
int kmd_popstrat_create(kmd_popstrat** out, int nb_controls, int nb_cases,
                        const uint64_t* total_controls, const uint64_t* total_cases,
                        const double* Z, int z_cols, int npc, const double* Y,
                        int standardize, int max_iter);
int kmd_popstrat_destroy(kmd_popstrat* ps);
/* --epsilon (src/cli.cpp:336-339 -> pop_strat_corrector::set_params, popstrat.hpp:162-175): a non-zero value
 * replaces the default 1e-30 below which |LLR| counts as zero (popstrat.hpp:321) */
int kmd_popstrat_set_epsilon(kmd_popstrat* ps, double epsilon);
/* introspection for tests: any output may be NULL.  alt_global is n x n_features_alt. */
int kmd_popstrat_info(const kmd_popstrat* ps, int* n_samples, int* n_features_alt, double* alt_global,
                      double* null_model, double* null_likelihood);
/* pop_strat_corrector::apply(KmerSign&) (popstrat.hpp:249-333) for n survivors:
 * d_pvalue[i] = chi2(1) tail of the logistic-regression likelihood ratio of survivor i.
 * d_counts: the survivors' count vectors as doubles (KmerSign::m_counts_ratio), either
 * survivor-major [n][S] as kmd_survivors_gather_counts writes them (sample_major = 0, ld
 * ignored) or sample-major [S][ld] (sample_major = 1, ld >= n). */
int kmd_popstrat_apply(const kmd_popstrat* ps, const double* d_counts, int sample_major, size_t ld,
                       size_t n, double* d_pvalue, void* stream);

/* ---- stage 2 front end: the principal components of the population structure ----------------
 * Replaces Sampler + EigGenoFile / EigSnpFile (include/kmdiff/popstrat.hpp:55-146: rows sampled
 * with probability --kmer-pca during stage 1, written as presence/absence), the external
 * `smartpca` run and evec2pca.perl (src/popstrat.cpp:97-134; parfile: usenorm YES,
 * numoutlieriter 0, numoutevec 10, popstrat.hpp:28-37) with smartpca's arithmetic as Hawk
 * modified it (thirdparty/hawk/EIG6.0.1-Hawk/src/eigensrc/smartpca.c: fvadjust :1694-1800,
 * getcolxz :2598-2700, main :880-1025): g = count > 0, x = (g - mean) / sqrt(p (1 - p)) with
 * p = 1 - sqrt(1 - mean) (diploid) or mean (`-V`), XTX = sum of x x^T over the sampled rows,
 * scaled by (n - 1) / trace, top eigenvectors at unit norm.
 * By nature not reproducible against the reference: it samples with one sequential RNG shared
 * by its partition threads; here a row is sampled iff hash(seed, k-mer) < rate.  The sign of
 * an eigenvector is the eigen-solver's there; here its largest component is positive. */
typedef struct kmd_pca kmd_pca;
/* capacity_rows: sampled rows the object holds to begin with (it grows by doubling) */
int kmd_pca_create(kmd_pca** out, int n_samples, double sample_rate, uint64_t seed, int diploid,
                   size_t capacity_rows);
void kmd_pca_destroy(kmd_pca* pca);
/* Sampler::sample over one tile (merge.hpp:150-152): records the presence pattern of the
 * sampled rows, in row order.  The tile needs its k-mer column. */
int kmd_pca_sample(kmd_pca* pca, const kmd_tile* tile, void* stream);
/* The same sampling for the fused merge (kmd_merge_filter: no matrix, no tile): the sampled k-mers are found
 * in the per-sample streams (arguments as kmd_merge_filter), in ascending k-mer order -- the rows, their
 * order and every sum over them are those kmd_pca_sample records from the merged matrix.  Synchronous. */
int kmd_pca_sample_streams(kmd_pca* pca, int n_samples, const uint64_t* d_kmers, const uint64_t* d_kmers_hi,
                           const uint32_t* d_counts, const uint64_t* offsets, void* stream);
int kmd_pca_count(const kmd_pca* pca, uint64_t* n_sampled);
/* xtx_host[S*S] (row-major, host) = sum over this object's sampled rows of x x^T, summed in a
 * fixed order.  Ranks add their matrices (rank order) before kmd_pca_eigen. */
int kmd_pca_gram(kmd_pca* pca, double* xtx_host, void* stream);
/* Scales xtx by (S - 1) / trace and solves it on the device (cyclic Jacobi, FP64):
 * eval_host[n_out] in decreasing order, evec_host[S][n_out] unit-norm columns -- the values
 * smartpca prints to .evec (pcs.evec holds them rounded to 4 decimals by evec2pca.perl). */
int kmd_pca_eigen(int n_samples, const double* xtx_host, int n_out, double* evec_host, double* eval_host);

/* ---- synthetic count matrices (benchmark / test support) ------------------
 * Counter-based generator, every cell a pure function of (seed, partition, row, sample);
 * the CPU oracle replays it.  d_kmer_lo / d_kmer_hi may be NULL.
 * partition: bits 0-7 the partition (< 256), bits 8-15 the PRESENCE PROFILE of the rows: 0 = SURVEY 8d's (every sample
 * absent with probability 0.3 in the two low rate classes: ~26 of 40 samples hold a row); KMD_SYNTH_MIXED (1) = every
 * second row RARE (present in one or two samples, counts of the two low rate classes) and the others COMMON (in 95 % of the samples) -- the shape of a
 * real partition's two populations, sample-specific k-mers and shared ones (bench.py's pipeline.sparse; not replayed
 * by the oracle: the device-built streams are held against the device-built matrix). */
#define KMD_SYNTH_MIXED 1u
#define KMD_SYNTH_PARTITION(partition, profile) ((uint32_t)(partition) | ((uint32_t)(profile) << 8))
int kmd_synth_fill(uint64_t seed, uint32_t partition, uint64_t row0, size_t n_rows, int nc,
                   int nk, int count_bytes, int layout, size_t ld, void* d_counts,
                   uint64_t* d_kmer_lo, uint64_t* d_kmer_hi, void* stream);
/* The same synthetic partition as kmtricks would hand it to the merge: per-sample (k-mer, count) streams -- the
 * records of sample s are the rows with a non-zero count in column s, in row order (ascending k-mer) --
 * concatenated in sample order, as kmd_merge_filter takes them.  Built on the device, chunk by chunk (no matrix of the
 * whole partition, no host copy: a configs[2] partition is 39 062 500 rows, ~10^9 records, 12 GB).  Two calls:
 *   d_kmers == NULL : offsets[0 .. nc+nk] (host) are computed; d_totals[s] += column sums (may be NULL);
 *   d_kmers != NULL : offsets is what the first call returned; d_kmers / d_counts (offsets[nc+nk] records) and
 *                     d_kmers_hi (two-limb k-mers: the high limbs; else NULL) are filled.
 * Synchronous. */
int kmd_synth_streams(uint64_t seed, uint32_t partition, uint64_t row0, size_t n_rows, int nc, int nk,
                      uint64_t* offsets, uint64_t* d_kmers, uint64_t* d_kmers_hi, uint32_t* d_counts,
                      uint64_t* d_totals, void* stream);
/* d_totals[s] += sum over rows of counts[.][s]  (uint64_t[nc+nk], caller zeroes);
 * the role of get_total_kmer (src/kmtricks_utils.cpp:78-139) for synthetic data. */
int kmd_column_sums(const void* d_counts, int count_bytes, int layout, size_t ld,
                    size_t n_rows, int n_samples, uint64_t* d_totals, void* stream);
/* Streaming-copy bandwidth probe (float4 copy of `bytes`), for the measured roofline. */
int kmd_copy_probe(void* d_dst, const void* d_src, size_t bytes, void* stream);
/* Streaming read of `bytes` with 4-, 8- or 16-byte loads per lane (width_bytes + 64: the same
 * with the non-temporal hint the filter kernel's column loads carry): a known byte count that
 * calibrates the FETCH_SIZE counter for the filter kernel's access width. */
int kmd_read_probe(const void* d_src, size_t bytes, int width_bytes, uint64_t* d_sink, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KMDIFF_HIP_H */
