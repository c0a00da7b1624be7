// kmd_shard.hip -- stage 3 across GPUs: the ONE exchange step of a sharded `kmdiff diff` run.
//
// The reference runs every partition in one address space, sums the partitions' counters with std::accumulate
// (include/kmdiff/merge.hpp:316, 402-413) and pops ALL survivors from one priority queue in ascending p
// (include/kmdiff/aggregator.hpp:286-310, 325-339), stopping BH / Holm at the first rejection.  With partition p on
// rank p % N the survivors live on N devices.  kmd_correct_sharded reproduces the global walk without moving them:
//   1. all-reduce of the counters: N = total k-mers, which every corrector needs (cmd/diff.hpp:249);
//   2. Bonferroni / Sidak / threshold: nothing more, every rank filters its own survivors (kmd_correct);
//   3. BH / Holm: all-gather of the ranks' 4096-bin p-value histograms (32 KB each), summed; the first bin the
//      ascending walk cannot accept wholesale (k_critical_bin, kmd_correct.hip) and the survivors before it;
//      only the p-values from that bin on are all-gathered (rank-major, local order kept) and walked exactly on
//      every rank (kmd_correct_from_rank, started at the rank the earlier bins consumed); every rank keeps its
//      own slice of the decisions.
// Ties: BH's and Holm's cuts grow with the rank, so a group of equal p-values is accepted or rejected as a whole
// wherever its members come from -- the decisions do not depend on how the ranks' lists are interleaved.
//
// The wire is a kmd_transport (two collectives on device buffers).  Two come with the library: an in-process one
// (N host threads of one process, one per GPU or all on one GPU: `kmdiff-hip diff --devices N`, the tests) here,
// and RCCL (one process per GPU: ncclAllReduce / ncclAllGather over xGMI) in libkmdiff_hip_rccl.so (kmd_rccl.cpp).
// A host with a wire of its own (torch.distributed in kmdiff_amd/dist.py, MPI ...) fills the struct itself.
#include "kmd_internal.h"

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

namespace {

constexpr uint32_t kBins = 4096;

__global__ void __launch_bounds__(256) k_sum_ranks(const unsigned long long* __restrict__ gathered, int world, size_t n,
                                                   unsigned long long* __restrict__ out)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long acc = 0;
  for (int r = 0; r < world; ++r) acc += gathered[(size_t)r * n + i];
  out[i] = acc;
}

// idx[0 .. *n_tail) = the indices i, ascending, of the p-values whose histogram bin is >= first_bin; one workgroup
// walks the list in order (survivors are few: this is microseconds) -- the order is part of the result
__global__ void __launch_bounds__(1024) k_tail_indices(const double* __restrict__ p, size_t n, uint32_t first_bin,
                                                       uint32_t* __restrict__ idx, unsigned long long* __restrict__ n_tail)
{
  __shared__ uint32_t s_wave[16];
  __shared__ unsigned long long s_base;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (size_t i0 = 0; i0 < n; i0 += 1024)
  {
    const size_t i = i0 + tid;
    bool in = false;
    if (i < n)
    {
      const unsigned long long bits = (unsigned long long)__double_as_longlong(p[i]);
      in = (uint32_t)((bits >> 51) & 4095ull) >= first_bin;
    }
    const unsigned long long m = __ballot(in);
    if (lane == 0) s_wave[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    unsigned long long at = s_base;
    for (uint32_t w = 0; w < wave; ++w) at += s_wave[w];
    if (in) idx[at + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)i;
    __syncthreads();
    if (tid == 0) { unsigned long long t = 0; for (int w = 0; w < 16; ++w) t += s_wave[w]; s_base += t; }
    __syncthreads();
  }
  if (tid == 0) *n_tail = s_base;
}

template <typename T>
__global__ void __launch_bounds__(256) k_take(const T* __restrict__ src, const uint32_t* __restrict__ idx, size_t n, T* __restrict__ dst)
{
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) dst[j] = src[idx[j]];
}

// the ranks' padded tails [world][pad] -> one list, rank-major: out[offs[r] + j] = in[r * pad + j], j < len[r]
template <typename T>
__global__ void __launch_bounds__(256) k_pack_tails(const T* __restrict__ in, size_t pad, const unsigned long long* __restrict__ offs, int world,
                                                    T* __restrict__ out)
{
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  if (r >= world) return;
  const unsigned long long len = offs[r + 1] - offs[r];
  if (j < len) out[offs[r] + j] = in[(size_t)r * pad + j];
}

// keep[i] = 1 everywhere (the bins before the critical one are accepted wholesale), then the tail's decisions;
// tallies[0] kept, [1] kept controls
__global__ void __launch_bounds__(256) k_keep_all(const int32_t* __restrict__ sign, size_t n, uint8_t* __restrict__ keep)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && keep) keep[i] = 1;
}

__global__ void __launch_bounds__(256) k_scatter_keep(const uint8_t* __restrict__ keep_tail, const uint32_t* __restrict__ idx, size_t n_tail,
                                                      uint8_t* __restrict__ keep)
{
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n_tail) keep[idx[j]] = keep_tail[j];
}

__global__ void __launch_bounds__(256) k_tally_keep(const uint8_t* __restrict__ keep, const int32_t* __restrict__ sign, size_t n,
                                                    unsigned long long* __restrict__ tallies)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool k = i < n && keep[i] != 0;
  const bool c = k && sign && sign[i] == KMD_SIGN_CONTROL;
  const unsigned long long km = __ballot(k), cm = __ballot(c);
  if ((threadIdx.x & 63) == 0 && km)
  {
    atomicAdd(&tallies[0], (unsigned long long)__popcll(km));
    if (cm) atomicAdd(&tallies[1], (unsigned long long)__popcll(cm));
  }
}

inline unsigned blocks_for(size_t n) { return (unsigned)std::max<size_t>(1, (n + 255) / 256); }

// scratch of one call: back to the cache when it dies (the caller has synchronised the stream on every way out)
struct scratch
{
  std::vector<void*> blocks;
  hipStream_t st;
  explicit scratch(hipStream_t s) : st(s) {}
  hipError_t take(void** out, size_t bytes)
  {
    const hipError_t e = kmd::scratch_alloc(out, bytes ? bytes : 1);
    if (e == hipSuccess) blocks.push_back(*out);
    return e;
  }
  ~scratch() { if (!blocks.empty()) (void)hipStreamSynchronize(st); for (void* b : blocks) kmd::scratch_free(b); }
};

#define KMD_T(call) do { const int rc__ = (call); if (rc__ != KMD_OK) return rc__; } while (0)

} // namespace

namespace {
int correct_sharded(const kmd_transport* t, int correction, double threshold, const uint64_t* counters_local,
                    uint64_t* counters_global, const double* d_pvalue, const int32_t* d_sign, size_t n,
                    uint8_t* d_keep, uint64_t* n_kept, uint64_t* n_control, uint64_t* n_case, void* stream)
{
  KMD_REQUIRE(counters_local, "kmd_correct_sharded: NULL counters");
  KMD_REQUIRE(correction >= KMD_CORR_NOTHING && correction <= KMD_CORR_HOLM, "kmd_correct_sharded: bad correction type");
  KMD_REQUIRE(n == 0 || d_pvalue, "kmd_correct_sharded: NULL p-values");
  KMD_REQUIRE(n < 0xFFFFFFFFull, "kmd_correct_sharded: too many survivors");
  const int world = t ? t->world : 1;
  KMD_REQUIRE(world >= 1 && (!t || (t->rank >= 0 && t->rank < world)), "kmd_correct_sharded: rank / world");
  KMD_REQUIRE(world == 1 || (t->allreduce_u64 && t->allgather), "kmd_correct_sharded: the transport lacks a collective");
  hipStream_t st = static_cast<hipStream_t>(stream);
  scratch sc(st);

  // 1. the counters (merge.hpp:316, 402-413)
  uint64_t g[KMD_NCOUNTERS];
  std::memcpy(g, counters_local, sizeof g);
  if (world > 1)
  {
    void* d_c = nullptr;
    KMD_HIP(sc.take(&d_c, sizeof g));
    KMD_HIP(hipMemcpyAsync(d_c, g, sizeof g, hipMemcpyHostToDevice, st));
    KMD_HIP(hipStreamSynchronize(st));
    KMD_T(t->allreduce_u64(t->ctx, static_cast<uint64_t*>(d_c), KMD_NCOUNTERS, stream));
    KMD_HIP(hipMemcpyAsync(g, d_c, sizeof g, hipMemcpyDeviceToHost, st));
    KMD_HIP(hipStreamSynchronize(st));
  }
  if (counters_global) std::memcpy(counters_global, g, sizeof g);
  const uint64_t total_kmers = g[KMD_CNT_TOTAL];

  // 2. one rank, or a corrector without memory: local
  if (world == 1 || (correction != KMD_CORR_BENJAMINI && correction != KMD_CORR_HOLM))
    return kmd_correct_from_rank(correction, threshold, total_kmers, 0, d_pvalue, d_sign, n, d_keep, n_kept, n_control, n_case, stream);

  // 3. BH / Holm: histograms -> critical bin -> the tails -> the exact walk
  void *d_hist = nullptr, *d_hists = nullptr;
  KMD_HIP(sc.take(&d_hist, kBins * 8));
  KMD_HIP(sc.take(&d_hists, (size_t)world * kBins * 8));
  KMD_HIP(hipMemsetAsync(d_hist, 0, kBins * 8, st));
  KMD_T(kmd_pvalue_histogram(d_pvalue, n, static_cast<uint64_t*>(d_hist), stream));
  KMD_HIP(hipStreamSynchronize(st));
  KMD_T(t->allgather(t->ctx, d_hist, d_hists, kBins * 8, stream));
  hipLaunchKernelGGL(k_sum_ranks, dim3(blocks_for(kBins)), dim3(256), 0, st, static_cast<const unsigned long long*>(d_hists), world, (size_t)kBins,
                     static_cast<unsigned long long*>(d_hist));
  KMD_HIP(hipGetLastError());
  uint32_t first_bin = 0;
  uint64_t before = 0;
  KMD_T(kmd_correct_critical_bin(correction, threshold, total_kmers, static_cast<const uint64_t*>(d_hist), &first_bin, &before, stream));

  // the local tail, in local order
  void *d_idx = nullptr, *d_small = nullptr;
  KMD_HIP(sc.take(&d_idx, std::max<size_t>(n, 1) * 4));
  KMD_HIP(sc.take(&d_small, ((size_t)world + 2) * 8 * 2));            // [n_tail | lengths of the ranks' tails | offsets]
  unsigned long long* d_ntail = static_cast<unsigned long long*>(d_small);
  hipLaunchKernelGGL(k_tail_indices, dim3(1), dim3(1024), 0, st, d_pvalue, n, first_bin, static_cast<uint32_t*>(d_idx), d_ntail);
  KMD_HIP(hipGetLastError());
  unsigned long long* d_lens = d_ntail + 1;
  KMD_HIP(hipStreamSynchronize(st));
  KMD_T(t->allgather(t->ctx, d_ntail, d_lens, 8, stream));
  std::vector<unsigned long long> lens((size_t)world), offs((size_t)world + 1, 0);
  KMD_HIP(hipMemcpyAsync(lens.data(), d_lens, (size_t)world * 8, hipMemcpyDeviceToHost, st));
  KMD_HIP(hipStreamSynchronize(st));
  unsigned long long pad = 1;
  for (int r = 0; r < world; ++r) { offs[(size_t)r + 1] = offs[(size_t)r] + lens[(size_t)r]; pad = std::max(pad, lens[(size_t)r]); }
  const size_t n_tail = (size_t)lens[(size_t)t->rank], n_all = (size_t)offs[(size_t)world];
  KMD_REQUIRE(n_all < 0xFFFFFFFFull, "kmd_correct_sharded: too many survivors in the tail");
  unsigned long long* d_offs = d_lens + world;
  KMD_HIP(hipMemcpyAsync(d_offs, offs.data(), ((size_t)world + 1) * 8, hipMemcpyHostToDevice, st));

  void *d_ps = nullptr, *d_ss = nullptr, *d_pg = nullptr, *d_sg = nullptr, *d_pa = nullptr, *d_sa = nullptr, *d_kt = nullptr;
  KMD_HIP(sc.take(&d_ps, (size_t)pad * 8)); KMD_HIP(sc.take(&d_ss, (size_t)pad * 4));
  KMD_HIP(sc.take(&d_pg, (size_t)world * pad * 8)); KMD_HIP(sc.take(&d_sg, (size_t)world * pad * 4));
  KMD_HIP(sc.take(&d_pa, std::max<size_t>(n_all, 1) * 8)); KMD_HIP(sc.take(&d_sa, std::max<size_t>(n_all, 1) * 4));
  KMD_HIP(sc.take(&d_kt, std::max<size_t>(n_all, 1)));
  KMD_HIP(hipMemsetAsync(d_ps, 0, (size_t)pad * 8, st));
  KMD_HIP(hipMemsetAsync(d_ss, 0, (size_t)pad * 4, st));
  if (n_tail)
  {
    hipLaunchKernelGGL((k_take<double>), dim3(blocks_for(n_tail)), dim3(256), 0, st, d_pvalue, static_cast<const uint32_t*>(d_idx), n_tail, static_cast<double*>(d_ps));
    if (d_sign)
      hipLaunchKernelGGL((k_take<int32_t>), dim3(blocks_for(n_tail)), dim3(256), 0, st, d_sign, static_cast<const uint32_t*>(d_idx), n_tail, static_cast<int32_t*>(d_ss));
    KMD_HIP(hipGetLastError());
  }
  KMD_HIP(hipStreamSynchronize(st));
  KMD_T(t->allgather(t->ctx, d_ps, d_pg, (size_t)pad * 8, stream));
  KMD_T(t->allgather(t->ctx, d_ss, d_sg, (size_t)pad * 4, stream));
  if (n_all)
  {
    hipLaunchKernelGGL((k_pack_tails<double>), dim3(blocks_for((size_t)pad), (unsigned)world), dim3(256), 0, st, static_cast<const double*>(d_pg), (size_t)pad,
                       d_offs, world, static_cast<double*>(d_pa));
    hipLaunchKernelGGL((k_pack_tails<int32_t>), dim3(blocks_for((size_t)pad), (unsigned)world), dim3(256), 0, st, static_cast<const int32_t*>(d_sg), (size_t)pad,
                       d_offs, world, static_cast<int32_t*>(d_sa));
    KMD_HIP(hipGetLastError());
  }
  // the walk over the gathered tail, the same on every rank (same list, same order), from the rank the earlier bins consumed
  KMD_T(kmd_correct_from_rank(correction, threshold, total_kmers, before, static_cast<const double*>(d_pa), static_cast<const int32_t*>(d_sa), n_all,
                              static_cast<uint8_t*>(d_kt), nullptr, nullptr, nullptr, stream));
  // this rank's decisions
  void* d_keep_own = nullptr;
  uint8_t* keep = d_keep;
  if (!keep) { KMD_HIP(sc.take(&d_keep_own, std::max<size_t>(n, 1))); keep = static_cast<uint8_t*>(d_keep_own); }
  void* d_t = nullptr;
  KMD_HIP(sc.take(&d_t, 16));
  KMD_HIP(hipMemsetAsync(d_t, 0, 16, st));
  uint64_t h_t[2] = { 0, 0 };
  if (n)
  {
    hipLaunchKernelGGL(k_keep_all, dim3(blocks_for(n)), dim3(256), 0, st, d_sign, n, keep);
    if (n_tail)
      hipLaunchKernelGGL(k_scatter_keep, dim3(blocks_for(n_tail)), dim3(256), 0, st, static_cast<const uint8_t*>(d_kt) + offs[(size_t)t->rank],
                         static_cast<const uint32_t*>(d_idx), n_tail, keep);
    hipLaunchKernelGGL(k_tally_keep, dim3(blocks_for(n)), dim3(256), 0, st, keep, d_sign, n, static_cast<unsigned long long*>(d_t));
    KMD_HIP(hipGetLastError());
    KMD_HIP(hipMemcpyAsync(h_t, d_t, 16, hipMemcpyDeviceToHost, st));
  }
  KMD_HIP(hipStreamSynchronize(st));
  if (n_kept) *n_kept = h_t[0];
  if (n_control) *n_control = h_t[1];
  if (n_case) *n_case = h_t[0] - h_t[1];
  return KMD_OK;
}
} // namespace

// Every rank calls this, and the collectives inside are matched calls: a rank that leaves early -- an argument refused,
// an allocation that failed, a collective that returned an error -- would leave the others waiting for it.  It tells them
// (kmd_transport::abort) on every error return once there is more than one rank.
extern "C" int kmd_correct_sharded(const kmd_transport* t, int correction, double threshold, const uint64_t* counters_local,
                                   uint64_t* counters_global, const double* d_pvalue, const int32_t* d_sign, size_t n,
                                   uint8_t* d_keep, uint64_t* n_kept, uint64_t* n_control, uint64_t* n_case, void* stream)
{
  const int rc = correct_sharded(t, correction, threshold, counters_local, counters_global, d_pvalue, d_sign, n, d_keep, n_kept, n_control, n_case, stream);
  if (rc != KMD_OK && t && t->world > 1 && t->abort) t->abort(t->ctx);
  return rc;
}

extern "C" int kmd_transport_abort(const kmd_transport* t)
{
  if (t && t->abort) t->abort(t->ctx);
  return KMD_OK;
}

// ---- the in-process transport: N host threads of one process, one per rank -----------------------------------
// Every rank's thread calls the collective; a generation barrier lines them up, the data moves with device
// copies (hipMemcpyPeer between devices, a plain copy on one).  Rank r's thread has made its device current.
namespace {

struct local_hub
{
  std::mutex mu;
  std::condition_variable cv;
  int world = 0, arrived = 0;
  unsigned long long generation = 0;
  std::vector<const void*> send;
  std::vector<int> dev;
  bool failed = false;                            // a rank's copy failed inside a collective (everybody still arrives)
  bool aborted = false;                           // a rank left for good (kmd_transport::abort): nobody waits any more
  int copying = 0;                                // ranks between the two barriers of a collective: reading the others' send buffers
  // false: a rank has given up -- the collective in progress, and every later one, fails on all ranks
  bool barrier()
  {
    std::unique_lock<std::mutex> lock(mu);
    if (aborted) return false;
    const unsigned long long gen = generation;
    if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); }
    else cv.wait(lock, [&] { return generation != gen || aborted; });
    return !aborted;
  }
  void abort()
  {
    std::lock_guard<std::mutex> lock(mu);
    aborted = true;
    cv.notify_all();
  }
  // a rank's copies out of the others' send buffers: counted in and out, so that a rank the abort wakes at the second
  // barrier does not return -- and have its caller free its send buffer -- while a peer's copy still reads it
  void copies_begin() { std::lock_guard<std::mutex> lock(mu); ++copying; }
  void copies_end() { std::lock_guard<std::mutex> lock(mu); --copying; cv.notify_all(); }
  void wait_for_copies() { std::unique_lock<std::mutex> lock(mu); cv.wait(lock, [&] { return copying == 0; }); }
};

struct local_rank
{
  std::shared_ptr<local_hub> hub;
  int rank = 0;
};

int local_allgather(void* ctx, const void* d_send, void* d_recv, size_t bytes, void* stream)
{
  local_rank* R = static_cast<local_rank*>(ctx);
  local_hub& H = *R->hub;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e == hipSuccess) e = hipStreamSynchronize(static_cast<hipStream_t>(stream));       // what is sent has been written
  {
    std::lock_guard<std::mutex> lock(H.mu);
    H.send[(size_t)R->rank] = d_send;
    H.dev[(size_t)R->rank] = dev;
    if (e != hipSuccess) H.failed = true;
  }
  if (!H.barrier()) { kmd::set_error("kmd_transport_local: another rank gave up"); return KMD_E_HIP; }
  // (on the caller's stream, and waited for there: a device-to-device hipMemcpy on the null stream may return before
  // the copy has run, and the callers' non-blocking streams are not ordered against the null stream)
  hipStream_t st = static_cast<hipStream_t>(stream);
  H.copies_begin();
  for (int q = 0; q < H.world && e == hipSuccess && bytes; ++q)
  {
    char* dst = static_cast<char*>(d_recv) + (size_t)q * bytes;
    if (H.dev[(size_t)q] == dev) e = hipMemcpyAsync(dst, H.send[(size_t)q], bytes, hipMemcpyDeviceToDevice, st);
    else e = hipMemcpyPeerAsync(dst, dev, H.send[(size_t)q], H.dev[(size_t)q], bytes, st);
  }
  const hipError_t e_sync = hipStreamSynchronize(st);           // (also after a failed enqueue: the copies before it have run)
  if (e == hipSuccess) e = e_sync;
  H.copies_end();
  if (e != hipSuccess) { std::lock_guard<std::mutex> lock(H.mu); H.failed = true; }
  const bool all_here = H.barrier();                           // nobody's send buffer is still being read
  if (!all_here)
  {
    // (woken by an abort: the ranks that passed the first barrier with this one may still be copying)
    H.wait_for_copies();
    kmd::set_error("kmd_transport_local: another rank gave up");
    return KMD_E_HIP;
  }
  if (e != hipSuccess) return kmd::hip_fail(e, "kmd_transport_local: copy", __FILE__, __LINE__);
  if (H.failed) { kmd::set_error("kmd_transport_local: another rank failed"); return KMD_E_HIP; }
  return KMD_OK;
}

int local_allreduce_u64(void* ctx, uint64_t* d_buf, size_t n, void* stream)
{
  local_rank* R = static_cast<local_rank*>(ctx);
  const int world = R->hub->world;
  void* d_all = nullptr;
  {
    const hipError_t e_alloc = kmd::scratch_alloc(&d_all, (size_t)world * n * 8);
    if (e_alloc != hipSuccess) { R->hub->abort(); return kmd::hip_fail(e_alloc, "kmd_transport_local: scratch", __FILE__, __LINE__); }
  }
  int rc = local_allgather(ctx, d_buf, d_all, n * 8, stream);
  if (rc == KMD_OK)
  {
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_sum_ranks, dim3(blocks_for(n)), dim3(256), 0, st, static_cast<const unsigned long long*>(d_all), world, n,
                       reinterpret_cast<unsigned long long*>(d_buf));
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) rc = kmd::hip_fail(e, "kmd_transport_local: sum", __FILE__, __LINE__);
  }
  else (void)hipStreamSynchronize(static_cast<hipStream_t>(stream));
  kmd::scratch_free(d_all);
  return rc;
}

void local_abort(void* ctx) { static_cast<local_rank*>(ctx)->hub->abort(); }

} // namespace

extern "C" int kmd_transport_local_create(int world, kmd_transport* out)
{
  KMD_REQUIRE(world >= 1 && world <= 1024 && out, "kmd_transport_local_create: arguments");
  auto hub = std::make_shared<local_hub>();
  hub->world = world;
  hub->send.assign((size_t)world, nullptr);
  hub->dev.assign((size_t)world, 0);
  for (int r = 0; r < world; ++r)
  {
    local_rank* R = new local_rank;
    R->hub = hub; R->rank = r;
    out[r].ctx = R; out[r].rank = r; out[r].world = world;
    out[r].allreduce_u64 = local_allreduce_u64;
    out[r].allgather = local_allgather;
    out[r].abort = local_abort;
  }
  return KMD_OK;
}

extern "C" int kmd_transport_local_destroy(int world, kmd_transport* t)
{
  if (!t) return KMD_OK;
  for (int r = 0; r < world; ++r) { delete static_cast<local_rank*>(t[r].ctx); t[r].ctx = nullptr; }
  return KMD_OK;
}
