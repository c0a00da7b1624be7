// kmd_pack.hip -- the compact transfer format of a partition's per-sample streams, and the kernel that unpacks it.
//
// The reference streams the LZ4 files of a partition straight into its merge (include/kmdiff/merge.hpp:265-266,
// cmd/diff.hpp:92-95).  Here the files are decoded on the host and the records cross PCIe: 12 bytes each as plain
// (k-mer, count) arrays -- 57 GB/s of link against 4 TB/s of merge kernel.  A sample's stream is sorted, so its keys
// are sent as deltas: per block of 256 records
//     [u64 first k-mer][u8 w][u8 0][u16 n_esc][u32 0]          16 bytes
//     [256 deltas of w bits, bit-packed, delta 0 = 0][8 bytes of slack]
//     [256 counts of one byte; 255 = "see the escape list"]
//     [n_esc x u32: the counts >= 255, in order][padding to 8 bytes]
// w = the width of the block's largest delta (k-mers are hashed into partitions by their minimizers: deltas of a
// sample's stream are ~2^62 / records, 22-45 bits), counts are almost always below 255.  4-6.5 bytes per record
// instead of 12.  kmd_pack_block (host, any thread) writes one block; k_unpack (one wave per block: a wave-wide
// prefix sum over the deltas) writes the 12-byte arrays kmd_merge_filter reads -- 0.2 ms of HBM traffic per 2 M-row
// partition against 5-6 ms of copy saved.  Differences are taken modulo 2^64: any key sequence round-trips (an
// unsorted one just packs badly).  One-limb k-mers (k <= 32) only; two-limb streams are sent as they are.
#include "kmd_internal.h"

#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace {

constexpr uint32_t kBlock = KMD_PACK_BLOCK;          // records per block
constexpr uint32_t kEscape = 255;

__device__ __forceinline__ unsigned long long wave_excl_scan_u64(unsigned long long v, uint32_t lane, unsigned long long& total)
{
  unsigned long long incl = v;
  for (int o = 1; o < 64; o <<= 1)
  {
    const unsigned long long t = __shfl_up(incl, (unsigned)o, 64);
    incl += lane >= (uint32_t)o ? t : 0ull;
  }
  total = __shfl(incl, 63, 64);
  return incl - v;
}

// One wave per block (four blocks per workgroup).  Lane l owns records 4 l .. 4 l + 3: their deltas come out of the
// block's bit-packed words (staged in LDS with coalesced loads), a wave-wide exclusive scan of the lanes' sums and
// the anchor give the keys; the counts' escapes are numbered by a second scan.
__global__ void __launch_bounds__(256) k_unpack(const unsigned char* __restrict__ packed, const uint32_t* __restrict__ block_off8,
                                                const unsigned long long* __restrict__ stream_base,   // [S + 1]: byte offset of stream s in `packed`; [S]: all bytes
                                                const unsigned long long* __restrict__ rec_off,       // [S + 1]: records
                                                const uint32_t* __restrict__ blk_prefix,              // [S + 1]: blocks
                                                uint32_t S, uint32_t n_blocks, unsigned long long* __restrict__ kmers,
                                                uint32_t* __restrict__ counts)
{
  __shared__ unsigned long long s_words[4][4 * 64 + 2];       // 256 deltas of up to 64 bits + the slack word
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t g = blockIdx.x * 4u + wave;
  if (g >= n_blocks) return;
  uint32_t lo = 0, hi = S;                                   // the stream of block g: the last s with blk_prefix[s] <= g
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (blk_prefix[mid] <= g) lo = mid; else hi = mid; }
  const uint32_t s = lo, b = g - blk_prefix[s];
  const unsigned long long n_s = rec_off[s + 1] - rec_off[s], first = (unsigned long long)b * kBlock;
  const uint32_t n = (uint32_t)(n_s - first < kBlock ? n_s - first : kBlock);
  const unsigned long long at = stream_base[s] + (unsigned long long)block_off8[g] * 8ull;
  // where the block ends: the next block of the stream, or the stream's end
  const unsigned long long end = g + 1 < blk_prefix[s + 1] ? stream_base[s] + (unsigned long long)block_off8[g + 1] * 8ull : stream_base[s + 1];
  const unsigned long long out = rec_off[s] + first + 4ull * lane;
  const unsigned char* blk = packed + at;
  uint32_t w = 0, n_esc = 0;
  bool sane = at + 16 <= end && end <= stream_base[S];
  if (sane)
  {
    w = blk[8]; n_esc = *reinterpret_cast<const unsigned short*>(blk + 10);
    // what kmd_pack_block wrote for this header, to the byte; anything else is damage: the block's records come out as
    // zeros -- wrong, but nothing outside the block is read and nothing outside its records written
    sane = w <= 64u && n_esc <= kBlock && at + ((16ull + (4ull * w + 1ull) * 8ull + kBlock + 4ull * n_esc + 7ull) & ~7ull) == end;
  }
  if (!sane)
  {
#pragma unroll
    for (int j = 0; j < 4; ++j) if (4u * lane + (uint32_t)j < n) { kmers[out + j] = 0ull; counts[out + j] = 0u; }
    return;
  }
  const unsigned long long anchor = *reinterpret_cast<const unsigned long long*>(blk);
  const unsigned long long* words = reinterpret_cast<const unsigned long long*>(blk + 16);
  const uint32_t n_words = 4u * w + 1u;
  unsigned long long* sw = s_words[wave];
  for (uint32_t i = lane; i < n_words; i += 64) sw[i] = words[i];
  const unsigned char* cb = blk + 16 + (size_t)n_words * 8;
  const uint32_t* esc = reinterpret_cast<const uint32_t*>(cb + kBlock);
  const uint32_t c4 = reinterpret_cast<const uint32_t*>(cb)[lane];          // this lane's four count bytes
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const unsigned long long mask = w >= 64 ? ~0ull : ((1ull << w) - 1ull);
  unsigned long long d[4], sum = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j)
  {
    const uint32_t bit = (4u * lane + (uint32_t)j) * w, at = bit >> 6, sh = bit & 63u;
    const unsigned long long w0 = sw[at], w1 = sw[at + 1];
    const unsigned long long v = w ? (((w0 >> sh) | (sh ? w1 << (64u - sh) : 0ull)) & mask) : 0ull;
    sum += v;                                                               // (modulo 2^64, as the packer subtracted)
    d[j] = sum;
  }
  unsigned long long total;
  const unsigned long long before = wave_excl_scan_u64(sum, lane, total) + anchor;
  uint32_t c[4], my_esc = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) { c[j] = (c4 >> (8 * j)) & 0xFFu; my_esc += c[j] == kEscape ? 1u : 0u; }
  unsigned long long esc_total;
  uint32_t e_at = (uint32_t)wave_excl_scan_u64(my_esc, lane, esc_total);
#pragma unroll
  for (int j = 0; j < 4; ++j) if (c[j] == kEscape) { c[j] = e_at < n_esc ? esc[e_at] : kEscape; ++e_at; }
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (4u * lane + (uint32_t)j < n) { kmers[out + j] = before + d[j]; counts[out + j] = c[j]; }
}

} // namespace

// (the host's side of the format -- kmd_pack_block_bound, kmd_pack_block, kmd_pack_stream -- is kmd_pack_host.cpp: plain C++)

// The kernel's small tables (where each stream starts, in bytes, records and blocks) travel through a page-locked
// ring kept per (device, stream): the upload is a true asynchronous copy -- a pageable source would make the call wait
// for the copies the caller has just enqueued on the stream -- and nothing is allocated per call.  A slot is reused
// four calls later; an event says its copy has been read (it has, long since: the wait is a formality).
namespace {
constexpr int kTableSlots = 4;
struct table_ring
{
  char* h = nullptr;                 // kTableSlots x bytes, page-locked
  char* d = nullptr;                 // kTableSlots x bytes, device
  size_t bytes = 0;
  hipEvent_t ev[kTableSlots] = {};
  bool used[kTableSlots] = {};
  int next = 0;
};
std::mutex g_ring_mu;
std::map<std::pair<int, hipStream_t>, table_ring> g_rings;

int ring_slot(hipStream_t st, size_t bytes, char** h, char** d, hipEvent_t* ev)
{
  int dev = 0;
  KMD_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_ring_mu);
  table_ring& R = g_rings[{ dev, st }];
  if (bytes > R.bytes)
  {
    for (int i = 0; i < kTableSlots; ++i) if (R.used[i]) { KMD_HIP(hipEventSynchronize(R.ev[i])); R.used[i] = false; }
    if (R.h) { (void)hipHostFree(R.h); (void)hipFree(R.d); R.h = nullptr; R.d = nullptr; }
    size_t cap = 4096;
    while (cap < bytes) cap <<= 1;
    void* p = nullptr;
    KMD_HIP(hipHostMalloc(&p, cap * kTableSlots, hipHostMallocDefault));
    R.h = static_cast<char*>(p);
    KMD_HIP(hipMalloc(&p, cap * kTableSlots));
    R.d = static_cast<char*>(p);
    R.bytes = cap;
    for (int i = 0; i < kTableSlots; ++i) if (!R.ev[i]) KMD_HIP(hipEventCreateWithFlags(&R.ev[i], hipEventDisableTiming));
  }
  const int i = R.next;
  R.next = (R.next + 1) % kTableSlots;
  if (R.used[i]) KMD_HIP(hipEventSynchronize(R.ev[i]));
  R.used[i] = true;
  *h = R.h + (size_t)i * R.bytes; *d = R.d + (size_t)i * R.bytes; *ev = R.ev[i];
  return KMD_OK;
}
} // namespace

// kmd_stream_destroy / kmd_release_cache: nothing of the caller's is in flight on the stream(s)
void kmd::unpack_tables_forget(hipStream_t stream, bool all)
{
  std::lock_guard<std::mutex> lock(g_ring_mu);
  for (auto it = g_rings.begin(); it != g_rings.end();)
    if (all || it->first.second == stream)
    {
      table_ring& R = it->second;
      for (int i = 0; i < kTableSlots; ++i) if (R.ev[i]) (void)hipEventDestroy(R.ev[i]);
      if (R.h) { (void)hipHostFree(R.h); (void)hipFree(R.d); }
      it = g_rings.erase(it);
    }
    else ++it;
}

extern "C" int kmd_unpack_streams(int n_samples, const void* d_packed, const uint64_t* stream_base, const uint32_t* d_block_off8,
                                  const uint64_t* offsets, uint64_t* d_kmers, uint32_t* d_counts, void* stream)
{
  KMD_REQUIRE(n_samples > 0 && n_samples <= 65536 && stream_base && offsets, "kmd_unpack_streams: arguments");
  const size_t S = (size_t)n_samples;
  std::vector<uint32_t> blk(S + 1, 0);
  for (size_t s = 0; s < S; ++s)
  {
    KMD_REQUIRE(offsets[s] <= offsets[s + 1], "kmd_unpack_streams: offsets must be ascending");
    KMD_REQUIRE((stream_base[s] & 7) == 0 && stream_base[s] <= stream_base[s + 1], "kmd_unpack_streams: stream_base must ascend in multiples of 8 bytes");
    const uint64_t nb = (offsets[s + 1] - offsets[s] + kBlock - 1) / kBlock;
    KMD_REQUIRE((uint64_t)blk[s] + nb < 0xFFFFFFFFull, "kmd_unpack_streams: too many blocks");
    blk[s + 1] = blk[s] + (uint32_t)nb;
  }
  const uint32_t n_blocks = blk[S];
  if (n_blocks == 0) return KMD_OK;
  KMD_REQUIRE(d_packed && d_block_off8 && d_kmers && d_counts, "kmd_unpack_streams: NULL device buffers");
  hipStream_t st = static_cast<hipStream_t>(stream);
  // [stream_base S + 1 | offsets S + 1 | blk_prefix S + 1 (u32)]: one upload
  const size_t words = (S + 1) + (S + 1) + (S + 2) / 2;
  char *h = nullptr, *d = nullptr;
  hipEvent_t ev = nullptr;
  int rc = ring_slot(st, words * 8, &h, &d, &ev);
  if (rc != KMD_OK) return rc;
  uint64_t* up = reinterpret_cast<uint64_t*>(h);
  std::memcpy(up, stream_base, (S + 1) * 8);
  std::memcpy(up + S + 1, offsets, (S + 1) * 8);
  std::memcpy(up + 2 * (S + 1), blk.data(), (S + 1) * 4);
  KMD_HIP(hipMemcpyAsync(d, h, words * 8, hipMemcpyHostToDevice, st));
  KMD_HIP(hipEventRecord(ev, st));
  const unsigned long long* d_base = reinterpret_cast<const unsigned long long*>(d);
  hipLaunchKernelGGL(k_unpack, dim3((n_blocks + 3) / 4), dim3(256), 0, st, static_cast<const unsigned char*>(d_packed), d_block_off8, d_base, d_base + S + 1,
                     reinterpret_cast<const uint32_t*>(d_base + 2 * (S + 1)), (uint32_t)S, n_blocks, reinterpret_cast<unsigned long long*>(d_kmers), d_counts);
  KMD_HIP(hipGetLastError());
  return KMD_OK;
}
