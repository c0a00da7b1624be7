// kmd_pack_host.cpp -- the host's side of the compact transfer format (kmd_pack.hip has the format and the device's
// unpack kernel): kmd_pack_block_bound, kmd_pack_block, kmd_pack_records, kmd_pack_stream.  Plain C++ (no device pass: the loops below are
// built twice, for AVX2 and for any x86-64, and picked at load time).
#include <algorithm>
#include <array>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <utility>

#include <immintrin.h>

#include "../../include/kmdiff_hip.h"

namespace {
constexpr uint32_t kBlock = KMD_PACK_BLOCK;          // records per block
constexpr uint32_t kEscape = 255;
} // namespace

extern "C" size_t kmd_pack_block_bound(void) { return 16 + (4 * 64 + 1) * 8 + kBlock + 4 * kBlock; }

// ---- the host's side: one block ------------------------------------------------------------------------------
// The decoder threads of `kmdiff-hip diff` pack while they decode: round 5 measured 31 % of a decoder thread's time here
// (2.8 ns per record: a shift, an or and a compare-and-branch per delta with a run-time width, a compare per count).
// Now (same bytes out -- tests/test_pack_format.py holds the format, the device's unpack kernel is unchanged):
//   * the width of a block's deltas is known before a bit is packed, so the packing itself is one of 64 functions
//     with the width as a template argument: 64 deltas -> W words in straight-line code, every shift an immediate,
//     no branch (64 deltas of W bits are W whole words: the block's four groups are packed independently);
//   * deltas + their OR, and the counts' narrowing to bytes, are loops the compiler vectorises (AVX2 where the CPU has
//     it: target_clones picks at load time); the escape list is collected in a second pass only if a count needs it.
namespace {

template <int W>
void pack_group(const uint64_t* __restrict__ d, uint64_t* __restrict__ out)
{
  if constexpr (W == 64) { std::memcpy(out, d, 64 * 8); return; }
  else
  {
    uint64_t acc = 0;
    int at = 0;
#pragma GCC unroll 64
    for (int j = 0; j < 64; ++j)
    {
      const int fill = (j * W) & 63;                         // (compile-time once unrolled)
      acc |= d[j] << fill;
      if (fill + W >= 64)
      {
        out[at++] = acc;
        acc = fill + W > 64 ? d[j] >> (64 - fill) : 0;
      }
    }
  }
}

using pack_group_fn = void (*)(const uint64_t*, uint64_t*);
template <int... Ws> constexpr std::array<pack_group_fn, sizeof...(Ws)> pack_table(std::integer_sequence<int, Ws...>)
{
  return { { &pack_group<Ws + 1>... } };
}
const std::array<pack_group_fn, 64> g_pack_group = pack_table(std::make_integer_sequence<int, 64>());

__attribute__((target_clones("avx2", "default")))
uint64_t block_deltas(const uint64_t* __restrict__ kmers, uint32_t n, uint64_t* __restrict__ delta)
{
  uint64_t all = 0;
  delta[0] = 0;
  for (uint32_t i = 1; i < n; ++i) { const uint64_t d = kmers[i] - kmers[i - 1]; delta[i] = d; all |= d; }
  for (uint32_t i = n; i < kBlock; ++i) delta[i] = 0;
  return all;
}

// counts as bytes (255 for everything from 255 up); returns the largest count
__attribute__((target_clones("avx2", "default")))
uint32_t block_counts(const uint32_t* __restrict__ counts, uint32_t n, unsigned char* __restrict__ cb)
{
  uint32_t top = 0;
  for (uint32_t i = 0; i < n; ++i)
  {
    const uint32_t c = counts[i];
    top = c > top ? c : top;
    cb[i] = (unsigned char)(c < kEscape ? c : kEscape);
  }
  return top;
}

} // namespace

extern "C" size_t kmd_pack_block(const uint64_t* kmers, const uint32_t* counts, uint32_t n, void* out)
{
  if (!kmers || !counts || !out || n == 0 || n > kBlock) return 0;
  unsigned char* o = static_cast<unsigned char*>(out);
  uint64_t delta[kBlock];
  const uint64_t all = block_deltas(kmers, n, delta);
  const uint32_t w = all ? 64u - (uint32_t)__builtin_clzll(all) : 0u;
  const uint32_t n_words = 4 * w + 1;
  std::memcpy(o, &kmers[0], 8);
  o[8] = (unsigned char)w; o[9] = 0; o[12] = o[13] = o[14] = o[15] = 0;
  uint64_t* words = reinterpret_cast<uint64_t*>(o + 16);
  if (w)
  {
    static_assert(kBlock == 256, "four groups of 64 deltas");
    const pack_group_fn fn = g_pack_group[w - 1];
    for (uint32_t g = 0; g < 4; ++g) fn(delta + 64 * g, words + (size_t)w * g);
  }
  words[n_words - 1] = 0;                                    // the slack word
  unsigned char* cb = o + 16 + (size_t)n_words * 8;
  uint32_t* esc = reinterpret_cast<uint32_t*>(cb + kBlock);
  uint32_t n_esc = 0;
  if (block_counts(counts, n, cb) >= kEscape)
    for (uint32_t i = 0; i < n; ++i) if (counts[i] >= kEscape) esc[n_esc++] = counts[i];
  std::memset(cb + n, 0, kBlock - n);
  const unsigned short ne = (unsigned short)n_esc;
  std::memcpy(o + 10, &ne, 2);
  size_t bytes = 16 + (size_t)n_words * 8 + kBlock + (size_t)n_esc * 4;
  while (bytes & 7) o[bytes++] = 0;
  return bytes;
}

// ---- records as a kmtricks k-mer file holds them: [k-mer, 8 bytes][count, 1 / 2 / 4 bytes], one behind the other -----
// What `kmdiff-hip diff`'s decoder threads hand over: the bytes the LZ4 decoder has just written.  Taking the records
// apart was a loop of two 8- and 4-byte copies per record in the caller -- 0.8 of the 1.55 ns a record cost to pack,
// more than the bit-packing itself.  With 4-byte counts (the reference's default build: MAX_C = 2^32 - 1,
// CMakeLists.txt:68-80) eight records are 96 bytes = three vectors: seven lane permutes and four blends put their
// k-mers into two vectors and their counts into one (AVX2; picked at load time).
namespace {

template <int CB>
void split_scalar(const unsigned char* q, uint32_t n, uint64_t* __restrict__ km, uint32_t* __restrict__ ct)
{
  for (uint32_t i = 0; i < n; ++i, q += 8 + CB)
  {
    std::memcpy(&km[i], q, 8);
    if constexpr (CB == 4) std::memcpy(&ct[i], q + 8, 4);
    else if constexpr (CB == 2) { uint16_t v; std::memcpy(&v, q + 8, 2); ct[i] = v; }
    else ct[i] = q[8];
  }
}

__attribute__((target("avx2")))
void split12_avx2(const unsigned char* q, uint32_t n, uint64_t* __restrict__ km, uint32_t* __restrict__ ct)
{
  // dwords of 8 records: record j = dwords 3j, 3j + 1 (k-mer), 3j + 2 (count); A = dwords 0-7, B = 8-15, C = 16-23
  const __m256i pk0a = _mm256_setr_epi32(0, 1, 3, 4, 6, 7, 0, 0), pk0b = _mm256_setr_epi32(0, 0, 0, 0, 0, 0, 1, 2);
  const __m256i pk1b = _mm256_setr_epi32(4, 5, 7, 0, 0, 0, 0, 0), pk1c = _mm256_setr_epi32(0, 0, 0, 0, 2, 3, 5, 6);
  const __m256i pca = _mm256_setr_epi32(2, 5, 0, 0, 0, 0, 0, 0), pcb = _mm256_setr_epi32(0, 0, 0, 3, 6, 0, 0, 0), pcc = _mm256_setr_epi32(0, 0, 0, 0, 0, 1, 4, 7);
  uint32_t j = 0;
  for (; j + 8 <= n; j += 8, q += 96)
  {
    const __m256i A = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(q));
    const __m256i B = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(q + 32));
    const __m256i C = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(q + 64));
    const __m256i k0 = _mm256_blend_epi32(_mm256_permutevar8x32_epi32(A, pk0a), _mm256_permutevar8x32_epi32(B, pk0b), 0xC0);
    const __m256i k1 = _mm256_blend_epi32(_mm256_permutevar8x32_epi32(B, pk1b), _mm256_permutevar8x32_epi32(C, pk1c), 0xF8);
    const __m256i c = _mm256_blend_epi32(_mm256_blend_epi32(_mm256_permutevar8x32_epi32(A, pca), _mm256_permutevar8x32_epi32(B, pcb), 0x1C),
                                         _mm256_permutevar8x32_epi32(C, pcc), 0xE0);
    _mm256_storeu_si256(reinterpret_cast<__m256i*>(km + j), k0);
    _mm256_storeu_si256(reinterpret_cast<__m256i*>(km + j + 4), k1);
    _mm256_storeu_si256(reinterpret_cast<__m256i*>(ct + j), c);
  }
  split_scalar<4>(q, n - j, km + j, ct + j);
}

using split_fn = void (*)(const unsigned char*, uint32_t, uint64_t*, uint32_t*);
split_fn pick_split12()
{
  __builtin_cpu_init();
  return __builtin_cpu_supports("avx2") ? &split12_avx2 : &split_scalar<4>;
}
const split_fn g_split12 = pick_split12();

} // namespace

extern "C" size_t kmd_pack_records(const void* records, uint32_t count_bytes, uint32_t n, void* out)
{
  if (!records || !out || n == 0 || n > kBlock) return 0;
  const unsigned char* q = static_cast<const unsigned char*>(records);
  uint64_t km[kBlock];
  uint32_t ct[kBlock];
  if (count_bytes == 4) g_split12(q, n, km, ct);
  else if (count_bytes == 2) split_scalar<2>(q, n, km, ct);
  else if (count_bytes == 1) split_scalar<1>(q, n, km, ct);
  else return 0;
  return kmd_pack_block(km, ct, n, out);
}

// a whole stream: its blocks one behind the other (what a host does while it decodes a sample's file)
extern "C" size_t kmd_pack_stream(const uint64_t* kmers, const uint32_t* counts, size_t n, void* out, size_t out_capacity, uint32_t* block_off8)
{
  if (n == 0) return 0;
  if (!kmers || !counts || !out || !block_off8) return 0;
  const size_t bound = kmd_pack_block_bound();
  char* o = static_cast<char*>(out);
  size_t at = 0, b = 0;
  alignas(8) char tmp[16 + (4 * 64 + 1) * 8 + kBlock + 4 * kBlock];
  for (size_t i = 0; i < n; i += kBlock, ++b)
  {
    const uint32_t m = (uint32_t)std::min<size_t>(kBlock, n - i);
    if (at / 8 > 0xFFFFFFFFull) return 0;                                  // (block_off8 is 32-bit: 32 GB of one stream)
    block_off8[b] = (uint32_t)(at / 8);
    if (out_capacity - at >= bound) at += kmd_pack_block(kmers + i, counts + i, m, o + at);
    else
    {
      const size_t got = kmd_pack_block(kmers + i, counts + i, m, tmp);    // the last blocks of a tight buffer: packed aside, copied if they fit
      if (got == 0 || got > out_capacity - at) return 0;
      std::memcpy(o + at, tmp, got);
      at += got;
    }
  }
  return at;
}

