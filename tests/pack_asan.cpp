// Test harness (no GPU; built with -fsanitize=address,undefined from kmdiff_amd/csrc/kmd_pack_host.cpp alone): the host
// packer on buffers that end where the contract says they end.  kmd_pack_records takes the bytes an LZ4 decoder leaves of a
// kmtricks k-mer file -- exactly n x (8 + count_bytes) of them, at any alignment -- and its AVX2 path reads 96 bytes at a
// time: one byte read beyond the records, or written beyond kmd_pack_block_bound(), ends this program.  Its output must be
// kmd_pack_block's for the same records; kmd_pack_stream must be the blocks one behind the other in a buffer of exactly
// the size it reports.
//   pack_asan <iterations> <rng seed>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/kmdiff_hip.h"

static uint64_t rng_state;
static uint64_t rnd()
{
  rng_state += 0x9E3779B97F4A7C15ull;
  uint64_t x = rng_state;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

int main(int argc, char** argv)
{
  if (argc != 3) { std::fprintf(stderr, "usage: pack_asan <iterations> <rng seed>\n"); return 2; }
  const int iters = std::atoi(argv[1]);
  rng_state = std::strtoull(argv[2], nullptr, 10);
  const size_t bound = kmd_pack_block_bound();
  size_t blocks = 0, bytes = 0;
  for (int it = 0; it < iters; ++it)
  {
    const uint32_t cb = 1u << (rnd() % 3);                                  // 1, 2, 4
    const uint32_t n = (it % 7 == 0) ? 256u : 1u + (uint32_t)(rnd() % 256);
    const size_t rec = 8 + cb;
    const int width = 1 + (int)(rnd() % 63);                                // bits of a delta
    const uint64_t cmax = cb == 4 ? 0xFFFFFFFFull : (1ull << (8 * cb)) - 1;
    std::vector<uint64_t> km(n);
    std::vector<uint32_t> ct(n);
    uint64_t k = rnd() >> 8;
    for (uint32_t i = 0; i < n; ++i)
    {
      k += 1 + (rnd() & ((1ull << width) - 1));
      km[i] = k;
      ct[i] = (uint32_t)((rnd() % 11 == 0) ? cmax - rnd() % 3 : 1 + rnd() % (cmax < 300 ? cmax : 300));
    }
    // the records, in a heap block of exactly their size, at a misaligned start now and then
    const size_t shift = rnd() % 4;
    unsigned char* raw = (unsigned char*)std::malloc(shift + n * rec);
    for (uint32_t i = 0; i < n; ++i) { std::memcpy(raw + shift + i * rec, &km[i], 8); std::memcpy(raw + shift + i * rec + 8, &ct[i], cb); }
    unsigned char* a = (unsigned char*)std::malloc(bound);
    unsigned char* b = (unsigned char*)std::malloc(bound);
    std::memset(a, 0xAA, bound); std::memset(b, 0x55, bound);
    const size_t na = kmd_pack_block(km.data(), ct.data(), n, a);
    const size_t nb = kmd_pack_records(raw + shift, cb, n, b);
    if (na == 0 || na != nb || na > bound || (na & 7) || std::memcmp(a, b, na) != 0)
    {
      std::fprintf(stderr, "pack_asan: kmd_pack_records != kmd_pack_block (n = %u, count_bytes = %u, width %d: %zu / %zu bytes)\n", n, cb, width, nb, na);
      return 1;
    }
    std::free(raw); std::free(a); std::free(b);
    ++blocks; bytes += na;
    if (it % 16 == 0)
    {
      // a stream of a few blocks into a buffer of exactly the bytes it takes
      const size_t m = 1 + rnd() % 1500;
      std::vector<uint64_t> sk(m); std::vector<uint32_t> sc(m);
      uint64_t kk = 0;
      for (size_t i = 0; i < m; ++i) { kk += 1 + (rnd() & 0xFFFFFF); sk[i] = kk; sc[i] = 1 + (uint32_t)(rnd() % 500); }
      const size_t nblk = (m + KMD_PACK_BLOCK - 1) / KMD_PACK_BLOCK;
      std::vector<uint32_t> off(nblk);
      std::vector<unsigned char> big(nblk * bound);
      const size_t need = kmd_pack_stream(sk.data(), sc.data(), m, big.data(), big.size(), off.data());
      if (need == 0) { std::fprintf(stderr, "pack_asan: kmd_pack_stream refused a buffer of the bound's size\n"); return 1; }
      unsigned char* tight = (unsigned char*)std::malloc(need);
      uint32_t* off2 = (uint32_t*)std::malloc(nblk * 4);
      const size_t got = kmd_pack_stream(sk.data(), sc.data(), m, tight, need, off2);
      if (got != need || std::memcmp(tight, big.data(), need) != 0 || std::memcmp(off2, off.data(), nblk * 4) != 0)
      {
        std::fprintf(stderr, "pack_asan: kmd_pack_stream into a tight buffer differs (%zu records: %zu / %zu bytes)\n", m, got, need);
        return 1;
      }
      if (need > 8 && kmd_pack_stream(sk.data(), sc.data(), m, tight, need - 8, off2) != 0)
      {
        std::fprintf(stderr, "pack_asan: kmd_pack_stream accepted a buffer 8 bytes short\n");
        return 1;
      }
      std::free(tight); std::free(off2);
    }
  }
  std::printf("pack_asan ok: %zu blocks, %zu bytes\n", blocks, bytes);
  return 0;
}
