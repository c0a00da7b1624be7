// Test harness (no GPU; built with -fsanitize=address,undefined): the host readers of kmdiff_amd/host/kmtricks_io.cpp on
// damaged files.  A reader may refuse a file (std::exception) or deliver fewer records; it may not read or write out of
// bounds, overflow, or hang.  Mutations of a seed file: truncation at a random length, 1..8 flipped bits, a block of
// random bytes, a header field overwritten with a large value.
//   io_fuzz kmers|matrix|survivors <seed file> <work file> <iterations> <rng seed>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>
#include <fstream>
#include <string>
#include <vector>

#include "../kmdiff_amd/host/kmtricks_io.hpp"

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
  if (argc != 6) { std::fprintf(stderr, "usage: io_fuzz kmers|matrix|survivors <seed file> <work file> <iterations> <rng seed>\n"); return 2; }
  const std::string what = argv[1], seed_path = argv[2], work = argv[3];
  const int iters = std::atoi(argv[4]);
  rng_state = std::strtoull(argv[5], nullptr, 10);
  std::ifstream in(seed_path, std::ios::binary);
  std::vector<char> seed((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  if (seed.size() < 64) { std::fprintf(stderr, "seed file too small\n"); return 2; }
  size_t refused = 0, read_ok = 0, records = 0;
  for (int it = 0; it < iters; ++it)
  {
    std::vector<char> f = seed;
    switch (rnd() % 4)
    {
      case 0: f.resize(rnd() % f.size()); break;                                            // truncated
      case 1: for (int b = 1 + (int)(rnd() % 8); b > 0; --b) f[rnd() % f.size()] ^= (char)(1u << (rnd() % 8)); break;
      case 2: { const size_t at = rnd() % f.size(), n = 1 + rnd() % 64; for (size_t i = at; i < f.size() && i < at + n; ++i) f[i] = (char)rnd(); break; }
      default: { const size_t at = rnd() % 64; const uint32_t v = (rnd() & 1) ? 0xFFFFFFFFu : (uint32_t)rnd(); if (at + 4 <= f.size()) std::memcpy(&f[at], &v, 4); break; }
    }
    { std::ofstream out(work, std::ios::binary | std::ios::trunc); out.write(f.data(), (std::streamsize)f.size()); }
    try
    {
      if (what == "kmers")
      {
        std::vector<uint64_t> km, kh; std::vector<uint32_t> ct;
        kmd_host::record_sink sink;
        sink.reserve = [&](kmd_host::record_sink& k, size_t n)
        {
          km.resize(n); ct.resize(n * k.nb_counts); if (k.slots == 2) kh.resize(n);
          k.kmers = km.data(); k.counts = ct.data(); k.kmers_hi = k.slots == 2 ? kh.data() : nullptr; k.capacity = n;
        };
        const kmd_host::kmer_file_info i1 = kmd_host::stream_kmer_file(work, 0, sink);
        records += i1.records;
        std::vector<uint64_t> k2, h2; std::vector<uint32_t> c2;
        records += kmd_host::read_kmer_file(work, 0, k2, c2, &h2);
      }
      else if (what == "matrix")
      {
        std::vector<uint64_t> km, kh; std::vector<uint32_t> ct;
        kmd_host::record_sink sink;
        sink.reserve = [&](kmd_host::record_sink& k, size_t n)
        {
          km.resize(n); ct.resize(n * k.nb_counts); if (k.slots == 2) kh.resize(n);
          k.kmers = km.data(); k.counts = ct.data(); k.kmers_hi = k.slots == 2 ? kh.data() : nullptr; k.capacity = n;
        };
        records += kmd_host::stream_matrix_file(work, sink).rows;
        records += kmd_host::read_matrix_file(work).kmers.size();
      }
      else
      {
        kmd_host::survivor_set s;
        records += kmd_host::read_survivor_file(work, s);
      }
      ++read_ok;
    }
    catch (const std::exception&) { ++refused; }
  }
  std::printf("%s: %d damaged files, %zu refused, %zu read (%zu records delivered)\n", what.c_str(), iters, refused, read_ok, records);
  return 0;
}
