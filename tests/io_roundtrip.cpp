// Test harness (no GPU): reads a file with the host readers of kmdiff_amd/host/kmtricks_io.cpp and
// writes it back with the host writers, so that tests/test_host_io.py can compare both directions
// with the independent Python reader/writer of tests/kmtricks_files.py.
//   io_roundtrip matrix <in> <out> | survivors <in> <out> | survivors16 <in> <out> | options <in> <out> | kmers <in> <out> | kff <in> <out>
#include <cstdio>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

#include "../kmdiff_amd/host/kmtricks_io.hpp"

int main(int argc, char** argv)
{
  if (argc != 4) { std::fprintf(stderr, "usage: io_roundtrip matrix|survivors|options <in> <out>\n"); return 2; }
  const std::string what = argv[1], in = argv[2], out = argv[3];
  try
  {
    if (what == "matrix")
    {
      const kmd_host::matrix_rows m = kmd_host::read_matrix_file(in);
      if (!m.kmers_hi.empty()) std::printf("two-limb ");
      std::printf("rows=%zu k=%u count_bytes=%u nb_counts=%u partition=%u\n", m.kmers.size(), m.kmer_size, m.count_bytes,
                  m.nb_counts, m.partition);
      {
        // the streaming reader (what the CLI uses) must deliver the same rows
        std::vector<uint64_t> km, kh; std::vector<uint32_t> ct;
        kmd_host::record_sink sink;
        sink.reserve = [&](kmd_host::record_sink& k, size_t n)
        {
          km.resize(n); ct.resize(n * k.nb_counts); if (k.slots == 2) kh.resize(n);
          k.kmers = km.data(); k.counts = ct.data(); k.kmers_hi = k.slots == 2 ? kh.data() : nullptr; k.capacity = n;
        };
        const kmd_host::matrix_file_info f = kmd_host::stream_matrix_file(in, sink);
        bool same = f.rows == m.kmers.size() && f.nb_counts == m.nb_counts && f.count_bytes == m.count_bytes &&
                    f.partition == m.partition && f.kmer_size == m.kmer_size && (f.slots == 2) == !m.kmers_hi.empty();
        for (size_t i = 0; same && i < f.rows; ++i)
        {
          same = km[i] == m.kmers[i] && (f.slots == 1 || kh[i] == m.kmers_hi[i]);
          for (uint32_t s2 = 0; same && s2 < f.nb_counts; ++s2) same = ct[i * f.nb_counts + s2] == m.counts[i * m.nb_counts + s2];
        }
        if (!same) { std::fprintf(stderr, "error: stream_matrix_file differs from read_matrix_file\n"); return 1; }
      }
      kmd_host::write_matrix_file(out, m);
    }
    else if (what == "survivors" || what == "survivors16")
    {
      kmd_host::survivor_set s;
      s.kmer_bytes = what == "survivors16" ? 16 : 8;
      const size_t n = kmd_host::read_survivor_file(in, s);
      std::printf("records=%zu n_counts=%zu\n", n, s.n_counts);
      kmd_host::write_survivor_file(out, s, 0, n);
    }
    else if (what == "kmers")
    {
      // the streaming reader (what the CLI uses) into plain vectors, next to the whole-file reader;
      // <out> = the streamed arrays as raw little-endian bytes: kmers, kmers_hi (two limbs only), counts
      std::vector<uint64_t> km, kh; std::vector<uint32_t> ct;
      kmd_host::record_sink sink;
      sink.reserve = [&](kmd_host::record_sink& k, size_t n)
      {
        const bool two = k.slots == 2;
        km.resize(n); ct.resize(n); if (two) kh.resize(n);
        k.kmers = km.data(); k.counts = ct.data(); k.kmers_hi = two ? kh.data() : nullptr; k.capacity = n;
      };
      const kmd_host::kmer_file_info f = kmd_host::stream_kmer_file(in, 0, sink);
      std::vector<uint64_t> km2, kh2; std::vector<uint32_t> ct2;
      const size_t n2 = kmd_host::read_kmer_file(in, 0, km2, ct2, &kh2);
      bool same = n2 == f.records;
      for (size_t i = 0; same && i < n2; ++i)
        same = km[i] == km2[i] && ct[i] == ct2[i] && (f.slots == 1 || kh[i] == kh2[i]);
      std::printf("records=%zu slots=%u count_bytes=%u same=%d\n", f.records, f.slots, f.count_bytes, (int)same);
      std::FILE* o = std::fopen(out.c_str(), "wb");
      if (!o) return 1;
      std::fwrite(km.data(), 8, f.records, o);
      if (f.slots == 2) std::fwrite(kh.data(), 8, f.records, o);
      std::fwrite(ct.data(), 4, f.records, o);
      std::fclose(o);
    }
    else if (what == "kff")
    {
      // <in>: text, first line k, then one "lo hi" pair (decimal) per k-mer; <out>: the KFF file
      std::FILE* i = std::fopen(in.c_str(), "r");
      if (!i) return 1;
      unsigned long long k = 0, lo = 0, hi = 0;
      if (std::fscanf(i, "%llu", &k) != 1) return 1;
      kmd_host::kff_writer w(out, (size_t)k);
      size_t n = 0;
      while (std::fscanf(i, "%llu %llu", &lo, &hi) == 2) { w.write(lo, hi); ++n; }
      std::fclose(i);
      w.close();
      std::printf("kmers=%zu k=%llu\n", n, k);
    }
    else if (what == "options")
    {
      kmd_host::resume_options o, same;
      if (!kmd_host::load_opt(in, o)) { std::fprintf(stderr, "short options.bin\n"); return 1; }
      same = o;
      std::printf("threshold=%.17g cutoff=%.17g correction=%d pop=%d kmer_pca=%.17g npc=%llu action_same=%u\n", o.threshold, o.cutoff,
                  o.correction, (int)o.pop_correction, o.kmer_pca, (unsigned long long)o.npc, kmd_host::compare_opt(o, same));
      kmd_host::dump_opt(o, out);
    }
    else return 2;
  }
  catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
  return 0;
}
