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
      {
        // the two other ways a caller can take the records (the CLI's packed transfer uses `raw`): chunk by chunk through
        // `consume` (arrays sized for a chunk, whole groups of 7 records taken, the rest carried), and as they are in the
        // file through `raw` -- both must deliver the file's records, in order, once
        std::vector<uint64_t> ck, ckh, got_k, got_kh; std::vector<uint32_t> cc, got_c;
        kmd_host::record_sink s2;
        s2.reserve = [&](kmd_host::record_sink& k, size_t n)
        {
          ck.resize(n); cc.resize(n); if (k.slots == 2) ckh.resize(n);
          k.kmers = ck.data(); k.counts = cc.data(); k.kmers_hi = k.slots == 2 ? ckh.data() : nullptr; k.capacity = n;
        };
        s2.consume = [&](kmd_host::record_sink& k, size_t held, bool last) -> size_t
        {
          const size_t take = last ? held : held / 7 * 7;
          got_k.insert(got_k.end(), k.kmers, k.kmers + take); got_c.insert(got_c.end(), k.counts, k.counts + take);
          if (k.kmers_hi) got_kh.insert(got_kh.end(), k.kmers_hi, k.kmers_hi + take);
          return take;
        };
        const kmd_host::kmer_file_info f2 = kmd_host::stream_kmer_file(in, 0, s2);
        same = same && f2.records == n2 && got_k.size() == n2 && got_c.size() == n2;
        for (size_t i = 0; same && i < n2; ++i) same = got_k[i] == km2[i] && got_c[i] == ct2[i] && (f.slots == 1 || got_kh[i] == kh2[i]);
        if (f.slots == 1)
        {
          std::vector<uint64_t> rk; std::vector<uint32_t> rc; size_t calls_after_end = 0; bool ended = false;
          kmd_host::record_sink s3;
          s3.raw = [&](const char* p, size_t n, uint32_t cb)
          {
            if (!p) { ended = true; return; }
            if (ended) ++calls_after_end;
            for (size_t i = 0; i < n; ++i)
            {
              uint64_t k; uint32_t c = 0;
              std::memcpy(&k, p + i * (8 + cb), 8); std::memcpy(&c, p + i * (8 + cb) + 8, cb);
              rk.push_back(k); rc.push_back(c);
            }
          };
          const kmd_host::kmer_file_info f3 = kmd_host::stream_kmer_file(in, 0, s3);
          same = same && ended && !calls_after_end && f3.records == n2 && rk.size() == n2;
          for (size_t i = 0; same && i < n2; ++i) same = rk[i] == km2[i] && rc[i] == ct2[i];
        }
      }
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
