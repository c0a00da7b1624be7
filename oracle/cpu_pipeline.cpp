// oracle/cpu_pipeline.cpp -- TEST / BENCH INFRASTRUCTURE, never shipped and never on the product's path: the CPU doing the
// job `kmdiff-hip diff` does on the same run directory, as the reference does it -- one task per partition on a pool of
// T threads (global_merge::merge, include/kmdiff/merge.hpp:239-307), each task
//     LZ4-decode the partition's per-sample k-mer files   (km::KmerReader / lz4_stream; here: liblz4 through
//                                                          kmdiff_amd/host/kmtricks_io.cpp, the reader the CLI itself uses)
//     S-way merge into rows                               (km::KmerMerger::merge, merge.hpp:265-289 -> kmdo_merge_partition)
//     Poisson likelihood-ratio test + threshold per row   (diff_observer::process, merge.hpp:68-103 -> kmdo_diff_partition)
// so that tools/cli_throughput.py --cpu-baseline and bench.py --e2e can put a like-for-like CPU number beside the
// command's (missing 2).  Only tools/, tests/ and bench.py's baseline legs run it.
//
// usage: cpu_pipeline <run_dir> <nb_controls> <nb_cases> <threads> [threshold = 0.05 / 100000] [log_factorial = 10000]
// prints ONE JSON line: partitions, rows, records, survivors, seconds of stage 1, rows_per_s, records_per_s, threads
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../kmdiff_amd/host/kmtricks_io.hpp"
extern "C" {
#include "kmd_oracle.h"
}

int main(int argc, char** argv)
{
  if (argc < 5) { std::fprintf(stderr, "usage: %s <run_dir> <nb_controls> <nb_cases> <threads> [threshold] [log_factorial]\n", argv[0]); return 2; }
  const std::string run = argv[1];
  const int nc = std::atoi(argv[2]), nk = std::atoi(argv[3]);
  const int T = std::max(1, std::atoi(argv[4]));
  const double threshold = argc > 5 ? std::atof(argv[5]) : 0.05 / 100000.0;
  const size_t lf_n = argc > 6 ? (size_t)std::atoll(argv[6]) : 10000;
  try
  {
    const kmd_host::kmtricks_config cfg = kmd_host::get_kmtricks_config(run);
    const std::vector<kmd_host::fof_entry> fof = kmd_host::read_fof(run);
    const int S = nc + nk;
    if ((int)fof.size() != S) throw std::runtime_error("fof: " + std::to_string(fof.size()) + " samples, expected " + std::to_string(S));
    if (cfg.kmer_size > 32) throw std::runtime_error("k > 32: not covered by this baseline");
    uint64_t tc = 0, tk = 0;                                      // get_total_kmer, kmtricks_utils.cpp:78-139
    for (int s = 0; s < S; ++s) (s < nc ? tc : tk) += kmd_host::sample_total(run, fof[(size_t)s], cfg.abundance_min);
    std::vector<double> lf(lf_n);
    kmdo_lf_build(lf_n, lf.data());

    std::atomic<size_t> next { 0 };
    std::atomic<uint64_t> rows { 0 }, records { 0 }, survivors { 0 };
    std::mutex err_mu;
    std::string err;
    const auto t0 = std::chrono::steady_clock::now();
    auto worker = [&]()
    {
      std::vector<uint64_t> kmers, kmer_out, surv_row;
      std::vector<uint32_t> counts, matrix;
      std::vector<double> sp, smc, smk;
      std::vector<int32_t> ss;
      for (;;)
      {
        const size_t p = next.fetch_add(1);
        if (p >= cfg.nb_partitions) return;
        try
        {
          kmers.clear(); counts.clear();
          std::vector<uint64_t> offsets((size_t)S + 1, 0);
          for (int s = 0; s < S; ++s)
          {
            kmd_host::read_kmer_file(kmd_host::kmer_file_path(run, p, fof[(size_t)s].id), cfg.kmer_size, kmers, counts);
            offsets[(size_t)s + 1] = kmers.size();
          }
          const size_t n = kmers.size();
          matrix.resize(n * (size_t)S);                           // (a row per record at most)
          kmer_out.resize(n);
          const size_t n_rows = kmdo_merge_partition(S, kmers.data(), counts.data(), offsets.data(), matrix.data(), kmer_out.data(), n);
          const size_t cap = n_rows / 8 + 1024;
          surv_row.resize(cap); sp.resize(cap); smc.resize(cap); smk.resize(cap); ss.resize(cap);
          kmdo_counters cnt;
          std::memset(&cnt, 0, sizeof cnt);
          const size_t found = kmdo_diff_partition(matrix.data(), 4, KMDO_LAYOUT_ROWS, (size_t)S, n_rows, nc, nk, tc, tk, lf.data(), lf_n, threshold,
                                                   surv_row.data(), sp.data(), ss.data(), smc.data(), smk.data(), cap, &cnt);
          rows += n_rows; records += n; survivors += found;
        }
        catch (const std::exception& e)
        {
          std::lock_guard<std::mutex> lock(err_mu);
          if (err.empty()) err = e.what();
          return;
        }
      }
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < T; ++t) pool.emplace_back(worker);
    for (auto& th : pool) th.join();
    if (!err.empty()) throw std::runtime_error(err);
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("{\"partitions\": %zu, \"rows\": %llu, \"records\": %llu, \"survivors\": %llu, \"threads\": %d, \"seconds\": %.6f, "
                "\"rows_per_s\": %.6e, \"records_per_s\": %.6e, \"what\": \"liblz4 decode + kmdo_merge_partition + kmdo_diff_partition, one task per partition\"}\n",
                (size_t)cfg.nb_partitions, (unsigned long long)rows.load(), (unsigned long long)records.load(), (unsigned long long)survivors.load(), T, sec,
                (double)rows.load() / sec, (double)records.load() / sec);
    return 0;
  }
  catch (const std::exception& e)
  {
    std::fprintf(stderr, "cpu_pipeline: %s\n", e.what());
    return 1;
  }
}
