// kmtricks_io.hpp -- host-side readers for the pieces of a kmtricks run directory that
// `kmdiff diff` consumes.  kmtricks is an un-vendored dependency of the reference; the byte
// layouts are the ones SURVEY.md 8f derives from the reference's fixture
// (tests/data_test/km_out_dir, kmtricks v1.1.1) and that tests/kmtricks_files.py round-trips.
//
// Replaces, for this path:
//   get_kmtricks_config   src/kmtricks_utils.cpp:29-69
//   get_total_kmer        src/kmtricks_utils.cpp:78-139   (km::HistReader, km::Fof)
//   get_partition_paths   src/kmtricks_utils.cpp:142-151  (km::KmDir::get_files_to_merge)
//   km::KmerReader / lz4_stream (kmtricks)
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace kmd_host {

struct kmtricks_config { size_t kmer_size = 0, abundance_min = 0, nb_partitions = 0; };

struct fof_entry { std::string id; size_t ab_min = 0; };

// kmdiff-count.opt (key abundance_min) or options.txt (key c_ab_min); #entries of counts/
kmtricks_config get_kmtricks_config(const std::string& run_dir);

// kmtricks.fof: "<id> : <files> [! <ab_min>]" per line, in sample order (controls first)
std::vector<fof_entry> read_fof(const std::string& run_dir);

// <run>/histograms/<id>.hist -> total k-mer abundance minus the abundances below ab_min
uint64_t sample_total(const std::string& run_dir, const fof_entry& sample, size_t abundance_min);

// <run>/counts/partition_<p>/<id>.kmer.lz4 -> records appended to kmers / counts
// (k <= 32: one 64-bit limb).  Returns the number of records read.
size_t read_kmer_file(const std::string& path, size_t expected_k, std::vector<uint64_t>& kmers,
                      std::vector<uint32_t>& counts);

std::string kmer_file_path(const std::string& run_dir, size_t partition, const std::string& id);

// 2-bit code A=0 C=1 T=2 G=3, first base most significant (km::Kmer::to_string)
std::string kmer_to_string(uint64_t kmer, size_t k);

} // namespace kmd_host
