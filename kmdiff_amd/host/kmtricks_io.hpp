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
#include <functional>
#include <cstdio>
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

// <run>/counts/partition_<p>/<id>.kmer.lz4 -> records appended to kmers / counts.  k <= 32: one
// 64-bit limb; 32 < k <= 64: two limbs, low limb first in the file (km::Kmer<64> holds one
// 128-bit integer and dumps its bytes; no fixture with k > 32 exists: unpinned), the high limbs
// go to kmers_hi (required then).  Returns the number of records read.
size_t read_kmer_file(const std::string& path, size_t expected_k, std::vector<uint64_t>& kmers,
                      std::vector<uint32_t>& counts, std::vector<uint64_t>* kmers_hi = nullptr);

// The same file in two steps, for hosts that place the records of many files in one buffer:
// the decoded payload with its layout, then the split of [k-mer limbs][count] records into arrays
// the caller owns (dst arrays hold at least `records` elements; hi may be NULL for one limb).
struct kmer_file_raw { std::vector<char> payload; uint32_t slots = 1, count_bytes = 4; size_t records = 0; };
kmer_file_raw decode_kmer_file(const std::string& path, size_t expected_k);
void split_records(const kmer_file_raw& f, uint64_t* kmers, uint64_t* kmers_hi, uint32_t* counts);

// The same file as a stream: the LZ4 frame is decoded chunk by chunk and the records go straight
// into arrays the caller owns and reuses from file to file (page-locked in the CLI), so that no
// pass over the data allocates.  `reserve(n)` must make the arrays hold at least n records (counts:
// n * nb_counts),
// keeping what is already there, and set the pointers (kmers_hi may stay NULL for one limb).
struct record_sink
{
  uint64_t* kmers = nullptr; uint64_t* kmers_hi = nullptr; uint32_t* counts = nullptr;
  size_t capacity = 0;                                    // records the arrays hold
  uint32_t slots = 1;                                     // limbs per k-mer of the file being read (set before reserve is called)
  uint32_t nb_counts = 1;                                 // counts per record: 1 for k-mer files, the samples of a matrix row
  std::function<void(record_sink&, size_t)> reserve;
  // optional (k-mer files): called after every decoded chunk with the number of records the arrays hold, from index 0;
  // returns how many of them, from the front, it has taken -- the rest moves to the front and the next chunk is put
  // behind it (last = the file is finished: what is held now is all there is).  With it the arrays need room for one
  // chunk, not for the file: the CLI packs the records for the transfer to the device as they are decoded.
  std::function<size_t(record_sink&, size_t held, bool last)> consume;
  // optional (k-mer files of one-limb k-mers): the decoded records as they are in the file -- n records of
  // [k-mer 8 bytes][count count_bytes], back to back -- instead of the split into arrays (reserve / consume are not
  // called then); a last call with n = 0 says the file is finished.  The CLI packs from these directly.
  std::function<void(const char* records, size_t n, uint32_t count_bytes)> raw;
  size_t file_size = 0;                                   // set by the reader before the first record: bytes of the file being read
  std::vector<char> in, out;                              // scratch of the decoder, kept between files
};
struct kmer_file_info { uint32_t slots = 1, count_bytes = 4; size_t records = 0; };
kmer_file_info stream_kmer_file(const std::string& path, size_t expected_k, record_sink& sink);
// <run>/matrices/* the same way: rows go to kmers[/kmers_hi] and, widened to 4 bytes, to
// counts[row * nb_counts + sample]; reserve(n) must provide n * nb_counts counts
struct matrix_file_info { uint32_t kmer_size = 0, slots = 1, count_bytes = 4, nb_counts = 0, partition = 0; size_t rows = 0; };
matrix_file_info stream_matrix_file(const std::string& path, record_sink& sink);

std::string kmer_file_path(const std::string& run_dir, size_t partition, const std::string& id);

// 2-bit code A=0 C=1 T=2 G=3, first base most significant (km::Kmer::to_string)
std::string kmer_to_string(uint64_t kmer, size_t k);
std::string kmer_to_string(uint64_t hi, uint64_t lo, size_t k);       // 32 < k <= 64

// ---- matrix files: <run>/matrices/* (the alternate feed, matrix_proxy::merge, merge.hpp:194-203)
// and positive_kmer_matrix/matrices/matrix_<p>.count.lz4 (--save-sk, merge.hpp:272-278).
// kmtricks source is absent and the reference holds no matrix fixture: the header is written BY
// ANALOGY with the k-mer file header the fixture pins (same 13-byte base header, then
// "matrix\0\0", u32 k, u32 kmer_slots, u32 count_bytes, u32 nb_counts, u32 id, u32 partition =
// 45 bytes) followed by one LZ4 frame of rows [u64 kmer x slots][count x nb_counts].  Unpinned.
struct matrix_rows
{
  uint32_t kmer_size = 0, count_bytes = 0, nb_counts = 0, partition = 0;
  std::vector<uint64_t> kmers;                 // low limb per row
  std::vector<uint64_t> kmers_hi;              // high limb per row when 32 < k <= 64, else empty
  std::vector<uint32_t> counts;                // [row][sample], widened
};
matrix_rows read_matrix_file(const std::string& path);
void write_matrix_file(const std::string& path, const matrix_rows& m);

// KffWriter (include/kmdiff/kff_utils.hpp:32-107) over kff-cpp-api's Kff_file / Section_GV / Section_Raw, written
// out by hand from the KFF 1.0 format description (kff-cpp-api is an un-vendored dependency of the reference;
// its pinned commit is not recoverable -- this writer is UNPINNED against it):
//   header   "KFF" | version 1.0 | encoding byte (A C G T -> 2 bits each; the reference passes {0, 1, 3, 2}:
//            0b00011110) | uniqueness 0 | canonicity 0 | free block size 0 (u32, big endian)
//   'v'      3 variables (u64 count, then name\0 + u64 value, big endian): k, max = 1, data_size = 0
//   'r'      u64 number of blocks; one block per k-mer: with max = 1 there is no count field, the sequence is
//            ceil(k / 4) bytes, 2 bits per nucleotide, first nucleotide most significant, right-aligned (the
//            FIRST byte holds the k % 4 leading nucleotides), no data bytes
//   footer   'v' with first_index = 0 and footer_size, then "KFF"
// The reference's codes (A 0, C 1, G 3, T 2) are kmtricks' own 2-bit codes, so a k-mer's compacted sequence is
// its packed value, big endian.
class kff_writer
{
 public:
  kff_writer(const std::string& path, size_t kmer_size);
  void write(uint64_t kmer_lo, uint64_t kmer_hi = 0);     // kmer_hi: bits 64.. for 32 < k <= 64
  void close();
  ~kff_writer() { if (m_open) close(); }
 private:
  struct impl;
  std::FILE* m_f = nullptr;
  size_t m_k = 0;
  uint64_t m_blocks = 0;
  long m_count_at = 0;
  bool m_open = false;
};
std::vector<std::string> matrix_paths(const std::string& run_dir);    // sorted; empty when there is no matrices/ content

// ---- survivor files: <out>/partitions/p<i>_uncorrected and p<i>_popstrat_uncorrected
// (FileAccumulator<KmerSign<KSIZE>>, accumulator.hpp:156-285): one LZ4 frame (lz4_stream) of
// records [kmer 8 B (16 B, low limb first, when 32 < k <= 64)][p f64][sign i32][mean_control f64][mean_case f64][n u16][n x f64 counts]
// (KmerSign::dump, kmer.hpp:113-127; the count block is there because WITH_POPSTRAT is ON by
// default, CMakeLists.txt:8).
struct survivor_set
{
  std::vector<uint64_t> kmer;
  std::vector<uint64_t> kmer_hi;               // filled when kmer_bytes == 16
  size_t kmer_bytes = 8;                       // 8 * ceil(KSIZE / 32): set before reading / writing
  std::vector<double> p, mean_control, mean_case;
  std::vector<int32_t> sign;
  std::vector<double> counts;                  // [record][n_counts]
  size_t n_counts = 0;
  size_t size() const { return p.size(); }
};
void write_survivor_file(const std::string& path, const survivor_set& s, size_t first, size_t count);
// appends the file's records to s (s.n_counts is set by the first record read)
size_t read_survivor_file(const std::string& path, survivor_set& s);

// ---- options.bin (dump_opt / load_opt / compare_opt, cmd/diff_opt.hpp:78-133): 37 bytes
struct resume_options
{
  double threshold = 0, cutoff = 0;
  int32_t correction = 0;
  bool pop_correction = false;
  double kmer_pca = 0;
  uint64_t npc = 0;
};
void dump_opt(const resume_options& o, const std::string& path);
bool load_opt(const std::string& path, resume_options& o);
// bit 0: redo stage 1, bit 1: redo the pop-strat stage, bit 2: redo the correction
unsigned compare_opt(const resume_options& opt, const resume_options& prev);

} // namespace kmd_host
