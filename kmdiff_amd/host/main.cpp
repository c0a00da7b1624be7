// kmdiff-hip -- the `kmdiff diff` command over libkmdiff_hip.so.
//
// Keeps the flag set of the reference's `diff` sub-command (src/cli.cpp:148-362) and the stage
// order of main_diff (include/kmdiff/cmd/diff.hpp:262-377): do_diff (merge + Poisson test +
// threshold, per partition) -> optional do_pop (re-test of the survivors) -> do_correction
// (corrector + control/case split) -> control_kmers.fasta / case_kmers.fasta.  Every number is
// computed by the HIP library through the C-ABI of include/kmdiff_hip.h; this file only moves
// bytes (kmtricks files -> HBM, survivors -> FASTA) and parses flags.  No CPU compute path:
// without a device the command fails.
//
// Also kept: the alternate feed from <run>/matrices (matrix_proxy, merge.hpp:194-203), the
// survivor files partitions/p<i>_uncorrected / p<i>_popstrat_uncorrected with options.bin and
// the stage-skip logic of main_diff (--keep-tmp; cmd/diff.hpp:278-370), --save-sk.
//
// --pop-correction takes its principal components from the device PCA (kmd_pca_*: rows sampled
// at rate --kmer-pca during stage 1, smartpca's normalisation and eigen-decomposition; written to
// popstrat/pcs.evec in evec2pca's format) or, with --pcs FILE, from a file computed elsewhere.
//
// --cmodel / --config load a user's IModel plugin the way plugin_manager does; its process() is
// called row by row on the host (it is arbitrary host code), merge and correction stay on the GPU.
//
// Not carried over (out of scope): `count`/`infos` sub-commands, progress bars; --covariates is parsed and
// refused with the reason (the reference's load_C never terminates with a file, src/popstrat.cpp:207).
#include <algorithm>
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <ctime>
#include <filesystem>
#include <functional>
#include <future>
#include <mutex>
#include <thread>
#include <fstream>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include <dlfcn.h>

#include "../../include/kmdiff_hip.h"
#include "imodel_abi.hpp"
#include "kmtricks_io.hpp"

namespace fs = std::filesystem;
using namespace kmd_host;

namespace {

struct diff_options                       // include/kmdiff/cmd/diff_opt.hpp:6-40
{
  std::string kmtricks_dir, output_directory = "./kmdiff_output", correction = "bonferroni", pcs;
  std::string model_lib_path, model_config;          // --cmodel / --config: a user's IModel plugin (cli.cpp:246-262)
  size_t nb_controls = 0, nb_cases = 0, cutoff = 100000, log_size = 10000, npc = 2, max_iteration = 0;
  double threshold = 0.05, epsilon = 0.0;
  bool pop_correction = false, stand = true, keep_tmp = false, save_sk = false, kff = false;
  bool matrix_path = std::getenv("KMD_HOST_MATRIX") != nullptr;   // --matrix-path: k-way merge into the count matrix, then K1 (default: merge fused with the test)
  double kmer_pca = 0.001;                  // proportion of k-mers sampled for the PCA (cli.cpp:286-289)
  size_t ploidy = 2, seed = 0;              // cli.cpp:298-302, :349-351
  size_t threads = std::max(1u, std::thread::hardware_concurrency());   // -t: host threads decoding the k-mer files (cli.cpp:72-76)
  bool verbose_timing = std::getenv("KMD_HOST_TIMING") != nullptr;   // dev: where stage 1 spends its time
  bool raw_transfer = std::getenv("KMD_RAW_TRANSFER") != nullptr;    // --raw-transfer: the streams cross PCIe as plain 12-byte records (default: packed, kmd_pack_block)
  int device = 0, devices = 1, verbose = 1;   // first GPU, number of GPUs (0 = all): partition p on GPU (device + p % devices)
};

// Errors travel as exceptions: a GPU worker thread reports through its worker_result (run_gpu_worker), the
// main thread prints and exits from main() -- nobody calls exit() while other workers, decoder tasks or
// copies from page-locked memory are still alive.
// KMD_HOST_TIMING: CPU time (CLOCK_THREAD_CPUTIME_ID -- not wall time: under a CPU quota a thread is off the core for much of
// its wall time) the decoder threads spent in all, and of it packing records for the transfer (kmd_pack_records)
std::atomic<uint64_t> g_pack_ns { 0 }, g_decode_ns { 0 };
const bool g_time_pack = std::getenv("KMD_HOST_TIMING") != nullptr;
inline uint64_t thread_cpu_ns()
{
  timespec ts;
  clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
  return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

[[noreturn]] void die(const std::string& msg)
{
  throw std::runtime_error(msg);
}

void ck(int rc, const char* what)
{
  if (rc != KMD_OK) die(std::string(what) + ": " + kmd_last_error() + " (" + kmd_status_string(rc) + ")");
}

void usage()
{
  std::puts("kmdiff-hip diff -d DIR -1 INT -2 INT [-o DIR] [-s FLOAT] [-u INT] [-c STR] [--log-factorial INT]\n"
            "                [--pop-correction --pcs FILE [--n-pc INT] [--max-iteration INT]] [--device INT] [-v STR]\n"
            "  -d/--km-run        kmtricks run directory\n"
            "  -o/--output-dir    output directory {./kmdiff_output}\n"
            "  -1/--nb-controls   number of controls\n"
            "  -2/--nb-cases      number of cases\n"
            "  -s/--significance  significance threshold {0.05}\n"
            "  -u/--cutoff        divide the significance threshold by N for the first pass {100000}\n"
            "  -c/--correction    bonferroni|benjamini|sidak|holm|disabled {bonferroni}\n"
            "  --log-factorial    size of the log-factorial table {10000}\n"
            "  --pop-correction   re-test the survivors with the population-stratification model\n"
            "  --pcs FILE         principal components computed elsewhere (pcs.evec: one row per sample, 10 columns);\n"
            "                     default: the device PCA over k-mers sampled at rate --kmer-pca {0.001}\n"
            "  --ploidy INT       2: diploid normalisation of the PCA, else haploid {2}\n"
            "  --random-seed INT  seed of the PCA row sampler {0}\n"
            "  --n-pc             number of principal components in [2, 10] {2}\n"
            "  --cmodel FILE      a model plugin (shared library exporting plugin_name / create32, imodel.hpp); its\n"
            "                     process() is called row by row ON THE HOST, as the reference does; --config STR is\n"
            "                     handed to its configure()\n"
            "  --device           GPU index (the first one with --devices) {0}\n"
            "  --devices INT      number of GPUs, 0 = all: partition p goes to GPU p mod N {1}\n"
            "  --keep-tmp         keep partitions/p<i>_uncorrected (+ options.bin): a later run resumes from them\n"
            "  --save-sk          write the significant rows to positive_kmer_matrix/matrices/matrix_<p>.count.lz4\n"
            "  --matrix-path      k-way merge into the count matrix, then the test (default: merge fused with the test, no\n"
            "                     matrix -- same output; --cmodel and the matrices/ feed always take the matrix path)\n"
            "  --raw-transfer     send the decoded streams to the GPU as plain (k-mer, count) arrays, 12 bytes per record\n"
            "                     (default for k <= 32: packed on the host as they are decoded -- first k-mer + bit-packed deltas\n"
            "                     + one-byte counts per 256 records, ~4-6 bytes per record -- and unpacked on the GPU; same output)\n"
            "  -t/--threads INT   host threads decoding the per-sample k-mer files {all}\n"
            "  -f/--kff-output    control_kmers.kff / case_kmers.kff (k-mers only) instead of the two FASTA files\n"
            "  --covariates FILE  known to the reference's command line; refused (its load_C never terminates with a file)\n"
            "  -m, -r: accepted for compatibility, ignored");
}

diff_options parse(int argc, char** argv)
{
  diff_options o;
  if (argc < 2 || std::string(argv[1]) != "diff") { usage(); std::exit(argc < 2 ? 1 : (std::string(argv[1]) == "--help" ? 0 : 1)); }
  auto need = [&](int& i) -> std::string { if (i + 1 >= argc) die(std::string("missing value for ") + argv[i]); return argv[++i]; };
  for (int i = 2; i < argc; ++i)
  {
    const std::string a = argv[i];
    if (a == "-d" || a == "--km-run") o.kmtricks_dir = need(i);
    else if (a == "-o" || a == "--output-dir") o.output_directory = need(i);
    else if (a == "-1" || a == "--nb-controls") o.nb_controls = std::stoull(need(i));
    else if (a == "-2" || a == "--nb-cases") o.nb_cases = std::stoull(need(i));
    else if (a == "-s" || a == "--significance") o.threshold = std::stod(need(i));
    else if (a == "-u" || a == "--cutoff") o.cutoff = std::stoull(need(i));
    else if (a == "-c" || a == "--correction") o.correction = need(i);
    else if (a == "--log-factorial") o.log_size = std::stoull(need(i));
    else if (a == "--pop-correction") o.pop_correction = true;
    else if (a == "--pcs") o.pcs = need(i);
    else if (a == "--cmodel") o.model_lib_path = need(i);
    else if (a == "--config") o.model_config = need(i);
    else if (a == "--n-pc") o.npc = std::stoull(need(i));
    else if (a == "--max-iteration") o.max_iteration = std::stoull(need(i));
    else if (a == "--device") o.device = std::stoi(need(i));
    else if (a == "--devices") o.devices = std::stoi(need(i));
    else if (a == "--keep-tmp") o.keep_tmp = true;
    else if (a == "--save-sk") o.save_sk = true;
    else if (a == "--no-matrix") {}                       // (the default now)
    else if (a == "--matrix-path") o.matrix_path = true;
    else if (a == "--raw-transfer") o.raw_transfer = true;
    else if (a == "--kmer-pca") o.kmer_pca = std::stod(need(i));
    else if (a == "--ploidy") o.ploidy = std::stoull(need(i));
    else if (a == "--random-seed") o.seed = std::stoull(need(i));
    else if (a == "-t" || a == "--threads") o.threads = std::max<size_t>(1, std::stoull(need(i)));
    else if (a == "--epsilon") o.epsilon = std::stod(need(i));         // -> pop_strat_corrector::s_epsilon (popstrat.hpp:162-175,321)
    else if (a == "-v" || a == "--verbose" || a == "--gender" || a == "--learning-rate") (void)need(i);
    else if (a == "--covariates")                                     // cli.cpp:310-315 (hidden, --pop-correction builds only)
    {
      // Known to the reference's command line, but not to its pop-strat code: load_C's inner loop over a sample's
      // covariates never ends once a file is given (src/popstrat.cpp:207: `for (j = 0; ctmp[...].size(); j++)`), so no
      // reference run with --covariates has ever produced a result to be matched.  Refused, with the reason.
      const std::string f = need(i);
      if (!fs::is_regular_file(f)) die("--covariates: " + f + " is not a file");       // (bc::check::is_file)
      die("--covariates " + f + ": not supported -- the reference's pop_strat_corrector::load_C never terminates with a covariates "
          "file (src/popstrat.cpp:207), so there is no reference behaviour to reproduce; run without it");
    }
    else if (a == "-f" || a == "--kff-output") o.kff = true;          // cli.cpp: control_kmers.kff / case_kmers.kff
    else if (a == "-m" || a == "--in-memory" || a == "-r" || a == "--cpr" ||
             a == "--stand" || a == "--irls") {}
    else if (a == "-h" || a == "--help") { usage(); std::exit(0); }
    else die("unknown option " + a);
  }
  if (o.kmtricks_dir.empty()) die("-d/--km-run is required");
  if (!o.nb_controls || !o.nb_cases) die("-1/--nb-controls and -2/--nb-cases are required");
  if (!(o.threshold >= 0.0 && o.threshold <= 1.0)) die("-s/--significance must be in [0, 1]");          // cli.cpp:191-199
  static const std::map<std::string, int> ok = { {"bonferroni", 1}, {"benjamini", 2}, {"sidak", 3}, {"holm", 4}, {"disabled", 0} };
  if (!ok.count(o.correction)) die("-c/--correction must be bonferroni|benjamini|sidak|holm|disabled");
  if (!(o.kmer_pca > 0.0 && o.kmer_pca <= 0.05)) die("--kmer-pca must be in (0, 0.05]");                 // cli.cpp:289
  if (o.npc < 2 || o.npc > 10) die("--n-pc must be in [2, 10]");
  return o;
}

int correction_type(const std::string& c)
{
  return c == "bonferroni" ? KMD_CORR_BONFERRONI : c == "benjamini" ? KMD_CORR_BENJAMINI :
         c == "sidak" ? KMD_CORR_SIDAK : c == "holm" ? KMD_CORR_HOLM : KMD_CORR_NOTHING;
}

// fmt's "{}" for a double (aggregator.hpp:51-55 formats m_mean_case with it): the shortest digits
// that round-trip, written in fixed notation -- without a trailing ".0" -- when the decimal exponent is
// in [-4, 16), in exponent notation (at least two exponent digits) outside; 10 -> "10", 1200 -> "1200",
// 1e16 -> "1e+16", 0.5 -> "0.5".  (fmt's write_float with the default exponent thresholds; the pinned
// fmt commit is not in the tree, 7.0+ behaviour assumed.)
std::string shortest(double v)
{
  if (v == 0) return std::signbit(v) ? "-0" : "0";
  if (!std::isfinite(v)) return std::isnan(v) ? "nan" : (v < 0 ? "-inf" : "inf");
  char buf[64];
  int prec = 1;
  for (; prec <= 17; ++prec)
  {
    std::snprintf(buf, sizeof buf, "%.*e", prec - 1, v);
    if (std::strtod(buf, nullptr) == v) break;
  }
  // buf = [-]d[.ddd]e[+-]XX
  std::string t(buf);
  const bool neg = t[0] == '-';
  if (neg) t.erase(0, 1);
  const size_t e = t.find('e');
  const int exp10 = std::atoi(t.c_str() + e + 1);
  std::string digits = t.substr(0, e);
  digits.erase(std::remove(digits.begin(), digits.end(), '.'), digits.end());
  while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
  std::string out;
  if (exp10 >= -4 && exp10 < 16)
  {
    if (exp10 < 0) out = "0." + std::string((size_t)(-exp10 - 1), '0') + digits;
    else if ((size_t)exp10 + 1 >= digits.size()) out = digits + std::string((size_t)exp10 + 1 - digits.size(), '0');
    else out = digits.substr(0, (size_t)exp10 + 1) + "." + digits.substr((size_t)exp10 + 1);
  }
  else
  {
    out = digits.substr(0, 1);
    if (digits.size() > 1) out += "." + digits.substr(1);
    char eb[16];
    std::snprintf(eb, sizeof eb, "e%c%02d", exp10 < 0 ? '-' : '+', std::abs(exp10));
    out += eb;
  }
  return neg ? "-" + out : out;
}

// Timer (src/time.cpp:8-49): wall time of a stage, logged like cmd/diff.hpp:158,197,221,372
struct stopwatch
{
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  double seconds() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

struct dev_buf
{
  void* p = nullptr; size_t cap = 0;
  void reserve(size_t bytes) { if (bytes > cap) { if (p) kmd_free(p); ck(kmd_malloc(&p, bytes), "kmd_malloc"); cap = bytes; } }
  ~dev_buf() { if (p) kmd_free(p); }
};


// ---- the run: what parse() and the run directory fix once, shared by the stages -------------
struct run_context
{
  diff_options opt;
  int ndev = 0;
  kmtricks_config cfg;
  bool two_limbs = false;                                // KSIZE 32 / 64, src/main.cc:75
  std::vector<fof_entry> fof;
  size_t S = 0;                                          // controls + cases
  std::vector<uint64_t> total_controls, total_cases;     // cmd/diff.hpp:111
  kmd_model* model0 = nullptr;                           // the model on the first device (cmd/diff.hpp:117-123)
  double first_threshold = 0;                            // cmd/diff.hpp:147
  std::shared_ptr<kmdiff::IModel<kmdiff::maxc32>> plugin;    // --cmodel
  size_t n_workers = 1;                                  // GPUs (folded onto the ones there are)
  std::string part_dir, pop_dir;
  bool want_counts = false;                              // survivors carry their count rows (pop-strat, --keep-tmp, --save-sk)
};

// ---- what stage 1 (or the files of a previous run) hands to the later stages
struct survivors_of_run
{
  survivor_set sv_all;                                   // survivors of all partitions, partition after partition
  std::vector<size_t> part_begin;                        // [nb_partitions + 1] into sv_all
  uint64_t total_kmers = 0, n_sig = 0, n_sig_control = 0, n_sig_case = 0;
  uint64_t n_near = 0;                               // rows decided with correctly rounded log / exp (KMD_CNT_NEAR_THRESHOLD)
  std::vector<double> Z_device;                          // [S][10] when the device PCA ran
  std::vector<uint64_t> worker_totals;                   // rows each GPU worker tested (empty after a resume: only their sum is on file)
  uint64_t records = 0, h2d_bytes = 0;                   // k-mer feed: records of the run, bytes they crossed PCIe in (packed transfer: ~4-6 each)
  bool packed = false;
};

// ---- host side of a partition ------------------------------------------------------------------
// Its S files are LZ4-decoded by up to -t threads (the reference spends -t on whole partitions,
// merge.hpp:239-307; here the device takes the partitions one after the other and the threads take
// the files), and partition p + 1 is decoded while the device works on partition p.
std::atomic<uint64_t> g_pin_ns { 0 }, g_pin_bytes { 0 };    // (KMD_HOST_TIMING) wall time inside page-locked allocations, bytes asked for
struct pinned                                           // page-locked staging array, grown as needed, reused
{
  void* p = nullptr; size_t cap = 0;
  bool owned = true;                                    // false: a piece of somebody else's allocation (partition_input::slab)
  void reserve(size_t bytes, size_t keep = 0)           // the first `keep` bytes survive
  {
    if (bytes <= cap) return;
    const auto t0 = std::chrono::steady_clock::now();
    void* q = nullptr;
    ck(kmd_malloc_host(&q, bytes), "kmd_malloc_host");
    if (keep) std::memcpy(q, p, keep);
    if (p && owned) kmd_free_host(p);
    p = q; cap = bytes; owned = true;
    g_pin_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    g_pin_bytes += bytes;
  }
  void borrow(void* q, size_t bytes) { if (p && owned) kmd_free_host(p); p = q; cap = bytes; owned = false; }
  ~pinned() { if (p && owned) kmd_free_host(p); }
};
struct sample_stream                                    // one sample's file of a partition, decoded
{
  pinned kmers, kmers_hi, counts;
  record_sink sink;
  size_t n = 0;
  // packed transfer (kmd_pack_block): the records are packed as they are decoded -- the plain arrays above then hold
  // one chunk of the file at a time and are ordinary memory
  pinned packed, table;                                 // the stream's blocks one after the other; where each begins (/ 8)
  size_t packed_bytes = 0, n_blocks = 0;
  std::vector<char> carry;                              // records (as in the file) of a block the last chunk left open
};
struct partition_input
{
  pinned slab;                                          // packed transfer: ONE page-locked allocation the streams' arrays are cut from
                                                        // (declared before them: released after them)
  std::vector<sample_stream> st;                        // the S streams, each in its own page-locked arrays; the
                                                        // matrices/ feed: one stream of rows (counts [row][sample])
  matrix_file_info m;                                   // matrices/ feed: what the file says
  std::vector<uint64_t> offs;
  size_t n = 0;
};

// Page-locked staging a worker is done with is released behind the run's back: un-pinning what a 20v20 run stages (~1.5 GB:
// three partitions' packed streams) is a quarter of a second of hipHostFree -- measured: 0.26 of the 0.63 s stage 1 took on
// eight 2 M-row partitions -- and nothing needs it done before the process ends.  main() owns the object: its threads are
// joined on every way out of main, before the runtime's own teardown.
class retired_staging
{
public:
  void take(std::vector<partition_input>&& slots)
  {
    std::lock_guard<std::mutex> g(mu_);
    threads_.emplace_back([held = std::move(slots)]() mutable { held.clear(); });
  }
  ~retired_staging() { for (auto& t : threads_) if (t.joinable()) t.join(); }
private:
  std::mutex mu_;
  std::vector<std::thread> threads_;
};
retired_staging* g_retired = nullptr;

class partition_loader
{
public:
  partition_loader(const run_context& C, std::vector<std::string> matrix_files)
    : C_(C), mpaths_(std::move(matrix_files)), threads_(std::max<size_t>(C.opt.threads / C.n_workers, 1)) {}
  bool from_matrix() const { return !mpaths_.empty(); }
  // the k-mer feed of one-limb k-mers crosses PCIe packed (kmd_pack.hip) unless --raw-transfer says otherwise
  bool packed_transfer() const { return !from_matrix() && !C_.two_limbs && !C_.opt.raw_transfer; }
  const std::vector<std::string>& matrix_files() const { return mpaths_; }

  // The S files of partition p are decoded in parallel, each as a stream (LZ4 chunk -> records ->
  // the sample's page-locked arrays, which live as long as the staging slot: nothing is allocated
  // per partition once the arrays have grown to the run's file sizes).
  void load(size_t p, partition_input* in) const
  {
    const stopwatch t;
    const uint64_t pin0 = g_pin_ns.load(), pinb0 = g_pin_bytes.load();
    load_files(p, in);
    if (C_.opt.verbose_timing && std::getenv("KMD_HOST_TIMING")[0] == '2')
      std::fprintf(stderr, "[kmdiff-hip] partition %zu decoded in %.3f s (page-locked allocations meanwhile, all threads: %.3f s for %.0f MB)\n", p, t.seconds(),
                   (double)(g_pin_ns.load() - pin0) * 1e-9, (double)(g_pin_bytes.load() - pinb0) / 1048576.0);
    std::lock_guard<std::mutex> g(mu_);
    busy_ += t.seconds();
  }
  double busy_seconds() const { std::lock_guard<std::mutex> g(mu_); return busy_; }
  size_t decoder_threads() const { return std::min(threads_, C_.S); }

private:
  void load_files(size_t p, partition_input* in) const
  {
    const size_t S = C_.S;
    const bool two_limbs = C_.two_limbs;
    const size_t n_streams = from_matrix() ? 1 : S;
    const bool packed = packed_transfer();
    if (in->st.size() != n_streams)
    {
      in->st = std::vector<sample_stream>(n_streams);
      for (auto& st : in->st)
      {
        sample_stream* self = &st;
        if (packed)
        {
          // The records are packed for the transfer as they leave the LZ4 decoder (kmd_pack_records): a block's 256
          // records are taken from the file's [k-mer][count] layout and packed
          // into the stream's page-locked bytes -- no pass over the data besides the decoder's own; what a chunk
          // leaves over (< 256 records) waits in `carry` for the next one.
          st.sink.raw = [self](const char* p, size_t n, uint32_t cb)
          {
            static const size_t bound = kmd_pack_block_bound();
            struct pack_clock                                // (one reading per decoded chunk, not per block)
            {
              uint64_t t0 = 0;
              pack_clock() { if (g_time_pack) t0 = thread_cpu_ns(); }
              ~pack_clock() { if (g_time_pack) g_pack_ns += thread_cpu_ns() - t0; }
            } clock_;
            const size_t rec = 8 + (size_t)cb;
            const bool last = p == nullptr;
            auto room = [&](size_t blocks)
            {
              if (self->packed_bytes + blocks * bound > self->packed.cap)
                self->packed.reserve(std::max(self->packed_bytes + blocks * bound, self->packed.cap + self->packed.cap / 2 + (1u << 20)), self->packed_bytes);
              if ((self->n_blocks + blocks) * 4 > self->table.cap)
                self->table.reserve(std::max((self->n_blocks + blocks) * 4, self->table.cap * 2 + 4096), self->n_blocks * 4);
            };
            auto pack = [&](const char* q, uint32_t m)
            {
              ((uint32_t*)self->table.p)[self->n_blocks++] = (uint32_t)(self->packed_bytes / 8);
              const size_t got = kmd_pack_records(q, cb, m, (char*)self->packed.p + self->packed_bytes);
              if (!got) throw std::runtime_error("kmd_pack_records failed");
              self->packed_bytes += got;
            };
            if (self->n_blocks == 0 && self->packed_bytes == 0)
            {
              // first records of a file: room for the whole of it in one go (page-locking is the fixed cost of a run -- ~4.5 GB
              // a second, and the first partition waits for it -- and growing a page-locked array copies it): LZ4 leaves
              // sorted k-mers + small counts at 0.65-0.75 of their 12 bytes, packed they take 4-6.5, i.e. 0.45-0.7 of the
              // file's size.  Five eighths of it and an eighth more (the files of a sample differ by a little from partition
              // to partition) is asked for -- the file's whole size and a quarter, until round 6, page-locked 1.1 GB for a
              // 20v20 run of 2 M-row partitions where 0.45 GB are used: 0.25 s, most of the 0.29 s partition 0 took --
              // and a stream that needs more grows (room()); the arrays stay for the next partition
              const size_t guess_records = self->sink.file_size * 3 / 2 / rec + 1024;
              const size_t want_p = self->sink.file_size / 8 * 5 + (1u << 16), want_t = (guess_records / KMD_PACK_BLOCK + 2) * 4;
              if (self->packed.cap < want_p) self->packed.reserve(want_p + want_p / 8, 0);
              if (self->table.cap < want_t) self->table.reserve(want_t + want_t / 4, 0);
            }
            room(n / KMD_PACK_BLOCK + 2);
            std::vector<char>& carry = self->carry;
            if (!carry.empty() && n)                    // fill the block the previous chunk left open
            {
              const size_t have = carry.size() / rec, take = std::min(n, (size_t)KMD_PACK_BLOCK - have);
              carry.insert(carry.end(), p, p + take * rec);
              p += take * rec; n -= take;
              if (carry.size() / rec == KMD_PACK_BLOCK) { pack(carry.data(), KMD_PACK_BLOCK); carry.clear(); }
            }
            for (; n >= KMD_PACK_BLOCK; n -= KMD_PACK_BLOCK, p += KMD_PACK_BLOCK * rec) pack(p, KMD_PACK_BLOCK);
            if (n) carry.insert(carry.end(), p, p + n * rec);
            if (last && !carry.empty()) { pack(carry.data(), (uint32_t)(carry.size() / rec)); carry.clear(); }
            if (self->packed_bytes / 8 > 0xFFFFFFFFull) throw std::runtime_error("a sample's packed stream exceeds 32 GB");
          };
          continue;
        }
        st.sink.reserve = [self](record_sink& k, size_t n)
        {
          const size_t keep = k.capacity;                // grows only while a file is being read: all of it is live
          const size_t per = (size_t)k.nb_counts * 4;    // bytes of counts per record
          self->kmers.reserve(n * 8, keep * 8); self->counts.reserve(n * per, keep * per);
          if (k.slots == 2) self->kmers_hi.reserve(n * 8, keep * 8);
          k.kmers = (uint64_t*)self->kmers.p; k.counts = (uint32_t*)self->counts.p;
          k.kmers_hi = k.slots == 2 ? (uint64_t*)self->kmers_hi.p : nullptr;
          k.capacity = n;
        };
      }
    }
    if (from_matrix())                                   // pre-merged rows (matrix_proxy::merge): one file, one thread
    {
      in->m = stream_matrix_file(mpaths_[p], in->st[0].sink);
      in->st[0].n = in->n = in->m.rows;
      return;
    }
    if (packed && !in->slab.p)
    {
      // The slot's page-locked arrays in ONE allocation, cut by the files' sizes (the sink's own estimate, below): eighty
      // streams each page-locking their two arrays from the decoder threads went through the runtime one after the other
      // (measured: 3.1 s of the 32 threads' wall time inside 160 allocations; 0.2 s in these two).  The first two
      // partitions still take 0.15 s to decode where the others take 0.026: what is left is the allocation itself --
      // 0.1 s per 309 MB in this process, whose memory cgroup is full of the run directory's page cache; 0.013 s in an
      // idle one (tools/archive/r06_pin_probe.py).  A stream that outgrows its piece gets an array of its own
      // (pinned::reserve), as before.
      auto up = [](size_t v) { return (v + 4095) / 4096 * 4096; };
      std::vector<size_t> wp(S), wt(S);
      size_t total = 0;
      for (size_t s2 = 0; s2 < S; ++s2)
      {
        std::error_code ec;
        const auto fsz = fs::file_size(kmer_file_path(C_.opt.kmtricks_dir, p, C_.fof[s2].id), ec);
        const size_t sz = ec ? 0 : (size_t)fsz;
        const size_t want_p = sz / 8 * 5 + (1u << 16), want_t = ((sz * 3 / 2 / 9 + 1024) / KMD_PACK_BLOCK + 2) * 4;     // (9: the shortest record)
        wp[s2] = up(want_p + want_p / 8); wt[s2] = up(want_t + want_t / 4);
        total += wp[s2] + wt[s2];
      }
      in->slab.reserve(total);
      char* at = (char*)in->slab.p;
      for (size_t s2 = 0; s2 < S; ++s2)
      {
        in->st[s2].packed.borrow(at, wp[s2]); at += wp[s2];
        in->st[s2].table.borrow(at, wt[s2]); at += wt[s2];
      }
    }
    for_samples([&](size_t s2)
    {
      in->st[s2].packed_bytes = 0; in->st[s2].n_blocks = 0; in->st[s2].carry.clear();
      const kmer_file_info f = stream_kmer_file(kmer_file_path(C_.opt.kmtricks_dir, p, C_.fof[s2].id), C_.cfg.kmer_size, in->st[s2].sink);
      if ((f.slots == 2) != two_limbs) throw std::runtime_error("k-mer width of a sample file differs from the run's");
      in->st[s2].n = f.records;
    });
    in->offs.assign(S + 1, 0);
    for (size_t s2 = 0; s2 < S; ++s2) in->offs[s2 + 1] = in->offs[s2] + in->st[s2].n;         // KmDir::get_files_to_merge order
    in->n = in->offs[S];
  }

  // body(s) for every sample on this worker's share of the -t threads; the first exception is rethrown
  void for_samples(const std::function<void(size_t)>& body) const
  {
    const size_t S = C_.S;
    std::atomic<size_t> next { 0 };
    std::mutex mu;
    std::exception_ptr err;
    auto work = [&]()
    {
      const uint64_t t0 = g_time_pack ? thread_cpu_ns() : 0;
      for (size_t s2; (s2 = next++) < S;)
      {
        try { body(s2); }
        catch (...) { std::lock_guard<std::mutex> g(mu); if (!err) err = std::current_exception(); }
      }
      if (g_time_pack) g_decode_ns += thread_cpu_ns() - t0;
    };
    std::vector<std::thread> pool;
    for (size_t t = 1; t < std::min(threads_, S); ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    if (err) std::rethrow_exception(err);
  }

  const run_context& C_;
  std::vector<std::string> mpaths_;
  size_t threads_;
  mutable std::mutex mu_;
  mutable double busy_ = 0;                              // seconds spent in load() (KMD_HOST_TIMING)
};

// ---- one GPU of stage 1 ------------------------------------------------------------------------
// One worker thread per GPU; partition p belongs to worker p % n_workers (the sharding of
// kmdiff_amd/dist.py, in one process).  A worker keeps its survivors in its own set; they are put
// in partition order afterwards.
struct worker_result
{
  survivor_set sv;
  std::vector<std::pair<size_t, size_t>> span;     // per partition of this worker: (begin, count) in sv
  uint64_t total = 0, n_sig = 0, n_ctrl = 0, n_case = 0, n_sampled = 0, n_near = 0;
  uint64_t records = 0, h2d_bytes = 0;              // k-mer feed: records of this worker's partitions, bytes they crossed PCIe in
  std::vector<double> xtx;                         // the worker's PCA Gram matrix
  std::string error;
};

// worker wi: partitions wi, wi + n_workers, ... of the n_units there are -> R
void gpu_worker_partitions(const run_context& C, const partition_loader& loader, const size_t wi, const size_t n_units, const bool run_pca,
                    const stopwatch& merge_time, worker_result& R)
{
  const diff_options& opt = C.opt;
  const kmtricks_config& cfg = C.cfg;
  const size_t S = C.S, n_workers = C.n_workers;
  const int ndev = C.ndev;
  const bool two_limbs = C.two_limbs, want_counts = C.want_counts;
  const std::vector<uint64_t>& total_controls = C.total_controls;
  const std::vector<uint64_t>& total_cases = C.total_cases;
  kmd_model* const model0 = C.model0;
  const double first_threshold = C.first_threshold;
  const std::shared_ptr<kmdiff::IModel<kmdiff::maxc32>>& plugin = C.plugin;
  const bool from_matrix = loader.from_matrix();
  const std::vector<std::string>& mpaths = loader.matrix_files();
  const size_t T = 4096;                                 // rows per block of the tiled layout K2 writes
  const int dev = (opt.device + (int)wi) % ndev;
  ck(kmd_set_device(dev), "kmd_set_device");
  kmd_model* model = wi == 0 ? model0 : nullptr;                                             // a model lives on one device
  if (wi != 0)
    ck(kmd_model_create(&model, (int)opt.nb_controls, (int)opt.nb_cases, total_controls.data(), total_cases.data(), opt.log_size),
       "kmd_model_create");
  kmd_pca* pca = nullptr;
  if (run_pca) ck(kmd_pca_create(&pca, (int)S, opt.kmer_pca, opt.seed, opt.ploidy == 2 ? 1 : 0, (size_t)1 << 20), "kmd_pca_create");   // grows
  survivor_set& sv_all = R.sv;                       // (this worker's)
  sv_all.n_counts = want_counts ? S : 0;
  sv_all.kmer_bytes = two_limbs ? 16 : 8;
  uint64_t total_kmers = 0, n_sig = 0, n_sig_control = 0, n_sig_case = 0;
  uint64_t n_near = 0;                               // rows decided with correctly rounded log / exp (KMD_CNT_NEAR_THRESHOLD)
  // what a partition brings to the device, twice: the set of partition t + 1 is filled by the copy
  // stream while the kernels of partition t read the other one.  K-mer feed: the S streams one after
  // the other (the input of K2); matrices/ feed: the rows themselves (the tile K1 reads).
  struct device_input
  {
    dev_buf kmers, kmers_hi, counts;
    dev_buf packed, table;                               // packed transfer: the streams' blocks and block tables as they arrive
    std::vector<uint64_t> offs;                          // k-mer feed: stream s is [offs[s], offs[s + 1])
    size_t n = 0;                                        // records (k-mer feed) or rows (matrices/)
  };
  device_input dset[2];
  dev_buf d_matrix, d_kmer_col, d_kmer_col_hi, d_cnt, d_srow, d_skmer, d_skmer_hi, d_sp, d_ssign, d_smc, d_smk, d_sc, d_sum_c, d_sum_k;
  // --no-matrix: the streams go straight through merge + test (kmd_merge_filter); the count rows of the
  // survivors (pop-strat, --keep-tmp, --save-sk) are looked up in the streams afterwards
  // (the default; --matrix-path keeps the k-way merge into the count matrix + K1, which a --cmodel plugin and
  // the matrices/ feed need anyway)
  const bool use_sums = !opt.matrix_path && !from_matrix && !plugin && S <= 1024;
  size_t fused_cap = (size_t)1 << 16;                    // survivor sink of the fused path (grows on demand)
  // a ring of staging sets: the partition being processed and `depth` more being decoded.  K-mer
  // files: one partition ahead, its S files on this worker's share of the -t threads (deeper
  // measured no gain: the decode is hidden already).  matrices/: one file = one LZ4 frame = one
  // thread per partition, so up to 8 partitions are decoded at once, like the reference's -t
  // threads each taking a partition (merge.hpp:239-307) -- but no more slots than a quarter of this
  // worker's partitions: every slot is page-locked once (~0.15 s per GB, and again to release),
  // which a short run does not earn back (12 partitions of 336 MB: depth 8 cost 1.1 s more than depth 1).
  const size_t my_units = (n_units + n_workers - 1 - wi) / n_workers;
  const size_t depth = std::getenv("KMD_RING_DEPTH") ? (size_t)std::max(1, std::atoi(std::getenv("KMD_RING_DEPTH"))) : !from_matrix ? 1 :
    std::max<size_t>(1, std::min({ std::max<size_t>(opt.threads / n_workers, 1), (size_t)8, my_units / 4 }));
  std::vector<partition_input> staging(depth + 1);
  std::vector<std::future<void>> ahead(depth + 1);
  // declared after the staging ring: on any way out the copies are waited for before their page-locked
  // sources are released
  void* copy_stream = nullptr;
  ck(kmd_stream_create(&copy_stream), "kmd_stream_create");
  struct stream_guard { void* s; ~stream_guard() { kmd_stream_sync(s); kmd_stream_destroy(s); } } copy_guard { copy_stream };
  double t_loader = 0, t_device = 0, t_first = 0, t_steady = 0, t_copy_wait = 0;      // waiting for the decoder / copies + kernels + survivors back / the part of
                                                       // the wait spent on the ring's first turn (its arrays get page-locked then)
  size_t issued = 0;                                   // partitions of this worker handed to the loader
  auto issue = [&]()
  {
    const size_t p_next = wi + issued * n_workers;
    if (p_next < n_units)
      ahead[issued % (depth + 1)] = std::async(std::launch::async, [&loader, p_next, in = &staging[issued % (depth + 1)]]() { loader.load(p_next, in); });
    ++issued;
  };
  // turn t of this worker = its t-th partition.  upload(t): wait for the decoder of turn t, then
  // enqueue its host-to-device copies (page-locked arrays -> device set t % 2) on the copy stream.
  auto upload = [&](size_t t)
  {
    const size_t p = wi + t * n_workers;
    if (p >= n_units) return;
    ahead[t % (depth + 1)].get();
    const partition_input& in = staging[t % (depth + 1)];
    device_input& D = dset[t % 2];
    D.n = in.n;
    if (from_matrix)
    {
      // pre-merged rows (matrix_proxy::merge): row-major counts go to the device as they are
      if (in.m.nb_counts != S) die(mpaths[p] + ": number of samples differs from -1 + -2");
      if (two_limbs != (in.m.slots == 2)) die(mpaths[p] + ": k-mer width differs from the run's");
      const sample_stream& rows = in.st[0];
      D.counts.reserve(D.n * S * 4); D.kmers.reserve(D.n * 8);
      ck(kmd_memcpy_h2d_async(D.counts.p, rows.counts.p, D.n * S * 4, copy_stream), "h2d");
      ck(kmd_memcpy_h2d_async(D.kmers.p, rows.kmers.p, D.n * 8, copy_stream), "h2d");
      if (two_limbs)
      {
        D.kmers_hi.reserve(D.n * 8);
        ck(kmd_memcpy_h2d_async(D.kmers_hi.p, rows.kmers_hi.p, D.n * 8, copy_stream), "h2d");
      }
      return;
    }
    D.offs = in.offs;                                    // the slot is refilled while this partition is merged
    D.kmers.reserve(D.n * 8); D.counts.reserve(D.n * 4);
    if (two_limbs) D.kmers_hi.reserve(D.n * 8);
    R.records += D.n;
    if (loader.packed_transfer())
    {
      // 4-6 bytes per record cross the link instead of 12; k_unpack, behind the copies on the copy stream, writes the
      // plain arrays the kernels read (merge.hpp:265-266: the reference streams its files straight into the merge)
      std::vector<uint64_t> base(S + 1, 0), blk(S + 1, 0);
      for (size_t s2 = 0; s2 < S; ++s2) { base[s2 + 1] = base[s2] + in.st[s2].packed_bytes; blk[s2 + 1] = blk[s2] + in.st[s2].n_blocks; }
      D.packed.reserve(std::max<size_t>(base[S], 8)); D.table.reserve(std::max<size_t>(blk[S] * 4, 8));
      for (size_t s2 = 0; s2 < S; ++s2)
      {
        const sample_stream& st = in.st[s2];
        if (!st.n) continue;
        ck(kmd_memcpy_h2d_async((char*)D.packed.p + base[s2], st.packed.p, st.packed_bytes, copy_stream), "h2d");
        ck(kmd_memcpy_h2d_async((char*)D.table.p + blk[s2] * 4, st.table.p, st.n_blocks * 4, copy_stream), "h2d");
      }
      ck(kmd_unpack_streams((int)S, D.packed.p, base.data(), (const uint32_t*)D.table.p, D.offs.data(), (uint64_t*)D.kmers.p,
                            (uint32_t*)D.counts.p, copy_stream), "kmd_unpack_streams");
      R.h2d_bytes += base[S] + blk[S] * 4;
      return;
    }
    R.h2d_bytes += D.n * (two_limbs ? 20 : 12);
    for (size_t s2 = 0; s2 < S; ++s2)                     // each stream to its place in the partition's arrays
    {
      const sample_stream& st = in.st[s2];
      ck(kmd_memcpy_h2d_async((char*)D.kmers.p + D.offs[s2] * 8, st.kmers.p, st.n * 8, copy_stream), "h2d");
      ck(kmd_memcpy_h2d_async((char*)D.counts.p + D.offs[s2] * 4, st.counts.p, st.n * 4, copy_stream), "h2d");
      if (two_limbs) ck(kmd_memcpy_h2d_async((char*)D.kmers_hi.p + D.offs[s2] * 8, st.kmers_hi.p, st.n * 8, copy_stream), "h2d");
    }
  };
  if (opt.verbose_timing) std::fprintf(stderr, "[kmdiff-hip] GPU %d: worker ready %.3f s into stage 1\n", dev, merge_time.seconds());
  for (size_t d = 0; d < depth + 1; ++d) issue();        // every slot of the ring gets a decoder
  upload(0);
  size_t turn = 0;
  for (size_t p = wi; p < n_units; p += n_workers, ++turn)
  {
    kmd_tile tile {};
    uint64_t n_rows = 0;
    bool sums_done = false;                              // --no-matrix took this partition
    // copies of this partition done: its page-locked slot goes to the decoder of turn + depth + 1,
    // and the next partition (decoded during the previous turn) is uploaded behind the kernels of this one
    const stopwatch t_wait;
    ck(kmd_stream_sync(copy_stream), "kmd_stream_sync");
    t_copy_wait += t_wait.seconds();
    issue();
    upload(turn + 1);
    t_loader += t_wait.seconds();
    if (turn <= depth) t_first += t_wait.seconds();
    const stopwatch t_dev;
    const device_input& D = dset[turn % 2];
    if (from_matrix)
    {
      n_rows = D.n;
      tile = kmd_tile { D.counts.p, 4, KMD_LAYOUT_ROWS, S, (const uint64_t*)D.kmers.p,
                        two_limbs ? (const uint64_t*)D.kmers_hi.p : nullptr, (size_t)n_rows, 0 };
    }
    else
    {
      const size_t n = D.n;
      if (use_sums && n) sums_done = true;               // merged and tested in one go below
      if (n && !sums_done)
      {
        d_matrix.reserve(std::max(((n + T - 1) / T) * T, n) * S * 4); d_kmer_col.reserve(n * 8);
        if (two_limbs) d_kmer_col_hi.reserve(n * 8);
        ck(kmd_merge_partition((int)S, (const uint64_t*)D.kmers.p, two_limbs ? (const uint64_t*)D.kmers_hi.p : nullptr,
                               (const uint32_t*)D.counts.p, D.offs.data(), 4, plugin ? KMD_LAYOUT_ROWS : KMD_LAYOUT_TILED,
                               plugin ? S : T, n, d_matrix.p,
                               (uint64_t*)d_kmer_col.p, two_limbs ? (uint64_t*)d_kmer_col_hi.p : nullptr, &n_rows, nullptr),
           "kmd_merge_partition");
      }
      tile = kmd_tile { d_matrix.p, 4, plugin ? KMD_LAYOUT_ROWS : KMD_LAYOUT_TILED, plugin ? S : T, (const uint64_t*)d_kmer_col.p,
                        two_limbs ? (const uint64_t*)d_kmer_col_hi.p : nullptr, (size_t)n_rows, 0 };
    }
    size_t ns = 0;
    const size_t base = sv_all.size();
    if (pca && sums_done)                                                                     // merge.hpp:150-152, from the streams
    {
      const device_input& Dp = dset[turn % 2];
      ck(kmd_pca_sample_streams(pca, (int)S, (const uint64_t*)Dp.kmers.p, two_limbs ? (const uint64_t*)Dp.kmers_hi.p : nullptr,
                                (const uint32_t*)Dp.counts.p, Dp.offs.data(), nullptr), "kmd_pca_sample_streams");
    }
    else if (n_rows && pca) ck(kmd_pca_sample(pca, &tile, nullptr), "kmd_pca_sample");        // merge.hpp:150-152
    bool unresolved_redo = false;                         // the fused pass left near-threshold rows undecided: the matrix way, in pieces
    uint64_t unresolved_rows = 0;                         // ... how many it said were beyond the list
    if (n_rows && plugin)
    {
      // diff_observer::process with the user's model (merge.hpp:68-103): the merged rows come
      // back to the host, row-major, and go through process() one by one
      std::vector<uint32_t> rows(n_rows * S);
      std::vector<uint64_t> km(n_rows), kmh(two_limbs ? n_rows : 0);
      ck(kmd_memcpy_d2h(rows.data(), tile.d_counts, n_rows * S * 4, nullptr), "d2h");
      ck(kmd_memcpy_d2h(km.data(), tile.d_kmer_lo, n_rows * 8, nullptr), "d2h");
      if (two_limbs) ck(kmd_memcpy_d2h(kmh.data(), tile.d_kmer_hi, n_rows * 8, nullptr), "d2h");
      std::vector<uint32_t> row(S);
      for (size_t i = 0; i < n_rows; ++i)
      {
        std::copy(rows.begin() + i * S, rows.begin() + (i + 1) * S, row.begin());
        kmdiff::Range<uint32_t> controls(row, 0, opt.nb_controls), cases(row, opt.nb_controls, opt.nb_cases);
        auto [pv, sg, mc, mk] = plugin->process(controls, cases);
        ++total_kmers;
        if (pv <= first_threshold)
        {
          sv_all.kmer.push_back(km[i]);
          if (two_limbs) sv_all.kmer_hi.push_back(kmh[i]);
          sv_all.p.push_back(pv); sv_all.sign.push_back((int32_t)sg);
          sv_all.mean_control.push_back(mc); sv_all.mean_case.push_back(mk);
          if (want_counts) for (size_t s2 = 0; s2 < S; ++s2) sv_all.counts.push_back((double)row[s2]);
          if (sg == kmdiff::Significance::CONTROL) ++n_sig_control; else ++n_sig_case;        // merge.hpp:95-98
          ++n_sig; ++ns;
        }
      }
    }
    else if (sums_done)
    {
      // KmerMerger::merge(diff_observer) in one call: streams in, survivors out (k-mer order restored
      // on the device); a sink that turns out too small is enlarged and the partition run again
      const device_input& D = dset[turn % 2];
      uint64_t c[KMD_NCOUNTERS];
      kmd_survivors sv {};
      for (;;)
      {
        d_srow.reserve(fused_cap * 8); d_skmer.reserve(fused_cap * 8); d_sp.reserve(fused_cap * 8); d_ssign.reserve(fused_cap * 4);
        d_smc.reserve(fused_cap * 8); d_smk.reserve(fused_cap * 8); d_cnt.reserve(KMD_NCOUNTERS * 8);
        if (two_limbs) d_skmer_hi.reserve(fused_cap * 8);
        ck(kmd_memset(d_cnt.p, 0, KMD_NCOUNTERS * 8, nullptr), "memset");
        sv = kmd_survivors { (uint64_t*)d_srow.p, (uint64_t*)d_skmer.p, two_limbs ? (uint64_t*)d_skmer_hi.p : nullptr, (double*)d_sp.p,
                             (int32_t*)d_ssign.p, (double*)d_smc.p, (double*)d_smk.p, fused_cap };
        ck(kmd_merge_filter(model, (int)S, (const uint64_t*)D.kmers.p, two_limbs ? (const uint64_t*)D.kmers_hi.p : nullptr,
                            (const uint32_t*)D.counts.p, D.offs.data(), first_threshold, &sv, (uint64_t*)d_cnt.p, &n_rows, nullptr),
           "kmd_merge_filter");
        ck(kmd_memcpy_d2h(c, d_cnt.p, sizeof c, nullptr), "d2h");
        if (c[KMD_CNT_SIG] <= fused_cap) break;
        fused_cap = (size_t)c[KMD_CNT_SIG] + (size_t)c[KMD_CNT_SIG] / 4;
      }
      if (c[KMD_CNT_NEAR_UNRESOLVED] != 0 && !two_limbs)
      {
        // More rows within 1e-8 of the threshold than one launch can list (4096; include/kmdiff_hip.h): the rows beyond
        // kept the device libm's decision.  The guard's guarantee is restored the long way -- and without a matrix
        // (round 5 merged the partition into n x S x 4 bytes here: tens of GB on exactly the jobs that take the fused
        // path because the matrix does not fit): the partition's rows as (k-mer, control sum, case sum), 24
        // bytes each (kmd_merge_sums), tested in pieces small enough for every near row to be listed -- as many pieces
        // as the count of unresolved rows asks for, twice as many while any is left.
        std::fprintf(stderr, "[kmdiff-hip] partition %zu: %llu near-threshold rows beyond the list: again, as rows of sums in pieces\n", p, (unsigned long long)c[KMD_CNT_NEAR_UNRESOLVED]);
        const size_t nr = (size_t)n_rows;
        d_kmer_col.reserve(nr * 8); d_sum_c.reserve(nr * 8); d_sum_k.reserve(nr * 8);
        uint64_t got = 0;
        ck(kmd_merge_sums((int)S, (int)opt.nb_controls, (const uint64_t*)D.kmers.p, nullptr, (const uint32_t*)D.counts.p, D.offs.data(), nr,
                          (uint64_t*)d_kmer_col.p, nullptr, (uint64_t*)d_sum_c.p, (uint64_t*)d_sum_k.p, &got, nullptr), "kmd_merge_sums");
        if (got != nr) die("kmd_merge_sums: " + std::to_string(got) + " rows, the fused pass counted " + std::to_string(nr));
        for (size_t pieces = std::max<size_t>(2, (size_t)((c[KMD_CNT_NEAR_UNRESOLVED] + 4095) / 4096) + 1);; pieces *= 2)
        {
          const size_t rows_per = (nr + pieces - 1) / pieces;
          ck(kmd_memset(d_cnt.p, 0, KMD_NCOUNTERS * 8, nullptr), "memset");
          for (size_t r0 = 0; r0 < nr; r0 += rows_per)
            ck(kmd_poisson_filter_sums(model, (const uint64_t*)d_kmer_col.p + r0, (const uint64_t*)d_sum_c.p + r0, (const uint64_t*)d_sum_k.p + r0,
                                       std::min(rows_per, nr - r0), first_threshold, &sv, (uint64_t*)d_cnt.p, nullptr), "kmd_poisson_filter_sums");
          ck(kmd_memcpy_d2h(c, d_cnt.p, sizeof c, nullptr), "d2h");
          if (c[KMD_CNT_SIG] > fused_cap) die("the pieces found more survivors than the fused pass");
          if (c[KMD_CNT_NEAR_UNRESOLVED] == 0)
          {
            std::fprintf(stderr, "[kmdiff-hip] partition %zu: every near-threshold row decided in %zu pieces of %zu rows\n", p, pieces, rows_per);
            break;
          }
          if (rows_per <= 4096) die("near-threshold rows left unresolved in a piece of 4096 rows");
        }
      }
      if (c[KMD_CNT_NEAR_UNRESOLVED] != 0)
      {
        // (two-limb k-mers: the rows-of-sums test carries one limb -- the partition is merged into a matrix and its rows
        // are tested in pieces, below)
        std::fprintf(stderr, "[kmdiff-hip] partition %zu: %llu near-threshold rows beyond the list: again, as a matrix in pieces\n", p, (unsigned long long)c[KMD_CNT_NEAR_UNRESOLVED]);
        const size_t n = D.n;
        d_matrix.reserve(std::max(((n + T - 1) / T) * T, n) * S * 4); d_kmer_col.reserve(n * 8);
        if (two_limbs) d_kmer_col_hi.reserve(n * 8);
        ck(kmd_merge_partition((int)S, (const uint64_t*)D.kmers.p, two_limbs ? (const uint64_t*)D.kmers_hi.p : nullptr,
                               (const uint32_t*)D.counts.p, D.offs.data(), 4, KMD_LAYOUT_TILED, T, n, d_matrix.p,
                               (uint64_t*)d_kmer_col.p, two_limbs ? (uint64_t*)d_kmer_col_hi.p : nullptr, &n_rows, nullptr),
           "kmd_merge_partition");
        tile = kmd_tile { d_matrix.p, 4, KMD_LAYOUT_TILED, T, (const uint64_t*)d_kmer_col.p,
                          two_limbs ? (const uint64_t*)d_kmer_col_hi.p : nullptr, (size_t)n_rows, 0 };
        sums_done = false;
        unresolved_redo = true;
        unresolved_rows = c[KMD_CNT_NEAR_UNRESOLVED];
      }
      else
      {
      ns = (size_t)c[KMD_CNT_SIG];
      ck(kmd_pvalues_refine(model, ns, (const double*)d_smc.p, (const double*)d_smk.p, (double*)d_sp.p, nullptr), "pvalues_refine");   // glibc's bits
      ck(kmd_survivors_sort_by_kmer(&sv, ns, nullptr), "sort_by_kmer");                   // reference push order
      sv_all.kmer.resize(base + ns); sv_all.p.resize(base + ns); sv_all.sign.resize(base + ns);
      sv_all.mean_control.resize(base + ns); sv_all.mean_case.resize(base + ns);
      if (two_limbs) sv_all.kmer_hi.resize(base + ns);
      if (ns)
      {
        ck(kmd_memcpy_d2h(sv_all.kmer.data() + base, d_skmer.p, ns * 8, nullptr), "d2h");
        if (two_limbs) ck(kmd_memcpy_d2h(sv_all.kmer_hi.data() + base, d_skmer_hi.p, ns * 8, nullptr), "d2h");
        ck(kmd_memcpy_d2h(sv_all.p.data() + base, d_sp.p, ns * 8, nullptr), "d2h");
        ck(kmd_memcpy_d2h(sv_all.sign.data() + base, d_ssign.p, ns * 4, nullptr), "d2h");
        ck(kmd_memcpy_d2h(sv_all.mean_control.data() + base, d_smc.p, ns * 8, nullptr), "d2h");
        ck(kmd_memcpy_d2h(sv_all.mean_case.data() + base, d_smk.p, ns * 8, nullptr), "d2h");
        if (want_counts)                                                                  // merge.hpp:91-92, from the streams
        {
          d_sc.reserve(ns * S * 8);
          ck(kmd_survivors_gather_counts_streams((int)S, (const uint64_t*)D.kmers.p, two_limbs ? (const uint64_t*)D.kmers_hi.p : nullptr,
                                                 (const uint32_t*)D.counts.p, D.offs.data(), (const uint64_t*)d_skmer.p,
                                                 two_limbs ? (const uint64_t*)d_skmer_hi.p : nullptr, nullptr, ns, (double*)d_sc.p, nullptr),
             "gather_counts_streams");
          sv_all.counts.resize((base + ns) * S);
          ck(kmd_memcpy_d2h(sv_all.counts.data() + base * S, d_sc.p, ns * S * 8, nullptr), "d2h");
        }
      }
      total_kmers += c[KMD_CNT_TOTAL]; n_sig += ns; n_sig_control += c[KMD_CNT_SIG_CONTROL]; n_sig_case += c[KMD_CNT_SIG_CASE];
      n_near += c[KMD_CNT_NEAR_THRESHOLD];
      }
    }
    if (n_rows && !plugin && !sums_done)
    {
      // survivor sink sized for the worst case of this partition (every row)
      d_srow.reserve(n_rows * 8); d_skmer.reserve(n_rows * 8); d_sp.reserve(n_rows * 8); d_ssign.reserve(n_rows * 4);
      d_smc.reserve(n_rows * 8); d_smk.reserve(n_rows * 8); d_cnt.reserve(KMD_NCOUNTERS * 8);
      if (two_limbs) d_skmer_hi.reserve(n_rows * 8);
      ck(kmd_memset(d_cnt.p, 0, KMD_NCOUNTERS * 8, nullptr), "memset");
      kmd_survivors sv { (uint64_t*)d_srow.p, (uint64_t*)d_skmer.p, two_limbs ? (uint64_t*)d_skmer_hi.p : nullptr, (double*)d_sp.p, (int32_t*)d_ssign.p,
                         (double*)d_smc.p, (double*)d_smk.p, (size_t)n_rows };
      // One launch lists up to 4096 rows within 1e-8 of the threshold for the correctly rounded second look; more than
      // that (KMD_CNT_NEAR_UNRESOLVED) and the partition is tested again in pieces, twice as many each time -- a piece
      // of 4096 rows cannot overflow the list.  (0 such rows in 10^10 synthetic ones; a threshold that IS some row's
      // p-value with thousands of equal rows gets here.)
      uint64_t c[KMD_NCOUNTERS];
      const size_t unit = tile.layout == KMD_LAYOUT_TILED ? tile.ld : 1;                  // pieces begin on a block of the tiled layout
      // (the fused pass said how many rows were beyond ITS list: with them spread evenly, that many lists' worth of pieces
      // -- plus one -- is where a retry has a chance; starting at 2 ran every doubling in between from scratch: ADVICE r5)
      for (size_t pieces = unresolved_redo ? std::max<size_t>(2, (size_t)((unresolved_rows + 4095) / 4096) + 1) : 1;; pieces *= 2)
      {
        size_t rows_per = ((size_t)n_rows + pieces - 1) / pieces;
        rows_per = std::max<size_t>((rows_per + unit - 1) / unit * unit, unit);
        if (pieces > 1) ck(kmd_memset(d_cnt.p, 0, KMD_NCOUNTERS * 8, nullptr), "memset");
        for (size_t r0 = 0; r0 < (size_t)n_rows; r0 += rows_per)
        {
          kmd_tile piece = tile;
          piece.n_rows = std::min(rows_per, (size_t)n_rows - r0);
          piece.row_base = tile.row_base + r0;
          const char* cp = (const char*)tile.d_counts;
          if (tile.layout == KMD_LAYOUT_TILED) cp += (r0 / tile.ld) * S * tile.ld * (size_t)tile.count_bytes;
          else if (tile.layout == KMD_LAYOUT_ROWS) cp += r0 * tile.ld * (size_t)tile.count_bytes;
          else cp += r0 * (size_t)tile.count_bytes;
          piece.d_counts = cp;
          if (tile.d_kmer_lo) piece.d_kmer_lo = tile.d_kmer_lo + r0;
          if (tile.d_kmer_hi) piece.d_kmer_hi = tile.d_kmer_hi + r0;
          ck(kmd_poisson_filter(model, &piece, first_threshold, &sv, (uint64_t*)d_cnt.p, nullptr), "kmd_poisson_filter");
        }
        ck(kmd_memcpy_d2h(c, d_cnt.p, sizeof c, nullptr), "d2h");
        if (c[KMD_CNT_NEAR_UNRESOLVED] == 0)
        {
          if (pieces > 1) std::fprintf(stderr, "[kmdiff-hip] partition %zu: every near-threshold row decided in %zu pieces of %zu rows\n", p, pieces, rows_per);
          break;
        }
        if (rows_per <= 4096) die("kmd_poisson_filter: near-threshold rows left undecided in a piece of <= 4096 rows");
      }
      ns = (size_t)c[KMD_CNT_SIG];
      ck(kmd_pvalues_refine(model, ns, (const double*)d_smc.p, (const double*)d_smk.p, (double*)d_sp.p, nullptr), "pvalues_refine");   // glibc's bits
      ck(kmd_survivors_sort_by_row(&sv, ns, nullptr), "sort_by_row");                     // reference push order
      sv_all.kmer.resize(base + ns); sv_all.p.resize(base + ns); sv_all.sign.resize(base + ns);
      sv_all.mean_control.resize(base + ns); sv_all.mean_case.resize(base + ns);
      if (two_limbs) sv_all.kmer_hi.resize(base + ns);
      if (ns)
      {
        ck(kmd_memcpy_d2h(sv_all.kmer.data() + base, d_skmer.p, ns * 8, nullptr), "d2h");
        if (two_limbs) ck(kmd_memcpy_d2h(sv_all.kmer_hi.data() + base, d_skmer_hi.p, ns * 8, nullptr), "d2h");
        ck(kmd_memcpy_d2h(sv_all.p.data() + base, d_sp.p, ns * 8, nullptr), "d2h");
        ck(kmd_memcpy_d2h(sv_all.sign.data() + base, d_ssign.p, ns * 4, nullptr), "d2h");
        ck(kmd_memcpy_d2h(sv_all.mean_control.data() + base, d_smc.p, ns * 8, nullptr), "d2h");
        ck(kmd_memcpy_d2h(sv_all.mean_case.data() + base, d_smk.p, ns * 8, nullptr), "d2h");
        if (want_counts)                                                                  // merge.hpp:91-92
        {
          d_sc.reserve(ns * S * 8);
          ck(kmd_survivors_gather_counts(&tile, (int)S, (const uint64_t*)d_srow.p, ns, (double*)d_sc.p, nullptr), "gather_counts");
          sv_all.counts.resize((base + ns) * S);
          ck(kmd_memcpy_d2h(sv_all.counts.data() + base * S, d_sc.p, ns * S * 8, nullptr), "d2h");
        }
      }
      total_kmers += c[KMD_CNT_TOTAL]; n_sig += ns; n_sig_control += c[KMD_CNT_SIG_CONTROL]; n_sig_case += c[KMD_CNT_SIG_CASE];
      n_near += c[KMD_CNT_NEAR_THRESHOLD];
    }
    t_device += t_dev.seconds();
    if (turn > depth) t_steady += t_wait.seconds();       // decoder wait + device work of this partition
    if (opt.verbose_timing && std::getenv("KMD_HOST_TIMING")[0] == '2')
      std::fprintf(stderr, "[kmdiff-hip] GPU %d: partition %zu done %.3f s into stage 1 (its turn: %.3f s waiting for decoder + copies, %.3f s copies + kernels + survivors)\n",
                   dev, p, merge_time.seconds(), t_wait.seconds() - t_dev.seconds(), t_dev.seconds());
    R.span.emplace_back(base, ns);
    if (opt.save_sk)                                                                      // merge.hpp:83-86,272-278
    {
      matrix_rows sk; sk.kmer_size = (uint32_t)cfg.kmer_size; sk.count_bytes = 4; sk.nb_counts = (uint32_t)S; sk.partition = (uint32_t)p;
      sk.kmers.assign(sv_all.kmer.begin() + base, sv_all.kmer.end());
      if (two_limbs) sk.kmers_hi.assign(sv_all.kmer_hi.begin() + base, sv_all.kmer_hi.end());
      sk.counts.resize(ns * S);
      for (size_t i = 0; i < ns * S; ++i) sk.counts[i] = (uint32_t)sv_all.counts[base * S + i];
      write_matrix_file(opt.output_directory + "/positive_kmer_matrix/matrices/matrix_" + std::to_string(p) + ".count.lz4", sk);
    }
  }
  R.total = total_kmers; R.n_sig = n_sig; R.n_ctrl = n_sig_control; R.n_case = n_sig_case; R.n_near = n_near;
  if (opt.verbose_timing)
  {
    std::fprintf(stderr, "[kmdiff-hip] GPU %d: waited %.3f s for the file decoder (%.3f s of it for the first %zu partitions, whose staging arrays "
                         "get page-locked), %.3f s in copies + kernels\n", dev, t_loader, t_first, depth + 1, t_device);
    std::fprintf(stderr, "[kmdiff-hip] GPU %d: last partition done %.3f s into stage 1; of the wait, %.3f s for the copies; decoders busy %.3f s in all on %zu threads = %.3f s of CPU, %.3f s of it (%.0f %%) packing for the transfer\n",
                 dev, merge_time.seconds(), t_copy_wait, loader.busy_seconds(), loader.decoder_threads(), (double)g_decode_ns.load() * 1e-9,
                 (double)g_pack_ns.load() * 1e-9, 100.0 * (double)g_pack_ns.load() / std::max<double>(1.0, (double)g_decode_ns.load()));
    if (turn > depth + 1)
      std::fprintf(stderr, "[kmdiff-hip] GPU %d: steady state %.2f ms per partition (%zu partitions after the first %zu)\n", dev,
                   1e3 * t_steady / (double)(turn - depth - 1), turn - depth - 1, depth + 1);
  }
  // (every decoder has been waited for and every copy has landed -- the last partition's kernels ran behind them -- so the
  // ring's page-locked arrays are nobody's any more)
  if (g_retired && !std::getenv("KMD_SYNC_RELEASE"))
  {
    ck(kmd_stream_sync(copy_stream), "kmd_stream_sync");
    for (auto& f : ahead) if (f.valid()) f.wait();
    g_retired->take(std::move(staging));
    staging.clear();
  }
  if (pca)
  {
    ck(kmd_pca_count(pca, &R.n_sampled), "kmd_pca_count");
    R.xtx.assign(S * S, 0.0);
    ck(kmd_pca_gram(pca, R.xtx.data(), nullptr), "kmd_pca_gram");
    kmd_pca_destroy(pca);
  }
  if (wi != 0) kmd_model_destroy(model);
}

// the thread body: a worker's failure is reported by the launching thread (R.error)
void run_gpu_worker(const run_context& C, const partition_loader& loader, const size_t wi, const size_t n_units, const bool run_pca,
                    const stopwatch& merge_time, worker_result& R)
{
  try { gpu_worker_partitions(C, loader, wi, n_units, run_pca, merge_time, R); }
  catch (const std::exception& e) { R.error = e.what(); }
}

// ---- stage 1: do_diff (cmd/diff.hpp:66-164): merge + Poisson test + threshold, partition after
// partition on every GPU of the run; with run_pca the rows are sampled for the device PCA as they pass
void do_diff(const run_context& C, survivors_of_run& O, const bool run_pca)
{
  const diff_options& opt = C.opt;
  const kmtricks_config& cfg = C.cfg;
  const size_t S = C.S, n_workers = C.n_workers;
  const int ndev = C.ndev;
  const bool two_limbs = C.two_limbs, want_counts = C.want_counts;
  const std::string& part_dir = C.part_dir;
  const std::string& pop_dir = C.pop_dir;
  survivor_set& sv_all = O.sv_all;
  std::vector<size_t>& part_begin = O.part_begin;
  uint64_t& total_kmers = O.total_kmers; uint64_t& n_sig = O.n_sig; uint64_t& n_sig_control = O.n_sig_control; uint64_t& n_sig_case = O.n_sig_case;
  std::vector<double>& Z_device = O.Z_device;

  // ---- stage 1: do_diff (cmd/diff.hpp:66-164), one partition after the other on this GPU
  std::fprintf(stderr, "[kmdiff-hip] Process partitions\n");
  const stopwatch merge_time;
  const partition_loader loader(C, matrix_paths(opt.kmtricks_dir));                       // cmd/diff.hpp:80-101
  const bool from_matrix = loader.from_matrix();
  const std::vector<std::string>& mpaths = loader.matrix_files();
  if (opt.save_sk)                                                                        // cmd/diff.hpp:52-64,137-143
  {
    const std::string sk = opt.output_directory + "/positive_kmer_matrix";
    fs::create_directories(sk + "/matrices");
    for (const char* f : { "/options.txt", "/kmtricks.fof" })
      if (fs::exists(opt.kmtricks_dir + f)) fs::copy(opt.kmtricks_dir + f, sk, fs::copy_options::overwrite_existing);
    for (const char* dname : { "/config_gatb", "/repartition_gatb" })
      if (fs::exists(opt.kmtricks_dir + dname))
        fs::copy(opt.kmtricks_dir + dname, sk + dname, fs::copy_options::recursive | fs::copy_options::overwrite_existing);
  }
  // one accumulator per entry of counts/ (cmd/diff.hpp:103-107); matrix files map onto them in order
  const size_t n_units = from_matrix ? std::min(mpaths.size(), cfg.nb_partitions) : cfg.nb_partitions;
  if (from_matrix && mpaths.size() > cfg.nb_partitions) die("more files in matrices/ than partitions in counts/");
  std::vector<worker_result> results(n_workers);
  {
    std::vector<std::thread> gpus;
    for (size_t wi = 1; wi < n_workers; ++wi)
      gpus.emplace_back([&, wi]() { run_gpu_worker(C, loader, wi, n_units, run_pca, merge_time, results[wi]); });
    run_gpu_worker(C, loader, 0, n_units, run_pca, merge_time, results[0]);
    for (auto& t : gpus) t.join();
    if (opt.verbose_timing) std::fprintf(stderr, "[kmdiff-hip] workers done %.3f s into stage 1 (device arrays released; the page-locked ones are released in the background)\n", merge_time.seconds());
    ck(kmd_set_device(opt.device % ndev), "kmd_set_device");
    for (auto& R : results) if (!R.error.empty()) die(R.error);
  }
  // survivors in partition order (the order one GPU would have produced)
  {
    std::vector<size_t> taken(n_workers, 0);
    for (size_t p = 0; p < n_units; ++p)
    {
      worker_result& R = results[p % n_workers];
      const auto [b, cnt] = R.span[taken[p % n_workers]++];
      sv_all.kmer.insert(sv_all.kmer.end(), R.sv.kmer.begin() + b, R.sv.kmer.begin() + b + cnt);
      if (two_limbs) sv_all.kmer_hi.insert(sv_all.kmer_hi.end(), R.sv.kmer_hi.begin() + b, R.sv.kmer_hi.begin() + b + cnt);
      sv_all.p.insert(sv_all.p.end(), R.sv.p.begin() + b, R.sv.p.begin() + b + cnt);
      sv_all.sign.insert(sv_all.sign.end(), R.sv.sign.begin() + b, R.sv.sign.begin() + b + cnt);
      sv_all.mean_control.insert(sv_all.mean_control.end(), R.sv.mean_control.begin() + b, R.sv.mean_control.begin() + b + cnt);
      sv_all.mean_case.insert(sv_all.mean_case.end(), R.sv.mean_case.begin() + b, R.sv.mean_case.begin() + b + cnt);
      if (want_counts) sv_all.counts.insert(sv_all.counts.end(), R.sv.counts.begin() + b * S, R.sv.counts.begin() + (b + cnt) * S);
      part_begin[p + 1] = sv_all.size();
    }
    for (auto& R : results) { total_kmers += R.total; n_sig += R.n_sig; n_sig_control += R.n_ctrl; n_sig_case += R.n_case; O.n_near += R.n_near; O.worker_totals.push_back(R.total); O.records += R.records; O.h2d_bytes += R.h2d_bytes; }
    O.packed = loader.packed_transfer();
  }
  for (size_t p = n_units; p < cfg.nb_partitions; ++p) part_begin[p + 1] = part_begin[n_units];
  if (run_pca)                                                                             // run_eigenstrat_smartpca
  {
    uint64_t n_sampled = 0;
    std::vector<double> xtx(S * S, 0.0);
    for (auto& R : results)                                                                // in worker order
    {
      n_sampled += R.n_sampled;
      for (size_t i = 0; i < S * S; ++i) xtx[i] += R.xtx[i];
    }
    const int n_out = (int)std::min<size_t>(S, 10);                                        // popstrat.cpp:118
    std::vector<double> ev(S * n_out), el(n_out);
    ck(kmd_pca_eigen((int)S, xtx.data(), n_out, ev.data(), el.data()), "kmd_pca_eigen");
    fs::create_directories(pop_dir);
    std::ofstream pf(pop_dir + "/pcs.evec");
    Z_device.assign(S * 10, 0.0);
    for (size_t i = 0; i < S; ++i)
    {
      for (int k = 0; k < n_out; ++k)
      {
        char b[32]; std::snprintf(b, sizeof b, "%.04f", ev[i * n_out + k]);                 // evec2pca.perl
        pf << ' ' << (ev[i * n_out + k] > 0 ? " " : "") << b;
        Z_device[i * 10 + k] = std::strtod(b, nullptr);                                    // what load_Z reads back
      }
      pf << '\n';
    }
    std::fprintf(stderr, "[kmdiff-hip] PCA done: %" PRIu64 " k-mers sampled, eigenvalues %.4f %.4f\n", n_sampled, el[0],
                 n_out > 1 ? el[1] : 0.0);
  }
  if (opt.keep_tmp)                                                                       // FileAccumulator, del = !keep_tmp
    for (size_t p = 0; p < cfg.nb_partitions; ++p)
      write_survivor_file(part_dir + "/p" + std::to_string(p) + "_uncorrected", sv_all, part_begin[p], part_begin[p + 1] - part_begin[p]);
  std::fprintf(stderr, "[kmdiff-hip] Partitions processed (%.3f s)\n", merge_time.seconds());                // cmd/diff.hpp:158
  std::fprintf(stderr, "[kmdiff-hip] %" PRIu64 "/%" PRIu64 " significant k-mers.\n", n_sig, total_kmers);      // cmd/diff.hpp:160
  std::fprintf(stderr, "[kmdiff-hip] Before correction: %" PRIu64 " (control), %" PRIu64 " (case).\n", n_sig_control, n_sig_case);
}

// ---- stage 1 skipped: the survivors of the previous run (cmd/diff.hpp:329-337).  The reference
// takes total_kmers from the loaded options, which options.bin does not hold (diff_opt.hpp:78-88);
// here it is kept in resume.txt next to options.bin.
void load_previous_survivors(const run_context& C, survivors_of_run& O)
{
  const diff_options& opt = C.opt;
  const kmtricks_config& cfg = C.cfg;
  const size_t S = C.S;
  const bool want_counts = C.want_counts;
  const std::string& part_dir = C.part_dir;
  survivor_set& sv_all = O.sv_all;
  std::vector<size_t>& part_begin = O.part_begin;
  uint64_t& total_kmers = O.total_kmers; uint64_t& n_sig = O.n_sig; uint64_t& n_sig_control = O.n_sig_control; uint64_t& n_sig_case = O.n_sig_case;

  std::fprintf(stderr, "[kmdiff-hip] Resume: partitions/p*_uncorrected of the previous run\n");
  for (size_t p = 0; p < cfg.nb_partitions; ++p)
  {
    read_survivor_file(part_dir + "/p" + std::to_string(p) + "_uncorrected", sv_all);
    part_begin[p + 1] = sv_all.size();
  }
  std::ifstream rs(opt.output_directory + "/resume.txt");
  if (!(rs >> total_kmers >> n_sig >> n_sig_control >> n_sig_case)) die("resume.txt of the previous run is missing");
  if (want_counts && sv_all.size() && sv_all.n_counts != S) die("previous run's survivor files hold another sample count");
}


// ---- the run directory and the devices -> run_context (src/main.cc:74-75, cmd/diff.hpp:109-133)
void open_run(run_context& C)
{
  diff_options& opt = C.opt;
  int& ndev = C.ndev;
  if (kmd_device_count(&ndev) != KMD_OK || ndev < 1) die("no HIP device: kmdiff-hip has no CPU path");
  ck(kmd_set_device(opt.device % ndev), "kmd_set_device");
  char dn[256]; ck(kmd_device_name(dn, sizeof dn), "kmd_device_name");
  std::fprintf(stderr, "[kmdiff-hip] device %d: %s\n", opt.device % ndev, dn);

  C.cfg = get_kmtricks_config(opt.kmtricks_dir);
  const kmtricks_config& cfg = C.cfg;                       // src/main.cc:74
  if (cfg.kmer_size > 64) die("k > 64 is not supported");
  C.two_limbs = cfg.kmer_size > 32;                                                // KSIZE 32 / 64, src/main.cc:75
  C.fof = read_fof(opt.kmtricks_dir);
  const std::vector<fof_entry>& fof = C.fof;
  C.S = opt.nb_controls + opt.nb_cases;
  const size_t S = C.S;
  if (fof.size() < S) die("kmtricks.fof has fewer samples than -1 + -2");
  C.total_controls.assign(opt.nb_controls, 0); C.total_cases.assign(opt.nb_cases, 0);
  std::vector<uint64_t>& total_controls = C.total_controls;
  std::vector<uint64_t>& total_cases = C.total_cases;       // cmd/diff.hpp:111
  for (size_t i = 0; i < opt.nb_controls; ++i) total_controls[i] = sample_total(opt.kmtricks_dir, fof[i], cfg.abundance_min);
  for (size_t i = 0; i < opt.nb_cases; ++i) total_cases[i] = sample_total(opt.kmtricks_dir, fof[opt.nb_controls + i], cfg.abundance_min);

  kmd_model*& model0 = C.model0;                                                              // cmd/diff.hpp:117-123
  ck(kmd_model_create(&model0, (int)opt.nb_controls, (int)opt.nb_cases, total_controls.data(), total_cases.data(), opt.log_size),
     "kmd_model_create");
  C.first_threshold = opt.threshold / (double)opt.cutoff;                       // cmd/diff.hpp:147
  // A user's model plugin (model_manager.hpp:33-94: dlopen, plugin_name, create<bits>, configure).
  // Arbitrary host code cannot run on the device: its rows are evaluated by the reference's own
  // loop -- process(controls, cases) per row, one instance shared by all workers (it has to be
  // re-entrant there too, merge.hpp:418) --, everything around it (merge, correction) stays on
  // the GPU.  Pop-strat correction is switched off with custom models, cmd/diff.hpp:127-132.
  std::shared_ptr<kmdiff::IModel<kmdiff::maxc32>>& plugin = C.plugin;
  if (!opt.model_lib_path.empty())
  {
    void* h = dlopen(opt.model_lib_path.c_str(), RTLD_LAZY);
    if (!h) die(std::string("--cmodel: ") + dlerror());
    auto name = reinterpret_cast<std::string (*)()>(dlsym(h, "plugin_name"));
    auto create = reinterpret_cast<kmdiff::IModel<kmdiff::maxc32>* (*)()>(dlsym(h, "create32"));
    if (!name || !create) die(std::string("--cmodel: ") + dlerror());
    std::fprintf(stderr, "[kmdiff-hip] model plugin: %s\n", name().c_str());
    plugin.reset(create());
    plugin->configure(opt.model_config);
    if (opt.pop_correction) std::fprintf(stderr, "[kmdiff-hip] warning: population stratification correction disabled with custom models.\n");
    opt.pop_correction = false;
  }
  C.n_workers = (size_t)std::max(1, opt.devices == 0 ? ndev : opt.devices);       // GPUs (folded onto the ones there are)
  C.part_dir = opt.output_directory + "/partitions";
  C.pop_dir = opt.output_directory + "/popstrat";
  C.want_counts = opt.pop_correction || opt.keep_tmp || opt.save_sk;
}

// ---- stage 2: do_pop (cmd/diff.hpp:167-224): the survivors' p-values from the pop-strat model, or
// (resume) from the p<i>_popstrat_uncorrected files of the previous run
void do_pop(const run_context& C, survivors_of_run& O, const bool run_stage2)
{
  const diff_options& opt = C.opt;
  const kmtricks_config& cfg = C.cfg;
  const size_t S = C.S;
  const std::vector<uint64_t>& total_controls = C.total_controls;
  const std::vector<uint64_t>& total_cases = C.total_cases;
  const std::string& part_dir = C.part_dir;
  const std::string& pop_dir = C.pop_dir;
  survivor_set& sv_all = O.sv_all;
  const std::vector<size_t>& part_begin = O.part_begin;
  const std::vector<double>& Z_device = O.Z_device;
  std::vector<double>& s_p = sv_all.p;
  const std::vector<double>& s_counts = sv_all.counts;
  const size_t n = s_p.size();
  if (opt.pop_correction && !run_stage2)
  {
    std::fprintf(stderr, "[kmdiff-hip] Resume: partitions/p*_popstrat_uncorrected of the previous run\n");
    survivor_set ps;
    for (size_t p = 0; p < cfg.nb_partitions; ++p) read_survivor_file(part_dir + "/p" + std::to_string(p) + "_popstrat_uncorrected", ps);
    if (ps.size() != n) die("previous run's pop-strat survivor files do not match");
    s_p = ps.p;
  }
  const stopwatch pop_time;
  if (run_stage2 && n)
  {
    std::vector<double> Z(S * 10, 0.0), Y(S, 0.0);
    if (!Z_device.empty()) Z = Z_device;
    else
    {
      const std::string zpath = opt.pcs.empty() ? pop_dir + "/pcs.evec" : opt.pcs;
      std::ifstream zin(zpath);
      if (!zin) die("cannot open " + zpath);
      const size_t per_row = opt.pcs.empty() ? std::min<size_t>(S, 10) : 10;
      for (size_t i = 0; i < S; ++i)
        for (size_t k = 0; k < per_row; ++k)
          if (!(zin >> Z[i * 10 + k])) die(zpath + ": expected " + std::to_string(per_row) + " values per sample");   // popstrat.cpp:153-161
    }
    for (size_t i = 0; i < opt.nb_controls; ++i) Y[i] = 1.0;                                                       // popstrat.cpp:168
    kmd_popstrat* ps = nullptr;
    ck(kmd_popstrat_create(&ps, (int)opt.nb_controls, (int)opt.nb_cases, total_controls.data(), total_cases.data(), Z.data(), 10,
                           (int)opt.npc, Y.data(), opt.stand ? 1 : 0, (int)opt.max_iteration), "kmd_popstrat_create");
    ck(kmd_popstrat_set_epsilon(ps, opt.epsilon), "kmd_popstrat_set_epsilon");             // cmd/diff.hpp:356
    dev_buf d_c, d_p; d_c.reserve(n * S * 8); d_p.reserve(n * 8);
    ck(kmd_memcpy_h2d(d_c.p, s_counts.data(), n * S * 8, nullptr), "h2d");
    ck(kmd_popstrat_apply(ps, (const double*)d_c.p, 0, 0, n, (double*)d_p.p, nullptr), "kmd_popstrat_apply");
    ck(kmd_memcpy_d2h(s_p.data(), d_p.p, n * 8, nullptr), "d2h");                                                  // ks.set_pval
    kmd_popstrat_destroy(ps);
    std::fprintf(stderr, "[kmdiff-hip] Population correction done. (%.3f s)\n", pop_time.seconds());              // cmd/diff.hpp:221
  }
  if (run_stage2 && opt.keep_tmp)
    for (size_t p = 0; p < cfg.nb_partitions; ++p)
      write_survivor_file(part_dir + "/p" + std::to_string(p) + "_popstrat_uncorrected", sv_all, part_begin[p], part_begin[p + 1] - part_begin[p]);
}

// ---- stage 3: do_correction (cmd/diff.hpp:227-260) and the writers (aggregator.hpp:26-71)
void do_correction(const run_context& C, survivors_of_run& O)
{
  const diff_options& opt = C.opt;
  const kmtricks_config& cfg = C.cfg;
  const bool two_limbs = C.two_limbs;
  const survivor_set& sv_all = O.sv_all;
  const uint64_t total_kmers = O.total_kmers, n_sig = O.n_sig, n_sig_control = O.n_sig_control, n_sig_case = O.n_sig_case;
  const std::vector<uint64_t>& s_kmer = sv_all.kmer;
  const std::vector<double>& s_p = sv_all.p;
  const std::vector<double>& s_mc = sv_all.mean_control;
  const std::vector<double>& s_mk = sv_all.mean_case;
  const std::vector<int32_t>& s_sign = sv_all.sign;
  const size_t n = s_p.size();
  dev_buf d_p, d_sign, d_keep;
  d_p.reserve(std::max<size_t>(n, 1) * 8); d_sign.reserve(std::max<size_t>(n, 1) * 4); d_keep.reserve(std::max<size_t>(n, 1));
  uint64_t kept = 0, c_controls = 0, c_cases = 0;
  std::vector<uint8_t> keep(n, 0);
  const size_t n_workers = C.n_workers;
  if (n_workers > 1)
  {
    // --devices N: the survivors of partition p belong to GPU p mod N (where stage 1 found them); every GPU decides
    // its own with kmd_correct_sharded -- the counters' all-reduce, and for BH / Holm the histogram all-gather and the
    // exact walk over the gathered tails (aggregator.hpp:286-310, 325-339) -- over the in-process transport (one host
    // thread per GPU; with one process per GPU it would be libkmdiff_hip_rccl.so's).  Same decisions as one list.
    std::vector<kmd_transport> T(n_workers);
    ck(kmd_transport_local_create((int)n_workers, T.data()), "kmd_transport_local_create");
    std::vector<std::vector<size_t>> mine(n_workers);
    for (size_t p = 0; p < cfg.nb_partitions; ++p)
      for (size_t i = O.part_begin[p]; i < O.part_begin[p + 1]; ++i) mine[p % n_workers].push_back(i);
    std::vector<std::string> errors(n_workers);
    std::vector<uint64_t> w_ctrl(n_workers, 0), w_case(n_workers, 0);
    auto rank_body = [&](size_t wi)
    {
      try
      {
        ck(kmd_set_device((opt.device + (int)wi) % C.ndev), "kmd_set_device");
        if (const char* e = std::getenv("KMD_TEST_FAIL_RANK"))          // dev / tests: this rank fails before the exchange
          if ((size_t)std::atoi(e) == wi) throw std::runtime_error("KMD_TEST_FAIL_RANK: rank " + std::to_string(wi) + " fails before the exchange");
        const size_t m = mine[wi].size();
        std::vector<double> lp(m); std::vector<int32_t> ls(m); std::vector<uint8_t> lk(m, 0);
        for (size_t j = 0; j < m; ++j) { lp[j] = s_p[mine[wi][j]]; ls[j] = s_sign[mine[wi][j]]; }
        dev_buf b_p, b_s, b_k;
        b_p.reserve(std::max<size_t>(m, 1) * 8); b_s.reserve(std::max<size_t>(m, 1) * 4); b_k.reserve(std::max<size_t>(m, 1));
        if (m)
        {
          ck(kmd_memcpy_h2d(b_p.p, lp.data(), m * 8, nullptr), "h2d");
          ck(kmd_memcpy_h2d(b_s.p, ls.data(), m * 4, nullptr), "h2d");
        }
        uint64_t local[KMD_NCOUNTERS] = { 0 }, global[KMD_NCOUNTERS] = { 0 }, k_ = 0;
        local[KMD_CNT_TOTAL] = O.worker_totals.size() == n_workers ? O.worker_totals[wi] : (wi == 0 ? total_kmers : 0);
        local[KMD_CNT_SIG] = m;
        ck(kmd_correct_sharded(&T[wi], correction_type(opt.correction), opt.threshold, local, global, (const double*)b_p.p, (const int32_t*)b_s.p, m,
                               (uint8_t*)b_k.p, &k_, &w_ctrl[wi], &w_case[wi], nullptr), "kmd_correct_sharded");
        if (global[KMD_CNT_TOTAL] != total_kmers) throw std::runtime_error("kmd_correct_sharded: the ranks' totals do not add up");
        if (m) ck(kmd_memcpy_d2h(lk.data(), b_k.p, m, nullptr), "d2h");
        for (size_t j = 0; j < m; ++j) keep[mine[wi][j]] = lk[j];
      }
      catch (const std::exception& e)
      {
        // (the other ranks may be inside a collective, waiting for this one: they are told, and fail instead)
        errors[wi] = e.what();
        kmd_transport_abort(&T[wi]);
      }
    };
    {
      std::vector<std::thread> ranks;
      for (size_t wi = 1; wi < n_workers; ++wi) ranks.emplace_back(rank_body, wi);
      rank_body(0);
      for (auto& t : ranks) t.join();
    }
    kmd_transport_local_destroy((int)n_workers, T.data());
    ck(kmd_set_device(opt.device % C.ndev), "kmd_set_device");
    // (the rank that failed first, not the ones that came back because it had: "another rank gave up")
    for (const std::string& e : errors) if (!e.empty() && e.find("another rank gave up") == std::string::npos) die(e);
    for (const std::string& e : errors) if (!e.empty()) die(e);
    for (size_t wi = 0; wi < n_workers; ++wi) { c_controls += w_ctrl[wi]; c_cases += w_case[wi]; }
    kept = c_controls + c_cases;
  }
  else if (n)
  {
    ck(kmd_memcpy_h2d(d_p.p, s_p.data(), n * 8, nullptr), "h2d");
    ck(kmd_memcpy_h2d(d_sign.p, s_sign.data(), n * 4, nullptr), "h2d");
    ck(kmd_correct(correction_type(opt.correction), opt.threshold, total_kmers, (const double*)d_p.p, (const int32_t*)d_sign.p, n,
                   (uint8_t*)d_keep.p, &kept, &c_controls, &c_cases, nullptr), "kmd_correct");
    ck(kmd_memcpy_d2h(keep.data(), d_keep.p, n, nullptr), "d2h");
  }
  fs::create_directories(opt.output_directory);
  // writers (aggregator.hpp:26-71).  BH/Holm emit in ascending p (the order the sorted
  // aggregator pops them); the stateless correctors in partition order.
  std::vector<size_t> order(n);
  for (size_t i = 0; i < n; ++i) order[i] = i;
  if (opt.correction == "benjamini" || opt.correction == "holm")
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return s_p[a] < s_p[b]; });
  // aggregator.hpp:197: ".kff" or ".fasta"; a KFF file holds the k-mers only (kff_utils.hpp:49-56)
  std::unique_ptr<kff_writer> kc, kk;
  std::ofstream fc, fk;
  if (opt.kff)
  {
    kc = std::make_unique<kff_writer>(opt.output_directory + "/control_kmers.kff", cfg.kmer_size);
    kk = std::make_unique<kff_writer>(opt.output_directory + "/case_kmers.kff", cfg.kmer_size);
  }
  else { fc.open(opt.output_directory + "/control_kmers.fasta"); fk.open(opt.output_directory + "/case_kmers.fasta"); }
  size_t ic = 0, ik = 0;
  for (size_t i : order)
  {
    if (!keep[i]) continue;
    const bool control = s_sign[i] == KMD_SIGN_CONTROL;                                     // aggregator.hpp:155-162
    std::ofstream& f = control ? fc : fk;
    size_t& idx = control ? ic : ik;
    if (opt.kff)
    {
      (control ? kc : kk)->write(s_kmer[i], two_limbs ? sv_all.kmer_hi[i] : 0);
      ++idx;
      continue;
    }
    char pv[64]; std::snprintf(pv, sizeof pv, "%g", s_p[i]);                               // {:g}
    f << '>' << idx << "_pval=" << pv << "_control=" << (uint64_t)s_mc[i] << "_case=" << shortest(s_mk[i]) << '\n'
      << (two_limbs ? kmer_to_string(sv_all.kmer_hi[i], s_kmer[i], cfg.kmer_size) : kmer_to_string(s_kmer[i], cfg.kmer_size)) << '\n';
    ++idx;
  }
  if (opt.kff) { kc->close(); kk->close(); }
  std::fprintf(stderr, "[kmdiff-hip] Significant k-mers: %" PRIu64 " (control), %" PRIu64 " (case).\n", c_controls, c_cases);   // cmd/diff.hpp:259
  // machine-readable summary for tests and scripts
  std::ofstream js(opt.output_directory + "/summary.json");
  js << "{\"total_kmers\": " << total_kmers << ", \"n_sig\": " << n_sig << ", \"n_sig_control\": " << n_sig_control
     << ", \"n_sig_case\": " << n_sig_case << ", \"kept\": " << kept << ", \"kept_control\": " << c_controls
     << ", \"kept_case\": " << c_cases << ", \"near_threshold\": " << O.n_near << ", \"kmer_size\": " << cfg.kmer_size
     << ", \"nb_partitions\": " << cfg.nb_partitions;
  if (O.records)                                         // (stage 1 ran on the k-mer feed)
  {
    char bpr[32]; std::snprintf(bpr, sizeof bpr, "%.3f", (double)O.h2d_bytes / (double)O.records);
    js << ", \"transfer\": {\"format\": \"" << (O.packed ? "packed" : "raw") << "\", \"records\": " << O.records << ", \"h2d_bytes\": " << O.h2d_bytes
       << ", \"bytes_per_record\": " << bpr << "}";
  }
  js << "}\n";
}

} // namespace

int main(int argc, char** argv)
{
  // test hook (tests/test_host_io.py): `kmdiff-hip fmt <double>...` prints the FASTA header's `case=` rendering of each
  if (argc >= 2 && std::string(argv[1]) == "fmt")
  {
    for (int i = 2; i < argc; ++i) std::printf("%s\n", shortest(std::strtod(argv[i], nullptr)).c_str());
    return 0;
  }
  run_context C;
  retired_staging retired;                               // (joined when main returns, whichever way)
  g_retired = &retired;
  try
  {
    C.opt = parse(argc, argv);
    const stopwatch whole_time;
    open_run(C);
    const diff_options& opt = C.opt;

    // ---- what a previous run left behind (cmd/diff.hpp:278-303)
    fs::create_directories(C.part_dir);
    resume_options ropt; ropt.threshold = opt.threshold; ropt.cutoff = (double)opt.cutoff;
    ropt.correction = correction_type(opt.correction); ropt.pop_correction = opt.pop_correction;
    ropt.kmer_pca = opt.kmer_pca; ropt.npc = opt.npc;
    resume_options prev;
    const bool prev_run = load_opt(opt.output_directory + "/options.bin", prev);
    auto all_exist = [&](const char* suffix)
    {
      for (size_t p = 0; p < C.cfg.nb_partitions; ++p)
        if (!fs::exists(C.part_dir + "/p" + std::to_string(p) + suffix)) return false;
      return true;
    };
    const unsigned action = prev_run ? compare_opt(ropt, prev) : 0;
    const bool prev_1 = prev_run && all_exist("_uncorrected");
    const bool prev_2 = prev_run && all_exist("_popstrat_uncorrected");

    survivors_of_run O;
    O.sv_all.n_counts = C.want_counts ? C.S : 0;
    O.sv_all.kmer_bytes = C.two_limbs ? 16 : 8;
    O.part_begin.assign(C.cfg.nb_partitions + 1, 0);
    // the device PCA needs stage 1 (it samples the rows as they pass); its result of a previous
    // run is popstrat/pcs.evec
    const bool device_pca = opt.pop_correction && opt.pcs.empty();
    const bool have_pcs = fs::exists(C.pop_dir + "/pcs.evec");
    const bool run_stage1 = !prev_1 || (action & 0b1) || (device_pca && !have_pcs);
    if (run_stage1) do_diff(C, O, /* run_pca */ device_pca);
    else load_previous_survivors(C, O);
    dump_opt(ropt, opt.output_directory + "/options.bin");
    {
      std::ofstream rs(opt.output_directory + "/resume.txt");
      rs << O.total_kmers << ' ' << O.n_sig << ' ' << O.n_sig_control << ' ' << O.n_sig_case << '\n';
    }
    if (opt.pop_correction) do_pop(C, O, /* run_stage2 */ !prev_2 || (action & 0b10) || run_stage1);       // cmd/diff.hpp:349
    do_correction(C, O);
    kmd_model_destroy(C.model0);
    std::fprintf(stderr, "[kmdiff-hip] Done in %.3f s.\n", whole_time.seconds());                                   // cmd/diff.hpp:372-376
  }
  catch (const std::exception& e)                                                             // src/main.cc:93-102 logs and exits
  {
    std::fprintf(stderr, "[kmdiff-hip] error: %s\n", e.what());
    return 1;
  }
  return 0;
}
