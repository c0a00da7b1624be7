#include "kmtricks_io.hpp"

#include <algorithm>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace fs = std::filesystem;

// liblz4 frame API (lz4frame.h), declared here so the build needs only the shared library
extern "C" {
typedef struct LZ4F_dctx_s LZ4F_dctx;
size_t LZ4F_createDecompressionContext(LZ4F_dctx** dctxPtr, unsigned version);
size_t LZ4F_freeDecompressionContext(LZ4F_dctx* dctx);
size_t LZ4F_decompress(LZ4F_dctx* dctx, void* dst, size_t* dstSizePtr, const void* src, size_t* srcSizePtr,
                       const void* options);
unsigned LZ4F_isError(size_t code);
const char* LZ4F_getErrorName(size_t code);
}

namespace kmd_host {

static std::string trim(const std::string& s)
{
  size_t b = s.find_first_not_of(" \t\r\n"), e = s.find_last_not_of(" \t\r\n");
  return b == std::string::npos ? std::string() : s.substr(b, e - b + 1);
}

static std::vector<std::string> split(const std::string& s, char c)
{
  std::vector<std::string> out;
  std::stringstream ss(s);
  for (std::string item; std::getline(ss, item, c);) out.push_back(item);
  return out;
}

kmtricks_config get_kmtricks_config(const std::string& run_dir)
{
  kmtricks_config cfg;
  std::string path = run_dir + "/kmdiff-count.opt", key = "abundance_min";
  if (!fs::exists(path)) { path = run_dir + "/options.txt"; key = "c_ab_min"; }
  std::ifstream in(path);
  for (std::string line; std::getline(in, line);)
  {
    if (line.find("kmer_size") == std::string::npos) continue;
    for (auto o : split(line, ','))
    {
      o = trim(o);
      auto kv = split(o, '=');
      if (kv.size() < 2) continue;
      if (o.find("kmer_size") != std::string::npos) cfg.kmer_size = std::stoull(kv[1]);
      if (o.find(key) != std::string::npos) cfg.abundance_min = std::stoull(kv[1]);
    }
  }
  if (fs::exists(run_dir + "/counts"))
    for (auto& e : fs::directory_iterator(run_dir + "/counts")) { (void)e; cfg.nb_partitions++; }
  if (!cfg.kmer_size || !cfg.nb_partitions)
    throw std::runtime_error("Unable to load config from " + path + ".");
  return cfg;
}

std::vector<fof_entry> read_fof(const std::string& run_dir)
{
  std::vector<fof_entry> out;
  std::ifstream in(run_dir + "/kmtricks.fof");
  if (!in) throw std::runtime_error("cannot open " + run_dir + "/kmtricks.fof");
  for (std::string line; std::getline(in, line);)
  {
    line = trim(line);
    if (line.empty()) continue;
    fof_entry e;
    auto colon = line.find(':');
    e.id = trim(line.substr(0, colon));
    auto bang = line.find('!');
    if (bang != std::string::npos) e.ab_min = std::stoull(trim(line.substr(bang + 1)));
    out.push_back(e);
  }
  return out;
}

static std::vector<char> slurp(const std::string& path)
{
  std::ifstream in(path, std::ios::binary);
  if (!in) throw std::runtime_error("cannot open " + path);
  return std::vector<char>((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
}

template <typename T> static T rd(const std::vector<char>& d, size_t off)
{
  if (off + sizeof(T) > d.size()) throw std::runtime_error("truncated kmtricks file");
  T v; std::memcpy(&v, d.data() + off, sizeof(T)); return v;
}

uint64_t sample_total(const std::string& run_dir, const fof_entry& sample, size_t abundance_min)
{
  const std::string path = run_dir + "/histograms/" + sample.id + ".hist";
  auto d = slurp(path);
  if (d.size() < 93 || std::memcmp(d.data(), "kmtricks", 8) != 0 || std::memcmp(d.data() + 13, "khist", 5) != 0)
    throw std::runtime_error(path + ": not a kmtricks histogram");
  const uint64_t lower = rd<uint64_t>(d, 29), upper = rd<uint64_t>(d, 37);
  uint64_t total = rd<uint64_t>(d, 53);
  const size_t n = (size_t)(upper - lower + 1);
  size_t ab_min = sample.ab_min ? sample.ab_min : abundance_min;        // kmtricks_utils.cpp:99-101
  for (size_t j = 1; j < ab_min; ++j)                                     // :105-108
    if (j - 1 < n) total -= (uint64_t)j * rd<uint64_t>(d, 93 + 8 * (j - 1));
  return total;
}

std::string kmer_file_path(const std::string& run_dir, size_t partition, const std::string& id)
{
  return run_dir + "/counts/partition_" + std::to_string(partition) + "/" + id + ".kmer.lz4";
}

size_t read_kmer_file(const std::string& path, size_t expected_k, std::vector<uint64_t>& kmers,
                      std::vector<uint32_t>& counts)
{
  auto d = slurp(path);
  if (d.size() < 41 || std::memcmp(d.data(), "kmtricks", 8) != 0 || std::memcmp(d.data() + 13, "kmer", 4) != 0)
    throw std::runtime_error(path + ": not a kmtricks k-mer file");
  const uint8_t compressed = rd<uint8_t>(d, 12);
  const uint32_t k = rd<uint32_t>(d, 21), slots = rd<uint32_t>(d, 25), cbytes = rd<uint32_t>(d, 29);
  if (expected_k && k != expected_k) throw std::runtime_error(path + ": k-mer size differs from the run's");
  if (slots != 1) throw std::runtime_error(path + ": k > 32 is not supported by this reader yet");
  if (cbytes != 1 && cbytes != 2 && cbytes != 4) throw std::runtime_error(path + ": bad count width");
  std::vector<char> raw;
  if (compressed)
  {
    LZ4F_dctx* ctx = nullptr;
    if (LZ4F_isError(LZ4F_createDecompressionContext(&ctx, 100))) throw std::runtime_error("LZ4F context");
    std::vector<char> buf(1 << 20);
    size_t pos = 41;
    while (pos < d.size())
    {
      size_t dn = buf.size(), sn = d.size() - pos;
      size_t r = LZ4F_decompress(ctx, buf.data(), &dn, d.data() + pos, &sn, nullptr);
      if (LZ4F_isError(r)) { LZ4F_freeDecompressionContext(ctx); throw std::runtime_error(path + ": " + LZ4F_getErrorName(r)); }
      raw.insert(raw.end(), buf.begin(), buf.begin() + dn);
      pos += sn;
      if (r == 0 && sn == 0) break;
    }
    LZ4F_freeDecompressionContext(ctx);
  }
  else raw.assign(d.begin() + 41, d.end());
  const size_t rec = 8 * slots + cbytes, n = raw.size() / rec;
  kmers.reserve(kmers.size() + n); counts.reserve(counts.size() + n);
  for (size_t i = 0; i < n; ++i)
  {
    uint64_t km; std::memcpy(&km, raw.data() + i * rec, 8);
    uint32_t c = 0; std::memcpy(&c, raw.data() + i * rec + 8, cbytes);
    kmers.push_back(km); counts.push_back(c);
  }
  return n;
}

std::string kmer_to_string(uint64_t kmer, size_t k)
{
  static const char code[4] = { 'A', 'C', 'T', 'G' };
  std::string s(k, 'A');
  for (size_t i = 0; i < k; ++i) s[i] = code[(kmer >> (2 * (k - 1 - i))) & 3];
  return s;
}

} // namespace kmd_host
