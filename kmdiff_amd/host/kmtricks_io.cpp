#include "kmtricks_io.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

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
typedef struct LZ4F_cctx_s LZ4F_cctx;
size_t LZ4F_createCompressionContext(LZ4F_cctx** cctxPtr, unsigned version);
size_t LZ4F_freeCompressionContext(LZ4F_cctx* cctx);
size_t LZ4F_compressBegin(LZ4F_cctx* cctx, void* dst, size_t dstCapacity, const void* prefs);
size_t LZ4F_compressBound(size_t srcSize, const void* prefs);
size_t LZ4F_compressUpdate(LZ4F_cctx* cctx, void* dst, size_t dstCapacity, const void* src, size_t srcSize,
                           const void* options);
size_t LZ4F_compressEnd(LZ4F_cctx* cctx, void* dst, size_t dstCapacity, const void* options);
}

namespace kmd_host {

constexpr uint32_t kMaxMatrixSamples = 1u << 20;       // rows of a count matrix: more samples than this is a damaged header

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
  std::ifstream in(path, std::ios::binary | std::ios::ate);
  if (!in) throw std::runtime_error("cannot open " + path);
  const std::streamsize size = in.tellg();
  std::vector<char> d((size_t)(size > 0 ? size : 0));
  in.seekg(0);
  if (size > 0 && !in.read(d.data(), size)) throw std::runtime_error("cannot read " + path);
  return d;
}

// one LZ4 frame (what lz4_stream and kmtricks write) starting at d[off]
static std::vector<char> lz4_frame_decode(const std::vector<char>& d, size_t off, const std::string& path)
{
  std::vector<char> raw;
  raw.reserve((d.size() - off) * 2);
  LZ4F_dctx* ctx = nullptr;
  if (LZ4F_isError(LZ4F_createDecompressionContext(&ctx, 100))) throw std::runtime_error("LZ4F context");
  std::vector<char> buf(1 << 20);
  size_t pos = off;
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
  return raw;
}

// payload -> one LZ4 frame appended to the stream (default preferences: 64 KB linked blocks)
static void lz4_frame_encode(std::ostream& out, const char* src, size_t n)
{
  LZ4F_cctx* ctx = nullptr;
  if (LZ4F_isError(LZ4F_createCompressionContext(&ctx, 100))) throw std::runtime_error("LZ4F context");
  const size_t chunk = 1 << 20;
  std::vector<char> buf(LZ4F_compressBound(chunk, nullptr) + 64);
  auto put = [&](size_t r)
  {
    if (LZ4F_isError(r)) { LZ4F_freeCompressionContext(ctx); throw std::runtime_error(std::string("LZ4F: ") + LZ4F_getErrorName(r)); }
    out.write(buf.data(), (std::streamsize)r);
  };
  put(LZ4F_compressBegin(ctx, buf.data(), buf.size(), nullptr));
  for (size_t pos = 0; pos < n; pos += chunk)
    put(LZ4F_compressUpdate(ctx, buf.data(), buf.size(), src + pos, std::min(chunk, n - pos), nullptr));
  put(LZ4F_compressEnd(ctx, buf.data(), buf.size(), nullptr));
  LZ4F_freeCompressionContext(ctx);
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
                      std::vector<uint32_t>& counts, std::vector<uint64_t>* kmers_hi)
{
  auto d = slurp(path);
  if (d.size() < 41 || std::memcmp(d.data(), "kmtricks", 8) != 0 || std::memcmp(d.data() + 13, "kmer", 4) != 0)
    throw std::runtime_error(path + ": not a kmtricks k-mer file");
  const uint8_t compressed = rd<uint8_t>(d, 12);
  const uint32_t k = rd<uint32_t>(d, 21), slots = rd<uint32_t>(d, 25), cbytes = rd<uint32_t>(d, 29);
  if (expected_k && k != expected_k) throw std::runtime_error(path + ": k-mer size differs from the run's");
  if (slots != 1 && slots != 2) throw std::runtime_error(path + ": k > 64 is not supported");
  if (slots == 2 && !kmers_hi) throw std::runtime_error(path + ": two-limb k-mers need a high-limb sink");
  if (cbytes != 1 && cbytes != 2 && cbytes != 4) throw std::runtime_error(path + ": bad count width");
  std::vector<char> raw;
  if (compressed) raw = lz4_frame_decode(d, 41, path);
  else raw.assign(d.begin() + 41, d.end());
  const size_t rec = 8 * slots + cbytes, n = raw.size() / rec;
  kmers.reserve(kmers.size() + n); counts.reserve(counts.size() + n);
  for (size_t i = 0; i < n; ++i)
  {
    uint64_t km; std::memcpy(&km, raw.data() + i * rec, 8);
    uint32_t c = 0; std::memcpy(&c, raw.data() + i * rec + 8 * slots, cbytes);
    kmers.push_back(km); counts.push_back(c);
    if (slots == 2) { uint64_t h; std::memcpy(&h, raw.data() + i * rec + 8, 8); kmers_hi->push_back(h); }
    else if (kmers_hi) kmers_hi->push_back(0);
  }
  return n;
}

// ---------------------------------------------------------------------------------------------
std::vector<std::string> matrix_paths(const std::string& run_dir)
{
  std::vector<std::string> out;
  const std::string dir = run_dir + "/matrices";
  if (fs::exists(dir))
    for (auto& e : fs::directory_iterator(dir)) if (fs::is_regular_file(e.path())) out.push_back(e.path().string());
  std::sort(out.begin(), out.end());
  return out;
}

matrix_rows read_matrix_file(const std::string& path)
{
  auto d = slurp(path);
  if (d.size() < 45 || std::memcmp(d.data(), "kmtricks", 8) != 0 || std::memcmp(d.data() + 13, "matrix", 6) != 0)
    throw std::runtime_error(path + ": not a kmtricks count matrix");
  matrix_rows m;
  const uint8_t compressed = rd<uint8_t>(d, 12);
  m.kmer_size = rd<uint32_t>(d, 21);
  const uint32_t slots = rd<uint32_t>(d, 25);
  m.count_bytes = rd<uint32_t>(d, 29);
  m.nb_counts = rd<uint32_t>(d, 33);
  m.partition = rd<uint32_t>(d, 41);
  if (slots != 1 && slots != 2) throw std::runtime_error(path + ": k > 64 is not supported");
  if (m.count_bytes != 1 && m.count_bytes != 2 && m.count_bytes != 4) throw std::runtime_error(path + ": bad count width");
  if (m.nb_counts > kMaxMatrixSamples) throw std::runtime_error(path + ": implausible number of samples in the header");
  std::vector<char> raw;
  if (compressed) raw = lz4_frame_decode(d, 45, path); else raw.assign(d.begin() + 45, d.end());
  const size_t kb = 8 * (size_t)slots, rec = kb + (size_t)m.count_bytes * m.nb_counts, n = raw.size() / rec;
  if (n * rec != raw.size()) throw std::runtime_error(path + ": truncated row");
  m.kmers.resize(n); m.counts.assign(n * m.nb_counts, 0);
  if (slots == 2) m.kmers_hi.resize(n);
  for (size_t i = 0; i < n; ++i)
  {
    const char* r = raw.data() + i * rec;
    std::memcpy(&m.kmers[i], r, 8);
    if (slots == 2) std::memcpy(&m.kmers_hi[i], r + 8, 8);
    for (uint32_t s = 0; s < m.nb_counts; ++s) std::memcpy(&m.counts[i * m.nb_counts + s], r + kb + (size_t)s * m.count_bytes, m.count_bytes);
  }
  return m;
}

void write_matrix_file(const std::string& path, const matrix_rows& m)
{
  std::ofstream out(path, std::ios::binary);
  if (!out) throw std::runtime_error("cannot write " + path);
  const uint32_t zero = 0, slots = m.kmers_hi.empty() ? 1 : 2;
  const uint8_t compressed = 1;
  out.write("kmtricks", 8); out.write((const char*)&zero, 4); out.write((const char*)&compressed, 1);
  out.write("matrix\0\0", 8);
  out.write((const char*)&m.kmer_size, 4); out.write((const char*)&slots, 4); out.write((const char*)&m.count_bytes, 4);
  out.write((const char*)&m.nb_counts, 4); out.write((const char*)&zero, 4); out.write((const char*)&m.partition, 4);
  const size_t kb = 8 * (size_t)slots, rec = kb + (size_t)m.count_bytes * m.nb_counts, n = m.kmers.size();
  std::vector<char> raw(n * rec);
  for (size_t i = 0; i < n; ++i)
  {
    char* r = raw.data() + i * rec;
    std::memcpy(r, &m.kmers[i], 8);
    if (slots == 2) std::memcpy(r + 8, &m.kmers_hi[i], 8);
    for (uint32_t s = 0; s < m.nb_counts; ++s) std::memcpy(r + kb + (size_t)s * m.count_bytes, &m.counts[i * m.nb_counts + s], m.count_bytes);
  }
  lz4_frame_encode(out, raw.data(), raw.size());
}

// ---------------------------------------------------------------------------------------------
void write_survivor_file(const std::string& path, const survivor_set& s, size_t first, size_t count)
{
  std::ofstream out(path, std::ios::binary);
  if (!out) throw std::runtime_error("cannot write " + path);
  const uint16_t nc = (uint16_t)s.n_counts;
  const size_t kb = s.kmer_bytes, rec = kb + 8 + 4 + 8 + 8 + 2 + 8 * (size_t)nc;
  std::vector<char> raw(count * rec);
  for (size_t i = 0; i < count; ++i)
  {
    char* r = raw.data() + i * rec;
    const size_t j = first + i;
    std::memcpy(r, &s.kmer[j], 8);
    if (kb == 16) std::memcpy(r + 8, &s.kmer_hi[j], 8);
    r += kb;
    std::memcpy(r, &s.p[j], 8); std::memcpy(r + 8, &s.sign[j], 4);
    std::memcpy(r + 12, &s.mean_control[j], 8); std::memcpy(r + 20, &s.mean_case[j], 8); std::memcpy(r + 28, &nc, 2);
    if (nc) std::memcpy(r + 30, &s.counts[j * nc], 8 * (size_t)nc);
  }
  lz4_frame_encode(out, raw.data(), raw.size());
}

size_t read_survivor_file(const std::string& path, survivor_set& s)
{
  auto d = slurp(path);
  const std::vector<char> raw = d.empty() ? std::vector<char>() : lz4_frame_decode(d, 0, path);
  size_t pos = 0, n = 0;
  const size_t kb = s.kmer_bytes, fixed = kb + 30;
  while (pos + fixed <= raw.size())
  {
    uint64_t km, kh = 0; double p, mc, mk; int32_t sg; uint16_t nc;
    std::memcpy(&km, &raw[pos], 8);
    if (kb == 16) std::memcpy(&kh, &raw[pos + 8], 8);
    const char* r = &raw[pos + kb];
    std::memcpy(&p, r, 8); std::memcpy(&sg, r + 8, 4);
    std::memcpy(&mc, r + 12, 8); std::memcpy(&mk, r + 20, 8); std::memcpy(&nc, r + 28, 2);
    if (s.size() == 0) s.n_counts = nc;
    if (nc != s.n_counts || pos + fixed + 8 * (size_t)nc > raw.size()) throw std::runtime_error(path + ": malformed survivor record");
    s.kmer.push_back(km); s.p.push_back(p); s.sign.push_back(sg); s.mean_control.push_back(mc); s.mean_case.push_back(mk);
    if (kb == 16) s.kmer_hi.push_back(kh);
    s.counts.resize(s.counts.size() + nc);
    if (nc) std::memcpy(&s.counts[s.counts.size() - nc], r + 30, 8 * (size_t)nc);
    pos += fixed + 8 * (size_t)nc; ++n;
  }
  if (pos != raw.size()) throw std::runtime_error(path + ": trailing bytes");
  return n;
}

// ---------------------------------------------------------------------------------------------
void dump_opt(const resume_options& o, const std::string& path)
{
  std::ofstream out(path, std::ios::binary);
  const uint8_t pc = o.pop_correction ? 1 : 0;
  out.write((const char*)&o.threshold, 8); out.write((const char*)&o.cutoff, 8); out.write((const char*)&o.correction, 4);
  out.write((const char*)&pc, 1); out.write((const char*)&o.kmer_pca, 8); out.write((const char*)&o.npc, 8);
}

bool load_opt(const std::string& path, resume_options& o)
{
  std::ifstream in(path, std::ios::binary);
  char b[37];
  if (!in.read(b, 37)) return false;
  uint8_t pc;
  std::memcpy(&o.threshold, b, 8); std::memcpy(&o.cutoff, b + 8, 8); std::memcpy(&o.correction, b + 16, 4);
  std::memcpy(&pc, b + 20, 1); std::memcpy(&o.kmer_pca, b + 21, 8); std::memcpy(&o.npc, b + 29, 8);
  o.pop_correction = pc != 0;
  return true;
}

unsigned compare_opt(const resume_options& opt, const resume_options& prev)      // cmd/diff_opt.hpp:106-133
{
  unsigned r = 0;
  if (opt.threshold != prev.threshold || opt.cutoff != prev.cutoff) r |= 0b1;
  if (prev.pop_correction && opt.pop_correction)
  {
    if (opt.kmer_pca != prev.kmer_pca) r |= 0b11;
    if (opt.npc != prev.npc) r |= 0b10;
  }
  if (!prev.pop_correction && opt.pop_correction) r |= 0b11;
  if (opt.correction != prev.correction) r |= 0b100;
  if (prev.pop_correction && !opt.pop_correction) r |= 0b100;
  return r;
}

kmer_file_raw decode_kmer_file(const std::string& path, size_t expected_k)
{
  auto d = slurp(path);
  if (d.size() < 41 || std::memcmp(d.data(), "kmtricks", 8) != 0 || std::memcmp(d.data() + 13, "kmer", 4) != 0)
    throw std::runtime_error(path + ": not a kmtricks k-mer file");
  kmer_file_raw f;
  const uint8_t compressed = rd<uint8_t>(d, 12);
  const uint32_t k = rd<uint32_t>(d, 21);
  f.slots = rd<uint32_t>(d, 25); f.count_bytes = rd<uint32_t>(d, 29);
  if (expected_k && k != expected_k) throw std::runtime_error(path + ": k-mer size differs from the run's");
  if (f.slots != 1 && f.slots != 2) throw std::runtime_error(path + ": k > 64 is not supported");
  if (f.count_bytes != 1 && f.count_bytes != 2 && f.count_bytes != 4) throw std::runtime_error(path + ": bad count width");
  if (compressed) f.payload = lz4_frame_decode(d, 41, path); else f.payload.assign(d.begin() + 41, d.end());
  f.records = f.payload.size() / (8 * (size_t)f.slots + f.count_bytes);
  return f;
}

void split_records(const kmer_file_raw& f, uint64_t* kmers, uint64_t* kmers_hi, uint32_t* counts)
{
  const size_t rec = 8 * (size_t)f.slots + f.count_bytes;
  const char* p = f.payload.data();
  for (size_t i = 0; i < f.records; ++i, p += rec)
  {
    std::memcpy(&kmers[i], p, 8);
    if (f.slots == 2) std::memcpy(&kmers_hi[i], p + 8, 8);
    else if (kmers_hi) kmers_hi[i] = 0;
    uint32_t c = 0;
    std::memcpy(&c, p + 8 * f.slots, f.count_bytes);
    counts[i] = c;
  }
}

// [limb(s)][count] records -> arrays, n whole records starting at p
template <int SLOTS, int CB>
static void split_fixed(const char* p, size_t n, uint64_t* kmers, uint64_t* kmers_hi, uint32_t* counts)
{
  constexpr size_t rec = 8 * SLOTS + CB;
  for (size_t i = 0; i < n; ++i, p += rec)
  {
    std::memcpy(&kmers[i], p, 8);
    if (SLOTS == 2) std::memcpy(&kmers_hi[i], p + 8, 8);
    else if (kmers_hi) kmers_hi[i] = 0;
    uint32_t c = 0;
    std::memcpy(&c, p + 8 * SLOTS, CB);
    counts[i] = c;
  }
}

namespace {

// A kmtricks file as a stream: header, then the payload (one LZ4 frame, or plain bytes) handed on in
// chunks of whole records.  `on_records(ptr, n)` gets n records starting at ptr; the bytes of a
// record cut by a chunk boundary are carried to the front of the next chunk.
struct payload_stream
{
  int fd = -1;
  std::string path;
  size_t file_size = 0;
  explicit payload_stream(const std::string& p) : path(p)
  {
    fd = ::open(p.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("cannot open " + p);
    struct stat st;
    if (::fstat(fd, &st) != 0) { ::close(fd); throw std::runtime_error("cannot stat " + p); }
    file_size = (size_t)st.st_size;
  }
  ~payload_stream() { if (fd >= 0) ::close(fd); }
  payload_stream(const payload_stream&) = delete;
  payload_stream& operator=(const payload_stream&) = delete;

  size_t read_some(char* dst, size_t want)
  {
    size_t got = 0;
    while (got < want)
    {
      const ssize_t r = ::read(fd, dst + got, want - got);
      if (r < 0) throw std::runtime_error("cannot read " + path);
      if (r == 0) break;
      got += (size_t)r;
    }
    return got;
  }

  // everything after the header; `in` / `out` are the caller's scratch vectors (kept between files)
  // returns the bytes left over after the last whole record
  size_t records(bool compressed, size_t rec, std::vector<char>& in, std::vector<char>& outv,
                 const std::function<void(const char*, size_t)>& on_records)
  {
    static const size_t chunk = []() { const char* e = std::getenv("KMD_IO_CHUNK_KB"); return (size_t)(e ? std::atoi(e) : 1024) << 10; }();
    if (in.size() < chunk) in.resize(chunk);
    if (outv.size() < chunk + rec) outv.resize(chunk + rec);
    char* const out = outv.data();
    size_t carry = 0;                                      // bytes of a record cut by the chunk boundary, at the front of out
    auto take = [&](size_t fresh)                          // out[0, carry + fresh) -> whole records
    {
      const size_t avail = carry + fresh, whole = avail / rec;
      if (whole) on_records(out, whole);
      carry = avail - whole * rec;
      if (carry) std::memmove(out, out + whole * rec, carry);
    };
    if (!compressed)
    {
      for (size_t got; (got = read_some(out + carry, chunk)) != 0;) take(got);
      return carry;
    }
    LZ4F_dctx* ctx = nullptr;
    if (LZ4F_isError(LZ4F_createDecompressionContext(&ctx, 100))) throw std::runtime_error("LZ4F context");
    struct freer { LZ4F_dctx* c; ~freer() { LZ4F_freeDecompressionContext(c); } } ctx_guard { ctx };
    bool done = false;
    for (size_t got; !done && (got = read_some(in.data(), chunk)) != 0;)
    {
      for (size_t pos = 0;;)
      {
        size_t dn = chunk, sn = got - pos;
        const size_t r = LZ4F_decompress(ctx, out + carry, &dn, in.data() + pos, &sn, nullptr);
        if (LZ4F_isError(r)) throw std::runtime_error(path + ": " + LZ4F_getErrorName(r));
        pos += sn;
        if (dn) take(dn);
        if (r == 0) { done = true; break; }                // end of the frame
        if (pos >= got && dn < chunk) break;               // input used up and nothing left to flush
        if (sn == 0 && dn == 0) throw std::runtime_error(path + ": LZ4 frame makes no progress");
      }
    }
    if (!done) throw std::runtime_error(path + ": truncated LZ4 frame (no end mark)");
    return carry;
  }
};

// room for `more` records after the `have` the sink holds: first guess from the file size (sorted
// k-mers + small counts shrink by 1.3-1.5 under LZ4), then by quarters -- page-locking the arrays is
// the fixed cost of a run (~0.15 s per GB, and again to release)
void grow_sink(record_sink& sink, size_t have, size_t more, size_t file_size, size_t rec)
{
  if (have + more <= sink.capacity) return;
  const size_t guess = sink.consume ? 0 : (file_size * 3 / 2) / rec + 1024;       // (a consumer empties the arrays chunk by chunk)
  sink.reserve(sink, std::max({ have + more, sink.capacity + sink.capacity / 4, guess }));
  if (sink.capacity < have + more || (sink.slots == 2 && !sink.kmers_hi)) throw std::runtime_error("record sink did not grow");
}

} // namespace

kmer_file_info stream_kmer_file(const std::string& path, size_t expected_k, record_sink& sink)
{
  payload_stream ps(path);
  char head[41];
  if (ps.read_some(head, 41) != 41 || std::memcmp(head, "kmtricks", 8) != 0 || std::memcmp(head + 13, "kmer", 4) != 0)
    throw std::runtime_error(path + ": not a kmtricks k-mer file");
  kmer_file_info f;
  const uint8_t compressed = (uint8_t)head[12];
  uint32_t k;
  std::memcpy(&k, head + 21, 4); std::memcpy(&f.slots, head + 25, 4); std::memcpy(&f.count_bytes, head + 29, 4);
  if (expected_k && k != expected_k) throw std::runtime_error(path + ": k-mer size differs from the run's");
  if (f.slots != 1 && f.slots != 2) throw std::runtime_error(path + ": k > 64 is not supported");
  if (f.count_bytes != 1 && f.count_bytes != 2 && f.count_bytes != 4) throw std::runtime_error(path + ": bad count width");
  if (sink.slots != f.slots || sink.nb_counts != 1) sink.capacity = 0;      // arrays sized for records of another shape
  sink.slots = f.slots; sink.nb_counts = 1;
  const size_t rec = 8 * (size_t)f.slots + f.count_bytes;
  sink.file_size = ps.file_size;
  // (a packed-transfer sink takes one-limb records only and has no arrays to grow: a two-limb file in a k <= 32 run is
  // refused here, by name -- it used to fall through to an empty sink.reserve and die of std::bad_function_call)
  if (sink.raw && f.slots != 1) throw std::runtime_error(path + ": k-mer width of a sample file differs from the run's");
  if (sink.raw)
  {
    ps.records(compressed != 0, rec, sink.in, sink.out, [&](const char* p, size_t whole) { sink.raw(p, whole, f.count_bytes); f.records += whole; });
    sink.raw(nullptr, 0, f.count_bytes);
    return f;
  }
  size_t held = 0;                                        // records in the sink's arrays (all of the file's so far, unless a consumer takes them)
  // (bytes after the last whole record are dropped, as read_kmer_file does)
  ps.records(compressed != 0, rec, sink.in, sink.out, [&](const char* p, size_t whole)
  {
    grow_sink(sink, held, whole, ps.file_size, rec);
    uint64_t* km = sink.kmers + held;
    uint64_t* kh = sink.kmers_hi ? sink.kmers_hi + held : nullptr;
    uint32_t* ct = sink.counts + held;
    if (f.slots == 1 && f.count_bytes == 4) split_fixed<1, 4>(p, whole, km, kh, ct);
    else if (f.slots == 1 && f.count_bytes == 2) split_fixed<1, 2>(p, whole, km, kh, ct);
    else if (f.slots == 1) split_fixed<1, 1>(p, whole, km, kh, ct);
    else if (f.count_bytes == 4) split_fixed<2, 4>(p, whole, km, kh, ct);
    else if (f.count_bytes == 2) split_fixed<2, 2>(p, whole, km, kh, ct);
    else split_fixed<2, 1>(p, whole, km, kh, ct);
    f.records += whole;
    held += whole;
    if (sink.consume)
    {
      const size_t taken = sink.consume(sink, held, false);
      if (taken > held) throw std::runtime_error("record sink took more than it was given");
      if (taken && taken < held)
      {
        std::memmove(sink.kmers, sink.kmers + taken, (held - taken) * 8);
        if (sink.kmers_hi) std::memmove(sink.kmers_hi, sink.kmers_hi + taken, (held - taken) * 8);
        std::memmove(sink.counts, sink.counts + taken, (held - taken) * 4);
      }
      held -= taken;
    }
  });
  if (sink.consume && sink.consume(sink, held, true) != held) throw std::runtime_error("record sink left records behind");
  return f;
}

matrix_file_info stream_matrix_file(const std::string& path, record_sink& sink)
{
  payload_stream ps(path);
  char head[45];
  if (ps.read_some(head, 45) != 45 || std::memcmp(head, "kmtricks", 8) != 0 || std::memcmp(head + 13, "matrix", 6) != 0)
    throw std::runtime_error(path + ": not a kmtricks count matrix");
  matrix_file_info f;
  const uint8_t compressed = (uint8_t)head[12];
  std::memcpy(&f.kmer_size, head + 21, 4); std::memcpy(&f.slots, head + 25, 4); std::memcpy(&f.count_bytes, head + 29, 4);
  std::memcpy(&f.nb_counts, head + 33, 4); std::memcpy(&f.partition, head + 41, 4);
  if (f.slots != 1 && f.slots != 2) throw std::runtime_error(path + ": k > 64 is not supported");
  if (f.count_bytes != 1 && f.count_bytes != 2 && f.count_bytes != 4) throw std::runtime_error(path + ": bad count width");
  // (a damaged header must not size the decoder's buffers: 2^32 - 1 samples would be an 8 GB row -- found by tests/io_fuzz.cpp)
  if (f.nb_counts > kMaxMatrixSamples) throw std::runtime_error(path + ": implausible number of samples in the header");
  if (sink.slots != f.slots || sink.nb_counts != f.nb_counts) sink.capacity = 0;   // arrays sized for rows of another shape
  sink.slots = f.slots; sink.nb_counts = f.nb_counts;
  const size_t kb = 8 * (size_t)f.slots, rec = kb + (size_t)f.count_bytes * f.nb_counts;
  const uint32_t S = f.nb_counts, cb = f.count_bytes;
  const size_t left = ps.records(compressed != 0, rec, sink.in, sink.out, [&](const char* p, size_t whole)
  {
    grow_sink(sink, f.rows, whole, ps.file_size, rec);
    uint64_t* km = sink.kmers + f.rows;
    uint64_t* kh = sink.kmers_hi ? sink.kmers_hi + f.rows : nullptr;
    uint32_t* ct = sink.counts + f.rows * (size_t)S;
    for (size_t i = 0; i < whole; ++i, p += rec, ct += S)
    {
      std::memcpy(&km[i], p, 8);
      if (f.slots == 2) std::memcpy(&kh[i], p + 8, 8);
      if (cb == 4) std::memcpy(ct, p + kb, (size_t)S * 4);                                // rows of 4-byte counts as they are
      else if (cb == 2) for (uint32_t s2 = 0; s2 < S; ++s2) { uint16_t v; std::memcpy(&v, p + kb + 2 * (size_t)s2, 2); ct[s2] = v; }
      else for (uint32_t s2 = 0; s2 < S; ++s2) ct[s2] = (uint8_t)p[kb + s2];
    }
    f.rows += whole;
  });
  if (left) throw std::runtime_error(path + ": truncated row");
  return f;
}

std::string kmer_to_string(uint64_t hi, uint64_t lo, size_t k)
{
  static const char code[4] = { 'A', 'C', 'T', 'G' };
  std::string s(k, 'A');
  for (size_t i = 0; i < k; ++i)
  {
    const size_t bit = 2 * (k - 1 - i);                  // position of the base in the 128-bit value
    s[i] = code[(bit >= 64 ? (hi >> (bit - 64)) : (lo >> bit)) & 3];
  }
  return s;
}

std::string kmer_to_string(uint64_t kmer, size_t k)
{
  static const char code[4] = { 'A', 'C', 'T', 'G' };
  std::string s(k, 'A');
  for (size_t i = 0; i < k; ++i) s[i] = code[(kmer >> (2 * (k - 1 - i))) & 3];
  return s;
}

} // namespace kmd_host


namespace kmd_host {

// ---- KFF 1.0 (see kmtricks_io.hpp) -------------------------------------------------------------
namespace {
void put_be(std::FILE* f, uint64_t v, int bytes)
{
  unsigned char b[8];
  for (int i = 0; i < bytes; ++i) b[i] = (unsigned char)(v >> (8 * (bytes - 1 - i)));
  if (std::fwrite(b, 1, (size_t)bytes, f) != (size_t)bytes) throw std::runtime_error("KFF: write failed");
}
void put_var(std::FILE* f, const char* name, uint64_t value)
{
  if (std::fwrite(name, 1, std::strlen(name) + 1, f) != std::strlen(name) + 1) throw std::runtime_error("KFF: write failed");
  put_be(f, value, 8);
}
}

kff_writer::kff_writer(const std::string& path, size_t kmer_size) : m_k(kmer_size)
{
  m_f = std::fopen(path.c_str(), "wb");
  if (!m_f) throw std::runtime_error("cannot write " + path);
  m_open = true;
  std::fputs("KFF", m_f);
  put_be(m_f, 1, 1); put_be(m_f, 0, 1);                   // version 1.0
  put_be(m_f, (0u << 6) | (1u << 4) | (3u << 2) | 2u, 1); // write_encoding({0, 1, 3, 2}): A C G T
  put_be(m_f, 0, 1); put_be(m_f, 0, 1);                   // k-mers not declared unique / canonical
  put_be(m_f, 0, 4);                                      // no free block
  std::fputc('v', m_f);                                   // Section_GV: k, max, data_size (kff_utils.hpp:43-47)
  put_be(m_f, 3, 8);
  put_var(m_f, "k", m_k); put_var(m_f, "max", 1); put_var(m_f, "data_size", 0);
  std::fputc('r', m_f);                                   // Section_Raw
  m_count_at = std::ftell(m_f);
  put_be(m_f, 0, 8);                                      // number of blocks: patched by close()
}

void kff_writer::write(uint64_t kmer_lo, uint64_t kmer_hi)
{
  const size_t nbytes = (m_k + 3) / 4;                    // write_compacted_sequence(encoded, k, nullptr)
  unsigned char b[16];
  for (size_t i = 0; i < nbytes; ++i)
  {
    const size_t shift = 8 * (nbytes - 1 - i);            // byte i holds bits [shift, shift + 8) of the 2k-bit value
    b[i] = (unsigned char)(shift >= 64 ? (kmer_hi >> (shift - 64)) : shift == 0 ? kmer_lo : ((kmer_lo >> shift) | (shift > 56 ? (kmer_hi << (64 - shift)) : 0)));
  }
  if (std::fwrite(b, 1, nbytes, m_f) != nbytes) throw std::runtime_error("KFF: write failed");
  ++m_blocks;
}

void kff_writer::close()
{
  if (!m_open) return;
  m_open = false;
  const long end = std::ftell(m_f);
  std::fseek(m_f, m_count_at, SEEK_SET);
  put_be(m_f, m_blocks, 8);
  std::fseek(m_f, end, SEEK_SET);
  std::fputc('v', m_f);                                   // footer
  put_be(m_f, 2, 8);
  put_var(m_f, "first_index", 0);
  put_var(m_f, "footer_size", 9 + 2 * (12 + 8));
  std::fputs("KFF", m_f);
  std::fclose(m_f);
  m_f = nullptr;
}

} // namespace kmd_host
