// kmd_pca.hip -- the population-structure PCA front end of the pop-strat stage, on the device.
//
// Replaces, for `kmdiff diff --pop-correction`:
//   Sampler / EigGenoFile / EigSnpFile   include/kmdiff/popstrat.hpp:55-146  (rows sampled with
//       probability kmer_pca during stage 1, written as presence/absence lines)
//   run_eigenstrat_smartpca              src/popstrat.cpp:97-134  (external `smartpca -p parfile`
//       with usenorm YES, numoutlieriter 0, numoutevec 10; then evec2pca.perl)
//   smartpca itself, as modified by Hawk  thirdparty/hawk/EIG6.0.1-Hawk/src/eigensrc/smartpca.c:
//       fvadjust :1694-1800   g = count > 0; mean mu; p = 1 - sqrt(1 - mu) (diploid) or mu (-V);
//                             x = (g - mu) / sqrt(p (1 - p))
//       getcolxz :2598-2700   a row is ignored only when no sample holds the k-mer
//       main     :880-1025    XTX += x x^T over the rows; XTX /= trace / (n - 1); eigvecs
//       main     :1140-1320   printed coordinates = eigenvectors scaled to unit norm
//
// What differs by nature: the reference samples with a sequential std::default_random_engine
// under a lock shared by the partition threads (order-dependent, not reproducible); here a row
// is sampled iff a hash of (seed, k-mer) falls below the rate -- a pure function of the row.
// Eigenvector signs are the eigen-solver's in smartpca; here the largest component is positive.
//
// Stages: k_pca_count / k_pca_emit  (two passes over the k-mer column, so that the sampled rows
//         land in row order: the Gram sums are then summed in a fixed order)
//         k_pca_gram + k_pca_reduce (32 x 32 tiles of XTX over slices of the sampled rows)
//         k_jacobi                  (parallel-ordered cyclic Jacobi, one workgroup, FP64)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string.h>

#include "kmd_internal.h"

#include <algorithm>
#include <vector>
#include <rocprim/rocprim.hpp>

struct kmd_pca
{
  int S = 0, W = 0;                 // samples, 32-bit words per presence pattern
  int diploid = 1;
  uint64_t seed = 0, thresh = 0;    // sampled iff hash < thresh (all_rows: every row)
  bool all_rows = false;
  size_t cap = 0, n = 0;            // sampled rows: capacity, recorded so far
  uint32_t* d_bits = nullptr;       // [cap][W]
  double* d_f = nullptr;            // [cap]  1 / sqrt(p (1 - p))
  double* d_mu = nullptr;           // [cap]  fraction of samples holding the k-mer
  int device = 0;
};

namespace {

__host__ __device__ inline uint64_t splitmix64(uint64_t x)
{
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__device__ __forceinline__ bool pca_sampled(uint64_t seed, uint64_t thresh, bool all_rows, uint64_t lo, uint64_t hi)
{
  if (all_rows) return true;
  return splitmix64(seed ^ splitmix64(lo) ^ (hi * 0x9E3779B97F4A7C15ull)) < thresh;
}

// sampled rows per wave tile (64 consecutive rows)
__global__ void __launch_bounds__(256) k_pca_count(const uint64_t* __restrict__ kmer_lo, const uint64_t* __restrict__ kmer_hi,
                                                   size_t n_rows, uint64_t seed, uint64_t thresh, bool all_rows,
                                                   uint32_t* __restrict__ wave_count)
{
  const size_t n_tiles = (n_rows + 63) / 64;
  const size_t n_waves = (size_t)gridDim.x * 4;
  const uint32_t lane = threadIdx.x & 63;
  for (size_t t = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < n_tiles; t += n_waves)
  {
    const size_t row = t * 64 + lane;
    const bool s = row < n_rows && pca_sampled(seed, thresh, all_rows, kmer_lo[row], kmer_hi ? kmer_hi[row] : 0ull);
    const unsigned long long m = __ballot(s);
    if (lane == 0) wave_count[t] = (uint32_t)__popcll(m);
  }
}

// presence pattern, mu and f of every sampled row, at base + (exclusive prefix of wave_count) + rank
template <typename CT>
__global__ void __launch_bounds__(256) k_pca_emit(const CT* __restrict__ counts, int layout, size_t ld, int S, int W,
                                                  const uint64_t* __restrict__ kmer_lo, const uint64_t* __restrict__ kmer_hi,
                                                  size_t n_rows, uint64_t seed, uint64_t thresh, bool all_rows, int diploid,
                                                  const uint32_t* __restrict__ wave_offset, size_t base, size_t cap,
                                                  uint32_t* __restrict__ bits, double* __restrict__ out_f,
                                                  double* __restrict__ out_mu)
{
  const size_t n_tiles = (n_rows + 63) / 64;
  const size_t n_waves = (size_t)gridDim.x * 4;
  const uint32_t lane = threadIdx.x & 63;
  for (size_t t = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < n_tiles; t += n_waves)
  {
    const size_t row = t * 64 + lane;
    const bool s = row < n_rows && pca_sampled(seed, thresh, all_rows, kmer_lo[row], kmer_hi ? kmer_hi[row] : 0ull);
    unsigned long long m = __ballot(s);
    size_t slot = base + wave_offset[t];
    while (m)                                            // wave-uniform: the sampled rows of this tile, in row order
    {
      const int src = __ffsll((long long)m) - 1;
      m &= m - 1;
      const size_t r = t * 64 + (size_t)src;
      if (slot < cap)
      {
        uint32_t present = 0;                            // samples holding the k-mer
        for (int w64 = 0; w64 * 64 < S; ++w64)
        {
          const int smp = w64 * 64 + (int)lane;
          const bool g = smp < S && counts[kmd::count_index(layout, ld, S, r, smp)] > 0;       // fvadjust: cc = cc > 0
          const unsigned long long pm = __ballot(g);
          present += (uint32_t)__popcll(pm);
          if (lane == 0)
          {
            bits[slot * W + 2 * w64] = (uint32_t)pm;
            if (2 * w64 + 1 < W) bits[slot * W + 2 * w64 + 1] = (uint32_t)(pm >> 32);
          }
        }
        if (lane == 0)
        {
          const double mu = (double)present / (double)S;                  // ymean
          const double p = diploid ? 1.0 - sqrt(1.0 - mu) : mu;           // smartpca.c:1784-1791
          const double y = p * (1.0 - p);
          out_mu[slot] = mu;
          out_f[slot] = y > 0.0 ? 1.0 / sqrt(y) : 1.0;                    // yfancy (x is all zero when y == 0)
        }
      }
      ++slot;
    }
  }
}

// partial[slice][i][j] = sum over the slice's rows of x_i x_j, x = (g - mu) f; 32 x 32 tile per block
__global__ void __launch_bounds__(1024) k_pca_gram(const uint32_t* __restrict__ bits, const double* __restrict__ f,
                                                   const double* __restrict__ mu, size_t n, int S, int W,
                                                   size_t rows_per_slice, double* __restrict__ partial)
{
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int bi = blockIdx.y, bj = blockIdx.x;
  const int i = bi * 32 + ty, j = bj * 32 + tx;
  const size_t r0 = (size_t)blockIdx.z * rows_per_slice;
  const size_t r1 = r0 + rows_per_slice < n ? r0 + rows_per_slice : n;
  double acc = 0.0;
  for (size_t r = r0; r < r1; ++r)
  {
    const uint32_t wi = bits[r * W + bi], wj = bits[r * W + bj];          // block-uniform
    const double fr = f[r], mr = mu[r];
    const double xi = ((double)((wi >> ty) & 1u) - mr) * fr;
    const double xj = ((double)((wj >> tx) & 1u) - mr) * fr;
    acc += xi * xj;
  }
  if (i < S && j < S) partial[((size_t)blockIdx.z * S + i) * S + j] = acc;
}

__global__ void __launch_bounds__(256) k_pca_reduce(const double* __restrict__ partial, size_t n_slices, size_t cells,
                                                    double* __restrict__ xtx)
{
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cells) return;
  double s = 0.0;
  for (size_t k = 0; k < n_slices; ++k) s += partial[k * cells + c];      // fixed order
  xtx[c] = s;
}

// Cyclic Jacobi with the round-robin (tournament) ordering: every round rotates m / 2 disjoint
// (p, q) planes at once.  One workgroup; A (m x m, symmetric, destroyed) and V in global memory.
// m is even (the caller pads an odd problem with a zero row / column, never rotated).
__global__ void __launch_bounds__(1024) k_jacobi(double* __restrict__ A, double* __restrict__ V, int m, int n_real,
                                                 int max_sweeps, int* __restrict__ sweeps_done)
{
  extern __shared__ double s_mem[];
  const int half = m / 2;
  double* cs = s_mem;                                   // [half][2]
  int* top = reinterpret_cast<int*>(cs + 2 * half);     // [half]
  int* bot = top + half;                                // [half]
  int* nxt = bot + half;                                // [m] scratch for the permutation
  __shared__ double s_red[32];
  __shared__ int s_stop;
  const int tid = threadIdx.x, nt = blockDim.x;

  for (int k = tid; k < half; k += nt) { top[k] = 2 * k; bot[k] = 2 * k + 1; }
  for (int e = tid; e < m * m; e += nt) V[e] = (e / m == e % m) ? 1.0 : 0.0;
  __syncthreads();

  int sweep = 0;
  for (; sweep < max_sweeps; ++sweep)
  {
    // convergence: off-diagonal mass against the diagonal
    double off = 0.0, dia = 0.0;
    for (int e = tid; e < m * m; e += nt)
    {
      const double a = A[e];
      if (e / m == e % m) dia += a * a; else off += a * a;
    }
    for (int o = 32; o > 0; o >>= 1) { off += __shfl_down(off, o, 64); dia += __shfl_down(dia, o, 64); }
    if ((tid & 63) == 0) { s_red[(tid >> 6) * 2] = off; s_red[(tid >> 6) * 2 + 1] = dia; }
    __syncthreads();
    if (tid == 0)
    {
      double o2 = 0.0, d2 = 0.0;
      for (int w = 0; w < (nt + 63) / 64; ++w) { o2 += s_red[2 * w]; d2 += s_red[2 * w + 1]; }
      s_stop = (o2 <= 1e-28 * d2 || o2 == 0.0) ? 1 : 0;       // off-diagonal mass at the rounding floor
    }
    __syncthreads();
    if (s_stop) break;

    for (int round = 0; round < m - 1; ++round)
    {
      // rotation angles of this round's planes
      for (int k = tid; k < half; k += nt)
      {
        int p = top[k], q = bot[k];
        if (p > q) { const int t = p; p = q; q = t; }
        double c = 1.0, s = 0.0;
        if (q < n_real)
        {
          const double apq = A[p * m + q];
          if (apq != 0.0)
          {
            const double theta = (A[q * m + q] - A[p * m + p]) / (2.0 * apq);
            const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            c = 1.0 / sqrt(t * t + 1.0);
            s = t * c;
          }
        }
        cs[2 * k] = c; cs[2 * k + 1] = s;
      }
      __syncthreads();
      // columns: A <- A J, V <- V J
      for (int e = tid; e < half * m; e += nt)
      {
        const int r = e / half, k = e - r * half;
        int p = top[k], q = bot[k];
        if (p > q) { const int t = p; p = q; q = t; }
        const double c = cs[2 * k], s = cs[2 * k + 1];
        const double arp = A[r * m + p], arq = A[r * m + q];
        A[r * m + p] = c * arp - s * arq;
        A[r * m + q] = s * arp + c * arq;
        const double vrp = V[r * m + p], vrq = V[r * m + q];
        V[r * m + p] = c * vrp - s * vrq;
        V[r * m + q] = s * vrp + c * vrq;
      }
      __syncthreads();
      // rows: A <- J^T A
      for (int e = tid; e < half * m; e += nt)
      {
        const int k = e / m, col = e - k * m;
        int p = top[k], q = bot[k];
        if (p > q) { const int t = p; p = q; q = t; }
        const double c = cs[2 * k], s = cs[2 * k + 1];
        const double apc = A[p * m + col], aqc = A[q * m + col];
        A[p * m + col] = c * apc - s * aqc;
        A[q * m + col] = s * apc + c * aqc;
      }
      __syncthreads();
      // next round of the tournament: top[0] stays, everybody else moves one seat
      for (int k = tid; k < half; k += nt)
      {
        nxt[k] = k == 0 ? top[0] : (k == 1 ? bot[0] : top[k - 1]);
        nxt[half + k] = k == half - 1 ? top[half - 1] : bot[k + 1];
      }
      __syncthreads();
      for (int k = tid; k < half; k += nt) { top[k] = nxt[k]; bot[k] = nxt[half + k]; }
      __syncthreads();
    }
  }
  if (tid == 0) *sweeps_done = sweep;
}

// room for `extra` more sampled rows (the store grows by doubling)
int pca_reserve(kmd_pca* P, size_t total, hipStream_t st)
{
  if (P->n + total > P->cap)                             // grow: at least double
  {
    const size_t want = std::max(P->cap * 2, P->n + (size_t)total);
    uint32_t* nb = nullptr; double *nf = nullptr, *nm = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&nb), want * (size_t)P->W * 4);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&nf), want * 8);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&nm), want * 8);
    if (e == hipSuccess && P->n)
    {
      e = hipMemcpyAsync(nb, P->d_bits, P->n * (size_t)P->W * 4, hipMemcpyDeviceToDevice, st);
      if (e == hipSuccess) e = hipMemcpyAsync(nf, P->d_f, P->n * 8, hipMemcpyDeviceToDevice, st);
      if (e == hipSuccess) e = hipMemcpyAsync(nm, P->d_mu, P->n * 8, hipMemcpyDeviceToDevice, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    if (e != hipSuccess)
    {
      if (nb) (void)hipFree(nb);
      if (nf) (void)hipFree(nf);
      if (nm) (void)hipFree(nm);
      kmd::set_error(std::string("kmd_pca_sample: cannot grow the sample store: ") + hipGetErrorString(e));
      return e == hipErrorOutOfMemory ? KMD_E_NOMEM : KMD_E_HIP;
    }
    (void)hipFree(P->d_bits); (void)hipFree(P->d_f); (void)hipFree(P->d_mu);
    P->d_bits = nb; P->d_f = nf; P->d_mu = nm; P->cap = want;
  }
  return KMD_OK;
}

// ---- sampling without a matrix (the fused merge, kmd_tilemerge.hip): the sampled k-mers are found in
// the per-sample streams themselves.  A record is kept when its k-mer is sampled (the same hash of
// (seed, k-mer) as k_pca_count); the kept k-mers are sorted and de-duplicated (a k-mer has a record in
// every sample that holds it) -- ascending k-mer order IS the row order of the merged matrix, so the
// sampled rows, their order and every sum over them are those of the matrix path; the presence pattern
// of a sampled k-mer is one binary search per sample.
__global__ void __launch_bounds__(256) k_pca_scan_streams(const uint64_t* __restrict__ lo, const uint64_t* __restrict__ hi, size_t n,
                                                          uint64_t seed, uint64_t thresh, bool all_rows,
                                                          uint64_t* __restrict__ out_lo, uint64_t* __restrict__ out_hi, size_t cap,
                                                          unsigned long long* __restrict__ counter)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t n_round = (n + stride - 1) / stride * stride;
  const uint32_t lane = threadIdx.x & 63;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride)
  {
    const bool in = i < n;
    const uint64_t k = in ? lo[i] : 0ull, kh = (in && hi) ? hi[i] : 0ull;
    const bool s = in && pca_sampled(seed, thresh, all_rows, k, kh);
    const unsigned long long m = __ballot(s);
    if (!m) continue;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(counter, (unsigned long long)__popcll(m));
    base = __shfl(base, 0, 64);
    const unsigned long long at = base + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
    if (s && out_lo && at < cap) { out_lo[at] = k; if (out_hi) out_hi[at] = kh; }
  }
}

__global__ void __launch_bounds__(256) k_pca_heads(const uint64_t* __restrict__ lo, const uint64_t* __restrict__ hi, size_t n,
                                                   uint32_t* __restrict__ flag)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = (i == 0 || lo[i] != lo[i - 1] || (hi && hi[i] != hi[i - 1])) ? 1u : 0u;
}

// one wave per distinct sampled k-mer: its presence pattern (binary search in every stream), mu and f
__global__ void __launch_bounds__(256) k_pca_presence(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ keys_hi,
                                                      const uint64_t* __restrict__ offs, int S, int W,
                                                      const uint64_t* __restrict__ s_lo, const uint64_t* __restrict__ s_hi,
                                                      const uint32_t* __restrict__ flag, const uint32_t* __restrict__ pos,
                                                      size_t n_sorted, int diploid, size_t base, size_t cap,
                                                      uint32_t* __restrict__ bits, double* __restrict__ out_f, double* __restrict__ out_mu)
{
  const size_t n_waves = (size_t)gridDim.x * 4;
  const uint32_t lane = threadIdx.x & 63;
  for (size_t j = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); j < n_sorted; j += n_waves)
  {
    if (!flag[j]) continue;                               // (wave-uniform) a repeat of the k-mer before it
    const size_t slot = base + pos[j];
    if (slot >= cap) continue;
    const uint64_t k = s_lo[j], kh = s_hi ? s_hi[j] : 0ull;
    uint32_t present = 0;
    for (int w64 = 0; w64 * 64 < S; ++w64)
    {
      const int smp = w64 * 64 + (int)lane;
      bool g = false;
      if (smp < S)
      {
        size_t lo = (size_t)offs[smp], hi = (size_t)offs[smp + 1];
        const size_t end = hi;
        while (lo < hi)
        {
          const size_t mid = lo + ((hi - lo) >> 1);
          const bool less = keys_hi ? (keys_hi[mid] < kh || (keys_hi[mid] == kh && keys[mid] < k)) : keys[mid] < k;
          if (less) lo = mid + 1; else hi = mid;
        }
        g = lo < end && keys[lo] == k && (!keys_hi || keys_hi[lo] == kh);     // a record = a count > 0 (fvadjust: cc = cc > 0)
      }
      const unsigned long long pm = __ballot(g);
      present += (uint32_t)__popcll(pm);
      if (lane == 0)
      {
        bits[slot * W + 2 * w64] = (uint32_t)pm;
        if (2 * w64 + 1 < W) bits[slot * W + 2 * w64 + 1] = (uint32_t)(pm >> 32);
      }
    }
    if (lane == 0)
    {
      const double mu = (double)present / (double)S;                  // ymean
      const double p = diploid ? 1.0 - sqrt(1.0 - mu) : mu;           // smartpca.c:1784-1791
      const double y = p * (1.0 - p);
      out_mu[slot] = mu;
      out_f[slot] = y > 0.0 ? 1.0 / sqrt(y) : 1.0;                    // yfancy (x is all zero when y == 0)
    }
  }
}

template <typename CT>
int pca_sample_tile(kmd_pca* P, const kmd_tile* tile, hipStream_t st)
{
  const size_t n_rows = tile->n_rows;
  const size_t n_tiles = (n_rows + 63) / 64;
  void *p_cnt = nullptr, *p_off = nullptr, *p_tmp = nullptr;
  struct guard { void** p; ~guard() { if (*p) kmd::scratch_free(*p); } } g1{ &p_cnt }, g2{ &p_off }, g3{ &p_tmp };
  KMD_HIP(kmd::scratch_alloc(&p_cnt, (n_tiles + 1) * 4));
  KMD_HIP(kmd::scratch_alloc(&p_off, (n_tiles + 1) * 4));
  uint32_t* cnt = static_cast<uint32_t*>(p_cnt);
  uint32_t* off = static_cast<uint32_t*>(p_off);
  KMD_HIP(hipMemsetAsync(cnt + n_tiles, 0, 4, st));
  size_t grid = (n_tiles + 3) / 4;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(k_pca_count, dim3((unsigned)grid), dim3(256), 0, st, tile->d_kmer_lo, tile->d_kmer_hi, n_rows, P->seed,
                     P->thresh, P->all_rows, cnt);
  KMD_HIP(hipGetLastError());
  size_t tmp = 0;
  KMD_HIP(rocprim::exclusive_scan(nullptr, tmp, cnt, off, 0u, n_tiles + 1, rocprim::plus<uint32_t>(), st));
  KMD_HIP(kmd::scratch_alloc(&p_tmp, tmp ? tmp : 1));
  KMD_HIP(rocprim::exclusive_scan(p_tmp, tmp, cnt, off, 0u, n_tiles + 1, rocprim::plus<uint32_t>(), st));
  uint32_t total = 0;
  KMD_HIP(hipMemcpyAsync(&total, off + n_tiles, 4, hipMemcpyDeviceToHost, st));
  KMD_HIP(hipStreamSynchronize(st));
  { const int rc_g = pca_reserve(P, total, st); if (rc_g != KMD_OK) return rc_g; }
  if (total)
  {
    hipLaunchKernelGGL((k_pca_emit<CT>), dim3((unsigned)grid), dim3(256), 0, st, static_cast<const CT*>(tile->d_counts),
                       tile->layout, tile->ld, P->S, P->W, tile->d_kmer_lo, tile->d_kmer_hi, n_rows, P->seed, P->thresh,
                       P->all_rows, P->diploid, off, P->n, P->cap, P->d_bits, P->d_f, P->d_mu);
    KMD_HIP(hipGetLastError());
    KMD_HIP(hipStreamSynchronize(st));
    P->n += total;
  }
  return KMD_OK;
}

} // namespace

extern "C" {

int kmd_pca_create(kmd_pca** out, int n_samples, double sample_rate, uint64_t seed, int diploid, size_t capacity_rows)
{
  KMD_REQUIRE(out, "kmd_pca_create: NULL out");
  KMD_REQUIRE(n_samples >= 2 && n_samples <= 1024, "kmd_pca_create: samples must be in [2, 1024]");
  KMD_REQUIRE(sample_rate > 0.0 && capacity_rows > 0, "kmd_pca_create: rate and capacity must be positive");
  kmd_pca* P = new kmd_pca();
  P->S = n_samples;
  P->W = ((n_samples + 63) / 64) * 2;
  P->diploid = diploid ? 1 : 0;
  P->seed = seed;
  P->all_rows = sample_rate >= 1.0;
  P->thresh = P->all_rows ? ~0ull : (uint64_t)ldexp(sample_rate, 64);
  P->cap = capacity_rows;
  hipError_t e = hipGetDevice(&P->device);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&P->d_bits), capacity_rows * (size_t)P->W * 4);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&P->d_f), capacity_rows * 8);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&P->d_mu), capacity_rows * 8);
  if (e != hipSuccess)
  {
    if (P->d_bits) (void)hipFree(P->d_bits);
    if (P->d_f) (void)hipFree(P->d_f);
    if (P->d_mu) (void)hipFree(P->d_mu);
    delete P;
    kmd::set_error(std::string("kmd_pca_create: ") + hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? KMD_E_NOMEM : KMD_E_HIP;
  }
  *out = P;
  return KMD_OK;
}

void kmd_pca_destroy(kmd_pca* P)
{
  if (!P) return;
  (void)hipFree(P->d_bits); (void)hipFree(P->d_f); (void)hipFree(P->d_mu);
  delete P;
}

int kmd_pca_sample(kmd_pca* P, const kmd_tile* tile, void* stream)
{
  KMD_REQUIRE(P && tile, "kmd_pca_sample: NULL");
  KMD_REQUIRE(tile->d_kmer_lo || tile->n_rows == 0, "kmd_pca_sample: the tile needs its k-mer column (rows are sampled by k-mer)");
  KMD_REQUIRE(kmd::layout_ok(tile->layout), "kmd_pca_sample: layout");
  if (tile->n_rows == 0) return KMD_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (tile->count_bytes)
  {
    case 1: return pca_sample_tile<uint8_t>(P, tile, st);
    case 2: return pca_sample_tile<uint16_t>(P, tile, st);
    case 4: return pca_sample_tile<uint32_t>(P, tile, st);
    default: kmd::set_error("kmd_pca_sample: count_bytes must be 1, 2 or 4"); return KMD_E_INVALID;
  }
}

// Sampler::sample for the fused merge (no matrix): the same rows, found in the streams (see k_pca_scan_streams).
int kmd_pca_sample_streams(kmd_pca* P, int n_samples, const uint64_t* d_kmers, const uint64_t* d_kmers_hi,
                           const uint32_t* d_counts, const uint64_t* offsets, void* stream)
{
  (void)d_counts;                                          // a record means count > 0: the keys are all that is read
  KMD_REQUIRE(P && offsets && n_samples == P->S, "kmd_pca_sample_streams: arguments");
  const size_t n = (size_t)offsets[n_samples];
  if (n == 0) return KMD_OK;
  KMD_REQUIRE(d_kmers, "kmd_pca_sample_streams: NULL k-mers");
  hipStream_t st = static_cast<hipStream_t>(stream);
  std::vector<void*> held;
  struct guard { std::vector<void*>& h; ~guard() { for (void* q : h) kmd::scratch_free(q); } } g{ held };
  auto take = [&](void** p, size_t bytes) -> hipError_t { const hipError_t e = kmd::scratch_alloc(p, bytes ? bytes : 1); if (e == hipSuccess) held.push_back(*p); return e; };
  void *p_cnt = nullptr, *p_offs = nullptr;
  KMD_HIP(take(&p_cnt, 8));
  KMD_HIP(take(&p_offs, ((size_t)n_samples + 1) * 8));
  KMD_HIP(hipMemsetAsync(p_cnt, 0, 8, st));
  KMD_HIP(hipMemcpyAsync(p_offs, offsets, ((size_t)n_samples + 1) * 8, hipMemcpyHostToDevice, st));
  unsigned long long* d_cnt = static_cast<unsigned long long*>(p_cnt);
  const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
  // pass 1: how many records are sampled; pass 2: list them
  hipLaunchKernelGGL(k_pca_scan_streams, dim3(grid), dim3(256), 0, st, d_kmers, d_kmers_hi, n, P->seed, P->thresh, P->all_rows,
                     (uint64_t*)nullptr, (uint64_t*)nullptr, (size_t)0, d_cnt);
  unsigned long long n_rec = 0;
  KMD_HIP(hipMemcpyAsync(&n_rec, d_cnt, 8, hipMemcpyDeviceToHost, st));
  KMD_HIP(hipStreamSynchronize(st));
  if (n_rec == 0) return KMD_OK;
  KMD_REQUIRE(n_rec < 0xFFFFFFFFull, "kmd_pca_sample_streams: too many sampled records");
  const bool two = d_kmers_hi != nullptr;
  void *p_lo = nullptr, *p_hi = nullptr, *p_lo2 = nullptr, *p_hi2 = nullptr, *p_flag = nullptr, *p_pos = nullptr, *p_tmp = nullptr;
  KMD_HIP(take(&p_lo, n_rec * 8)); KMD_HIP(take(&p_lo2, n_rec * 8));
  if (two) { KMD_HIP(take(&p_hi, n_rec * 8)); KMD_HIP(take(&p_hi2, n_rec * 8)); }
  KMD_HIP(take(&p_flag, (n_rec + 1) * 4)); KMD_HIP(take(&p_pos, (n_rec + 1) * 4));
  KMD_HIP(hipMemsetAsync(p_cnt, 0, 8, st));
  hipLaunchKernelGGL(k_pca_scan_streams, dim3(grid), dim3(256), 0, st, d_kmers, d_kmers_hi, n, P->seed, P->thresh, P->all_rows,
                     static_cast<uint64_t*>(p_lo), static_cast<uint64_t*>(p_hi), (size_t)n_rec, d_cnt);
  KMD_HIP(hipGetLastError());
  // ascending (hi, lo): LSD -- by the low limb carrying the high one, then stably by the high limb
  uint64_t *lo = static_cast<uint64_t*>(p_lo), *hi = static_cast<uint64_t*>(p_hi), *lo2 = static_cast<uint64_t*>(p_lo2), *hi2 = static_cast<uint64_t*>(p_hi2);
  size_t tmp = 0, tmp2 = 0;
  if (!two)
  {
    KMD_HIP(rocprim::radix_sort_keys(nullptr, tmp, lo, lo2, (size_t)n_rec, 0, 64, st));
    KMD_HIP(take(&p_tmp, tmp));
    KMD_HIP(rocprim::radix_sort_keys(p_tmp, tmp, lo, lo2, (size_t)n_rec, 0, 64, st));
    lo = lo2;
  }
  else
  {
    KMD_HIP(rocprim::radix_sort_pairs(nullptr, tmp, lo, lo2, hi, hi2, (size_t)n_rec, 0, 64, st));
    KMD_HIP(take(&p_tmp, tmp));
    KMD_HIP(rocprim::radix_sort_pairs(p_tmp, tmp, lo, lo2, hi, hi2, (size_t)n_rec, 0, 64, st));
    KMD_HIP(rocprim::radix_sort_pairs(nullptr, tmp2, hi2, hi, lo2, lo, (size_t)n_rec, 0, 64, st));
    void* p_tmp2 = nullptr;
    KMD_HIP(take(&p_tmp2, tmp2));
    KMD_HIP(rocprim::radix_sort_pairs(p_tmp2, tmp2, hi2, hi, lo2, lo, (size_t)n_rec, 0, 64, st));
  }
  uint32_t* flag = static_cast<uint32_t*>(p_flag);
  uint32_t* pos = static_cast<uint32_t*>(p_pos);
  hipLaunchKernelGGL(k_pca_heads, dim3((unsigned)((n_rec + 255) / 256)), dim3(256), 0, st, lo, two ? hi : nullptr, (size_t)n_rec, flag);
  KMD_HIP(hipMemsetAsync(flag + n_rec, 0, 4, st));
  size_t tmp3 = 0;
  void* p_tmp3 = nullptr;
  KMD_HIP(rocprim::exclusive_scan(nullptr, tmp3, flag, pos, 0u, (size_t)n_rec + 1, rocprim::plus<uint32_t>(), st));
  KMD_HIP(take(&p_tmp3, tmp3));
  KMD_HIP(rocprim::exclusive_scan(p_tmp3, tmp3, flag, pos, 0u, (size_t)n_rec + 1, rocprim::plus<uint32_t>(), st));
  uint32_t total = 0;
  KMD_HIP(hipMemcpyAsync(&total, pos + n_rec, 4, hipMemcpyDeviceToHost, st));
  KMD_HIP(hipStreamSynchronize(st));
  { const int rc_g = pca_reserve(P, total, st); if (rc_g != KMD_OK) return rc_g; }
  const unsigned grid_p = (unsigned)std::min<size_t>(((size_t)n_rec + 3) / 4, 8192);
  hipLaunchKernelGGL(k_pca_presence, dim3(grid_p), dim3(256), 0, st, d_kmers, d_kmers_hi, static_cast<const uint64_t*>(p_offs), P->S, P->W,
                     lo, two ? hi : nullptr, flag, pos, (size_t)n_rec, P->diploid, P->n, P->cap, P->d_bits, P->d_f, P->d_mu);
  KMD_HIP(hipGetLastError());
  KMD_HIP(hipStreamSynchronize(st));
  P->n += total;
  return KMD_OK;
}

int kmd_pca_count(const kmd_pca* P, uint64_t* n_sampled)
{
  KMD_REQUIRE(P && n_sampled, "kmd_pca_count: NULL");
  *n_sampled = P->n;
  return KMD_OK;
}

int kmd_pca_gram(kmd_pca* P, double* xtx_host, void* stream)
{
  KMD_REQUIRE(P && xtx_host, "kmd_pca_gram: NULL");
  const size_t S = (size_t)P->S, cells = S * S;
  if (P->n == 0) { std::memset(xtx_host, 0, cells * 8); return KMD_OK; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  size_t rows_per_slice = 2048, n_slices = (P->n + rows_per_slice - 1) / rows_per_slice;
  if (n_slices > 256) { n_slices = 256; rows_per_slice = (P->n + n_slices - 1) / n_slices; n_slices = (P->n + rows_per_slice - 1) / rows_per_slice; }
  void *p_partial = nullptr, *p_xtx = nullptr;
  struct guard { void** p; ~guard() { if (*p) kmd::scratch_free(*p); } } g1{ &p_partial }, g2{ &p_xtx };
  KMD_HIP(kmd::scratch_alloc(&p_partial, n_slices * cells * 8));
  KMD_HIP(kmd::scratch_alloc(&p_xtx, cells * 8));
  const unsigned T = (unsigned)((S + 31) / 32);
  hipLaunchKernelGGL(k_pca_gram, dim3(T, T, (unsigned)n_slices), dim3(1024), 0, st, P->d_bits, P->d_f, P->d_mu, P->n, P->S, P->W,
                     rows_per_slice, static_cast<double*>(p_partial));
  KMD_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_pca_reduce, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, static_cast<const double*>(p_partial),
                     n_slices, cells, static_cast<double*>(p_xtx));
  KMD_HIP(hipGetLastError());
  KMD_HIP(hipMemcpyAsync(xtx_host, p_xtx, cells * 8, hipMemcpyDeviceToHost, st));
  KMD_HIP(hipStreamSynchronize(st));
  return KMD_OK;
}

int kmd_pca_eigen(int n_samples, const double* xtx_host, int n_out, double* evec_host, double* eval_host)
{
  KMD_REQUIRE(n_samples >= 2 && n_samples <= 1024 && xtx_host && evec_host && eval_host, "kmd_pca_eigen: arguments");
  KMD_REQUIRE(n_out >= 1 && n_out <= n_samples, "kmd_pca_eigen: n_out must be in [1, samples]");
  const int S = n_samples, m = (S + 1) & ~1;
  // smartpca.c:1019-1023: XTX /= trace / (n - 1)
  double trace = 0.0;
  for (int i = 0; i < S; ++i) trace += xtx_host[(size_t)i * S + i];
  const double y = trace / (double)(S - 1);
  KMD_REQUIRE(y > 0.0 && std::isfinite(y), "kmd_pca_eigen: the Gram matrix has no positive trace (no sampled rows?)");
  std::vector<double> a((size_t)m * m, 0.0);
  for (int i = 0; i < S; ++i)
    for (int j = 0; j < S; ++j) a[(size_t)i * m + j] = xtx_host[(size_t)i * S + j] * (1.0 / y);
  double *d_a = nullptr, *d_v = nullptr;
  int* d_sweeps = nullptr;
  KMD_HIP(hipMalloc(reinterpret_cast<void**>(&d_a), (size_t)m * m * 8));
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_v), (size_t)m * m * 8);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_sweeps), 4);
  if (e == hipSuccess) e = hipMemcpy(d_a, a.data(), (size_t)m * m * 8, hipMemcpyHostToDevice);
  int sweeps = -1;
  std::vector<double> v((size_t)m * m);
  if (e == hipSuccess)
  {
    const size_t lds = (size_t)m * 8 + (size_t)m * 4 + (size_t)m * 4 + 64;       // cs + top/bot + nxt
    hipLaunchKernelGGL(k_jacobi, dim3(1), dim3(1024), lds, nullptr, d_a, d_v, m, S, 60, d_sweeps);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(a.data(), d_a, (size_t)m * m * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(v.data(), d_v, (size_t)m * m * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(&sweeps, d_sweeps, 4, hipMemcpyDeviceToHost);
  }
  (void)hipFree(d_a); if (d_v) (void)hipFree(d_v); if (d_sweeps) (void)hipFree(d_sweeps);
  if (e != hipSuccess) { kmd::set_error(std::string("kmd_pca_eigen: ") + hipGetErrorString(e)); return KMD_E_HIP; }
  KMD_REQUIRE(sweeps >= 0 && sweeps < 60, "kmd_pca_eigen: Jacobi did not converge");
  // eigenvalues in decreasing order (smartpca.c:1026); unit-norm vectors, largest component positive
  std::vector<int> order(S);
  for (int i = 0; i < S; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int x, int z) { return a[(size_t)x * m + x] > a[(size_t)z * m + z]; });
  for (int k = 0; k < n_out; ++k)
  {
    const int col = order[k];
    eval_host[k] = a[(size_t)col * m + col];
    double norm2 = 0.0, big = 0.0;
    for (int i = 0; i < S; ++i)
    {
      const double x = v[(size_t)i * m + col];
      norm2 += x * x;
      if (fabs(x) > fabs(big)) big = x;
    }
    const double scale = (big < 0.0 ? -1.0 : 1.0) / sqrt(norm2);
    for (int i = 0; i < S; ++i) evec_host[(size_t)i * n_out + k] = v[(size_t)i * m + col] * scale;
  }
  return KMD_OK;
}

} // extern "C"
