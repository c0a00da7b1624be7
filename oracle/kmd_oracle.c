/* oracle/kmd_oracle.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 * See oracle/kmd_oracle.h for scope, pinning status and who may load this library.
 * Compiled WITHOUT FP contraction (-ffp-contract=off) and without -march: plain IEEE
 * binary64, the arithmetic SURVEY.md 8c's known answers were produced with.
 */
#define _GNU_SOURCE
#include "kmd_oracle.h"
#include "../include/kmdiff_synth_tables.h"   /* the synthetic input's tables: data shared with the device generator */

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------------ */
/* R5  log-factorial table                                                               */
/* ------------------------------------------------------------------------------------ */

/* src/log_factorial_table.cpp:13-22 -- descending sum log(k)+log(k-1)+...+log(2) */
static double lf_descending(uint64_t k)
{
  double res = 0;
  while (k > 1) { res += log((double)k); k--; }
  return res;
}

/* src/log_factorial_table.cpp:5-11 */
void kmdo_lf_build(size_t size, double* out)
{
  for (size_t i = 0; i < size; i++) out[i] = lf_descending(i);
}

/* include/kmdiff/log_factorial_table.hpp:14-18 -- table hit, else the O(k) loop */
double kmdo_lf_at(const double* table, size_t size, uint64_t k)
{
  if (k < size) return table[k];
  return lf_descending(k);
}

/* ------------------------------------------------------------------------------------ */
/* R6  chi-square upper tail (Cephes igamc as shipped in alglib)                         */
/* ------------------------------------------------------------------------------------ */

/* thirdparty/alglib/src/specialfunctions.cpp:3744-3865 (Cephes lgam) */
double kmdo_lngamma(double x, double* sgngam)
{
  const double logpi = 1.14472988584940017414;
  const double ls2pi = 0.91893853320467274178;
  double tmp;
  *sgngam = 1;
  if (x < -34.0)
  {
    double q = -x;
    double w = kmdo_lngamma(q, &tmp);
    double p = floor(q);
    long i = lround(p);
    *sgngam = (i % 2 == 0) ? -1 : 1;
    double z = q - p;
    if (z > 0.5) { p = p + 1; z = p - q; }
    z = q * sin(3.14159265358979323846 * z);
    return logpi - log(z) - w;
  }
  if (x < 13)
  {
    double z = 1, p = 0, u = x;
    while (u >= 3) { p = p - 1; u = x + p; z = z * u; }
    while (u < 2)  { z = z / u; p = p + 1; u = x + p; }
    if (z < 0) { *sgngam = -1; z = -z; } else { *sgngam = 1; }
    if (u == 2) return log(z);
    p = p - 2;
    x = x + p;
    double b = -1378.25152569120859100;
    b = -38801.6315134637840924 + x * b;
    b = -331612.992738871184744 + x * b;
    b = -1162370.97492762307383 + x * b;
    b = -1721737.00820839662146 + x * b;
    b = -853555.664245765465627 + x * b;
    double c = 1;
    c = -351.815701436523470549 + x * c;
    c = -17064.2106651881159223 + x * c;
    c = -220528.590553854454839 + x * c;
    c = -1139334.44367982507207 + x * c;
    c = -2532523.07177582951285 + x * c;
    c = -2018891.41433532773231 + x * c;
    p = x * b / c;
    return log(z) + p;
  }
  double q = (x - 0.5) * log(x) - x + ls2pi;
  if (x > 100000000) return q;
  double p = 1 / (x * x);
  if (x >= 1000.0)
  {
    q = q + ((7.9365079365079365079365 * 0.0001 * p - 2.7777777777777777777778 * 0.001) * p
             + 0.0833333333333333333333) / x;
  }
  else
  {
    double a = 8.11614167470508450300 * 0.0001;
    a = -5.95061904284301438324 * 0.0001 + p * a;
    a = 7.93650340457716943945 * 0.0001 + p * a;
    a = -2.77777777730099687205 * 0.001 + p * a;
    a = 8.33333333333331927722 * 0.01 + p * a;
    q = q + a / x;
  }
  return q;
}

/* specialfunctions.cpp:4578-4619 (Cephes igam, power series) */
double kmdo_igam(double a, double x)
{
  const double eps = 0.000000000000001;
  double tmp;
  if (x <= 0 || a <= 0) return 0;
  if (x > 1 && x > a) return 1 - kmdo_igamc(a, x);
  double ax = a * log(x) - x - kmdo_lngamma(a, &tmp);
  if (ax < -709.78271289338399) return 0;
  ax = exp(ax);
  double r = a, c = 1, ans = 1;
  do { r = r + 1; c = c * x / r; ans = ans + c; } while (c / ans > eps);
  return ans * ax / a;
}

/* specialfunctions.cpp:4655-4739 (Cephes igamc, continued fraction) */
double kmdo_igamc(double a, double x)
{
  const double eps = 0.000000000000001;
  const double big = 4503599627370496.0;
  const double biginv = 2.22044604925031308085 * 0.0000000000000001;
  double tmp;
  if (x <= 0 || a <= 0) return 1;
  if (x < 1 || x < a) return 1 - kmdo_igam(a, x);
  double ax = a * log(x) - x - kmdo_lngamma(a, &tmp);
  if (ax < -709.78271289338399) return 0;
  ax = exp(ax);
  double y = 1 - a, z = x + y + 1, c = 0;
  double pkm2 = 1, qkm2 = x, pkm1 = x + 1, qkm1 = z * x;
  double ans = pkm1 / qkm1, t;
  do
  {
    c = c + 1; y = y + 1; z = z + 2;
    double yc = y * c;
    double pk = pkm1 * z - pkm2 * yc;
    double qk = qkm1 * z - qkm2 * yc;
    if (qk != 0) { double r = pk / qk; t = fabs((ans - r) / r); ans = r; }
    else t = 1;
    pkm2 = pkm1; pkm1 = pk; qkm2 = qkm1; qkm1 = qk;
    if (fabs(pk) > big)
    {
      pkm2 *= biginv; pkm1 *= biginv; qkm2 *= biginv; qkm1 *= biginv;
    }
  } while (t > eps);
  return ans * ax;
}

/* specialfunctions.cpp:9559-9567; the ae_assert (x>=0 && v>=1) throws in the reference */
double kmdo_chisqc(double v, double x)
{
  if (!(x >= 0 && v >= 1)) return NAN;
  return kmdo_igamc(v / 2.0, x / 2.0);
}

/* ------------------------------------------------------------------------------------ */
/* R4  PoissonLikelihood::process                                                        */
/* ------------------------------------------------------------------------------------ */

typedef struct
{
  const double* lf; size_t lf_n;
  int nc, nk;
  uint64_t tc, tk;
} model_t;

/* model.hpp:133-138 */
static double poisson_prob(const model_t* m, int k, double lambda)
{
  if (lambda <= 0) return 0;
  if (k < 0) k = 0;
  return (-lambda + (k * log(lambda) - kmdo_lf_at(m->lf, m->lf_n, (uint64_t)k)));
}

static inline double count_at(const void* counts, int count_bytes, int layout, size_t ld,
                              size_t row, size_t s)
{
  size_t idx = (layout == KMDO_LAYOUT_ROWS) ? row * ld + s : s * ld + row;
  switch (count_bytes)
  {
    case 1: return ((const uint8_t*)counts)[idx];
    case 2: return ((const uint16_t*)counts)[idx];
    default: return ((const uint32_t*)counts)[idx];
  }
}

/* model.hpp:142-176 -- one row */
static void process_row(const model_t* m, const void* counts, int count_bytes, int layout,
                        size_t ld, size_t row, double* p, int32_t* sign, double* mc, double* mk)
{
  /* compute_sum_e, model.hpp:70-81: accumulate in double, sample order */
  double mean_control = 0, mean_case = 0;
  for (int i = 0; i < m->nc; i++) mean_control += count_at(counts, count_bytes, layout, ld, row, i);
  for (int i = 0; i < m->nk; i++) mean_case += count_at(counts, count_bytes, layout, ld, row, m->nc + i);

  double mean = (mean_control + mean_case) / (double)(m->tc + m->tk);            /* :147 */
  double null_h = 0, alt_h = 0;
  alt_h += poisson_prob(m, (int)mean_control, mean_control);                     /* :152 */
  alt_h += poisson_prob(m, (int)mean_case, mean_case);                           /* :153 */
  null_h += poisson_prob(m, (int)mean_control, mean * (double)m->tc);            /* :155 */
  null_h += poisson_prob(m, (int)mean_case, mean * (double)m->tk);               /* :156 */
  double lr = alt_h - null_h;                                                    /* :158 */
  if (lr < 0) lr = 0;                                                            /* :160 */
  *p = kmdo_chisqc(1, 2 * lr);                                                   /* :161 */
  mean_control = mean_control * (double)m->tk / (double)m->tc;                   /* :165 */
  if (mean_control < mean_case) *sign = KMDO_SIGN_CASE;                          /* :167-172 */
  else if (mean_control > mean_case) *sign = KMDO_SIGN_CONTROL;
  else *sign = KMDO_SIGN_NO;
  *mc = mean_control; *mk = mean_case;                                           /* :174 */
}

void kmdo_poisson_rows(const void* counts, int count_bytes, int layout, size_t ld,
                       size_t n_rows, int nc, int nk, uint64_t tc, uint64_t tk,
                       const double* lf, size_t lf_n,
                       double* p, int32_t* sign, double* mean_ctrl, double* mean_case)
{
  model_t m = { lf, lf_n, nc, nk, tc, tk };
  for (size_t r = 0; r < n_rows; r++)
    process_row(&m, counts, count_bytes, layout, ld, r, &p[r], &sign[r], &mean_ctrl[r], &mean_case[r]);
}

/* ------------------------------------------------------------------------------------ */
/* R3  diff_observer::process over one partition                                         */
/* ------------------------------------------------------------------------------------ */

size_t kmdo_diff_partition(const void* counts, int count_bytes, int layout, size_t ld,
                           size_t n_rows, int nc, int nk, uint64_t tc, uint64_t tk,
                           const double* lf, size_t lf_n, double threshold,
                           uint64_t* surv_row, double* surv_p, int32_t* surv_sign,
                           double* surv_mc, double* surv_mk, size_t cap,
                           kmdo_counters* counters)
{
  model_t m = { lf, lf_n, nc, nk, tc, tk };
  kmdo_counters c = { 0, 0, 0, 0 };
  for (size_t r = 0; r < n_rows; r++)
  {
    double p, mc, mk; int32_t sign;
    process_row(&m, counts, count_bytes, layout, ld, r, &p, &sign, &mc, &mk);   /* merge.hpp:73 */
    c.total++;                                                                    /* :76 */
    if (p <= threshold)                                                           /* :78 */
    {
      if (c.n_sig < cap)
      {
        if (surv_row) surv_row[c.n_sig] = r;
        if (surv_p) surv_p[c.n_sig] = p;
        if (surv_sign) surv_sign[c.n_sig] = sign;
        if (surv_mc) surv_mc[c.n_sig] = mc;
        if (surv_mk) surv_mk[c.n_sig] = mk;
      }
      if (sign == KMDO_SIGN_CONTROL) c.n_sig_control++; else c.n_sig_case++;      /* :95-98 */
      c.n_sig++;                                                                  /* :101 */
    }
  }
  if (counters) *counters = c;
  return c.n_sig;
}

/* ------------------------------------------------------------------------------------ */
/* R8  correctors and aggregator decisions                                               */
/* ------------------------------------------------------------------------------------ */

void kmdo_corrector_init(kmdo_corrector* c, int type, double threshold, uint64_t total)
{
  c->type = type; c->threshold = threshold; c->total = total; c->rank = 1;  /* corrector.hpp:31 */
}

int kmdo_corrector_apply(kmdo_corrector* c, double p)
{
  switch (c->type)
  {
    case KMDO_CORR_BONFERRONI:                                   /* corrector.cpp:9-12 */
      return p < (c->threshold / c->total);
    case KMDO_CORR_BENJAMINI:                                    /* corrector.cpp:27-35 */
      if (p < ((c->rank / (double)c->total) * c->threshold)) { c->rank++; return 1; }
      return 0;
    case KMDO_CORR_SIDAK:                                        /* corrector.cpp:50-53 */
      return p < (1 - pow(1 - c->threshold, 1.0 / c->total));
    case KMDO_CORR_HOLM:                                         /* corrector.cpp:68-71 */
      return p < (c->threshold / c->total--);
    default:                                                     /* corrector.cpp:85-88 */
      return p < c->threshold;
  }
}

typedef struct { double p; size_t i; } pidx_t;
static int pidx_cmp(const void* a, const void* b)
{
  const pidx_t* x = a; const pidx_t* y = b;
  if (x->p < y->p) return -1;
  if (x->p > y->p) return 1;
  return (x->i > y->i) - (x->i < y->i);
}

size_t kmdo_aggregate(int type, double threshold, uint64_t total_kmers,
                      const double* p, size_t n, uint8_t* keep)
{
  kmdo_corrector c;
  kmdo_corrector_init(&c, type, threshold, total_kmers);       /* cmd/diff.hpp:249 */
  size_t kept = 0;
  memset(keep, 0, n);
  if (type == KMDO_CORR_BENJAMINI || type == KMDO_CORR_HOLM)    /* aggregator.hpp:358-360 */
  {
    /* sorted_aggregator::run, aggregator.hpp:286-310: ascending p, stop at first reject */
    pidx_t* v = malloc((n ? n : 1) * sizeof(pidx_t));
    for (size_t i = 0; i < n; i++) { v[i].p = p[i]; v[i].i = i; }
    qsort(v, n, sizeof(pidx_t), pidx_cmp);
    for (size_t i = 0; i < n; i++)
    {
      if (!kmdo_corrector_apply(&c, v[i].p)) break;
      keep[v[i].i] = 1; kept++;
    }
    free(v);
  }
  else
  {
    /* aggregator::worker, aggregator.hpp:146-166: stateless predicate per survivor */
    for (size_t i = 0; i < n; i++)
      if (kmdo_corrector_apply(&c, p[i])) { keep[i] = 1; kept++; }
  }
  return kept;
}

/* ------------------------------------------------------------------------------------ */
/* Synthetic count matrices.  Integer-only, counter-based, replayable:    */
/* every cell is a pure function of (seed, partition, row, sample).                      */
/*                                                                                       */
/*   h_row   = mix(mix(seed ^ C_PART*(part+1)) ^ C_ROW*(row+1))                          */
/*   class   = heavy-tailed rate class from h_row bits [0,16): weights .40 .30 .15 .10    */
/*             .04 .01 -> lambda index base {1,3,5,7,11,17} (lambda_j = 0.5*2^(j/2))      */
/*   big     = (h_row >> 16) % 1000000 == 1      -> base index 24 (count sums > 10000)    */
/*   planted = (h_row >> 36) % 10000 == 0        -> +6 index steps (x8) on cases, or on   */
/*             controls when bit 63 of h_row is set                                       */
/*   depth_s = mix(seed ^ C_DEPTH*(s+1)) % 3     -> +0/+1/+2 index steps per sample       */
/*   cell    = h = mix(h_row ^ C_CELL*(s+1)); classes 0,1 are zero-inflated: absent when  */
/*             (h >> 32) & 0xFFFF < 19661 (p = 0.3); else inverse-CDF Poisson draw of the */
/*             low 32 bits of h in table j (kmdiff_synth_tables.h); clamped to the count type    */
/*   all-zero rows get count 1 in sample h_row % S (the merge never emits empty rows)     */
/*   kmer    = part*2^54 + row*2^21 + 1 + (mix(h_row ^ C_KMER) & 0xFFFFF): strictly       */
/*             increasing in row, < 4^31; for k > 32 this is the HIGH limb and the low    */
/*             limb is mix(h_row ^ C_KMER2)                                               */
/* ------------------------------------------------------------------------------------ */

#define C_PART  0xA0761D6478BD642FULL
#define C_ROW   0xE7037ED1A0B428DBULL
#define C_DEPTH 0x8EBC6AF09C88C6E3ULL
#define C_CELL  0x589965CC75374CC3ULL
#define C_KMER  0x1D8E4E27C47D124FULL
#define C_KMER2 0xEB44ACCAB455D165ULL

static inline uint64_t mix64(uint64_t x)   /* splitmix64 finaliser */
{
  x += 0x9E3779B97F4A7C15ULL;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
  return x ^ (x >> 31);
}

static inline uint64_t row_hash(uint64_t seed, uint32_t part, uint64_t row)
{
  return mix64(mix64(seed ^ (C_PART * ((uint64_t)part + 1))) ^ (C_ROW * (row + 1)));
}

static inline int synth_class_base(uint64_t h)
{
  static const int base[6] = { 1, 3, 5, 7, 11, 17 };
  uint32_t u = (uint32_t)(h & 0xFFFF);
  int cls = u < 26214 ? 0 : u < 45875 ? 1 : u < 55705 ? 2 : u < 62259 ? 3 : u < 64880 ? 4 : 5;
  if ((h >> 16) % 1000000ULL == 1) return 24 | (6 << 8);
  return base[cls] | (cls << 8);
}

static inline uint32_t synth_poisson(int j, uint32_t u)
{
  const unsigned int* thr = KMD_SYNTH_THR + KMD_SYNTH_OFF[j];
  uint32_t lo = 0, hi = KMD_SYNTH_LEN[j];      /* count of thresholds <= u */
  while (lo < hi)
  {
    uint32_t mid = (lo + hi) >> 1;
    if (thr[mid] <= u) lo = mid + 1; else hi = mid;
  }
  return KMD_SYNTH_C0[j] + lo;
}

static inline uint32_t synth_cell(uint64_t seed, uint64_t h, int cb, int s, int is_case)
{
  int j = cb & 0xFF, cls = cb >> 8;
  uint64_t hc = mix64(h ^ (C_CELL * ((uint64_t)s + 1)));
  if (cls <= 1 && ((hc >> 32) & 0xFFFF) < 19661) return 0;
  j += (int)(mix64(seed ^ (C_DEPTH * ((uint64_t)s + 1))) % 3);
  if ((h >> 36) % 10000ULL == 0)
  {
    int boost_controls = (int)(h >> 63);
    if (boost_controls != is_case) j += 6;
  }
  if (j > KMD_SYNTH_NJ - 1) j = KMD_SYNTH_NJ - 1;
  return synth_poisson(j, (uint32_t)hc);
}

uint64_t kmdo_synth_kmer(uint64_t seed, uint32_t part, uint64_t row, uint64_t* hi)
{
  uint64_t h = row_hash(seed, part, row);
  uint64_t v = ((uint64_t)part << 54) + (row << 21) + 1 + (mix64(h ^ C_KMER) & 0xFFFFF);
  if (hi) { *hi = v; return mix64(h ^ C_KMER2); }
  return v;
}

static inline void store_count(void* counts, int count_bytes, size_t idx, uint32_t v)
{
  switch (count_bytes)
  {
    case 1: ((uint8_t*)counts)[idx] = v > 0xFF ? 0xFF : (uint8_t)v; break;
    case 2: ((uint16_t*)counts)[idx] = v > 0xFFFF ? 0xFFFF : (uint16_t)v; break;
    default: ((uint32_t*)counts)[idx] = v;
  }
}

void kmdo_synth_rows(uint64_t seed, uint32_t part, uint64_t row0, size_t n_rows, int nc, int nk,
                     int count_bytes, int layout, size_t ld, void* counts,
                     uint64_t* kmer_lo, uint64_t* kmer_hi)
{
  const int S = nc + nk;
  for (size_t r = 0; r < n_rows; r++)
  {
    uint64_t row = row0 + r;
    uint64_t h = row_hash(seed, part, row);
    int cb = synth_class_base(h);
    int any = 0;
    for (int s = 0; s < S; s++)
    {
      uint32_t v = synth_cell(seed, h, cb, s, s >= nc);
      any |= (v != 0);
      size_t idx = (layout == KMDO_LAYOUT_ROWS) ? r * ld + s : (size_t)s * ld + r;
      store_count(counts, count_bytes, idx, v);
    }
    if (!any)
    {
      int s = (int)(h % (uint64_t)S);
      size_t idx = (layout == KMDO_LAYOUT_ROWS) ? r * ld + s : (size_t)s * ld + r;
      store_count(counts, count_bytes, idx, 1);
    }
    if (kmer_lo)
    {
      uint64_t hi;
      if (kmer_hi) { kmer_lo[r] = kmdo_synth_kmer(seed, part, row, &hi); kmer_hi[r] = hi; }
      else kmer_lo[r] = kmdo_synth_kmer(seed, part, row, NULL);
    }
  }
}

/* ------------------------------------------------------------------------------------ */
/* CPU baseline driver: global_merge::merge (merge.hpp:239-317) -- one task per partition */
/* on a pool of n_threads; each task streams ROW-MAJOR rows (the shape matrix_proxy hands */
/* the observer, merge.hpp:194-203) through process_row with the tail function evaluated  */
/* for every row, exactly like the reference.  A worker generates its partition into a    */
/* thread-local buffer (not timed), then runs the timed test loop over it; the return     */
/* value is the largest per-thread sum of test-loop times, i.e. the parallel makespan.    */
/* ------------------------------------------------------------------------------------ */

typedef struct
{
  uint64_t seed; int n_parts; size_t rows; int nc, nk, count_bytes;
  uint64_t tc, tk; const double* lf; size_t lf_n; double threshold;
  kmdo_counters* per_part; double* thread_secs;
  int next, next_tid; pthread_mutex_t mu;
} bench_t;

static double now_s(void)
{
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static void* bench_worker(void* arg)
{
  bench_t* b = arg;
  pthread_mutex_lock(&b->mu);
  int tid = b->next_tid++;
  pthread_mutex_unlock(&b->mu);
  const size_t S = (size_t)(b->nc + b->nk);
  void* mat = malloc(b->rows * S * (size_t)b->count_bytes);
  double secs = 0;
  for (;;)
  {
    pthread_mutex_lock(&b->mu);
    int p = b->next++;
    pthread_mutex_unlock(&b->mu);
    if (p >= b->n_parts) break;
    kmdo_synth_rows(b->seed, (uint32_t)p, 0, b->rows, b->nc, b->nk, b->count_bytes,
                    KMDO_LAYOUT_ROWS, S, mat, NULL, NULL);
    double t0 = now_s();
    kmdo_diff_partition(mat, b->count_bytes, KMDO_LAYOUT_ROWS, S, b->rows, b->nc, b->nk,
                        b->tc, b->tk, b->lf, b->lf_n, b->threshold,
                        NULL, NULL, NULL, NULL, NULL, 0, &b->per_part[p]);
    secs += now_s() - t0;
  }
  b->thread_secs[tid] = secs;
  free(mat);
  return NULL;
}

double kmdo_bench_partitions(uint64_t seed, int n_parts, size_t rows_per_part, int nc, int nk,
                             int count_bytes, uint64_t tc, uint64_t tk, size_t lf_n,
                             double threshold, int n_threads, kmdo_counters* counters)
{
  bench_t b;
  memset(&b, 0, sizeof b);
  b.seed = seed; b.n_parts = n_parts; b.rows = rows_per_part; b.nc = nc; b.nk = nk;
  b.count_bytes = count_bytes; b.tc = tc; b.tk = tk; b.lf_n = lf_n; b.threshold = threshold;
  double* lf = malloc(sizeof(double) * (lf_n ? lf_n : 1));
  kmdo_lf_build(lf_n, lf);
  b.lf = lf;
  b.per_part = calloc((size_t)n_parts, sizeof(kmdo_counters));
  b.thread_secs = calloc((size_t)n_threads, sizeof(double));
  pthread_mutex_init(&b.mu, NULL);
  pthread_t* th = malloc(sizeof(pthread_t) * (size_t)n_threads);
  for (int i = 0; i < n_threads; i++) pthread_create(&th[i], NULL, bench_worker, &b);
  for (int i = 0; i < n_threads; i++) pthread_join(th[i], NULL);
  free(th);
  double makespan = 0;
  for (int i = 0; i < n_threads; i++) if (b.thread_secs[i] > makespan) makespan = b.thread_secs[i];
  kmdo_counters c = { 0, 0, 0, 0 };
  for (int p = 0; p < n_parts; p++)
  {
    c.total += b.per_part[p].total; c.n_sig += b.per_part[p].n_sig;
    c.n_sig_control += b.per_part[p].n_sig_control; c.n_sig_case += b.per_part[p].n_sig_case;
  }
  if (counters) *counters = c;
  free(b.per_part); free(b.thread_secs); free(lf);
  pthread_mutex_destroy(&b.mu);
  return makespan;
}

/* ------------------------------------------------------------------------------------ */
/* R9  pop-strat re-test: IRLS logistic regression + LRT                                 */
/* Matrices are dense row-major doubles.                                                 */
/* ------------------------------------------------------------------------------------ */

/* Test instrument, not the reference's code: when armed (kmdo_sigmoid_jitter), the pow() below is moved by one ulp, up or
 * down, in about one call in 2^log2_every -- the difference two correct libms may show.  It answers "is this row's p-value
 * a property of the data or of the last bit of pow()": pop-strat's default design is not standardised (a column of ~1e7
 * beside one of ~1e-6, Hessian condition number beyond 1e16), and IRLS on it can amplify one ulp to any size. */
static uint64_t g_jitter_state = 0;
static int g_jitter_log2 = 0;
void kmdo_sigmoid_jitter(uint64_t seed, int log2_every) { g_jitter_state = seed; g_jitter_log2 = log2_every; }

/* src/linear_model.cpp:191-195 */
double kmdo_sigmoid(double x)
{
  double e = M_E;
  double pw = pow(e, -x);
  if (g_jitter_state)
  {
    g_jitter_state ^= g_jitter_state << 13; g_jitter_state ^= g_jitter_state >> 7; g_jitter_state ^= g_jitter_state << 17;
    if ((g_jitter_state >> 20 & ((1ull << g_jitter_log2) - 1)) == 0)
      pw = nextafter(pw, (g_jitter_state >> 60 & 1) ? INFINITY : 0.0);
  }
  return 1.0 / (1.0 + pw);
}

/* src/linear_model.cpp:94-132 -- Doolittle, no pivoting */
void kmdo_lu(const double* a, int n, double* lower, double* upper)
{
  memset(lower, 0, sizeof(double) * (size_t)n * (size_t)n);
  memset(upper, 0, sizeof(double) * (size_t)n * (size_t)n);
  for (int i = 0; i < n; i++)
  {
    for (int k = i; k < n; k++)
    {
      double sum = 0.0;
      for (int j = 0; j < i; j++) sum += lower[i * n + j] * upper[j * n + k];
      upper[i * n + k] = a[i * n + k] - sum;
    }
    for (int k = i; k < n; k++)
    {
      if (i == k) lower[i * n + i] = 1;
      else
      {
        double sum = 0;
        for (int j = 0; j < i; j++) sum += lower[k * n + j] * upper[j * n + i];
        lower[k * n + i] = (a[k * n + i] - sum) / upper[i * n + i];
      }
    }
  }
}

/* src/linear_model.cpp:134-189 -- per-column forward/back substitution; det is the
 * running product over EVERY column solve (so it is det^n), used only for ==0 / NaN */
int kmdo_inverse(const double* a, int n, double* inv)
{
  double* lower = malloc(sizeof(double) * (size_t)n * (size_t)n * 2);
  double* upper = lower + (size_t)n * (size_t)n;
  double* y = malloc(sizeof(double) * (size_t)n * 2);
  double* x = y + n;
  kmdo_lu(a, n, lower, upper);
  double det = 1;
  for (int invc = 0; invc < n; invc++)
  {
    for (int j = 0; j < n; j++) { y[j] = 0; x[j] = 0; }
    det *= lower[0];
    y[0] = (invc == 0) ? 1 : 0;
    for (int row = 1; row < n; row++)
    {
      double sum = 0;
      for (int col = 0; col < n; col++) sum += lower[row * n + col] * y[col];
      y[row] = ((invc == row) ? 1 : 0) - sum;
      det *= lower[row * n + row];
    }
    x[n - 1] = y[n - 1] / upper[(n - 1) * n + (n - 1)];
    det *= upper[(n - 1) * n + (n - 1)];
    for (int row = n - 2; row > -1; row--)
    {
      double sum = 0;
      for (int col = row + 1; col < n; col++) sum += upper[row * n + col] * x[col];
      x[row] = (y[row] - sum) / upper[row * n + row];
      det *= upper[row * n + row];
    }
    for (int j = 0; j < n; j++) inv[j * n + invc] = x[j];
  }
  int flags = 0;
  if (det == 0) flags = 1; else if (isnan(det)) flags = 2;
  free(lower); free(y);
  return flags;
}

/* src/linear_model.cpp:297-410.  x is n x f row-major.  Returns the iteration count
 * (`ein`); weight[] (f) is what the reference returns as `weight`. */
int kmdo_glm_irls(const double* x, const double* y, int n, int f, int max_iters,
                  double* weight, double* ret_error, int* flags)
{
  const double epsilon = 1e-6;
  int iter = 0, ein = 0, fl = 0;
  double rerr = 0;
  double* eta = malloc(sizeof(double) * (size_t)n * 4);
  double* mu = eta + n; double* S = mu + n; double* z = S + n;
  int* good = malloc(sizeof(int) * (size_t)n);
  double* H = malloc(sizeof(double) * (size_t)f * (size_t)f * 2);
  double* Hinv = H + (size_t)f * (size_t)f;
  double* xtsz = malloc(sizeof(double) * (size_t)f * 2);
  double* w = xtsz + f;
  for (int j = 0; j < f; j++) { weight[j] = 1; w[j] = 1; }
  for (int i = 0; i < n; i++)
  {
    mu[i] = (y[i] + 0.5) / 2;
    eta[i] = log(mu[i] / (1 - mu[i]));
  }
  double prev_error = 1e18;
  for (;;)
  {
    double error = 0.0;
    int ng = 0;
    for (int i = 0; i < n; i++)
    {
      double g = mu[i] * (1.0 - mu[i]);
      if (g > 1e-305)
      {
        good[ng] = i;
        z[ng] = eta[i] + (y[i] - mu[i]) / (g + 1e-305);
        S[ng] = g;
        ng++;
      }
      error += (y[i] - mu[i]) * (y[i] - mu[i]);
    }
    if (ng == 0) break;
    error /= n;
    rerr = error;
    if (fabs(error - prev_error) < epsilon) break;
    prev_error = error;
    /* hessian = X^T (S X): res[a][b] = sum_k X[k][a] * (S[k]*X[k][b]), k ascending */
    for (int a = 0; a < f; a++)
      for (int b = 0; b < f; b++)
      {
        double r = 0.0;
        for (int k = 0; k < ng; k++)
          r = r + x[good[k] * f + a] * (S[k] * x[good[k] * f + b]);
        H[a * f + b] = r;
      }
    int inv_flags = kmdo_inverse(H, f, Hinv);
    if (inv_flags) { fl = inv_flags; rerr = prev_error; break; }
    for (int a = 0; a < f; a++)
    {
      double r = 0.0;
      for (int k = 0; k < ng; k++) r = r + x[good[k] * f + a] * (S[k] * z[k]);
      xtsz[a] = r;
    }
    for (int a = 0; a < f; a++)
    {
      double r = 0.0;
      for (int k = 0; k < f; k++) r = r + Hinv[a * f + k] * xtsz[k];
      w[a] = r;
    }
    iter += 1;
    ein = iter;
    if (iter >= max_iters) break;
    prev_error = error;
    rerr = prev_error;
    for (int j = 0; j < f; j++) weight[j] = w[j];
    for (int i = 0; i < n; i++)
    {
      double e = 0;
      for (int j = 0; j < f; j++) e += x[i * f + j] * w[j];
      eta[i] = e;
      mu[i] = kmdo_sigmoid(e);
    }
  }
  if (ret_error) *ret_error = rerr;
  if (flags) *flags = fl;
  free(eta); free(good); free(H); free(xtsz);
  return ein;
}

/* include/kmdiff/popstrat.hpp:249-333.  alt_global is n x f (last column overwritten with
 * counts[i]/totals[i]); the null design is its first f-1 columns (popstrat.cpp:270-311:
 * both are filled from the same sources and standardised by the same expression). */
double kmdo_popstrat_pvalue(const double* alt_global, int n, int f, const double* y,
                            const double* totals, const double* counts,
                            const double* null_model, int max_iters)
{
  double* local = malloc(sizeof(double) * (size_t)n * (size_t)f);
  double* model = malloc(sizeof(double) * (size_t)f);
  memcpy(local, alt_global, sizeof(double) * (size_t)n * (size_t)f);
  for (int i = 0; i < n; i++) local[i * f + f - 1] = counts[i] / totals[i];     /* :254-257 */
  kmdo_glm_irls(local, y, n, f, max_iters, model, NULL, NULL);                   /* :260-261 */
  double alt_l = 1.0;
  for (int r = 0; r < n; r++)                                                    /* :267-287 */
  {
    double s = 0.0;
    for (int i = 0; i < f; i++) s += model[i] * local[r * f + i];
    double p = kmdo_sigmoid(s);
    if (y[r] == 1) alt_l = alt_l * p; else alt_l *= 1.0 - p;
  }
  double null_l = 1.0;
  for (int r = 0; r < n; r++)                                                    /* :289-310 */
  {
    double s = 0.0;
    for (int i = 0; i < f - 1; i++) s += null_model[i] * alt_global[r * f + i];
    double p = kmdo_sigmoid(s);
    if (y[r] == 1) null_l *= p; else null_l *= 1.0 - p;
  }
  if (null_l == 0.0 && alt_l == 0.0) { null_l = 0.001; alt_l = 1.0; }           /* :312-316 */
  double ratio = null_l / alt_l;
  double llr = -2.0 * log(ratio);
  if (fabs(llr) < 1e-30 || llr < 0.0 || isnan(alt_l)) llr = 0.0;                 /* :321-326 */
  free(local); free(model);
  return kmdo_chisqc(1, llr);                                                    /* :328 */
}

/* src/popstrat.cpp:136-151 (ctor), 270-311 (init_global_features without covariates / sex:
 * load_C never terminates with a file and load_ginfo never finds a sex column, SURVEY.md
 * 8a R9) and 327-370 (standardize, quirks and all).  null_out: n x (2+npc), alt_out:
 * n x (3+npc) with the last (k-mer) column left 0. */
void kmdo_popstrat_features(int nc, int nk, const uint64_t* totals_c, const uint64_t* totals_k,
                            const double* Z, int z_cols, int npc, int standardize,
                            double* null_out, double* alt_out, double* totals_out)
{
  const int n = nc + nk, fn = 1 + npc + 0 + 1, fa = fn + 1;
  for (int i = 0; i < nc; i++) totals_out[i] = (double)totals_c[i];
  for (int i = 0; i < nk; i++) totals_out[nc + i] = (double)totals_k[i];
  memset(null_out, 0, sizeof(double) * (size_t)n * (size_t)fn);
  memset(alt_out, 0, sizeof(double) * (size_t)n * (size_t)fa);
  for (int i = 0; i < n; i++)
  {
    null_out[i * fn] = 1; alt_out[i * fa] = 1;                                   /* :282-283 */
    for (int z = 0; z < npc; z++)                                                /* :285-289 */
    {
      null_out[i * fn + z + 1] = Z[i * z_cols + z];
      alt_out[i * fa + z + 1] = Z[i * z_cols + z];
    }
    null_out[i * fn + 1 + npc] = totals_out[i];                                  /* :305-308 (m_unkg != 0) */
    alt_out[i * fa + 1 + npc] = totals_out[i];
  }
  if (!standardize) return;
  double* means = calloc((size_t)fn, sizeof(double));
  /* :330 sized by ROWS -- and filled by column (:349): with fn > n the reference writes past the end (undefined behaviour
   * there); the entries beyond n exist here and are never read (the division indexes rows) */
  double* stddev = calloc((size_t)(n > fn ? n : fn), sizeof(double));
  for (int i = 0; i < n; i++) for (int j = 0; j < fn; j++) means[j] += null_out[i * fn + j];
  for (int j = 1; j < fn; j++) means[j] /= fn;                                   /* :342 divides by ncols */
  for (int i = 0; i < n; i++) for (int j = 1; j < fn; j++)
    stddev[j] += pow(null_out[i * fn + j] - means[j], 2);                        /* :349 indexed by column */
  for (int j = 1; j < fn; j++) { stddev[j] /= n; stddev[j] = sqrt(stddev[j]); }  /* :353-357 */
  for (int i = 0; i < n; i++) for (int j = 1; j < fn; j++)
    if (fabs(stddev[i]) > 1e-305)                                                /* :363 indexed by ROW */
    {
      null_out[i * fn + j] = (null_out[i * fn + j] - means[j]) / stddev[i];
      alt_out[i * fa + j] = (alt_out[i * fa + j] - means[j]) / stddev[i];
    }
  free(means); free(stddev);
}


/* R1  km::KmerMerger::merge as kmdiff drives it (include/kmdiff/merge.hpp:265-289; kmtricks
 * un-vendored: contract per SURVEY.md 8a R1): S streams sorted by ascending k-mer -> one
 * row per distinct k-mer, ascending, counts row-major [row][S] (0 when absent).  Returns
 * the number of rows (at most cap rows are written). */
size_t kmdo_merge_partition(int S, const uint64_t* kmers, const uint32_t* counts, const uint64_t* offsets,
                            uint32_t* matrix_rows, uint64_t* kmer_out, size_t cap)
{
  return kmdo_merge_partition2(S, kmers, NULL, counts, offsets, matrix_rows, kmer_out, NULL, cap);
}

/* the same for k-mers of two 64-bit limbs (32 < k <= 64): (hi, lo) compared as one 128-bit
 * number; hi == NULL means one limb */
size_t kmdo_merge_partition2(int S, const uint64_t* kmers, const uint64_t* kmers_hi, const uint32_t* counts,
                             const uint64_t* offsets, uint32_t* matrix_rows, uint64_t* kmer_out,
                             uint64_t* kmer_hi_out, size_t cap)
{
  uint64_t* pos = malloc(sizeof(uint64_t) * (size_t)S);
  for (int s = 0; s < S; s++) pos[s] = offsets[s];
  size_t row = 0;
  for (;;)
  {
    int any = 0;
    uint64_t best = 0, best_hi = 0;
    for (int s = 0; s < S; s++)
      if (pos[s] < offsets[s + 1])
      {
        uint64_t lo = kmers[pos[s]], hi = kmers_hi ? kmers_hi[pos[s]] : 0;
        if (!any || hi < best_hi || (hi == best_hi && lo < best)) { best = lo; best_hi = hi; any = 1; }
      }
    if (!any) break;
    if (row < cap)
    {
      if (kmer_out) kmer_out[row] = best;
      if (kmer_hi_out) kmer_hi_out[row] = best_hi;
      for (int s = 0; s < S; s++) matrix_rows[row * (size_t)S + (size_t)s] = 0;
    }
    for (int s = 0; s < S; s++)
      if (pos[s] < offsets[s + 1] && kmers[pos[s]] == best && (!kmers_hi || kmers_hi[pos[s]] == best_hi))
      {
        if (row < cap) matrix_rows[row * (size_t)S + (size_t)s] = counts[pos[s]];
        pos[s]++;
      }
    row++;
  }
  free(pos);
  return row;
}
