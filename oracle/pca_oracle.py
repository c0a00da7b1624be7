"""CPU restatement (numpy) of the PCA front end of the pop-strat stage -- TEST INFRASTRUCTURE
ONLY: imported by tests/ as the checker of kmdiff_amd's device PCA (kmd_pca_*), never by the
product.

Follows smartpca as Hawk modified it (thirdparty/hawk/EIG6.0.1-Hawk/src/eigensrc/smartpca.c)
with the parameters kmdiff writes to parfile.txt (include/kmdiff/popstrat.hpp:28-37: usenorm YES,
numoutlieriter 0, numoutevec 10):
  fvadjust  smartpca.c:1694-1800  cc = cc > 0; ymean; p = 1 - sqrt(1 - ymean) if diploid else
                                  ymean; yfancy = 1 / sqrt(p (1 - p)) when that is positive
  getcolxz  smartpca.c:2598-2700  x = (cc - ymean) * yfancy; the row is dropped when no sample
                                  holds the k-mer (n0 = n1 = present samples, t == 0)
  main      smartpca.c:880-1025   XTX += x x^T; XTX /= trace(XTX) / (n - 1); eigvecs, decreasing
  main      smartpca.c:1140-1320  printed coordinates = unit-norm eigenvectors ("%10.4f");
                                  evec2pca.perl reprints them with "%.04f" -> pcs.evec

Parity unpinned: smartpca cannot be built here (GSL / LAPACK absent) and the reference samples
rows with a sequential RNG shared by its threads (not reproducible).  The row sampler below is the
build's own definition (hash of seed and k-mer), the same one kmd_pca.hip uses.  Eigenvector sign:
largest component positive (smartpca: whatever its eigen-solver returns)."""
import numpy as np

M64 = (1 << 64) - 1


def splitmix64(x):
    x = (np.asarray(x, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15))
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def sampled_mask(seed, rate, kmer_lo, kmer_hi=None):
    """Rows sampled for the PCA: hash(seed, k-mer) < rate * 2^64 (rate >= 1: every row)."""
    kmer_lo = np.asarray(kmer_lo, dtype=np.uint64)
    if rate >= 1.0:
        return np.ones(len(kmer_lo), dtype=bool)
    hi = np.zeros_like(kmer_lo) if kmer_hi is None else np.asarray(kmer_hi, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = splitmix64(np.uint64(seed) ^ splitmix64(kmer_lo) ^ (hi * np.uint64(0x9E3779B97F4A7C15)))
    import math
    thresh = np.uint64(int(math.ldexp(rate, 64)))
    return h < thresh


def normalised_rows(counts_rows, diploid=True):
    """x of getcolxz for every row of counts_rows [n][S]."""
    g = (np.asarray(counts_rows) > 0).astype(np.float64)
    mu = g.mean(axis=1)
    p = 1.0 - np.sqrt(1.0 - mu) if diploid else mu
    y = p * (1.0 - p)
    f = np.where(y > 0.0, 1.0 / np.sqrt(np.where(y > 0.0, y, 1.0)), 1.0)
    x = (g - mu[:, None]) * f[:, None]
    x[mu == 0.0] = 0.0
    return x


def gram(counts_rows, diploid=True):
    x = normalised_rows(counts_rows, diploid)
    return x.T @ x


def eigen(xtx, n_out=10):
    xtx = np.asarray(xtx, dtype=np.float64)
    S = xtx.shape[0]
    a = xtx / (np.trace(xtx) / (S - 1))
    w, v = np.linalg.eigh(a)
    order = np.argsort(-w, kind="stable")[:n_out]
    w, v = w[order], v[:, order]
    for k in range(v.shape[1]):
        big = v[np.argmax(np.abs(v[:, k])), k]
        v[:, k] *= (-1.0 if big < 0 else 1.0) / np.linalg.norm(v[:, k])
    return v, w


def pcs_evec_lines(evec):
    """The lines of pcs.evec: evec2pca.perl prints every coordinate as ' ' + (' ' if > 0) + '%.04f'."""
    out = []
    for row in np.asarray(evec):
        out.append("".join(" " + (" " if x > 0 else "") + "%.04f" % x for x in row))
    return out
