"""The PCA front end of the pop-strat stage on the device (kmd_pca_*) against the numpy
restatement of Hawk's smartpca in oracle/pca_oracle.py: which rows are sampled, the Gram matrix,
the eigen-decomposition, and that planted population structure comes out as the first component."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pca_oracle as PO  # noqa: E402

pytestmark = pytest.mark.gpu
SEED = 0x6B6D64696666


@pytest.fixture(scope="module")
def K():
    import kmdiff_amd
    return kmdiff_amd


def structured_counts(rng, n, S, n_pop_a):
    """Presence depends on the population for 20 % of the rows: two populations of samples."""
    base = rng.random((n, 1)) * 0.8 + 0.1
    shift = np.where(rng.random((n, 1)) < 0.2, rng.normal(0, 0.35, (n, 1)), 0.0)
    prob = np.clip(np.concatenate([np.repeat(base + shift, n_pop_a, axis=1), np.repeat(base - shift, S - n_pop_a, axis=1)], axis=1), 0.02, 0.98)
    counts = (rng.random((n, S)) < prob) * rng.integers(1, 50, (n, S))
    counts[counts.sum(axis=1) == 0, 0] = 1                      # the merge never emits an all-zero row
    return counts.astype(np.uint32)


@pytest.mark.parametrize("layout_name,S,diploid", [("rows", 12, True), ("tiled", 40, True), ("soa", 70, False), ("tiled", 131, True)])
def test_sampling_and_gram_match_oracle(K, layout_name, S, diploid):
    rng = np.random.default_rng(S)
    n = 30_000
    counts = structured_counts(rng, n, S, S // 2)
    kmers = np.sort(rng.integers(0, 1 << 62, n, dtype=np.uint64))
    layout = {"rows": K.LAYOUT_ROWS, "tiled": K.LAYOUT_TILED, "soa": K.LAYOUT_SOA}[layout_name]
    mat = K.CountMatrix.from_host(counts if layout == K.LAYOUT_ROWS else np.ascontiguousarray(counts.T), layout, kmer_lo=kmers)
    rate = 0.05
    pca = K.PopulationPCA(S, rate, seed=SEED, diploid=diploid, capacity=n)
    pca.sample(mat)
    mask = PO.sampled_mask(SEED, rate, kmers)
    assert pca.count() == int(mask.sum()) and 0.03 * n < mask.sum() < 0.07 * n
    want = PO.gram(counts[mask], diploid)
    got = pca.gram()
    assert np.allclose(got, got.T, rtol=0, atol=1e-9 * np.abs(want).max())
    assert np.allclose(got, want, rtol=1e-12, atol=1e-9)
    # a second tile accumulates (partitions pass one after the other)
    pca.sample(mat)
    assert pca.count() == 2 * int(mask.sum())
    assert np.allclose(pca.gram(), 2 * want, rtol=1e-12, atol=1e-9)


def test_every_row_and_growth(K):
    rng = np.random.default_rng(3)
    n, S = 500, 6
    counts = structured_counts(rng, n, S, 3)
    kmers = np.arange(n, dtype=np.uint64) * 977
    mat = K.CountMatrix.from_host(counts, K.LAYOUT_ROWS, kmer_lo=kmers)
    pca = K.PopulationPCA(S, 1.0, capacity=100)                 # rate 1: every row; the store grows past its first size
    pca.sample(mat)
    assert pca.count() == n and np.allclose(pca.gram(), PO.gram(counts), rtol=1e-12, atol=1e-10)
    for _ in range(3):
        pca.sample(mat)
    assert pca.count() == 4 * n and np.allclose(pca.gram(), 4 * PO.gram(counts), rtol=1e-12, atol=1e-9)
    nokmers = K.CountMatrix.from_host(counts, K.LAYOUT_ROWS)
    with pytest.raises(K.KmdError):
        K.PopulationPCA(S, 0.5, capacity=n).sample(nokmers)


@pytest.mark.parametrize("S", [2, 7, 40, 101, 200])
def test_eigen_matches_numpy(K, S):
    rng = np.random.default_rng(100 + S)
    counts = structured_counts(rng, 4000, S, max(1, S // 3))
    xtx = PO.gram(counts)
    n_out = min(10, S)
    evec, evals = K.pca_eigen(xtx, n_out)
    wv, ww = PO.eigen(xtx, n_out)
    assert np.allclose(evals, ww, rtol=1e-10, atol=1e-12)
    assert np.allclose(np.linalg.norm(evec, axis=0), 1.0, rtol=0, atol=1e-12)
    a = xtx / (np.trace(xtx) / (S - 1))
    for k in range(n_out):                                      # A v = lambda v, whatever the neighbours' gaps
        assert np.allclose(a @ evec[:, k], evals[k] * evec[:, k], rtol=0, atol=1e-9 * max(1.0, abs(evals[0])))
    gaps = np.abs(np.diff(ww))
    for k in range(min(3, n_out)):                              # well separated components agree entry by entry
        near = min(gaps[k - 1] if k > 0 else np.inf, gaps[k] if k < len(gaps) else np.inf)
        if near > 1e-3 * ww[0]:
            assert np.allclose(evec[:, k], wv[:, k], rtol=0, atol=1e-8)


def test_planted_populations_are_the_first_component(K):
    rng = np.random.default_rng(8)
    n, S, a = 60_000, 24, 10
    counts = structured_counts(rng, n, S, a)
    kmers = np.sort(rng.integers(0, 1 << 62, n, dtype=np.uint64))
    mat = K.CountMatrix.from_host(np.ascontiguousarray(counts.T), K.LAYOUT_TILED, kmer_lo=kmers)
    pca = K.PopulationPCA(S, 0.2, seed=5, capacity=n)
    pca.sample(mat)
    evec, evals = K.pca_eigen(pca.gram(), 10)
    pc1 = evec[:, 0]
    assert (np.sign(pc1[:a]) == np.sign(pc1[0])).all() and (np.sign(pc1[a:]) == -np.sign(pc1[0])).all()
    assert evals[0] > 1.5 * evals[1]
    lines = PO.pcs_evec_lines(evec)
    assert len(lines) == S and all(len(l.split()) == 10 for l in lines)
