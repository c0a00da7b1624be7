"""The synthetic count-matrix generator as restated by the CPU oracle."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 0x6B6D64696666


def test_one_copy_of_the_generator_tables():
    """The Poisson inverse-CDF tables that define the synthetic input are ONE header, included by the device generator
    and by the oracle's independent replay (round 3 had two generated copies)."""
    assert os.path.exists(os.path.join(ROOT, "include", "kmdiff_synth_tables.h"))
    assert not os.path.exists(os.path.join(ROOT, "oracle", "synth_tables.h")) and not os.path.exists(os.path.join(ROOT, "kmdiff_amd", "csrc", "kmd_synth_tables.h"))
    assert "kmdiff_synth_tables.h" in open(os.path.join(ROOT, "oracle", "kmd_oracle.c")).read()
    assert "kmdiff_synth_tables.h" in open(os.path.join(ROOT, "kmdiff_amd", "csrc", "kmd_api.hip")).read()


def test_rows_are_pure_functions_of_their_index(oracle):
    c1, lo1, _ = oracle.synth_rows(SEED, 3, 0, 2000, 4, 4)
    c2, lo2, _ = oracle.synth_rows(SEED, 3, 500, 700, 4, 4)
    assert (c1[500:1200] == c2).all() and (lo1[500:1200] == lo2).all()
    c3, _, _ = oracle.synth_rows(SEED, 4, 0, 2000, 4, 4)
    assert (c1 != c3).any()


def test_kmers_strictly_increasing_and_in_range(oracle):
    _, lo, _ = oracle.synth_rows(SEED, 255, 0, 5000, 2, 2)
    assert (np.diff(lo.astype(np.int64)) > 0).all()
    assert int(lo.max()) < 4 ** 31
    _, lo, hi = oracle.synth_rows(SEED, 7, 10 ** 8, 1000, 2, 2, kmer_limbs=2)
    assert (np.diff(hi.astype(np.int64)) > 0).all() and int(hi.max()) < 4 ** 31


def test_no_empty_rows_and_plausible_statistics(oracle):
    c, _, _ = oracle.synth_rows(SEED, 0, 0, 200000, 20, 20)
    assert (c.sum(axis=1) > 0).all()
    m = c.mean()
    assert 1.0 < m < 12.0
    # per-sample depth differs by design (three depth steps)
    col = c.mean(axis=0)
    assert col.max() / col.min() > 1.2


def test_narrow_counts_saturate(oracle):
    c32, _, _ = oracle.synth_rows(SEED, 1, 0, 300000, 4, 4, count_bytes=4)
    c8, _, _ = oracle.synth_rows(SEED, 1, 0, 300000, 4, 4, count_bytes=1)
    assert (np.minimum(c32, 255) == c8).all()
    assert c32.max() > 255
