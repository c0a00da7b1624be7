"""Parity of the fused merge + test (kmd_merge_filter, kmd_merge_sums: kmdiff_amd/csrc/kmd_tilemerge.hip)
with the CPU oracle: km::KmerMerger::merge(diff_observer) of one partition (merge.hpp:265-289, 68-103).
The oracle side is kmdo_merge_partition[2] followed by kmdo_diff_partition on the merged matrix.
Bars as in test_gpu_parity.py: k-mers, sums, sign, means, counters bit-exact; p within 1e-10 abs / 1e-9 rel.
"""
import numpy as np
import pytest

import oracle_lib as OL
from test_gpu_parity import assert_p_close, totals_of

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    import kmdiff_amd as K
    assert K.device_count() >= 1, "no GPU: the HIP path has no CPU fallback"
    return K


def make_streams(rng, universe, S, presence, count_hi=300, hi=None, empty=(), ones_in=()):
    streams = []
    for s in range(S):
        pick = rng.random(len(universe)) < (presence[s] if hasattr(presence, "__len__") else presence)
        if s in empty:
            pick[:] = False
        if len(ones_in):
            pick[-1] = s in ones_in
        cnt = rng.integers(1, count_hi, int(pick.sum())).astype(np.uint32)
        streams.append((universe[pick], cnt) if hi is None else (universe[pick], cnt, hi[pick]))
    return streams


def run_fused(K, oracle, streams, nc, thr, two=False, lf_n=10000):
    """device: streams -> survivors; oracle: merge, then diff_partition on the merged matrix"""
    S = len(streams)
    if two:
        want, wlo, whi = oracle.merge_partition2(streams)
    else:
        want, wlo = oracle.merge_partition(streams)
        whi = None
    tcs, tks = totals_of(want, nc)
    ref = oracle.diff_partition(want, OL.LAYOUT_ROWS, nc, S - nc, int(tcs.sum()), int(tks.sum()), oracle.lf_table(lf_n), thr)
    model = K.PoissonLikelihood(nc, S - nc, tcs, tks, lf_n)
    acc = K.SurvivorAccumulator(max(want.shape[0], 1), kmer_limbs=2 if two else 1)
    obs = K.diff_observer(model, acc, thr)
    ss = K.StreamSet(streams)
    n_rows = K.merge_filter(ss, obs)
    n = acc.finish(by_kmer=True)
    got = acc.get()
    c = acc.read_counters()
    assert n_rows == want.shape[0] and int(c[0]) == want.shape[0]
    assert (int(c[0]), int(c[1]), int(c[2]), int(c[3])) == ref["counters"]
    rr = ref["row"].astype(np.int64)
    assert n == len(rr)
    assert got["kmer_lo"].tolist() == wlo[rr].tolist()
    if two:
        assert got["kmer_hi"].tolist() == whi[rr].tolist()
    else:
        assert got["row"].tolist() == wlo[rr].tolist()
    assert got["sign"].tolist() == ref["sign"].tolist()
    assert got["mean_control"].tolist() == ref["mean_control"].tolist()
    assert got["mean_case"].tolist() == ref["mean_case"].tolist()
    assert_p_close(got["pvalue"], ref["pvalue"])
    if n:
        counts = K.gather_counts_streams(ss, None, n, acc.bufs["kmer_lo"], acc.bufs["kmer_hi"] if two else None)
        assert (counts == want[rr].astype(np.float64)).all()
    return want, ref


@pytest.mark.parametrize("S,nc,presence", [(9, 4, 0.5), (40, 20, 0.65), (40, 20, 0.05), (33, 1, 0.3), (64, 63, 0.2), (105, 50, 0.4),
                                           (200, 100, 0.1), (256, 128, 0.02), (300, 100, 0.03), (2, 1, 0.9), (3, 2, 1.0)])
def test_merge_filter_equals_merge_then_diff(K, oracle, S, nc, presence):
    """Random k-mers over the 62-bit range, every sample present with its own probability, a sample
    without k-mers, the all-ones k-mer (the table's empty marker) in a third of the samples."""
    rng = np.random.default_rng(7000 + S)
    universe = np.unique(np.concatenate([rng.integers(0, 1 << 62, int(150_000 / (S * presence)) + 3000, dtype=np.uint64),
                                         np.array([0, 2 ** 64 - 1], dtype=np.uint64)]))
    pres = np.clip(presence * rng.uniform(0.5, 1.5, S), 0.01, 1.0)
    streams = make_streams(rng, universe, S, pres, empty=(2,) if S > 3 else (), ones_in=tuple(range(0, S, 3)))
    want, ref = run_fused(K, oracle, streams, nc, 0.01)
    assert len(ref["row"]) > 20 or S <= 2


@pytest.mark.parametrize("thr", [1e-2, 1e-4, 5e-7, 1e-12])
@pytest.mark.parametrize("scale", [1, 3])
def test_merge_filter_every_pair_of_small_sums(K, oracle, thr, scale):
    """Every (control sum, case sum) in [0, 100)^2 is a row: the candidate cut of each threshold runs through the middle of
    them, one-sided rows and balanced ones -- where the pre-filter's two stages (the chi-square bound in double precision,
    the likelihood ratio in single precision, kmd_tilemerge.hip row_may_pass_kl) must let every candidate through."""
    a, b = np.meshgrid(np.arange(100, dtype=np.uint32), np.arange(100, dtype=np.uint32), indexing="ij")
    a, b = a.ravel()[1:], (b.ravel()[1:] * np.uint32(scale)).astype(np.uint32)
    keys = np.sort(np.random.default_rng(99).choice(1 << 40, len(a), replace=False).astype(np.uint64))
    streams = [(keys[a > 0], a[a > 0]), (keys[b > 0], b[b > 0])]
    want, ref = run_fused(K, oracle, streams, 1, thr)
    assert 50 < len(ref["row"]) < len(a) - 50
    import os
    os.environ["KMD_PREFILTER_KL_OFF"] = "1"                   # the same without the second stage: the same survivors
    try:
        run_fused(K, oracle, streams, 1, thr)
    finally:
        del os.environ["KMD_PREFILTER_KL_OFF"]


@pytest.mark.parametrize("presence", [0.6, 0.1])
@pytest.mark.parametrize("two", [False, True])
def test_merge_filter_counts_too_large_for_32_bit_sums(K, oracle, two, presence):
    """The table keeps a k-mer's two sums in 32 bits; a tile that meets a count of 2^22 or more (1024 samples of
    smaller counts cannot overflow) says so and is redone with 64-bit sums.  Here: a few k-mers whose counts add up
    past 2^32 within the controls, the rest ordinary; one- and two-limb k-mers; rows of many records and of few (for
    which the plan takes the 4096-slot table -- which, with two limbs AND 64-bit sums, would not fit the LDS: those
    tiles are redone on the smaller one)."""
    rng = np.random.default_rng(4242 + two)
    S, nc = 12, 6
    universe = np.unique(rng.integers(0, 1 << 62, 60_000, dtype=np.uint64))
    hi = np.sort(rng.integers(0, 1 << 40, len(universe), dtype=np.uint64)) if two else None
    if two:
        order = np.lexsort((universe, hi)); universe, hi = universe[order], hi[order]
    streams = make_streams(rng, universe, S, presence, hi=hi)
    big = set(universe[rng.choice(len(universe), 40 if presence > 0.5 else 600, replace=False)].tolist())
    for s in range(S):
        km, cnt = streams[s][0], streams[s][1]
        hit = np.isin(km, np.fromiter(big, dtype=np.uint64))
        cnt = cnt.copy()
        cnt[hit] = np.uint32(3_000_000_000) if s < nc else np.uint32(5_000_000)
        streams[s] = (km, cnt) + tuple(streams[s][2:])
    want, ref = run_fused(K, oracle, streams, nc, 0.01, two=two)
    sums_c = want[:, :nc].sum(axis=1, dtype=np.uint64)
    assert (sums_c > 2 ** 32).sum() >= (30 if presence > 0.5 else 10)      # the case is really there


def test_merge_sums_rows_are_compact_and_exact(K, oracle):
    """kmd_merge_sums: exactly one entry per distinct k-mer, no holes; KMD_E_OVERFLOW reports the rows needed."""
    rng = np.random.default_rng(99)
    S, nc = 24, 11
    universe = np.unique(rng.integers(0, 1 << 60, 90_000, dtype=np.uint64))
    streams = make_streams(rng, universe, S, 0.3, count_hi=2 ** 32 - 1)
    want, kmers = oracle.merge_partition(streams)
    sums = K.merge_sums(streams, nc)
    km, sc, sk, _ = sums.to_host()
    assert sums.n_rows == want.shape[0] == len(km)
    order = np.argsort(km, kind="stable")
    assert (km[order] == kmers).all()
    assert (sc[order] == want[:, :nc].sum(axis=1, dtype=np.uint64)).all()
    assert (sk[order] == want[:, nc:].sum(axis=1, dtype=np.uint64)).all()
    with pytest.raises(K.KmdError):
        K.merge_sums(streams, nc, row_capacity=want.shape[0] - 5)


@pytest.mark.parametrize("kind", ["gap", "dense", "runs", "dup-heavy"])
def test_merge_filter_clustered_keys(K, oracle, kind):
    """Key distributions the data splitters (every r-th key of the LONGEST stream) cannot balance: the
    tiles over capacity are cut again on the device, level by level -- no fallback, same results.
      gap      : the other samples hold 60 k consecutive k-mers where the longest one has none
      dense    : everything inside a few thousand consecutive integers (slices end at single k-mers)
      runs     : 200 dense runs scattered over the range
      dup-heavy: 40 samples, every k-mer in every sample"""
    rng = np.random.default_rng({"gap": 1, "dense": 2, "runs": 3, "dup-heavy": 4}[kind])
    if kind == "gap":
        S, nc = 12, 6
        base = np.unique(rng.integers(0, 1 << 62, 60_000, dtype=np.uint64))
        gap = (np.uint64(1) << np.uint64(61)) + np.arange(60_000, dtype=np.uint64) * np.uint64(5)
        base = base[(base < gap[0]) | (base > gap[-1])]
        streams = []
        for s in range(S):
            if s == 0:
                km = base                                           # the longest stream: nothing in the gap
            else:
                km = np.unique(np.concatenate([base[rng.random(len(base)) < 0.3], gap[rng.random(len(gap)) < 0.8]]))
            streams.append((km, rng.integers(1, 200 if s < nc else 30, len(km)).astype(np.uint32)))
    elif kind == "dense":
        S, nc = 20, 10
        universe = np.uint64(123456789) + np.arange(6000, dtype=np.uint64)
        streams = make_streams(rng, universe, S, 0.9)
    elif kind == "runs":
        S, nc = 16, 8
        starts = np.sort(rng.integers(0, 1 << 61, 200, dtype=np.uint64))
        universe = np.unique((starts[:, None] + np.arange(400, dtype=np.uint64)[None, :] * np.uint64(3)).ravel())
        streams = make_streams(rng, universe, S, rng.uniform(0.2, 0.9, S))
    else:
        S, nc = 40, 20
        universe = np.unique(rng.integers(0, 1 << 62, 9000, dtype=np.uint64))
        streams = make_streams(rng, universe, S, 1.0)
    run_fused(K, oracle, streams, nc, 1e-3)


@pytest.mark.parametrize("S,nc,hi_bits", [(6, 3, 6), (40, 20, 62), (100, 50, 30)])
def test_merge_filter_two_limb_kmers(K, oracle, S, nc, hi_bits):
    """32 < k <= 64: (hi, lo) pairs.  Families of k-mers that share a LOW limb under different high limbs
    (one table slot would take both: the tile must be cut until they part), low limbs of all ones, the
    all-ones k-mer."""
    rng = np.random.default_rng(8000 + S)
    n = 60_000
    hi = rng.integers(0, 1 << hi_bits, n, dtype=np.uint64)
    lo = rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64)
    lo[:2000] = lo[2000:4000]                                        # same low limb ...
    hi[:2000] = hi[2000:4000] + np.uint64(1)                         # ... next high limb: neighbours in the order
    lo[4000:4020] = np.uint64(2 ** 64 - 1)                           # low limb = the empty marker, assorted high limbs
    hi[4019] = np.uint64((1 << hi_bits) - 1)                          # the largest k-mer there is at this width
    order = np.lexsort((lo, hi))
    hi, lo = hi[order], lo[order]
    keep = np.ones(n, dtype=bool)
    keep[1:] = (hi[1:] != hi[:-1]) | (lo[1:] != lo[:-1])
    hi, lo = hi[keep], lo[keep]
    streams = make_streams(rng, lo, S, rng.uniform(0.1, 0.6, S), hi=hi)
    run_fused(K, oracle, streams, nc, 1e-2, two=True)
    # the rows themselves
    want, wlo, whi = oracle.merge_partition2(streams)
    sums = K.merge_sums(streams, nc)
    km, sc, sk, _, kh = sums.to_host()
    order = np.lexsort((km, kh))
    assert len(km) == want.shape[0]
    assert (km[order] == wlo).all() and (kh[order] == whi).all()
    assert (sc[order] == want[:, :nc].sum(axis=1, dtype=np.uint64)).all()
    assert (sk[order] == want[:, nc:].sum(axis=1, dtype=np.uint64)).all()


def test_merge_filter_tiny_and_empty_inputs(K, oracle):
    tcs, tks = np.array([10], np.uint64), np.array([10, 10], np.uint64)
    model = K.PoissonLikelihood(1, 2, tcs, tks, 100)
    acc = K.SurvivorAccumulator(16)
    obs = K.diff_observer(model, acc, 1.0)
    assert K.merge_filter([(np.zeros(0, np.uint64), np.zeros(0, np.uint32))] * 3, obs) == 0
    streams = [(np.array([5, 9], np.uint64), np.array([2, 3], np.uint32)), (np.array([9], np.uint64), np.array([7], np.uint32)),
               (np.zeros(0, np.uint64), np.zeros(0, np.uint32))]
    assert K.merge_filter(streams, obs) == 2
    assert acc.finish(by_kmer=True) == 2                             # threshold 1: every row survives
    got = acc.get()
    assert got["kmer_lo"].tolist() == [5, 9] and got["mean_case"].tolist() == [0.0, 7.0]


@pytest.mark.parametrize("thr", [1.0, 0.5, 1e-30, 0.0])
def test_merge_filter_thresholds_without_prefilter(K, oracle, thr):
    """Thresholds at which the chi-square pre-filter is off (every row is evaluated exactly; the
    workgroup queue overflows into immediate evaluation) and at which nothing survives."""
    rng = np.random.default_rng(5)
    S, nc = 10, 5
    universe = np.unique(rng.integers(0, 1 << 62, 30_000, dtype=np.uint64))
    streams = make_streams(rng, universe, S, 0.5)
    run_fused(K, oracle, streams, nc, thr)


@pytest.mark.parametrize("two", [False, True])
def test_pca_sampling_from_the_streams_equals_sampling_the_matrix(K, oracle, two):
    """kmd_pca_sample_streams (the fused path has no matrix): the same sampled rows, in the same order, hence
    bit for bit the same Gram matrix as kmd_pca_sample over the merged matrix."""
    rng = np.random.default_rng(21 + two)
    S = 23
    n = 40_000
    lo = np.unique(rng.integers(0, 1 << 62, n, dtype=np.uint64))
    hi = np.sort(rng.integers(0, 50, len(lo), dtype=np.uint64)) if two else None
    if two:
        order = np.lexsort((lo, hi)); lo, hi = lo[order], hi[order]
    streams = make_streams(rng, lo, S, rng.uniform(0.2, 0.8, S), hi=hi)
    mat = K.merge_partition(streams, count_bytes=4, layout=K.LAYOUT_TILED)
    a = K.PopulationPCA(S, 0.02, seed=7)
    a.sample(mat)
    b = K.PopulationPCA(S, 0.02, seed=7)
    b.sample_streams(streams)
    assert a.count() == b.count() and a.count() > 200
    assert (a.gram() == b.gram()).all()


def test_merge_filter_at_the_sample_limit(K, oracle):
    """1024 samples is what the tile's segment tables hold (more: refused, not mangled); runs of a handful of
    records each -- the sub-group path."""
    rng = np.random.default_rng(1024)
    S, nc = 1024, 500
    universe = np.unique(rng.integers(0, 1 << 62, 9_000, dtype=np.uint64))
    streams = make_streams(rng, universe, S, rng.uniform(0.02, 0.3, S), count_hi=40, empty=(7, 1000))
    run_fused(K, oracle, streams, nc, 1e-3)
    model = K.PoissonLikelihood(513, 512, np.ones(513, np.uint64), np.ones(512, np.uint64), 100)
    acc = K.SurvivorAccumulator(16)
    with pytest.raises(K.KmdError):
        K.merge_filter(streams + [streams[0]], K.diff_observer(model, acc, 0.5))


@pytest.mark.parametrize("S,n", [(16, 1_300_000), (20, 1_300_000), (23, 2_000_000), (20, 1_150_000)])
def test_disjoint_kmers_between_one_and_two_grids_of_tiles(K, S, n):
    """Every record a row of its own (disjoint k-mers, rho = 1), at the sizes where the plan asks for a little more
    than one persistent grid's worth of tiles: the plan used to round the tile count up to the next multiple of the
    grid (halving the tiles), pick sub-groups of lanes per run from the halved tiles -- and the host had not launched
    that instantiation: zero rows, KMD_OK.  Rows must equal records, every count accounted for."""
    rng = np.random.default_rng(S * 1000 + n % 997)
    keys = np.unique(rng.integers(0, 1 << 62, n + n // 50, dtype=np.uint64))[:n]
    owner = rng.integers(0, S, len(keys))
    streams = []
    for s in range(S):
        km = keys[owner == s]
        streams.append((km, rng.integers(1, 50, len(km)).astype(np.uint32)))
    nc = S // 2
    sums = K.merge_sums(streams, nc, row_capacity=len(keys))
    km, sc, sk, _ = sums.to_host()
    assert sums.n_rows == len(keys)
    assert np.array_equal(np.sort(km), keys)
    want_c = sum(int(streams[s][1].sum(dtype=np.uint64)) for s in range(nc))
    want_k = sum(int(streams[s][1].sum(dtype=np.uint64)) for s in range(nc, S))
    assert int(sc.sum(dtype=np.uint64)) == want_c and int(sk.sum(dtype=np.uint64)) == want_k
    tot = np.array([int(t[1].sum(dtype=np.uint64)) for t in streams], dtype=np.uint64)
    model = K.PoissonLikelihood(nc, S - nc, tot[:nc], tot[nc:], 10000)
    acc = K.SurvivorAccumulator(1 << 16)
    assert K.merge_filter(K.StreamSet(streams), K.diff_observer(model, acc, 1e-9)) == len(keys)
    assert int(acc.read_counters()[0]) == len(keys)


def test_merge_filter_partitions_in_flight(K, oracle):
    """Three different partitions at once, each from a host thread and on a stream of its own (scratch per call,
    near-threshold list per stream): the same survivors as one after the other."""
    import ctypes as C
    import threading
    lib = K._native.lib()
    S, nc = 16, 8
    jobs = []
    for j in range(3):
        rng = np.random.default_rng(300 + j)
        universe = np.unique(rng.integers(0, 1 << 62, 120_000 + 30_000 * j, dtype=np.uint64))
        streams = make_streams(rng, universe, S, 0.55)
        want, wlo = oracle.merge_partition(streams)
        tcs, tks = totals_of(want, nc)
        ref = oracle.diff_partition(want, OL.LAYOUT_ROWS, nc, S - nc, int(tcs.sum()), int(tks.sum()), oracle.lf_table(10000), 0.01)
        model = K.PoissonLikelihood(nc, S - nc, tcs, tks, 10000)
        acc = K.SurvivorAccumulator(4 * want.shape[0])
        st = C.c_void_p()
        assert lib.kmd_stream_create(C.byref(st)) == 0
        jobs.append({"ss": K.StreamSet(streams), "obs": K.diff_observer(model, acc, 0.01), "acc": acc, "st": st,
                     "want_rows": want.shape[0], "want_kmers": wlo[ref["row"].astype(np.int64)].tolist(), "ref": ref, "rows": []})

    # (the workers leave as they finish.  Round 4 saw this test abort the process 3 times in ~45 suite runs and held the
    # threads at a barrier on a guess; the cause was elsewhere -- the first filter launch on a fresh stream could run
    # before its near-threshold list was initialised, see test_first_filter_launch_on_fresh_streams)
    def work(job, reps):
        for _ in range(reps):
            job["rows"].append(K.merge_filter(job["ss"], job["obs"], stream=job["st"]))
    th = [threading.Thread(target=work, args=(job, 4)) for job in jobs]
    [t.start() for t in th]
    [t.join() for t in th]
    for job in jobs:
        assert lib.kmd_stream_sync(job["st"]) == 0
        assert job["rows"] == [job["want_rows"]] * 4
        n = job["acc"].finish(by_kmer=True)
        got = job["acc"].get()
        assert n == 4 * len(job["want_kmers"])
        assert got["kmer_lo"].tolist() == sorted(job["want_kmers"] * 4)
        c = job["acc"].read_counters()
        assert (int(c[0]), int(c[1]), int(c[2]), int(c[3])) == tuple(4 * v for v in job["ref"]["counters"])
        assert lib.kmd_stream_destroy(job["st"]) == 0


def _stress(args, env_extra, timeout=240):
    """tools/stress_inflight.py in a child process (a GPU queue error aborts the process that owns the queue)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("KMD_ABORT_TRACE", None)                     # (conftest's descriptor is not the child's)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(root, "tools", "stress_inflight.py")] + args, env=env, cwd=root,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_first_filter_launch_on_fresh_streams(K):
    """The abort of round 4, root cause and regression.  A stream's near-threshold list is allocated at the
    first filter launch on it; round 4 zeroed its count with hipMemset -- the NULL stream, against which the callers'
    non-blocking streams are not ordered -- so k_resolve_near on a new stream could read whatever the allocation held,
    'resolve' thousands of garbage entries and die of a memory aperture violation (the HSA queue error handler then
    aborts the process).  It took fresh streams + memory that had been used before (kmd_release_cache between test
    modules) + other host threads keeping the GPU busy: 1 run in 3 of `stress_inflight.py --threads 6 --new-streams
    --release`, 1 in 15 of the suite.  KMD_TEST_NEAR_INIT=1 fills every fresh list with ones before it is initialised:
    an initialisation that is not ordered before the kernels then fails EVERY time (mode 2 = round 4's: 6 of 6
    processes died on the box; KMD_TEST_DEMONSTRATE_ABORT=1 shows it here)."""
    import os
    r = _stress(["--threads", "6", "--new-streams", "--iters", "10", "--reps", "3"], {"KMD_TEST_NEAR_INIT": "1"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    r = _stress(["--threads", "6", "--new-streams", "--release", "--iters", "10", "--reps", "3"], {})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    if os.environ.get("KMD_TEST_DEMONSTRATE_ABORT"):
        r = _stress(["--threads", "6", "--new-streams", "--iters", "10", "--reps", "3"], {"KMD_TEST_NEAR_INIT": "2"})
        # (either of the runtime's two last words, depending on where the garbage address falls)
        assert r.returncode != 0 and ("APERTURE_VIOLATION" in r.stderr or "Memory access fault by GPU" in r.stderr), (r.returncode, r.stderr[-2000:])


def test_merge_filter_batch_equals_single_calls(K, oracle):
    """kmd_merge_filter_batch: nine partitions of different sizes and kinds (an empty one, one whose tiles must be cut
    again, two-limb k-mers apart) through the batch entry point -- the same survivors, bit for bit, and the same
    counters and row counts as nine single calls; partitions that share a sink add up."""
    rng = np.random.default_rng(515)
    S, nc = 12, 5
    sets, want = [], []
    for j in range(9):
        if j == 3:
            streams = [(np.zeros(0, np.uint64), np.zeros(0, np.uint32)) for _ in range(S)]
        elif j == 5:                                             # dense clusters: tiles are cut again (the slow way inside the batch)
            starts = np.sort(rng.integers(0, 1 << 61, 40, dtype=np.uint64))
            universe = np.unique((starts[:, None] + np.arange(3000, dtype=np.uint64)[None, :]).ravel())
            streams = make_streams(rng, universe, S, rng.uniform(0.3, 0.9, S))
        else:
            universe = np.unique(rng.integers(0, 1 << 62, 20_000 + 15_000 * j, dtype=np.uint64))
            streams = make_streams(rng, universe, S, rng.uniform(0.2, 0.8, S))
        sets.append(K.StreamSet(streams))
        want.append(streams)
    tot = np.zeros(S, dtype=np.uint64)
    for streams in want:
        tot += np.array([int(t[1].sum(dtype=np.uint64)) for t in streams], dtype=np.uint64)
    model = K.PoissonLikelihood(nc, S - nc, tot[:nc], tot[nc:], 10000)
    thr = 1e-6
    single = []
    for ss in sets:
        acc = K.SurvivorAccumulator(1 << 18)
        rows = K.merge_filter(ss, K.diff_observer(model, acc, thr)) if ss.total else 0
        n = acc.finish(by_kmer=True)
        single.append((rows, n, acc.get(), [int(x) for x in acc.read_counters()[:4]]))
    accs = [K.SurvivorAccumulator(1 << 18) for _ in sets]
    rows_b = K.merge_filter_batch(sets, [K.diff_observer(model, a, thr) for a in accs])
    assert sum(s[1] for s in single) > 200
    for j, a in enumerate(accs):
        n = a.finish(by_kmer=True)
        got = a.get()
        assert rows_b[j] == single[j][0] and n == single[j][1], j
        assert [int(x) for x in a.read_counters()[:4]] == single[j][3], j
        for key in ("kmer_lo", "pvalue", "sign", "mean_control", "mean_case"):
            assert got[key].tolist() == single[j][2][key].tolist(), (j, key)
    # one sink for all of them
    shared = K.SurvivorAccumulator(1 << 21)
    obs = K.diff_observer(model, shared, thr)
    rows_s = K.merge_filter_batch(sets, [obs] * len(sets))
    n = shared.finish(by_kmer=True)
    assert rows_s == rows_b and n == sum(s[1] for s in single)
    assert [int(x) for x in shared.read_counters()[:4]] == [sum(s[3][i] for s in single) for i in range(4)]
    assert sorted(shared.get()["kmer_lo"].tolist()) == sorted(k for s in single for k in s[2]["kmer_lo"].tolist())
    # bad input in one partition: the batch says so, the others are done all the same
    bad = K.StreamSet([(np.array([5, 3], dtype=np.uint64), np.array([1, 1], dtype=np.uint32))] + [(np.zeros(0, np.uint64), np.zeros(0, np.uint32))] * (S - 1))
    bad.offs = bad.offs.copy(); bad.offs[1] = 3; bad.offs[2:] = 2                      # offsets not ascending
    acc2 = [K.SurvivorAccumulator(1 << 18) for _ in range(3)]
    with pytest.raises(K.KmdError):
        K.merge_filter_batch([sets[0], bad, sets[1]], [K.diff_observer(model, a, thr) for a in acc2])
    assert acc2[0].finish(by_kmer=True) == single[0][1] and acc2[2].finish(by_kmer=True) == single[1][1]


def test_merge_filter_batch_of_partitions_whose_plans_differ(K, oracle):
    """A batch launches, for the partitions behind the first ones, only the instantiation of the merge kernel the plans so
    far took (whole waves or sub-groups per run; the 2048- or the 4096-slot table): a guess.  Here it is wrong over and
    over -- 80 samples, partitions of long runs, of short runs, of rows of two records, interleaved -- and nothing of a
    partition whose merge did not run may reach the sink (the candidates' kernels are gated on it): the same survivors
    and counters as single calls, which guess too (the shape of the last call on the device)."""
    rng = np.random.default_rng(8181)
    S, nc = 80, 40
    kinds = [0.9, 0.9, 0.9, 0.9, 0.9, 0.9, 0.9, 0.03, 0.9, 0.03, 0.03, 0.9, 0.025, 0.9]
    sets, host, tot = [], [], np.zeros(S, dtype=np.uint64)
    for j, pres in enumerate(kinds):
        universe = np.unique(rng.integers(0, 1 << 62, int(60_000 / (S * pres)) + 2000, dtype=np.uint64))
        streams = make_streams(rng, universe, S, np.clip(pres * rng.uniform(0.7, 1.3, S), 0.005, 1.0))
        sets.append(K.StreamSet(streams))
        host.append(streams)
        tot += np.array([int(t[1].sum(dtype=np.uint64)) for t in streams], dtype=np.uint64)
    model = K.PoissonLikelihood(nc, S - nc, tot[:nc], tot[nc:], 10000)
    thr = 1e-3
    single = []
    for ss in sets:
        acc = K.SurvivorAccumulator(1 << 18)
        rows = K.merge_filter(ss, K.diff_observer(model, acc, thr))
        n = acc.finish(by_kmer=True)
        single.append((rows, n, acc.get(), [int(x) for x in acc.read_counters()[:4]]))
    assert sum(s[1] for s in single) > 100
    for _ in range(2):
        accs = [K.SurvivorAccumulator(1 << 18) for _ in sets]
        rows_b = K.merge_filter_batch(sets, [K.diff_observer(model, a, thr) for a in accs])
        for j, a in enumerate(accs):
            n = a.finish(by_kmer=True)
            got = a.get()
            assert rows_b[j] == single[j][0] and n == single[j][1], j
            assert [int(x) for x in a.read_counters()[:4]] == single[j][3], j
            for key in ("kmer_lo", "pvalue", "sign", "mean_control", "mean_case"):
                assert got[key].tolist() == single[j][2][key].tolist(), (j, key)
    # single calls against the oracle, the table shape changing from call to call
    for j in (0, 7, 1, 12):
        run_fused(K, oracle, host[j], nc, thr)


@pytest.mark.parametrize("env", [{"KMD_TILE_CAND_CAP": "500"}, {"KMD_TILE_FILL": "60000", "KMD_TILE_LOAD_PCT": "50"},
                                 {"KMD_TILE_G": "3"}, {"KMD_TILE_SUM64": "1"}, {"KMD_TILE_SHAPE": "1024x4096"}])
def test_merge_filter_forced_paths(K, oracle, env):
    """The ways round that ordinary inputs rarely take, forced through the library's development switches: a
    candidate list that overflows (the evaluation enqueued behind the merge must hold still, the merge runs again
    with the size it reported), tiles far too large for the table (every one gives up and is cut again), the
    sub-group path on long runs, 64-bit sums from the start, the other table shape."""
    import os
    rng = np.random.default_rng(77)
    S, nc = 10, 5
    universe = np.unique(rng.integers(0, 1 << 62, 50_000, dtype=np.uint64))
    streams = make_streams(rng, universe, S, 0.6)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        run_fused(K, oracle, streams, nc, 0.3)                      # ~ every third row survives: a long candidate list
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_fused_equals_matrix_path_at_bench_size(K):
    """The bench's `pipeline` partition (20v20, 4 M rows, 104 M records): the fused merge + test and the merge into a
    count matrix followed by the test on it give the same survivors -- k-mers, p-values, signs bit for bit (one code
    evaluates both) -- and the same counters; every record is accounted for (sum of the rows' two sums = sum of the
    counts handed in)."""
    rows, nc, nk = 4_000_000, 20, 20
    S = nc + nk
    mat = K.synth_matrix(0x6B6D64696666, 0, rows, nc, nk, 4, K.LAYOUT_ROWS)
    host, lo = mat.to_host(), mat.kmers_to_host()[0]
    del mat
    streams = [(lo[host[:, s] > 0], host[host[:, s] > 0, s]) for s in range(S)]
    tot = host.sum(axis=0, dtype=np.uint64)
    model = K.PoissonLikelihood(nc, nk, tot[:nc], tot[nc:], 10000)
    thr = 1e-4
    ss = K.StreamSet(streams)
    a = K.SurvivorAccumulator(rows // 50)
    assert K.merge_filter(ss, K.diff_observer(model, a, thr)) == rows
    na = a.finish(by_kmer=True)
    ga, ca = a.get(), a.read_counters()
    out = K.merge_partition(streams, count_bytes=4, layout=K.LAYOUT_TILED)
    assert out.n_rows == rows
    b = K.SurvivorAccumulator(rows // 50)
    K.diff_observer(model, b, thr).process(out)
    nb = b.finish()
    gb, cb = b.get(), b.read_counters()
    assert na == nb > 1000 and [int(x) for x in ca[:4]] == [int(x) for x in cb[:4]] and int(ca[6]) == int(cb[6]) == 0
    assert ga["kmer_lo"].tolist() == gb["kmer_lo"].tolist()          # (rows ascend with their k-mers)
    assert ga["pvalue"].tolist() == gb["pvalue"].tolist() and ga["sign"].tolist() == gb["sign"].tolist()
    assert ga["mean_case"].tolist() == gb["mean_case"].tolist() and ga["mean_control"].tolist() == gb["mean_control"].tolist()
    sums = K.merge_sums(ss, nc, row_capacity=rows)
    km, sc, sk, _ = sums.to_host()
    assert sums.n_rows == rows and int(sc.sum(dtype=np.uint64)) == int(tot[:nc].sum()) and int(sk.sum(dtype=np.uint64)) == int(tot[nc:].sum())
    assert np.array_equal(np.sort(km), lo)


def test_merge_filter_random_small_cases(K, oracle):
    """Forty seeded small partitions of every shape at once: 2..14 samples, 0..3000 k-mers spread over the whole range
    or packed into a few thousand consecutive integers, presence per sample from almost none to all, empty samples,
    counts of 3e9 here and there, thresholds from 1 to 1e-6."""
    rng = np.random.default_rng(20261002)
    for case in range(40):
        S = int(rng.integers(2, 15))
        nc = int(rng.integers(1, S))
        n = int(rng.integers(0, 3000))
        if rng.random() < 0.4:
            base = int(rng.integers(0, 1 << 60))
            universe = np.unique(np.uint64(base) + rng.integers(0, 4 * n + 8, n).astype(np.uint64))
        else:
            universe = np.unique(rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64))
        if case % 7 == 0 and len(universe):
            universe = np.unique(np.concatenate([universe, np.array([0, 2 ** 64 - 1], dtype=np.uint64)]))
        pres = rng.uniform(0.02, 1.0, S)
        empty = tuple(int(x) for x in rng.choice(S, size=int(rng.integers(0, max(1, S // 3))), replace=False)) if S > 2 else ()
        # (count sums stay below ~3e4 or go past 2^31: the oracle follows the reference's O(sum) loop for every sum
        # between the table's end and 2^31, seconds per row at 1e9)
        streams = make_streams(rng, universe, S, pres, count_hi=int(rng.integers(2, 2000)), empty=empty)
        if sum(len(t[0]) for t in streams) == 0:
            continue
        if case % 5 == 0:
            for s in range(S):
                km, cnt = streams[s]
                if len(cnt):
                    cnt = cnt.copy()
                    cnt[rng.integers(0, len(cnt), 3)] = np.uint32(3_000_000_000)
                    streams[s] = (km, cnt)
        want, _ = oracle.merge_partition(streams)
        tcs, tks = totals_of(want, nc)
        if int(tcs.sum()) == 0 or int(tks.sum()) == 0:
            continue                                                 # (the model needs counts on both sides)
        run_fused(K, oracle, streams, nc, float(rng.choice([1.0, 0.2, 1e-2, 1e-6])), lf_n=int(rng.choice([10000, 200])))

