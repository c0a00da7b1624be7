"""bench.py's contract (one JSON line, the keys the driver reads) at N=1, and the N=2 code path --
partition sharding, counter all-reduce, sharded BH/Holm -- run as two ranks folded onto the one
GPU of the test box with gloo as the wire (KMD_BENCH_OVERSUBSCRIBE / KMD_BENCH_BACKEND; RCCL itself
is exercised by the driver's multi-GPU run)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROWS = 2_000_000


def free_port():
    """a TCP port nobody holds right now (the rendezvous of a torchrun started by a test: fixed numbers collide on shared boxes)"""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def last_json(out):
    lines = [l for l in out.strip().split("\n") if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_single_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rows", str(ROWS), "--steps", "4", "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = last_json(r.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["vs_baseline"] is None
    assert d["scaling"] == "weak" and d["unit"] == "k-mers/s" and "workload" in d["config"]
    assert d["config"]["counters"]["total"] == ROWS * 4
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]
    assert d["rccl_ranks"] == 1 and d["per_rank"]["ms_per_step_max"] == d["ms_per_step"]
    assert d["transport"] is None and len(d["per_rank"]["kernel_ms"]) == 1 and 0 < d["scaling_efficiency"] <= 1.0
    pl = d["pipeline"]
    for leg in (pl, pl["overlapped"], pl["batched"]):
        rf = leg["roofline"]
        assert rf["bound"] == "hbm" and 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert pl["batched"]["partitions"] == 12 and pl["batched"]["ms_per_partition"] > 0
    # the streams -> survivors leg runs on a WHOLE configs[2] partition (what the reference merges per task) and says so
    assert pl["rows"] == 39_062_500 and pl["records"] > 900_000_000 and "configs[2] partition" in pl["config"]["workload"]
    assert pl["small"]["rows"] == 4_000_000 and 0 < pl["small"]["roofline"]["frac"] < 1
    # the mixed partition (every second row in one or two samples, the others in 95 % of them): its own roofline blocks
    sp = pl["sparse"]
    assert sp["rows"] == 39_062_500 and 600_000_000 < sp["records"] < 900_000_000 and "MIXED" in sp["config"]["workload"]
    assert 0 < sp["roofline"]["frac"] < 1 and 0 < sp["batched"]["roofline"]["frac"] < 1
    # the path the command takes, link included: within reach of the link's ceiling, far below the resident rate, never `value`
    fi = pl["feed_inclusive"]
    assert "error" not in fi, fi
    assert fi["rows"] == 39_062_500 and 3.5 < fi["bytes_per_record"] < 7 and fi["n_sig"] == pl["n_sig"]
    assert 0.5 < fi["frac_of_link_ceiling"] <= 1.05 and fi["kmers_per_s"] < 0.2 * pl["kmers_per_s"] and d["value"] > 10 * fi["kmers_per_s"]


def test_bench_e2e_leg_with_the_cpu_pipeline_beside_it():
    """`bench.py --e2e` (never `value`): `kmdiff-hip diff` on a fabricated run directory and, on the same files, the CPU doing the
    same job (oracle/cpu_pipeline: liblz4 decode + the oracle's merge + the oracle's test) -- here on a small directory."""
    env = dict(os.environ, KMD_BENCH_E2E_PARTS="2", KMD_BENCH_E2E_ROWS="300000")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rows", str(ROWS), "--steps", "2", "--warmup", "1", "--no-pipeline", "--no-cpu-baseline", "--e2e"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    e = last_json(r.stdout)["e2e"]
    assert "error" not in e, e
    assert e["run_dir"]["partitions"] == 2 and e["run_dir"]["rows_per_partition"] == 300000
    assert e["kmdiff_hip_diff"]["rows_per_s"] > 0 and e["cpu_baseline_e2e"]["value"] > 0 and e["cpu_baseline_e2e"]["kind"] == "port"
    assert e["cpu_baseline_e2e"]["cores"] >= 1 and e["gpu_over_cpu"] > 1.0


@pytest.mark.parametrize("correction", ["bonferroni", "benjamini", "holm"])
def test_bench_two_ranks_on_one_gpu(correction):
    env = dict(os.environ, KMD_BENCH_OVERSUBSCRIBE="1", KMD_BENCH_BACKEND="gloo")
    port = free_port()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows",
                        str(ROWS), "--steps", "4", "--warmup", "1", "--correction", correction],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 2 and "cpu_baseline" not in d
    c = d["config"]["counters"]
    assert c["total"] == ROWS * 4 * 2 and c["n_sig"] == c["n_sig_control"] + c["n_sig_case"]
    assert 0 < c["kept_after_correction"] <= c["n_sig"]


@pytest.mark.parametrize("correction", ["benjamini", "holm"])
def test_bench_eight_ranks_folded_onto_this_gpu(correction):
    """The width the driver's scaling run ends at (`bench.py --gpus 8`), folded onto the GPU(s) of this box with gloo
    as the wire: eight processes, partition p on rank p % 8, the counters' all-reduce and the sharded BH / Holm walk
    over eight ranks' tails; one JSON line with the whole job's totals."""
    env = dict(os.environ, KMD_BENCH_OVERSUBSCRIBE="1", KMD_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    rows = 500_000
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--rows", str(rows), "--steps", "3", "--warmup", "1",
                        "--correction", correction], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 8 and d["rccl_ranks"] == 8 and d["backend"] == "gloo" and d["scaling"] == "weak"
    # the N > 1 line checks itself: every rank's own kernel time and step time, the wire it used, and what the barriers and
    # the exchange cost against the ranks' own kernels
    pr = d["per_rank"]
    assert len(pr["kernel_ms"]) == 8 and len(pr["ms_per_step"]) == 8 and all(x > 0 for x in pr["kernel_ms"])
    assert max(pr["kernel_ms"]) == pr["kernel_ms_max"] and min(pr["kernel_ms"]) == pr["kernel_ms_min"]
    assert d["transport"] == "torch" and d["transport_fallback"] is None and 0 < d["scaling_efficiency"] <= 1.0
    c = d["config"]["counters"]
    assert c["total"] == rows * 3 * 8 and c["n_sig"] == c["n_sig_control"] + c["n_sig_case"] and 0 < c["kept_after_correction"] <= c["n_sig"]
    assert abs(d["value"] - rows * 3 * 8 / (d["ms_per_step"] * 3e-3)) <= 1e-6 * d["value"]


def test_bench_a_rank_that_dies_ends_the_job():
    """A peer that dies before the exchange (KMD_BENCH_TEST_DIE_RANK): the launcher comes back with a non-zero status
    in bounded time -- torchrun ends the other ranks, and a collective nobody answers times out after
    KMD_BENCH_COLLECTIVE_TIMEOUT_S -- instead of the surviving rank waiting in its all-reduce."""
    import time
    env = dict(os.environ, KMD_BENCH_OVERSUBSCRIBE="1", KMD_BENCH_BACKEND="gloo", KMD_BENCH_TEST_DIE_RANK="1", KMD_BENCH_COLLECTIVE_TIMEOUT_S="60")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "500000", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode != 0 and time.time() - t0 < 300
    assert not [l for l in r.stdout.split("\n") if l.startswith("{")]            # no line from a job that lost a rank


def test_bench_rccl_transport_refuses_a_folded_run():
    """`--transport rccl` is the library's own RCCL communicator: one GPU per rank.  Folded onto one GPU (gloo as the
    job's backend) it says so and leaves with an error on every rank instead of timing anything."""
    env = dict(os.environ, KMD_BENCH_OVERSUBSCRIBE="1", KMD_BENCH_BACKEND="gloo", KMD_BENCH_COLLECTIVE_TIMEOUT_S="60")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "500000", "--steps", "2", "--warmup", "1",
                        "--transport", "rccl"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode != 0 and "--transport rccl needs one GPU per rank" in (r.stdout + r.stderr)
    assert not [l for l in r.stdout.split("\n") if l.startswith("{")]


def test_bench_two_gpus_over_the_librarys_own_rccl_communicator():
    """`bench.py --gpus 2 --transport rccl` on two real GPUs (skipped on a one-GPU box): the exchange over
    libkmdiff_hip_rccl.so's communicator instead of torch.distributed's."""
    import kmdiff_amd as K
    if K.device_count() < 2:
        pytest.skip("one GPU on this box")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                             "KMD_BENCH_OVERSUBSCRIBE", "KMD_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", str(ROWS), "--steps", "4",
                        "--warmup", "1", "--correction", "holm", "--transport", "rccl"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["transport"] == "rccl" and len(d["per_rank"]["kernel_ms"]) == 2


def test_two_ranks_through_rccl_on_this_box():
    """tests/nccl_probe2.py: kmd_correct_sharded over RCCL with two ranks -- torch.distributed's communicator and the
    library's own (libkmdiff_hip_rccl.so).  Two GPUs: must pass.  One GPU: both ranks are put on it, which RCCL may
    refuse; the outcome is recorded either way (DESIGN 7 quotes it)."""
    import kmdiff_amd as K
    fold = K.device_count() < 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if fold:
        env["KMD_PROBE_FOLD"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(free_port()), os.path.join(ROOT, "tests", "nccl_probe2.py")], capture_output=True, text=True, timeout=600,
                       cwd=ROOT, env=env)
    out = r.stdout + r.stderr
    note = os.path.join(ROOT, "gpurun_out", "nccl_two_ranks.txt")
    try:
        os.makedirs(os.path.dirname(note), exist_ok=True)
        open(note, "w").write("fold=%s rc=%d\n%s\n" % (fold, r.returncode, out[-6000:]))
    except OSError:
        pass
    if r.returncode == 0:
        assert "nccl probe2 ok: 2 ranks" in r.stdout
        return
    assert fold, out[-3000:]                                  # on two real GPUs it has to work
    reason = [l for l in out.split("\n") if "uplicate GPU" in l or "invalid usage" in l or "NCCL" in l or "RCCL" in l]
    pytest.skip("RCCL with two ranks on ONE GPU did not run here (rc %d): %s" % (r.returncode, (reason or ["?"])[0][:300]))


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (how the driver runs the N = 1 line): bench.py starts
    the two ranks itself, as a child process, before touching the GPU runtime; one JSON line comes back."""
    env = dict(os.environ, KMD_BENCH_OVERSUBSCRIBE="1", KMD_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", str(ROWS), "--steps", "4",
                        "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["backend"] == "gloo"
    assert d["config"]["counters"]["total"] == ROWS * 4 * 2
    pr = d["per_rank"]
    assert 0 < pr["ms_per_step_min"] <= pr["ms_per_step_max"] == d["ms_per_step"]


def test_bench_two_gpus_over_rccl():
    """The same entry point on two real GPUs with RCCL as the wire (skipped on a one-GPU box)."""
    import kmdiff_amd as K
    if K.device_count() < 2:
        pytest.skip("one GPU on this box")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                             "KMD_BENCH_OVERSUBSCRIBE", "KMD_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", str(ROWS), "--steps", "4",
                        "--warmup", "1", "--correction", "benjamini"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["backend"] == "nccl"


def test_collectives_through_rccl():
    """tests/nccl_probe.py: the dtypes and calls of kmdiff_amd/dist.py over backend "nccl" (= RCCL)."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                        "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "tests", "nccl_probe.py")],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "nccl probe ok" in r.stdout, (r.stdout + r.stderr)[-3000:]
