"""The N>1 host path on CPU: two processes over gloo exercise the partition map and the
exchange plumbing of kmdiff_amd/dist.py (no GPU, no compute)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kmdiff_amd import dist as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # stage-1 counters of this rank's partitions (total, n_sig, n_ctrl, n_case)
        mine = D.local_partitions(7, rank, world)
        local = np.array([1000 * len(mine) + rank, 10 + rank, 4 + rank, 6], dtype=np.uint64)
        g = D.allreduce_counters(local)
        # survivors' p-values, different length per rank (rank 1 has none when world == 2)
        p_local = torch.arange(3 * (1 - rank), dtype=torch.float64) + 10.0 * rank
        cat, offs = D.allgather_varlen(p_local)
        mx = D.max_over_ranks(0.5 + rank)
        # the ranks' PCA Gram matrices, added in rank order
        gram = D.sum_in_rank_order(np.array([[0.1, 1e-17], [1e-17, 0.3]]) * (1 + rank * 1e16))
        D.barrier()
        q.put((rank, mine, g.tolist(), cat.tolist(), offs, mx, gram.tolist()))
    finally:
        dist.destroy_process_group()


def test_partition_map_covers_every_partition_once():
    for world in (1, 2, 3, 8):
        seen = sorted(p for r in range(world) for p in D.local_partitions(256, r, world))
        assert seen == list(range(256))
        assert all(D.partition_owner(p, world) == r for r in range(world) for p in D.local_partitions(256, r, world))
    assert [len(D.local_partitions(256, r, 8)) for r in range(8)] == [32] * 8


def test_single_process_collectives_are_identity():
    c = np.array([5, 1, 1, 0], dtype=np.uint64)
    assert D.allreduce_counters(c).tolist() == [5, 1, 1, 0]
    t = torch.tensor([0.1, 0.2], dtype=torch.float64)
    cat, offs = D.allgather_varlen(t)
    assert cat.tolist() == [0.1, 0.2] and offs == [0, 2]
    assert D.max_over_ranks(1.5) == 1.5
    assert D.sum_in_rank_order(np.eye(3)).tolist() == np.eye(3).tolist()


@pytest.mark.timeout(120)
def test_two_ranks_gloo_exchange():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=90) for _ in range(world))
    [p.join(30) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (r0, mine0, g0, cat0, offs0, mx0, gram0), (r1, mine1, g1, cat1, offs1, mx1, gram1) = res
    a = np.array([[0.1, 1e-17], [1e-17, 0.3]])
    assert gram0 == gram1 == (a + a * (1 + 1e16)).tolist()           # rank 0 first: bitwise the same on both
    assert mine0 == [0, 2, 4, 6] and mine1 == [1, 3, 5]
    assert g0 == g1 == [4000 + 3001, 21, 9, 12]
    assert cat0 == cat1 == [0.0, 1.0, 2.0] and offs0 == offs1 == [0, 3, 3]
    assert mx0 == mx1 == 1.5


@pytest.mark.timeout(300)
def test_bench_starts_its_ranks_as_children_without_a_launcher():
    """`python bench.py --gpus 2` run bare (no torchrun, as the driver runs the N = 1 line) must start the two ranks
    itself.  Without a GPU each rank stops at "needs an MI355X" -- after the rendezvous environment was set up by
    the child launcher, which is what this checks -- and bench.py leaves with the launcher's non-zero code."""
    import os
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=280, cwd=root, env=env)
    assert r.returncode != 0
    # (the launcher ends the sibling as soon as the first rank has failed: one of the two messages may never be written)
    assert r.stderr.count("bench.py needs an MI355X") >= 1 and "local_rank: 1" in r.stderr, r.stderr[-2000:]
    # and a mismatch between the launcher's world size and --gpus is refused before anything touches the GPU
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=120,
                       cwd=root, env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE (2) != --gpus (4)" in r.stderr
