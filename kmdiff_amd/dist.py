"""Multi-GPU layer of the `diff` hot path: one process per GPU, partitions sharded across
ranks, and the single exchange step the path has.

The reference runs one ThreadPool task per partition in one address space
(include/kmdiff/merge.hpp:259-307) and reduces the per-partition results with
std::accumulate (merge.hpp:316, 402-413) and one global priority queue
(include/kmdiff/aggregator.hpp:325-339).  Partitions are independent (a k-mer lives in
exactly one partition), so here:

  * partition p is owned by rank p % world (no data-path collective in stage 1);
  * after stage 1, ONE all-reduce of the four counters gives N = total_kmers, which every
    corrector needs (cmd/diff.hpp:249);
  * Bonferroni / Sidak / threshold then filter locally;
  * Benjamini-Hochberg / Holm need the global ascending-p order: an all-gather of the
    survivors' p-values (KBs-MBs, latency-bound on xGMI), after which every rank runs the
    same device correction (kmd_correct) over the global list and keeps its own slice.

Collectives go through torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" in the CPU
tests).  Tensors are CUDA tensors under nccl and CPU tensors under gloo; nothing here
computes p-values or decisions on the CPU -- the decisions come from kmd_correct.
"""
import numpy as np
import torch
import torch.distributed as dist


def partition_owner(partition, world_size):
    return partition % world_size


def local_partitions(n_partitions, rank, world_size):
    """Partitions processed by `rank` (round-robin, like the pool's task order)."""
    return list(range(rank, n_partitions, world_size))


def _dev():
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def allreduce_counters(counters):
    """Sum of (total, n_sig, n_sig_control, n_sig_case, ...) over ranks.
    counters: array-like of uint64.  Returns numpy uint64."""
    c = np.asarray(counters, dtype=np.uint64)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return c.copy()
    t = torch.from_numpy(c.astype(np.int64)).to(_dev())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy().astype(np.uint64)


def allreduce_totals(totals):
    """Per-sample k-mer totals summed over the ranks' partitions (the role of
    get_total_kmer, src/kmtricks_utils.cpp:78-139, for sharded synthetic data)."""
    return allreduce_counters(totals)


def allgather_varlen(t):
    """All-gather of 1-D tensors of different lengths.  Returns (concatenation in rank
    order, offsets[world+1])."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return t, [0, int(t.numel())]
    world = dist.get_world_size()
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    m = max(max(sizes), 1)
    pad = torch.zeros(m, dtype=t.dtype, device=t.device)
    pad[:t.numel()] = t
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    cat = torch.cat([o[:s] for o, s in zip(out, sizes)])
    offs = [0]
    for s in sizes:
        offs.append(offs[-1] + s)
    return cat, offs


def max_over_ranks(x):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device=_dev())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def correct_sharded(K, correction, threshold, local_counters, pvalue_buf, sign_buf, n_local):
    """Stage 3 across ranks.  Returns (keep mask for the local survivors, global counters,
    (n_control, n_case) kept locally)."""
    import ctypes as C
    if isinstance(correction, str):
        correction = K.CORRECTION_BY_NAME[correction.lower()]
    g = allreduce_counters(local_counters)
    total_kmers = int(g[0])
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1 or correction not in (K.CORR_BENJAMINI, K.CORR_HOLM):
        keep, n_ctrl, n_case = K.aggregate(correction, threshold, total_kmers, pvalue_buf, sign_buf, n_local)
        return keep, g, (n_ctrl, n_case)
    # BH / Holm: global ascending-p walk (aggregator.hpp:286-310) over all ranks' survivors
    dev = _dev()
    if dev.type != "cuda":
        raise RuntimeError("correct_sharded(BH/Holm) needs the HIP library: no CPU decision path")
    if n_local:
        p_local = torch.as_tensor(_CudaView(pvalue_buf.ptr, n_local, "<f8"), device=dev)
        s_local = torch.as_tensor(_CudaView(sign_buf.ptr, n_local, "<i4"), device=dev)
    else:
        p_local = torch.empty(0, dtype=torch.float64, device=dev)
        s_local = torch.empty(0, dtype=torch.int32, device=dev)
    p_all, offs = allgather_varlen(p_local)
    s_all, _ = allgather_varlen(s_local)
    rank = dist.get_rank()
    torch.cuda.synchronize()
    keep_all = torch.empty(max(int(p_all.numel()), 1), dtype=torch.uint8, device=dev)
    nk, nc, nca = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    K._native.check(K._native.lib().kmd_correct(
        int(correction), float(threshold), total_kmers, p_all.data_ptr() if p_all.numel() else None,
        s_all.data_ptr() if p_all.numel() else None, int(p_all.numel()), keep_all.data_ptr(),
        C.byref(nk), C.byref(nc), C.byref(nca), None), "kmd_correct")
    keep = keep_all[offs[rank]:offs[rank + 1]].cpu().numpy()
    mine_sign = s_local.cpu().numpy()
    n_ctrl = int(((mine_sign == 0) & (keep == 1)).sum())
    return keep, g, (n_ctrl, int(keep.sum()) - n_ctrl)


class _CudaView:
    """Zero-copy view of library-owned HBM for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2}
