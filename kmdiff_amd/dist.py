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
  * Benjamini-Hochberg / Holm need the global ascending-p order: an all-gather of the ranks'
    4096-bin p-value histograms (32 KB each) locates the first bin the walk cannot accept
    wholesale; only the p-values from that bin on are all-gathered and walked exactly, on
    every rank, by the device corrector (kmd_correct_from_rank).

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


def all_gather(t):
    """dist.all_gather of equally shaped tensors; the result lives where `t` lives.  Under gloo
    (tests, 1-GPU dry runs of the N>1 code) device tensors travel through host memory."""
    world = dist.get_world_size()
    wire = t if (dist.get_backend() == "nccl" or t.device.type == "cpu") else t.cpu()
    out = [torch.empty_like(wire) for _ in range(world)]
    dist.all_gather(out, wire.contiguous())
    return [o.to(t.device) for o in out]


def allgather_varlen(t):
    """All-gather of 1-D tensors of different lengths.  Returns (concatenation in rank
    order, offsets[world+1])."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return t, [0, int(t.numel())]
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [int(s.item()) for s in all_gather(n)]
    m = max(max(sizes), 1)
    pad = torch.zeros(m, dtype=t.dtype, device=t.device)
    pad[:t.numel()] = t
    out = all_gather(pad)
    cat = torch.cat([o[:s] for o, s in zip(out, sizes)])
    offs = [0]
    for s in sizes:
        offs.append(offs[-1] + s)
    return cat, offs


def sum_in_rank_order(a):
    """Sum of a float64 array over the ranks, added in rank order on every rank (bitwise the
    same everywhere, unlike an all-reduce whose order is the library's): the ranks' Gram
    matrices of the pop-strat PCA (kmd_pca_gram) before kmd_pca_eigen."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return a.copy()
    parts = all_gather(torch.from_numpy(a).to(_dev()))
    out = parts[0].cpu().numpy().copy()
    for p in parts[1:]:
        out += p.cpu().numpy()
    return out


def max_over_ranks(x):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device=_dev())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


N_HIST_BINS = 4096


def pvalue_histogram(K, p_t):
    """4096-bin log-spaced histogram (device) of a CUDA float64 tensor of p-values."""
    hist = torch.zeros(N_HIST_BINS, dtype=torch.int64, device=p_t.device)
    if p_t.numel():
        K._native.check(K._native.lib().kmd_pvalue_histogram(p_t.data_ptr(), int(p_t.numel()), hist.data_ptr(), None),
                        "kmd_pvalue_histogram")
    torch.cuda.synchronize()
    return hist


def critical_bin(K, correction, threshold, total_kmers, hist_global):
    """(first bin the BH/Holm walk cannot accept wholesale, survivors before it) -- device."""
    import ctypes as C
    b, before = C.c_uint32(0), C.c_uint64(0)
    K._native.check(K._native.lib().kmd_correct_critical_bin(int(correction), float(threshold), int(total_kmers),
                                                             hist_global.data_ptr(), C.byref(b), C.byref(before), None),
                    "kmd_correct_critical_bin")
    return int(b.value), int(before.value)


def tail_of(p_t, first_bin):
    """Indices of the p-values whose histogram bin is >= first_bin (device ops)."""
    bins = (p_t.view(torch.int64) >> 51) & (N_HIST_BINS - 1)
    return torch.nonzero(bins >= first_bin, as_tuple=False).flatten()


def walk_tail(K, correction, threshold, total_kmers, rank_offset, p_all, s_all):
    """The exact ascending walk (kmd_correct_from_rank) over the gathered tail."""
    import ctypes as C
    n = int(p_all.numel())
    keep = torch.zeros(max(n, 1), dtype=torch.uint8, device=p_all.device)
    nk, nc, nca = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    K._native.check(K._native.lib().kmd_correct_from_rank(
        int(correction), float(threshold), int(total_kmers), int(rank_offset), p_all.data_ptr() if n else None,
        s_all.data_ptr() if n else None, n, keep.data_ptr(), C.byref(nk), C.byref(nc), C.byref(nca), None),
        "kmd_correct_from_rank")
    return keep[:n]


def correct_sharded(K, correction, threshold, local_counters, pvalue_buf, sign_buf, n_local):
    """Stage 3 across ranks.  Returns (keep mask for the local survivors, global counters,
    (n_control, n_case) kept locally).

    BH / Holm walk the survivors of ALL ranks in ascending p and stop at the first rejection
    (aggregator.hpp:286-310).  Reproduced without moving every survivor: all-gather the 32 KB
    p-value histograms, find the first bin the walk cannot accept wholesale, all-gather only
    the p-values from that bin on, and walk those exactly on every rank (same list, same
    order: rank-major, then local order) starting at the rank the earlier bins consumed."""
    if isinstance(correction, str):
        correction = K.CORRECTION_BY_NAME[correction.lower()]
    g = allreduce_counters(local_counters)
    total_kmers = int(g[0])
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1 or correction not in (K.CORR_BENJAMINI, K.CORR_HOLM):
        keep, n_ctrl, n_case = K.aggregate(correction, threshold, total_kmers, pvalue_buf, sign_buf, n_local)
        return keep, g, (n_ctrl, n_case)
    if not torch.cuda.is_available():
        raise RuntimeError("correct_sharded(BH/Holm) needs the HIP library: no CPU decision path")
    dev = torch.device("cuda", torch.cuda.current_device())      # compute device (the wire may be gloo)
    if n_local:
        p_local = torch.as_tensor(_CudaView(pvalue_buf.ptr, n_local, "<f8"), device=dev)
        s_local = torch.as_tensor(_CudaView(sign_buf.ptr, n_local, "<i4"), device=dev)
    else:
        p_local = torch.empty(0, dtype=torch.float64, device=dev)
        s_local = torch.empty(0, dtype=torch.int32, device=dev)
    hist = pvalue_histogram(K, p_local)
    hists = all_gather(hist)                                       # per-rank histograms over xGMI
    hist_global = torch.stack(hists).sum(dim=0).contiguous()
    first_bin, before = critical_bin(K, correction, threshold, total_kmers, hist_global)
    idx = tail_of(p_local, first_bin)
    p_all, offs = allgather_varlen(p_local[idx].contiguous())
    s_all, _ = allgather_varlen(s_local[idx].contiguous())
    torch.cuda.synchronize()
    keep_tail = walk_tail(K, correction, threshold, total_kmers, before, p_all, s_all)
    rank = dist.get_rank()
    keep_t = torch.ones(n_local, dtype=torch.uint8, device=dev)    # bins before the critical one: accepted
    keep_t[idx] = keep_tail[offs[rank]:offs[rank + 1]]
    keep = keep_t.cpu().numpy()
    mine_sign = s_local.cpu().numpy()
    n_ctrl = int(((mine_sign == 0) & (keep == 1)).sum())
    return keep, g, (n_ctrl, int(keep.sum()) - n_ctrl)


class _CudaView:
    """Zero-copy view of library-owned HBM for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2}
