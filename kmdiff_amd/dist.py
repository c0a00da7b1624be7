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
All of that is kmd_correct_sharded in the HIP library (kmd_shard.hip) since round 4; this module supplies the
wire -- a kmd_transport whose two collectives are torch.distributed's (backend "nccl" = RCCL on ROCm, "gloo" in
the CPU tests) -- and the small host-side reductions around it.  Nothing here computes p-values or decisions.
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


def torch_transport(K):
    """A kmd_transport (include/kmdiff_hip.h) whose two collectives are torch.distributed's: under backend "nccl"
    (= RCCL) the library's device buffers are handed to RCCL as they are (zero-copy views); under "gloo" (CPU tests,
    one-GPU dry runs of the N > 1 code) they travel through host memory.  Returns (Transport, keep-alive tuple)."""
    import ctypes as C
    N = K._native
    lib = N.lib()
    nccl = dist.get_backend() == "nccl"
    world, rank = dist.get_world_size(), dist.get_rank()

    def view(ptr, n, typestr, dtype):
        dev = torch.device("cuda", torch.cuda.current_device())
        return torch.as_tensor(_CudaView(ptr, n, typestr), device=dev) if n else torch.empty(0, dtype=dtype, device=dev)

    def to_host(ptr, nbytes):
        a = np.empty(nbytes, dtype=np.uint8)
        if nbytes:
            N.check(lib.kmd_memcpy_d2h(a.ctypes.data, ptr, nbytes, None), "d2h")
        return a

    def allreduce(ctx, d_buf, n, stream):
        try:
            if nccl:
                t = view(d_buf, n, "<i8", torch.int64)
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                torch.cuda.synchronize()
            else:
                t = torch.from_numpy(to_host(d_buf, 8 * n).view(np.int64).copy())
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                a = t.numpy()
                N.check(lib.kmd_memcpy_h2d(d_buf, a.ctypes.data, a.nbytes, None), "h2d")
            return 0
        except Exception:                                    # (no exception may cross the C boundary)
            import traceback
            traceback.print_exc()
            return -2

    def allgather(ctx, d_send, d_recv, nbytes, stream):
        try:
            if nbytes == 0:
                return 0
            if nccl:
                src = view(d_send, nbytes, "|u1", torch.uint8)
                dst = view(d_recv, nbytes * world, "|u1", torch.uint8)
                dist.all_gather_into_tensor(dst, src)
                torch.cuda.synchronize()
            else:
                src = torch.from_numpy(to_host(d_send, nbytes))
                out = [torch.empty_like(src) for _ in range(world)]
                dist.all_gather(out, src)
                a = torch.cat(out).numpy()
                N.check(lib.kmd_memcpy_h2d(d_recv, a.ctypes.data, a.nbytes, None), "h2d")
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return -2

    fa, fg = N.ALLREDUCE_FN(allreduce), N.ALLGATHER_FN(allgather)
    return N.Transport(None, rank, world, fa, fg, N.ABORT_FN()), (fa, fg)      # (no abort: torch.distributed's own timeouts apply)


def correct_sharded(K, correction, threshold, local_counters, pvalue_buf, sign_buf, n_local, transport=None):
    """Stage 3 across ranks: kmd_correct_sharded (kmdiff_amd/csrc/kmd_shard.hip) -- the counter all-reduce, and for
    BH / Holm the histogram all-gather, the critical bin and the exact walk over the gathered tails -- with
    torch.distributed as the wire (torch_transport), or the caller's `transport` (a _native.Transport).
    Returns (keep mask for the local survivors, global counters, (n_control, n_case) kept locally)."""
    import ctypes as C
    N = K._native
    if isinstance(correction, str):
        correction = K.CORRECTION_BY_NAME[correction.lower()]
    local = np.zeros(N.NCOUNTERS, dtype=np.uint64)
    lc = np.asarray(local_counters, dtype=np.uint64)
    local[:min(len(lc), N.NCOUNTERS)] = lc[:N.NCOUNTERS]
    world = dist.get_world_size() if dist.is_initialized() else 1
    alive = None
    if transport is None and world > 1:
        transport, alive = torch_transport(K)
    g = np.zeros(N.NCOUNTERS, dtype=np.uint64)
    keep = K.DeviceBuffer(max(n_local, 1))
    nk, nc, nca = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    N.check(N.lib().kmd_correct_sharded(C.byref(transport) if transport is not None else None, int(correction), float(threshold),
                                        local.ctypes.data, g.ctypes.data, pvalue_buf.ptr if n_local else None,
                                        sign_buf.ptr if (n_local and sign_buf) else None, int(n_local), keep.ptr,
                                        C.byref(nk), C.byref(nc), C.byref(nca), None), "kmd_correct_sharded")
    del alive
    return keep.to_host(np.uint8, n_local), g[:max(len(lc), 4)], (int(nc.value), int(nca.value))


class _CudaView:
    """Zero-copy view of library-owned HBM for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2}
