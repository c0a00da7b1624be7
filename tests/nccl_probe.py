"""Run under torch.distributed.run with --nproc-per-node 1 on a GPU box: the collectives of
kmdiff_amd/dist.py through RCCL itself (backend "nccl"), with the tensor types the N>1 path uses.
World size 1 is all a 1-GPU box allows; it still goes through RCCL's communicator and kernels."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmdiff_amd import dist as D  # noqa: E402

local_rank = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local_rank)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
dev = torch.device("cuda", local_rank)
assert D._dev() == dev
# counters: uint64 carried as int64, SUM
t = torch.from_numpy(np.array([625000000, 86438, 43090, 43348], dtype=np.uint64).astype(np.int64)).to(dev)
dist.all_reduce(t, op=dist.ReduceOp.SUM)
assert t.cpu().tolist() == [625000000, 86438, 43090, 43348]
# elapsed time: float64, MAX
m = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(m, op=dist.ReduceOp.MAX)
assert float(m.item()) == 1.25
# histograms (int64[4096]) and p-value tails (float64, int32) through the all-gather helper
h = torch.arange(4096, dtype=torch.int64, device=dev)
assert torch.equal(D.all_gather(h)[0], h)
p = torch.rand(1000, dtype=torch.float64, device=dev)
s = torch.randint(0, 3, (1000,), dtype=torch.int32, device=dev)
assert torch.equal(D.all_gather(p)[0], p) and torch.equal(D.all_gather(s)[0], s)
n = torch.tensor([p.numel()], dtype=torch.int64, device=dev)
assert int(D.all_gather(n)[0].item()) == 1000
# the Gram matrix of the PCA
g = np.random.default_rng(0).random((40, 40))
assert (D.all_gather(torch.from_numpy(g).to(dev))[0].cpu().numpy() == g).all()
# the wire of kmd_correct_sharded (round 4): dist.torch_transport's two collectives, called the way the library calls them
# -- on ITS device buffers, handed to RCCL as zero-copy views
import kmdiff_amd as K  # noqa: E402
tr, alive = D.torch_transport(K)
assert (tr.rank, tr.world) == (0, 1)
a = K.DeviceBuffer.from_host(np.arange(4096, dtype=np.uint64))
assert tr.allreduce_u64(tr.ctx, a.ptr, 4096, None) == 0
assert a.to_host(np.uint64, 4096).tolist() == list(range(4096))
b = K.DeviceBuffer(4096 * 8)
assert tr.allgather(tr.ctx, a.ptr, b.ptr, 4096 * 8, None) == 0
assert b.to_host(np.uint64, 4096).tolist() == list(range(4096))
assert tr.allgather(tr.ctx, a.ptr, b.ptr, 0, None) == 0
# ... and kmd_correct_sharded over it (one rank: the local walk, the counters back as they went)
pv = np.sort(np.random.default_rng(1).uniform(0, 1e-6, 500)) ** 2
sg = np.zeros(500, dtype=np.int32)
dp, ds = K.DeviceBuffer.from_host(pv), K.DeviceBuffer.from_host(sg)
keep, gc, (n_ctrl, n_case) = D.correct_sharded(K, "holm", 0.05, np.array([10 ** 8, 500, 500, 0], dtype=np.uint64), dp, ds, 500, transport=tr)
want, _, _ = K.aggregate("holm", 0.05, 10 ** 8, dp, ds, 500)
assert keep.tolist() == want.tolist() and int(gc[0]) == 10 ** 8 and n_ctrl == int(want.sum()) and n_case == 0
dist.barrier()
dist.destroy_process_group()
print("nccl probe ok")
