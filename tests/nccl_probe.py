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
dist.barrier()
dist.destroy_process_group()
print("nccl probe ok")
