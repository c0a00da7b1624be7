"""Run under torch.distributed.run with --nproc-per-node 2: kmd_correct_sharded over RCCL with TWO ranks -- on two GPUs
where the box has them, or (KMD_PROBE_FOLD=1) both ranks on GPU 0, which RCCL may refuse ("Duplicate GPU detected"): the
caller records the outcome either way (tests/test_gpu_bench.py::test_two_ranks_through_rccl_on_this_box).
Both wires are driven: dist.torch_transport (torch.distributed's communicator, what bench.py uses) and
libkmdiff_hip_rccl.so's own communicator (kmd_transport_rccl_init: what a C++ host with one process per GPU uses)."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmdiff_amd import dist as D  # noqa: E402
import kmdiff_amd as K  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
local_rank = 0 if os.environ.get("KMD_PROBE_FOLD") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local_rank)
import datetime  # noqa: E402
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=120))
N = K._native
N.check(N.lib().kmd_set_device(local_rank))
# the same survivor list on every rank (same seed), cut in `world` pieces: rank r owns piece r
rng = np.random.default_rng(3)
p_all = np.sort(rng.uniform(0, 1e-6, 5000)) ** 2
rng.shuffle(p_all)
s_all = rng.integers(0, 3, len(p_all)).astype(np.int32)
cuts = np.linspace(0, len(p_all), world + 1).astype(int)
mine = slice(cuts[rank], cuts[rank + 1])
total = 10 ** 9
for name in ("benjamini", "holm", "bonferroni"):
    want, _, _ = K.aggregate(name, 0.05, total, K.DeviceBuffer.from_host(p_all), K.DeviceBuffer.from_host(s_all), len(p_all))
    dp, ds = K.DeviceBuffer.from_host(p_all[mine]), K.DeviceBuffer.from_host(s_all[mine])
    local = np.array([total // world + (1 if rank < total % world else 0), mine.stop - mine.start, 0, 0], dtype=np.uint64)
    keep, g, _ = D.correct_sharded(K, name, 0.05, local, dp, ds, mine.stop - mine.start)
    assert int(g[0]) == total and int(g[1]) == len(p_all), g
    assert keep.tolist() == want[mine].tolist(), name
# ... and through the library's own RCCL communicator (the unique id travels over torch.distributed's store)
R = N.rccl_lib()
ident = (C.c_uint8 * 128)()
if rank == 0:
    assert R.kmd_rccl_unique_id(ident) == 0, R.kmd_rccl_last_error()
t_id = torch.tensor(list(ident), dtype=torch.uint8, device=torch.device("cuda", local_rank))
dist.broadcast(t_id, src=0)
ident = (C.c_uint8 * 128)(*t_id.cpu().tolist())
tr = N.Transport()
assert R.kmd_transport_rccl_init(C.byref(tr), world, rank, ident) == 0, R.kmd_rccl_last_error()
want, _, _ = K.aggregate("benjamini", 0.05, total, K.DeviceBuffer.from_host(p_all), K.DeviceBuffer.from_host(s_all), len(p_all))
dp, ds = K.DeviceBuffer.from_host(p_all[mine]), K.DeviceBuffer.from_host(s_all[mine])
local = np.array([total // world + (1 if rank < total % world else 0), mine.stop - mine.start, 0, 0], dtype=np.uint64)
keep, g, _ = D.correct_sharded(K, "benjamini", 0.05, local, dp, ds, mine.stop - mine.start, transport=tr)
assert int(g[0]) == total and keep.tolist() == want[mine].tolist()
R.kmd_transport_rccl_destroy(C.byref(tr))
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print("nccl probe2 ok: %d ranks on %s" % (world, "one GPU" if os.environ.get("KMD_PROBE_FOLD") == "1" else "%d GPUs" % world))
