// kmd_rccl.cpp -- libkmdiff_hip_rccl.so: the RCCL transport of kmd_correct_sharded (include/kmdiff_hip.h,
// kmd_transport): one process per GPU, the job's single exchange step -- the counter all-reduce and the all-gather of
// the ranks' p-value histograms and tails (the reference's std::accumulate, merge.hpp:316, and its one
// global priority queue, aggregator.hpp:286-310, 325-339) -- as ncclAllReduce / ncclAllGather over xGMI.
// A library of its own so that libkmdiff_hip.so does not pull librccl into hosts that run one GPU.
// Messages are KB to MB: latency-bound, one call each, no bucketing to tune.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/kmdiff_hip.h"
#include "../../include/kmdiff_hip_rccl.h"

namespace {

thread_local std::string g_rccl_error;

int fail(const char* what, const char* msg)
{
  g_rccl_error = std::string(what) + ": " + msg;
  return KMD_E_HIP;
}

#define KMD_NCCL(call) do { const ncclResult_t r__ = (call); if (r__ != ncclSuccess) return fail(#call, ncclGetErrorString(r__)); } while (0)
#define KMD_HIPR(call) do { const hipError_t e__ = (call); if (e__ != hipSuccess) return fail(#call, hipGetErrorString(e__)); } while (0)

struct rccl_ctx { ncclComm_t comm; bool own; bool aborted; };

int rccl_allreduce_u64(void* ctx, uint64_t* d_buf, size_t n, void* stream)
{
  rccl_ctx* R = static_cast<rccl_ctx*>(ctx);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (!R->comm || R->aborted) return fail("kmd_transport_rccl", "the communicator was aborted");
  KMD_NCCL(ncclAllReduce(d_buf, d_buf, n, ncclUint64, ncclSum, R->comm, st));
  KMD_HIPR(hipStreamSynchronize(st));
  return KMD_OK;
}

int rccl_allgather(void* ctx, const void* d_send, void* d_recv, size_t bytes, void* stream)
{
  rccl_ctx* R = static_cast<rccl_ctx*>(ctx);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (!R->comm || R->aborted) return fail("kmd_transport_rccl", "the communicator was aborted");
  if (bytes) KMD_NCCL(ncclAllGather(d_send, d_recv, bytes, ncclInt8, R->comm, st));
  KMD_HIPR(hipStreamSynchronize(st));
  return KMD_OK;
}

// kmd_transport::abort: this rank cannot go on.  ncclCommAbort tears the communicator down without waiting for the
// peers; theirs then fail (or time out) in RCCL instead of waiting for a collective this rank will never join.
// A communicator that was only WRAPPED (kmd_transport_rccl_wrap) belongs to the host -- torch.distributed's, say --
// which will destroy it itself: aborting it here would leave the host a freed handle.  This transport then
// only refuses further collectives; the peers are released by the host's own timeout or abort.
void rccl_abort(void* ctx)
{
  rccl_ctx* R = static_cast<rccl_ctx*>(ctx);
  R->aborted = true;
  if (R->own && R->comm) { (void)ncclCommAbort(R->comm); R->comm = nullptr; }
}

} // namespace

extern "C" {

const char* kmd_rccl_last_error(void) { return g_rccl_error.c_str(); }

int kmd_rccl_unique_id(void* id128)
{
  if (!id128) return fail("kmd_rccl_unique_id", "NULL");
  ncclUniqueId id;
  KMD_NCCL(ncclGetUniqueId(&id));
  static_assert(sizeof id == KMD_RCCL_UNIQUE_ID_BYTES, "ncclUniqueId size");
  std::memcpy(id128, &id, sizeof id);
  return KMD_OK;
}

int kmd_transport_rccl_init(kmd_transport* out, int world, int rank, const void* id128)
{
  if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return fail("kmd_transport_rccl_init", "arguments");
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof id);
  ncclComm_t comm = nullptr;
  KMD_NCCL(ncclCommInitRank(&comm, world, id, rank));            // (the calling thread's current device)
  rccl_ctx* R = new rccl_ctx { comm, true, false };
  out->ctx = R; out->rank = rank; out->world = world;
  out->allreduce_u64 = rccl_allreduce_u64; out->allgather = rccl_allgather; out->abort = rccl_abort;
  return KMD_OK;
}

int kmd_transport_rccl_wrap(kmd_transport* out, void* nccl_comm)
{
  if (!out || !nccl_comm) return fail("kmd_transport_rccl_wrap", "NULL");
  ncclComm_t comm = static_cast<ncclComm_t>(nccl_comm);
  int rank = 0, world = 0;
  KMD_NCCL(ncclCommUserRank(comm, &rank));
  KMD_NCCL(ncclCommCount(comm, &world));
  rccl_ctx* R = new rccl_ctx { comm, false, false };
  out->ctx = R; out->rank = rank; out->world = world;
  out->allreduce_u64 = rccl_allreduce_u64; out->allgather = rccl_allgather; out->abort = rccl_abort;
  return KMD_OK;
}

int kmd_transport_rccl_destroy(kmd_transport* t)
{
  if (!t || !t->ctx) return KMD_OK;
  rccl_ctx* R = static_cast<rccl_ctx*>(t->ctx);
  if (R->own && R->comm) (void)ncclCommDestroy(R->comm);
  delete R;
  t->ctx = nullptr;
  return KMD_OK;
}

} // extern "C"
