/* kmdiff_hip_rccl.h -- libkmdiff_hip_rccl.so: the RCCL transport of kmd_correct_sharded (kmdiff_hip.h).
 *
 * One process per GPU.  The job's single exchange step -- the reference's std::accumulate over the
 * partitions' counters (include/kmdiff/merge.hpp:316, 402-413) and its one global priority queue
 * (include/kmdiff/aggregator.hpp:286-310, 325-339) -- travels as ncclAllReduce / ncclAllGather over xGMI.
 * Kept out of libkmdiff_hip.so so that a one-GPU host does not load librccl.
 *
 *   rank 0:  kmd_rccl_unique_id(id);  ... hand the 128 bytes to the other ranks (a file, a socket, MPI, the launcher) ...
 *   every rank, its device current:  kmd_transport_rccl_init(&t, world, rank, id);
 *                                    kmd_correct_sharded(&t, ...);
 *                                    kmd_transport_rccl_destroy(&t);
 * A host that has a communicator already (ncclComm_t) wraps it: kmd_transport_rccl_wrap. */
#ifndef KMDIFF_HIP_RCCL_H
#define KMDIFF_HIP_RCCL_H

#include "kmdiff_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define KMD_RCCL_UNIQUE_ID_BYTES 128

const char* kmd_rccl_last_error(void);
int kmd_rccl_unique_id(void* id128);
int kmd_transport_rccl_init(kmd_transport* out, int world, int rank, const void* id128);
/* (a wrapped communicator stays the host's: kmd_transport.abort on it only makes this transport refuse further
 * collectives -- it never calls ncclCommAbort on a handle it does not own; _destroy leaves it alone as well) */
int kmd_transport_rccl_wrap(kmd_transport* out, void* nccl_comm);
int kmd_transport_rccl_destroy(kmd_transport* t);

#ifdef __cplusplus
}
#endif
#endif /* KMDIFF_HIP_RCCL_H */
