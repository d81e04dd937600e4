// comm_internal.hpp -- the collective layer behind rsem_comm (C ABI: include/rsem_hip.h, implementation: comm.hip).
//
// Two kinds of communicator share one interface:
//   RCCL  : one rank per GPU (a thread of one process, or one process per GPU), ncclCommInitRank on a shared id;
//           collectives are enqueued on the caller's HIP stream (xGMI between the GPUs of a node).
//   LOCAL : a group of ranks inside ONE process that may share a device (RCCL refuses two ranks on one GPU): the
//           exchange goes through the ranks' device buffers with host barriers.  It exists so that the sharded code
//           paths (row split, per-round reduction, stop rule on every rank) can be exercised on a single-GPU box; it
//           is not a transport anybody should run at scale.
#pragma once
#include "common.hpp"

struct rsem_comm;

namespace rsem {

// in-place sum over all ranks of n doubles at d_buf, result on every rank; ordered on `st`
int comm_allreduce_sum_f64(rsem_comm* c, double* d_buf, size_t n, hipStream_t st);
// in-place sum over all ranks, result on `root` only (the other ranks' buffers are unspecified afterwards)
int comm_reduce_sum_f64(rsem_comm* c, double* d_buf, size_t n, int root, hipStream_t st);
// false for NULL and for a one-rank communicator (unless RSEM_COMM_FORCE is set): the collectives are no-ops then
bool comm_active(const rsem_comm* c);
int comm_rank(const rsem_comm* c);
int comm_world(const rsem_comm* c);

}  // namespace rsem
