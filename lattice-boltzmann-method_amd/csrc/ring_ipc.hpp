// Internal: the peer-mapped transport of the slab ring (capi_ring_ipc.hip) -- halo messages are STORED
// straight into the neighbour's receive window (device memory shared through hipIpcMemHandle_t: another
// GPU over xGMI, or another process on the same GPU) and announced by sequence words; no RCCL.
// "peer-mapped direct stores are the fallback if RCCL latency breaks the weak-scaling target".
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace lbm {

struct IpcTransport;

// prev / next: neighbour ranks (-1 = none; may equal `rank`: a self ring).  slot_doubles: capacity of one
// receive slot (longer messages travel in pieces).  Collective over the ranks of the ring (host rendezvous
// through a POSIX shared-memory segment named after id128: one node).
int ipc_create(IpcTransport** out, const unsigned char* id128, int rank, int nranks, int prev, int next,
               size_t slot_doubles);
void ipc_destroy(IpcTransport* t);
// Enqueue on `st`: my two messages into the neighbours' windows, then their two messages out of mine into
// recv_prev / recv_next.  A count of 0 = no message in that direction (both ends must agree).
int ipc_sendrecv(IpcTransport* t, const double* send_prev, size_t n_send_prev, double* recv_prev, size_t n_recv_prev,
                 const double* send_next, size_t n_send_next, double* recv_next, size_t n_recv_next, hipStream_t st);
// 0, or the code of the first wait that gave up (a neighbour that never delivered): 1 = data, 2 = acknowledgement
int ipc_status(const IpcTransport* t);
// 1 if the receive window lives in ordinary (cached) device memory because the uncached allocation was refused and the
// caller had allowed that ("ring_ipc_cached_ok")
int ipc_window_cached(const IpcTransport* t);

}  // namespace lbm
