// C ABI, part 7: the slab ring in C++ -- one process per GPU, RCCL send/recv over xGMI on packed
// halo buffers, overlapped with the interior launch on a second HIP stream.  This is the native
// counterpart of pylbm/slab.py (which drives the same kernels through torch.distributed); both
// implement the block binding of test/decompose_domain.cpp:181-187 generalised to D ghost rows.
//
// Two transports carry the packed messages (lbm_ring_create_ex): RCCL send / recv (the default; dlopen()ed on
// first use -- the librccl.so.1 of the hosting process if one is already mapped, e.g. PyTorch's, else the ROCm
// one: liblbm_hip.so itself has no link-time dependency on it) and peer-mapped direct stores into the
// neighbour's receive window (capi_ring_ipc.hip; also works between processes that share ONE GPU, which RCCL
// refuses).  What travels and when is the same for both.
#include <dlfcn.h>
#include <fcntl.h>
#include <unistd.h>

#include <cstdlib>
#include <cstring>
#include <new>

#include "d2q9.hpp"
#include "internal.hpp"
#include "ring_ipc.hpp"
#include "slab_ibm.hpp"

namespace {

// the slice of rccl.h this file needs (kept local so the build does not require the header)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;            // ncclSuccess = 0
const int kNcclFloat64 = 8;          // ncclDataType_t ncclFloat64 / ncclDouble

struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  // optional (checked once per batch by lbm_ring_status; absent in a very old runtime: then the status stays LBM_OK)
  ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
};

Rccl g_rccl;

int load_rccl() {
  if (g_rccl.h) return LBM_OK;
  // the soname first: inside a process that already maps an RCCL (PyTorch's librccl.so.1) that is the one to join --
  // the bare "librccl.so" can resolve to /opt/rocm/lib and put a SECOND runtime beside it
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) {
    lbm::set_error("lbm_ring: cannot dlopen librccl.so (%s)", dlerror());
    return LBM_ERR_HIP;
  }
#define LBM_SYM(field, name)                                                         \
  *(void**)(&g_rccl.field) = dlsym(h, name);                                         \
  if (!g_rccl.field) {                                                               \
    lbm::set_error("lbm_ring: librccl.so lacks %s", name);                           \
    return LBM_ERR_HIP;                                                              \
  }
  LBM_SYM(GetUniqueId, "ncclGetUniqueId")
  LBM_SYM(CommInitRank, "ncclCommInitRank")
  LBM_SYM(CommDestroy, "ncclCommDestroy")
  LBM_SYM(Send, "ncclSend")
  LBM_SYM(Recv, "ncclRecv")
  LBM_SYM(GroupStart, "ncclGroupStart")
  LBM_SYM(GroupEnd, "ncclGroupEnd")
  LBM_SYM(GetErrorString, "ncclGetErrorString")
#undef LBM_SYM
  *(void**)(&g_rccl.CommGetAsyncError) = dlsym(h, "ncclCommGetAsyncError");
  *(void**)(&g_rccl.CommAbort) = dlsym(h, "ncclCommAbort");
  g_rccl.h = h;
  return LBM_OK;
}

#define LBM_CHECK_NCCL(expr)                                                                  \
  do {                                                                                        \
    ncclResult_t r_ = (expr);                                                                 \
    if (r_ != 0) {                                                                            \
      lbm::set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
      return LBM_ERR_HIP;                                                                     \
    }                                                                                         \
  } while (0)

}  // namespace

struct lbm_ring {
  int transport;                 // LBM_RING_RCCL / LBM_RING_IPC
  ncclComm_t comm;               // RCCL
  lbm::IpcTransport* ipc;        // peer-mapped windows
  size_t bufsz;                  // doubles each send / receive buffer holds
  int skipped;                   // launches without an exchange since the last one (their ghost rows are used up)
  int rank, nranks, next, prev;  // neighbours, -1 = none (chain end)
  lbm_geom g;                    // slab geometry (ghost = halo depth)
  size_t msg;                    // doubles per packed message
  hipStream_t edge;              // edge rows, pack, send/recv, unpack
  hipStream_t aux;               // immersed-boundary rows + forcing chain (created on first use)
  hipEvent_t main_done, edge_done, aux_done;
  double *send_next, *send_prev, *recv_prev, *recv_next;
  // lbm_ring_profile(1): timed events around the three phases of the last launch-step
  int closed;                    // periodic ring (every rank has both neighbours), not a chain with two ends
  int valid;                     // ghost rows per side known to be current (set by an exchange, used up by launches without one)
  int profile;
  hipEvent_t t_edge0, t_edge1, t_xchg1, t_main0, t_main1;
  int rccl_failed;               // ncclCommGetAsyncError has reported an error: the communicator is aborted, not destroyed
};

using namespace lbm;

extern "C" {

static int default_transport() {
  const char* e = std::getenv("LBM_RING_TRANSPORT");
  if (e && (!std::strcmp(e, "ipc") || !std::strcmp(e, "peer") || !std::strcmp(e, "1"))) return LBM_RING_IPC;
  return LBM_RING_RCCL;
}

int lbm_ring_unique_id_ex(unsigned char* id128, int transport) {
  LBM_REQUIRE(id128, "lbm_ring_unique_id: NULL buffer");
  if (transport == LBM_RING_DEFAULT) transport = default_transport();
  if (transport == LBM_RING_IPC) {
    // any 128 bytes no other ring of this node uses: they name the rendezvous segment
    const int fd = open("/dev/urandom", O_RDONLY);
    const bool ok = fd >= 0 && read(fd, id128, 128) == 128;
    if (fd >= 0) close(fd);
    LBM_REQUIRE(ok, "lbm_ring_unique_id: cannot read /dev/urandom");
    return LBM_OK;
  }
  LBM_REQUIRE(transport == LBM_RING_RCCL, "lbm_ring_unique_id: unknown transport %d", transport);
  int rc = load_rccl();
  if (rc) return rc;
  ncclUniqueId id;
  LBM_CHECK_NCCL(g_rccl.GetUniqueId(&id));
  std::memcpy(id128, id.internal, 128);
  return LBM_OK;
}
int lbm_ring_unique_id(unsigned char* id128) { return lbm_ring_unique_id_ex(id128, LBM_RING_DEFAULT); }

int lbm_ring_create_ex(lbm_ring** out, const unsigned char* id128, int rank, int nranks, const lbm_geom* slab,
                       int periodic, int transport) {
  LBM_REQUIRE(out && id128 && slab, "lbm_ring_create: NULL argument");
  LBM_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "lbm_ring_create: rank %d of %d", rank, nranks);
  LBM_REQUIRE(slab->ghost >= 1 && slab->ghost <= 15, "lbm_ring_create: slab needs 1..15 ghost rows (ghost=%d)", slab->ghost);
  if (transport == LBM_RING_DEFAULT) transport = default_transport();
  LBM_REQUIRE(transport == LBM_RING_RCCL || transport == LBM_RING_IPC, "lbm_ring_create: unknown transport %d", transport);
  if (transport == LBM_RING_RCCL) {
    int rc = load_rccl();
    if (rc) return rc;
  }
  lbm_ring* rg = new (std::nothrow) lbm_ring();
  LBM_REQUIRE(rg, "lbm_ring_create: out of host memory");
  std::memset(rg, 0, sizeof *rg);
  rg->transport = transport;
  rg->rank = rank;
  rg->nranks = nranks;
  rg->next = (periodic || rank < nranks - 1) ? (rank + 1) % nranks : -1;
  rg->prev = (periodic || rank > 0) ? (rank + nranks - 1) % nranks : -1;
  rg->closed = periodic ? 1 : 0;
  rg->g = *slab;
  rg->msg = (size_t)lbm_halo_rows(slab->ghost) * slab->C;
  // room for the two-colour message of the two-phase step when the slab has its 3 ghost rows
  size_t bufsz = (size_t)lbm_halo_rows(LBM_HALO_FULL(slab->ghost)) * slab->C;  // complete ghost rows (walls)
  if (slab->ghost == 3 && bufsz < 2 * (size_t)lbm_halo_rows(LBM_HALO_TWO_PHASE) * slab->C) bufsz = 2 * (size_t)lbm_halo_rows(LBM_HALO_TWO_PHASE) * slab->C;
  rg->bufsz = bufsz;
  if (transport == LBM_RING_RCCL) {
    ncclUniqueId id;
    std::memcpy(id.internal, id128, 128);
    ncclResult_t nr = g_rccl.CommInitRank(&rg->comm, nranks, id, rank);
    if (nr != 0) {
      set_error("ncclCommInitRank failed: %s", g_rccl.GetErrorString(nr));
      delete rg;
      return LBM_ERR_HIP;
    }
  } else {
    int rc = ipc_create(&rg->ipc, id128, rank, nranks, rg->prev, rg->next, bufsz);
    if (rc) {
      delete rg;
      return rc;
    }
  }
  int lo = 0, hi = 0;
  hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);
  if (e == hipSuccess) e = hipStreamCreateWithPriority(&rg->edge, hipStreamNonBlocking, hi);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&rg->main_done, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&rg->edge_done, hipEventDisableTiming);
  for (double** p : {&rg->send_next, &rg->send_prev, &rg->recv_prev, &rg->recv_next})
    if (e == hipSuccess) e = hipMalloc(p, bufsz * sizeof(double));
  if (e != hipSuccess) {
    set_error("lbm_ring_create: %s", hipGetErrorString(e));
    lbm_ring_destroy(rg);
    return LBM_ERR_HIP;
  }
  *out = rg;
  return LBM_OK;
}

int lbm_ring_create(lbm_ring** out, const unsigned char* id128, int rank, int nranks,
                    const lbm_geom* slab, int periodic) {
  return lbm_ring_create_ex(out, id128, rank, nranks, slab, periodic, LBM_RING_DEFAULT);
}

int lbm_ring_transport(const lbm_ring* rg) { return rg ? rg->transport : LBM_ERR_INVALID; }

// 0 while every message has arrived.  Peer-mapped transport: its bounded waits report here when a neighbour never
// delivered (1) or never acknowledged (2) -- the launch chain still drains, its results are void.  RCCL:
// ncclCommGetAsyncError (a failed peer, a broken link: errors RCCL detects asynchronously and that no launch call
// returns); meant to be read once per batch of launches, not per step.
int lbm_ring_status(const lbm_ring* rg) {
  LBM_REQUIRE(rg, "lbm_ring_status: NULL ring");
  if (rg->transport == LBM_RING_RCCL) {
    if (rg->rccl_failed) {
      set_error("lbm_ring: RCCL reported an asynchronous error on rank %d earlier", rg->rank);
      return LBM_ERR_STATE;
    }
    if (rg->comm && g_rccl.CommGetAsyncError) {
      ncclResult_t async = 0;
      const ncclResult_t r = g_rccl.CommGetAsyncError(rg->comm, &async);
      const int kNcclInProgress = 7;
      if (r != 0 || (async != 0 && async != kNcclInProgress)) {
        const_cast<lbm_ring*>(rg)->rccl_failed = 1;
        set_error("lbm_ring: RCCL asynchronous error on rank %d: %s", rg->rank, g_rccl.GetErrorString(r != 0 ? r : async));
        return LBM_ERR_STATE;
      }
    }
    return LBM_OK;
  }
  const int st = rg->ipc ? ipc_status(rg->ipc) : 0;
  if (st) set_error("lbm_ring: a neighbour of rank %d never %s within the time limit (\"ring_ipc_timeout_ms\")", rg->rank,
                    st == 1 ? "delivered its message" : "acknowledged a message");
  return st ? LBM_ERR_STATE : LBM_OK;
}

int lbm_ring_window_cached(const lbm_ring* rg) { return rg && rg->ipc ? ipc_window_cached(rg->ipc) : 0; }

int lbm_ring_destroy(lbm_ring* rg) {
  if (!rg) return LBM_OK;
  if (rg->edge) (void)hipStreamSynchronize(rg->edge);
  for (double* p : {rg->send_next, rg->send_prev, rg->recv_prev, rg->recv_next})
    if (p) (void)hipFree(p);
  if (rg->main_done) (void)hipEventDestroy(rg->main_done);
  if (rg->edge_done) (void)hipEventDestroy(rg->edge_done);
  if (rg->aux_done) (void)hipEventDestroy(rg->aux_done);
  for (hipEvent_t ev : {rg->t_edge0, rg->t_edge1, rg->t_xchg1, rg->t_main0, rg->t_main1})
    if (ev) (void)hipEventDestroy(ev);
  if (rg->aux) {
    (void)hipStreamSynchronize(rg->aux);
    (void)hipStreamDestroy(rg->aux);
  }
  if (rg->edge) (void)hipStreamDestroy(rg->edge);
  // a communicator with an asynchronous error is aborted: ncclCommDestroy would wait for its outstanding operations
  if (rg->comm && rg->rccl_failed && g_rccl.CommAbort) (void)g_rccl.CommAbort(rg->comm);
  else if (rg->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(rg->comm);
  if (rg->ipc) ipc_destroy(rg->ipc);
  delete rg;
  return LBM_OK;
}

// One message to and one from each neighbour, enqueued on the ring's edge stream (counts in doubles; 0 = none)
static int ring_transfer(lbm_ring* rg, const double* send_prev, size_t n_send_prev, double* recv_prev, size_t n_recv_prev,
                         const double* send_next, size_t n_send_next, double* recv_next, size_t n_recv_next) {
  if (rg->transport == LBM_RING_IPC)
    return ipc_sendrecv(rg->ipc, send_prev, n_send_prev, recv_prev, n_recv_prev, send_next, n_send_next, recv_next, n_recv_next, rg->edge);
  LBM_CHECK_NCCL(g_rccl.GroupStart());
  // sends (to next, to prev), receives (from prev, from next): with two ranks both neighbours
  // are the same peer and messages match in issue order
  if (rg->next >= 0 && n_send_next) LBM_CHECK_NCCL(g_rccl.Send(send_next, n_send_next, kNcclFloat64, rg->next, rg->comm, rg->edge));
  if (rg->prev >= 0 && n_send_prev) LBM_CHECK_NCCL(g_rccl.Send(send_prev, n_send_prev, kNcclFloat64, rg->prev, rg->comm, rg->edge));
  if (rg->prev >= 0 && n_recv_prev) LBM_CHECK_NCCL(g_rccl.Recv(recv_prev, n_recv_prev, kNcclFloat64, rg->prev, rg->comm, rg->edge));
  if (rg->next >= 0 && n_recv_next) LBM_CHECK_NCCL(g_rccl.Recv(recv_next, n_recv_next, kNcclFloat64, rg->next, rg->comm, rg->edge));
  LBM_CHECK_NCCL(g_rccl.GroupEnd());
  return LBM_OK;
}

// Bring the ghost rows of `lattice` (and of `lattice2`, if given: the second colour of the
// two-phase model travels in the same message) up to date: pack, one send + one recv per neighbour
// in one RCCL group, unpack -- all enqueued on the ring's edge stream, after the work already
// enqueued on `after`.
static int ring_exchange(lbm_ring* rg, double* lattice, double* lattice2, lbm_stream_t after, bool full = false) {
  // two lattices = the two colours; full = complete ghost rows (multi-step launches with walls)
  const int G = lattice2 ? LBM_HALO_TWO_PHASE : (full ? LBM_HALO_FULL(rg->g.ghost) : rg->g.ghost);
  const size_t msg = (size_t)lbm_halo_rows(G) * rg->g.C;
  LBM_REQUIRE(msg * (lattice2 ? 2 : 1) <= rg->bufsz, "lbm_ring: message of %zu doubles, buffers of %zu", msg * (lattice2 ? 2 : 1), rg->bufsz);
  rg->valid = lattice2 ? 3 : rg->g.ghost;  // these ghost rows are current again
  rg->skipped = 0;
  if (as_stream(after) != rg->edge) {
    LBM_CHECK_HIP(hipEventRecord(rg->main_done, as_stream(after)));
    LBM_CHECK_HIP(hipStreamWaitEvent(rg->edge, rg->main_done, 0));
  }
  double* lats[2] = {lattice, lattice2};
  const int nl = lattice2 ? 2 : 1;
  const size_t count = msg * nl;
  for (int k = 0; k < nl; ++k) {
    if (rg->next >= 0) {
      int rc = lbm_halo_pack(rg->send_next + k * msg, lats[k], &rg->g, G, 1, rg->edge);
      if (rc) return rc;
    }
    if (rg->prev >= 0) {
      int rc = lbm_halo_pack(rg->send_prev + k * msg, lats[k], &rg->g, G, 0, rg->edge);
      if (rc) return rc;
    }
  }
  {
    const size_t np = rg->prev >= 0 ? count : 0, nn = rg->next >= 0 ? count : 0;
    int rc = ring_transfer(rg, rg->send_prev, np, rg->recv_prev, np, rg->send_next, nn, rg->recv_next, nn);
    if (rc) return rc;
  }
  for (int k = 0; k < nl; ++k) {
    if (rg->prev >= 0) {
      int rc = lbm_halo_unpack(lats[k], rg->recv_prev + k * msg, &rg->g, G, 0, rg->edge);
      if (rc) return rc;
    }
    if (rg->next >= 0) {
      int rc = lbm_halo_unpack(lats[k], rg->recv_next + k * msg, &rg->g, G, 1, rg->edge);
      if (rc) return rc;
    }
  }
  return LBM_OK;
}

int lbm_ring_exchange(lbm_ring* rg, double* lattice, lbm_stream_t after) {
  LBM_REQUIRE(rg && lattice, "lbm_ring_exchange: NULL argument");
  return ring_exchange(rg, lattice, nullptr, after);
}

int lbm_ring_exchange_full(lbm_ring* rg, double* lattice, lbm_stream_t after) {
  LBM_REQUIRE(rg && lattice, "lbm_ring_exchange_full: NULL argument");
  return ring_exchange(rg, lattice, nullptr, after, true);
}

int lbm_ring_exchange2(lbm_ring* rg, double* lattice_a, double* lattice_b, lbm_stream_t after) {
  LBM_REQUIRE(rg && lattice_a && lattice_b, "lbm_ring_exchange2: NULL argument");
  LBM_REQUIRE(rg->g.ghost == 3, "lbm_ring_exchange2: two-phase exchange needs 3 ghost rows (ring has %d)", rg->g.ghost);
  return ring_exchange(rg, lattice_a, lattice_b, after);
}

// One overlapped step of the two-phase (colour-gradient) slab: the fused one-launch step on the
// edge rows (edge stream) and on the interior rows (main stream, concurrently), ONE exchange of
// the 3 ghost rows of both colours behind the edge rows.  The slab geometry must carry 3 ghost
// rows; bc = the physical edges of the GLOBAL domain (NULL = the driver's walls,
// lbm_cg_default_bc), seams become HALO.
int lbm_ring_cg_step(lbm_ring* rg, double* dst_r, double* dst_b, const double* src_r,
                     const double* src_b, const lbm_bc* bc, const lbm_cg_params* prm, int edge_rows,
                     lbm_stream_t main_s) {
  LBM_REQUIRE(rg && dst_r && dst_b && src_r && src_b && prm, "lbm_ring_cg_step: NULL argument");
  const int R = rg->g.R, G = rg->g.ghost;
  LBM_REQUIRE(G == 3, "lbm_ring_cg_step: the two-phase step needs 3 ghost rows (ring has %d)", G);
  if (edge_rows < G) edge_rows = G;
  LBM_REQUIRE(2 * edge_rows < R, "lbm_ring_cg_step: edge_rows=%d too large for %d rows", edge_rows, R);
  hipStream_t main = as_stream(main_s);
  lbm_bc b;
  if (bc) b = *bc;
  else lbm_cg_default_bc(&b);
  if (rg->prev >= 0) b.row_lo = LBM_EDGE_HALO;
  if (rg->next >= 0) b.row_hi = LBM_EDGE_HALO;
  auto rows = [&](int r0, int r1, hipStream_t st) -> int {
    return lbm_cg_step_fused(dst_r, dst_b, src_r, src_b, &rg->g, &b, prm, r0, r1, nullptr, nullptr,
                             nullptr, nullptr, nullptr, st);
  };
  LBM_CHECK_HIP(hipEventRecord(rg->main_done, main));
  LBM_CHECK_HIP(hipStreamWaitEvent(rg->edge, rg->main_done, 0));
  int rc = LBM_OK;
  if (rg->prev >= 0 || rg->next >= 0) {
    if (tuning("ring_cg_parts", 1) && tuning("cg_tile", 4) == 4 && tuning("cg_strip", 0) == 0) {
      // round 4: TWO compute launches per step instead of three row ranges of two each -- the frame of the slab (its wall /
      // copy columns AND the first and last edge rows, whose results the neighbours wait for) on the ring's stream with the
      // exchange behind it, the inner rectangle on the caller's stream beside both
      rc = lbm_cg_step_fused_part(dst_r, dst_b, src_r, src_b, &rg->g, &b, prm, LBM_CG_PART_FRAME, edge_rows, nullptr, nullptr, nullptr,
                                  nullptr, nullptr, rg->edge);
      if (!rc) rc = lbm_cg_step_fused_part(dst_r, dst_b, src_r, src_b, &rg->g, &b, prm, LBM_CG_PART_INNER, edge_rows, nullptr, nullptr,
                                           nullptr, nullptr, nullptr, main);
    } else {
      rc = rows(0, edge_rows, rg->edge);
      if (!rc) rc = rows(R - edge_rows, R, rg->edge);
      if (!rc) rc = rows(edge_rows, R - edge_rows, main);  // interior overlaps the exchange
    }
    if (!rc) rc = ring_exchange(rg, dst_r, dst_b, rg->edge);
  } else {
    rc = rows(0, R, main);
  }
  if (rc) return rc;
  LBM_CHECK_HIP(hipEventRecord(rg->edge_done, rg->edge));
  LBM_CHECK_HIP(hipStreamWaitEvent(main, rg->edge_done, 0));
  return LBM_OK;
}

}  // extern "C"

// One launch-step of a slab with overlap: edge rows (edge stream), interior rows (main stream,
// concurrently), halo exchange of dst behind the edge rows.  `rows(r0, r1, stream)` launches the
// model's kernel on a row range.  On return `main` has been made to wait for everything: the next
// call may follow immediately.
template <class Rows, class Edges>
static int ring_step(lbm_ring* rg, double* dst, int edge_rows, hipStream_t main, Rows&& rows, Edges&& edges, bool full = false) {
  const int R = rg->g.R;
#ifdef LBM_EXPERIMENTS
  if (tuning("ring_edges_main", 0)) {
    // Variant (opt-in): edge rows on the caller's stream in front of the interior, only the exchange on the ring's
    // stream.  Measured at N = 1 with self send / recv (profiles/r02_ring_dissect.txt): what a launch-step loses against
    // one launch over all rows (1.92 ms) is NOT the exchange -- pack, RCCL and unpack together add nothing measurable --
    // but the fork / join itself: one event record + wait pair each way per step costs 0.11 ms with the default schedule
    // below and 0.18-0.20 ms with this one (device-scope release events change neither figure).
    const bool prof = rg->profile != 0;
    if (prof) LBM_CHECK_HIP(hipEventRecord(rg->t_edge0, main));
    int rc = edges(main);
    if (rc) return rc;
    if (prof) LBM_CHECK_HIP(hipEventRecord(rg->t_edge1, main));
    LBM_CHECK_HIP(hipEventRecord(rg->main_done, main));
    LBM_CHECK_HIP(hipStreamWaitEvent(rg->edge, rg->main_done, 0));
    if (prof) LBM_CHECK_HIP(hipEventRecord(rg->t_main0, main));
    rc = rows(edge_rows, R - edge_rows, main);
    if (rc) return rc;
    if (prof) LBM_CHECK_HIP(hipEventRecord(rg->t_main1, main));
    rc = ring_exchange(rg, dst, nullptr, rg->edge, full);
    if (rc) return rc;
    if (prof) LBM_CHECK_HIP(hipEventRecord(rg->t_xchg1, rg->edge));
    LBM_CHECK_HIP(hipEventRecord(rg->edge_done, rg->edge));
    LBM_CHECK_HIP(hipStreamWaitEvent(main, rg->edge_done, 0));
    return LBM_OK;
  }
#endif  // LBM_EXPERIMENTS
  // edge stream starts after everything previously enqueued on main (src complete)
  LBM_CHECK_HIP(hipEventRecord(rg->main_done, main));
  LBM_CHECK_HIP(hipStreamWaitEvent(rg->edge, rg->main_done, 0));
  const bool prof = rg->profile != 0;
  if (prof) LBM_CHECK_HIP(hipEventRecord(rg->t_edge0, rg->edge));
  int rc = edges(rg->edge);  // rows [0, edge_rows) and [R - edge_rows, R)
  if (rc) return rc;
  if (prof) {
    LBM_CHECK_HIP(hipEventRecord(rg->t_edge1, rg->edge));
    LBM_CHECK_HIP(hipEventRecord(rg->t_main0, main));
  }
  rc = rows(edge_rows, R - edge_rows, main);  // interior overlaps the exchange
  if (rc) return rc;
  if (prof) LBM_CHECK_HIP(hipEventRecord(rg->t_main1, main));
  rc = ring_exchange(rg, dst, nullptr, rg->edge, full);
  if (rc) return rc;
  if (prof) LBM_CHECK_HIP(hipEventRecord(rg->t_xchg1, rg->edge));
  LBM_CHECK_HIP(hipEventRecord(rg->edge_done, rg->edge));
  LBM_CHECK_HIP(hipStreamWaitEvent(main, rg->edge_done, 0));
  return LBM_OK;
}

// A launch that exchanges still READS n_steps ghost rows of src.  After launches without an exchange only rg->valid of them
// are current: a deeper launch than the rows left (depths 2,2,2,5 on 10 ghost rows leave 4) refreshes src's ghost rows first.
static int ring_refresh_if_used_up(lbm_ring* rg, const double* src, int n_steps, bool full, hipStream_t main) {
  if (!rg->skipped || rg->valid >= n_steps) return LBM_OK;
  int rc = ring_exchange(rg, const_cast<double*>(src), nullptr, main, full);
  if (rc) return rc;
  LBM_CHECK_HIP(hipEventRecord(rg->edge_done, rg->edge));
  LBM_CHECK_HIP(hipStreamWaitEvent(main, rg->edge_done, 0));
  return LBM_OK;
}

static int ring_bgk_step(lbm_ring* rg, double* dst, const double* src, const lbm_bc* bc,
                         const lbm_bgk_params* prm, int n_steps, int edge_rows, lbm_stream_t main_s, bool may_skip,
                         bool force_full = false) {
  LBM_REQUIRE(rg && dst && src && prm, "lbm_ring_bgk_step: NULL argument");
  const int R = rg->g.R, G = rg->g.ghost;
  LBM_REQUIRE(n_steps >= 1 && n_steps <= G, "lbm_ring_bgk_step: %d steps with %d ghost rows", n_steps, G);
  if (edge_rows < G) edge_rows = G;
  LBM_REQUIRE(2 * edge_rows < R, "lbm_ring_bgk_step: edge_rows=%d too large for %d rows", edge_rows, R);
  lbm_bc b = bc ? *bc : lbm_bc{0, 0, 0, 0, 0, 1.0, 1.0, 0.0, 0.0};
  if (rg->prev >= 0) b.row_lo = LBM_EDGE_HALO;
  if (rg->next >= 0) b.row_hi = LBM_EDGE_HALO;
  // ghost = m x n_steps on a closed ring: ONE exchange per m launches.  What a launch with an exchange costs over a
  // plain launch is the fork / join of the two streams (0.11 ms, profiles/r02_ring_dissect.txt), not the bytes, so the
  // m - 1 launches in between run as ONE plain launch on the caller's stream over the owned rows plus the ghost rows
  // the later launches of the period still read (n_steps fewer per side each time; 2 x 5 extra rows in 1024 at m = 2),
  // and the last one is the overlapped launch-step below with all m x n_steps ghost rows in the message.  Every rank
  // takes the same branch: the decision depends on the ring's shape (rg->closed -- not "this rank has both neighbours":
  // the middle ranks of a CHAIN have them too, its end ranks do not) and on the call sequence (rg->valid: ghost rows
  // current after the last exchange minus what launches without one have used up; depths may vary from call to call).
  if (may_skip && n_steps > 1 && rg->closed && tuning("ring_period", 0) != 1 && rg->valid >= 2 * n_steps) {
    // enough current ghost rows for this launch AND a later one: no exchange now.  e rows per side stay current
    const int e = rg->valid - n_steps;
    lbm_geom g2 = rg->g;
    g2.plane_stride = make_geom(rg->g).plane;
    g2.R = R + 2 * e;
    g2.ghost = G - e;
    rg->valid = e;
    rg->skipped += 1;
    return lbm_bgk_stream_collide_xn(dst, src, &g2, &b, prm, n_steps, 0, g2.R, main_s);
  }
  // walls + several steps per launch: the NEXT launch reads complete ghost rows (force_full: a neighbour that owns an
  // immersed-boundary band always sends and expects complete rows, whatever the columns are)
  // (G > 1, not n_steps > 1: a single-step launch may be followed by a multi-step one that reads what travels now)
  const bool full = force_full || (G > 1 && (bc_is_wall(b.row_lo) || bc_is_wall(b.row_hi) || bc_is_wall(b.col_lo) || bc_is_wall(b.col_hi)));
  {
    int rc = ring_refresh_if_used_up(rg, src, n_steps, full, as_stream(main_s));
    if (rc) return rc;
  }
  auto rows = [&](int r0, int r1, hipStream_t st) -> int {
    if (n_steps == 1) return lbm_bgk_stream_collide(dst, src, &rg->g, &b, prm, r0, r1, nullptr, nullptr, st);
    return lbm_bgk_stream_collide_xn(dst, src, &rg->g, &b, prm, n_steps, r0, r1, st);
  };
  auto edges = [&](hipStream_t st) -> int {  // both ends in ONE dispatch where the window kernel runs them
    if (n_steps > 1 && edge_rows <= 192) return lbm_bgk_stream_collide_xn2(dst, src, &rg->g, &b, prm, n_steps, 0, edge_rows, R - edge_rows, st);
    int rc = rows(0, edge_rows, st);
    return rc ? rc : rows(R - edge_rows, R, st);
  };
  return ring_step(rg, dst, edge_rows, as_stream(main_s), rows, edges, full);
}

extern "C" {

// BGK: n_steps = 1: single-step kernel (ghost >= 1); n_steps >= 2: sliding-window kernel
// (ghost >= n_steps; ghost = m x n_steps on a closed ring: one exchange per m launches).
int lbm_ring_bgk_step(lbm_ring* rg, double* dst, const double* src, const lbm_bc* bc,
                      const lbm_bgk_params* prm, int n_steps, int edge_rows, lbm_stream_t main_s) {
  return ring_bgk_step(rg, dst, src, bc, prm, n_steps, edge_rows, main_s, true);
}

// KBC: the same schedule (n_steps 1, or 2..4 through the sliding window with the reassociated
// collision)
int lbm_ring_kbc_step(lbm_ring* rg, double* dst, const double* src, const lbm_bc* bc,
                      const lbm_kbc_params* prm, int n_steps, int edge_rows, lbm_stream_t main_s) {
  LBM_REQUIRE(rg && dst && src && prm, "lbm_ring_kbc_step: NULL argument");
  const int R = rg->g.R, G = rg->g.ghost;
  LBM_REQUIRE(n_steps >= 1 && n_steps <= G && n_steps <= 4, "lbm_ring_kbc_step: %d steps with %d ghost rows (max 4)", n_steps, G);
  if (edge_rows < G) edge_rows = G;
  LBM_REQUIRE(2 * edge_rows < R, "lbm_ring_kbc_step: edge_rows=%d too large for %d rows", edge_rows, R);
  lbm_bc b = bc ? *bc : lbm_bc{0, 0, 0, 0, 0, 1.0, 1.0, 0.0, 0.0};
  if (rg->prev >= 0) b.row_lo = LBM_EDGE_HALO;
  if (rg->next >= 0) b.row_hi = LBM_EDGE_HALO;
  if (n_steps > 1 && rg->closed && tuning("ring_period", 0) != 1 && rg->valid >= 2 * n_steps) {
    // ghost = m x n_steps rows on a closed ring: no exchange on this launch (see ring_bgk_step)
    const int e = rg->valid - n_steps;
    lbm_geom g2 = rg->g;
    g2.plane_stride = make_geom(rg->g).plane;
    g2.R = R + 2 * e;
    g2.ghost = G - e;
    rg->valid = e;
    rg->skipped += 1;
    return lbm_kbc_stream_collide_xn(dst, src, &g2, &b, prm, n_steps, 0, g2.R, main_s);
  }
  {
    int rc = ring_refresh_if_used_up(rg, src, n_steps, false, as_stream(main_s));
    if (rc) return rc;
  }
  auto rows = [&](int r0, int r1, hipStream_t st) -> int {
    if (n_steps == 1) return lbm_kbc_stream_collide(dst, src, &rg->g, &b, prm, r0, r1, nullptr, nullptr, st);
    return lbm_kbc_stream_collide_xn(dst, src, &rg->g, &b, prm, n_steps, r0, r1, st);
  };
  auto edges = [&](hipStream_t st) -> int {
    int rc = rows(0, edge_rows, st);
    return rc ? rc : rows(R - edge_rows, R, st);
  };
  return ring_step(rg, dst, edge_rows, as_stream(main_s), rows, edges);
}

// One overlapped single-step launch of a BGK slab that may own an immersed boundary (config 5:
// cylinder_test.cpp:88-164 over slabs).  Three concurrent chains:
//   edge stream : the rows at both slab ends, then the halo exchange of dst;
//   aux stream  : (owning rank) the ROI rows with rho, u -> multi-direct forcing (lbm_ibm_force,
//                 ~10 small dependent launches) -> Guo source on the ROI of dst (:110-127);
//   main stream : every other row.
// The forcing chain is latency-bound (~0.3 ms for 942 markers) and would otherwise serialise
// behind the lattice update on the one rank that owns the boundary -- the straggler of a weak-
// scaling run.  The ROI never contains a slab's first or last row, so the rows that travel
// carry no source.  ib = NULL: ranks that do not own the boundary (rho, u unused).
// rho [R][C], u [2][R][C], written on the ROI rows only.
int lbm_ring_bgk_step_ibm(lbm_ring* rg, double* dst, const double* src, const lbm_bc* bc,
                          const lbm_bgk_params* prm, int edge_rows, lbm_ibm* ib, double guo_a,
                          double guo_b, double* rho, double* u, lbm_stream_t main_s) {
  LBM_REQUIRE(rg && dst && src && prm, "lbm_ring_bgk_step_ibm: NULL argument");
  LBM_REQUIRE(!ib || (rho && u), "lbm_ring_bgk_step_ibm: the owning rank needs rho and u buffers");
  const int R = rg->g.R, G = rg->g.ghost;
  LBM_REQUIRE(G >= 1, "lbm_ring_bgk_step_ibm: slab without ghost rows");
  if (edge_rows < 1) edge_rows = 1;
  LBM_REQUIRE(2 * edge_rows < R, "lbm_ring_bgk_step_ibm: edge_rows=%d too large for %d rows", edge_rows, R);
  hipStream_t main = as_stream(main_s);
  lbm_bc b = bc ? *bc : lbm_bc{0, 0, 0, 0, 0, 1.0, 1.0, 0.0, 0.0};
  if (rg->prev >= 0) b.row_lo = LBM_EDGE_HALO;
  if (rg->next >= 0) b.row_hi = LBM_EDGE_HALO;
  int q0 = 0, q1 = 0, c0 = 0, c1 = 0;  // ROI rows [q0, q1)
  if (ib) {
    int rc = lbm_ibm_roi(ib, &q0, &q1, &c0, &c1);
    if (rc) return rc;
    LBM_REQUIRE(q0 >= 1 && q1 <= R - 1 && q0 < q1, "lbm_ring_bgk_step_ibm: ROI rows [%d, %d) touch the slab edge", q0, q1);
    if (!rg->aux) {
      int lo = 0, hi = 0;
      LBM_CHECK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
      LBM_CHECK_HIP(hipStreamCreateWithPriority(&rg->aux, hipStreamNonBlocking, hi));
      LBM_CHECK_HIP(hipEventCreateWithFlags(&rg->aux_done, hipEventDisableTiming));
    }
  }
  auto rows = [&](int r0, int r1, bool mom, hipStream_t st) -> int {
    if (r0 >= r1) return LBM_OK;
    return lbm_bgk_stream_collide(dst, src, &rg->g, &b, prm, r0, r1, mom ? rho : nullptr, mom ? u : nullptr, st);
  };
  // rows [r0, r1) minus the ROI rows (which the aux chain computes)
  auto rows_outside_roi = [&](int r0, int r1, hipStream_t st) -> int {
    if (!ib || q1 <= r0 || q0 >= r1) return rows(r0, r1, false, st);
    int rc = rows(r0, q0 < r0 ? r0 : q0, false, st);
    if (!rc) rc = rows(q1 > r1 ? r1 : q1, r1, false, st);
    return rc;
  };
  LBM_CHECK_HIP(hipEventRecord(rg->main_done, main));
  LBM_CHECK_HIP(hipStreamWaitEvent(rg->edge, rg->main_done, 0));
  int rc = rows_outside_roi(0, edge_rows, rg->edge);
  if (!rc) rc = rows_outside_roi(R - edge_rows, R, rg->edge);
  if (!rc) rc = ring_exchange(rg, dst, nullptr, rg->edge);
  if (rc) return rc;
  LBM_CHECK_HIP(hipEventRecord(rg->edge_done, rg->edge));
  if (ib) {
    LBM_CHECK_HIP(hipStreamWaitEvent(rg->aux, rg->main_done, 0));
    rc = rows(q0, q1, true, rg->aux);
    if (!rc) rc = lbm_ibm_step(ib, dst, &rg->g, u, rho, prm->omega, guo_a, guo_b, rg->aux);
    if (rc) return rc;
    LBM_CHECK_HIP(hipEventRecord(rg->aux_done, rg->aux));
  }
  rc = rows_outside_roi(edge_rows, R - edge_rows, main);
  if (rc) return rc;
  LBM_CHECK_HIP(hipStreamWaitEvent(main, rg->edge_done, 0));
  if (ib) LBM_CHECK_HIP(hipStreamWaitEvent(main, rg->aux_done, 0));
  return LBM_OK;
}

// Phase timing of launch-steps (diagnosis of a scaling run): on = 1 makes lbm_ring_bgk_step /
// lbm_ring_kbc_step record timed events around the edge rows, the exchange (pack + send/recv +
// unpack) and the interior rows of every launch-step; lbm_ring_last_timing waits for the last one
// and returns {edge_rows_ms, exchange_ms, interior_ms, span_ms} (span: first edge row -> the later of
// exchange end / interior end).  Off (default) records nothing.
int lbm_ring_profile(lbm_ring* rg, int on) {
  LBM_REQUIRE(rg, "lbm_ring_profile: NULL ring");
  if (on && !rg->t_edge0)
    for (hipEvent_t* ev : {&rg->t_edge0, &rg->t_edge1, &rg->t_xchg1, &rg->t_main0, &rg->t_main1})
      LBM_CHECK_HIP(hipEventCreate(ev));
  rg->profile = on ? 1 : 0;
  return LBM_OK;
}

int lbm_ring_last_timing(lbm_ring* rg, double* out4) {
  LBM_REQUIRE(rg && out4, "lbm_ring_last_timing: NULL argument");
  LBM_REQUIRE(rg->profile && rg->t_edge0, "lbm_ring_last_timing: profiling is off (lbm_ring_profile)");
  LBM_CHECK_HIP(hipEventSynchronize(rg->t_xchg1));
  LBM_CHECK_HIP(hipEventSynchronize(rg->t_main1));
  float edge = 0, xchg = 0, inner = 0, span_x = 0, span_m = 0;
  LBM_CHECK_HIP(hipEventElapsedTime(&edge, rg->t_edge0, rg->t_edge1));
  LBM_CHECK_HIP(hipEventElapsedTime(&xchg, rg->t_edge1, rg->t_xchg1));
  LBM_CHECK_HIP(hipEventElapsedTime(&inner, rg->t_main0, rg->t_main1));
  LBM_CHECK_HIP(hipEventElapsedTime(&span_x, rg->t_edge0, rg->t_xchg1));
  LBM_CHECK_HIP(hipEventElapsedTime(&span_m, rg->t_edge0, rg->t_main1));
  out4[0] = edge;
  out4[1] = xchg;
  out4[2] = inner;
  out4[3] = span_x > span_m ? span_x : span_m;
  return LBM_OK;
}

// ---- config 5 over slabs at multi-step speed (capi_slab_ibm.hip holds the per-rank engine) ----------------
// One send + one recv per neighbour in one RCCL group on the ring's edge stream, ordered after `after`.
static int ring_sendrecv(lbm_ring* rg, const double* send_prev, size_t n_send_prev, double* recv_prev, size_t n_recv_prev,
                         const double* send_next, size_t n_send_next, double* recv_next, size_t n_recv_next,
                         hipStream_t after) {
  LBM_CHECK_HIP(hipEventRecord(rg->main_done, after));
  LBM_CHECK_HIP(hipStreamWaitEvent(rg->edge, rg->main_done, 0));
  {
    int rc = ring_transfer(rg, send_prev, rg->prev >= 0 ? n_send_prev : 0, recv_prev, rg->prev >= 0 ? n_recv_prev : 0, send_next,
                           rg->next >= 0 ? n_send_next : 0, recv_next, rg->next >= 0 ? n_recv_next : 0);
    if (rc) return rc;
  }
  LBM_CHECK_HIP(hipEventRecord(rg->edge_done, rg->edge));
  LBM_CHECK_HIP(hipStreamWaitEvent(after, rg->edge_done, 0));
  return LBM_OK;
}

// the priming exchange of lbm_slab_ibm_prime_* (once per run: its buffers are allocated here and freed
// again; co-owners swap all their owned band rows, which the ring's block-sized buffers cannot hold)
static int ring_ibm_prime(lbm_ring* rg, lbm_slab_ibm* sl, double* lattice, double* post, lbm_stream_t main_s);
int lbm_ring_ibm_prime(lbm_ring* rg, lbm_slab_ibm* sl, double* lattice, lbm_stream_t main_s) {
  return ring_ibm_prime(rg, sl, lattice, nullptr, main_s);
}
// the driver's first iteration over the ring: `pre` = pre-collision state (its ghost rows are filled
// here), `post` receives the post-collision state incl. forcing and source, ghost rows current
int lbm_ring_ibm_start(lbm_ring* rg, lbm_slab_ibm* sl, double* post, double* pre, lbm_stream_t main_s) {
  LBM_REQUIRE(post, "lbm_ring_ibm_start: NULL argument");
  return ring_ibm_prime(rg, sl, pre, post, main_s);
}
static int ring_ibm_prime(lbm_ring* rg, lbm_slab_ibm* sl, double* lattice, double* post, lbm_stream_t main_s) {
  LBM_REQUIRE(rg && sl && lattice, "lbm_ring_ibm_prime: NULL argument");
  LBM_REQUIRE((!sl->has_prev || rg->prev >= 0) && (!sl->has_next || rg->next >= 0),
              "lbm_ring_ibm_prime: the slab has a neighbour the ring does not know");
  LBM_REQUIRE(rg->g.ghost == sl->D, "lbm_ring_ibm_prime: ring with %d ghost rows, blocks of %d steps", rg->g.ghost, sl->D);
  hipStream_t main = as_stream(main_s);
  long long cnt[2][2];  // [side][send / recv]
  for (int side = 0; side < 2; ++side) {
    int rc = lbm_slab_ibm_prime_counts(sl, side, &cnt[side][0], &cnt[side][1]);
    if (rc) return rc;
  }
  double* buf[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
  hipError_t e = hipSuccess;
  for (int side = 0; side < 2; ++side)
    for (int k = 0; k < 2; ++k)
      if (cnt[side][k] > 0 && e == hipSuccess) e = hipMalloc(&buf[side][k], (size_t)cnt[side][k] * sizeof(double));
  int rc = LBM_OK;
  if (e != hipSuccess) {
    set_error("lbm_ring_ibm_prime: %s", hipGetErrorString(e));
    rc = LBM_ERR_HIP;
  }
  if (!rc) rc = lbm_slab_ibm_prime_pack(sl, lattice, buf[0][0], buf[1][0], main_s);
  if (!rc) rc = ring_sendrecv(rg, buf[0][0], (size_t)cnt[0][0], buf[0][1], (size_t)cnt[0][1], buf[1][0], (size_t)cnt[1][0],
                              buf[1][1], (size_t)cnt[1][1], main);
  if (!rc) rc = post ? lbm_slab_ibm_start_finish(sl, post, lattice, buf[0][1], buf[1][1], main_s)
                     : lbm_slab_ibm_prime_finish(sl, lattice, buf[0][1], buf[1][1], main_s);
  if (!rc && hipStreamSynchronize(main) != hipSuccess) {
    set_error("lbm_ring_ibm_prime: stream synchronisation failed");
    rc = LBM_ERR_HIP;
  }
  for (int side = 0; side < 2; ++side)
    for (int k = 0; k < 2; ++k)
      if (buf[side][k]) (void)hipFree(buf[side][k]);
  return rc;
}

// One block of sl->D steps of a BGK slab in the ring.  Slabs without valid band rows run the overlapped
// schedule of lbm_ring_bgk_step (edge rows -> exchange of complete ghost rows beside the interior);
// (co-)owners run the band chain beside their far rows and exchange behind both -- the band chain
// (D dependent forced steps, latency-bound) is their critical path either way.
int lbm_ring_bgk_block_ibm(lbm_ring* rg, lbm_slab_ibm* sl, double* dst, const double* src, int edge_rows,
                           lbm_stream_t main_s) {
  LBM_REQUIRE(rg && sl && dst && src, "lbm_ring_bgk_block_ibm: NULL argument");
  LBM_REQUIRE(rg->g.R == sl->g.R && rg->g.C == sl->g.C && rg->g.ghost == sl->g.ghost && rg->g.ghost == sl->D,
              "lbm_ring_bgk_block_ibm: ring and slab geometries differ (ghost rows must equal the block depth)");
  LBM_REQUIRE((!sl->has_prev || rg->prev >= 0) && (!sl->has_next || rg->next >= 0),
              "lbm_ring_bgk_block_ibm: the slab has a neighbour the ring does not know");
  // (every launch with its exchange: the band's co-owners exchange every block too)
  // (complete ghost rows on every seam: that is what an owner next door sends and unpacks, also with periodic columns)
  if (!sl->owner) return ring_bgk_step(rg, dst, src, &sl->bc_global, &sl->prm, sl->D, edge_rows, main_s, false, true);
  hipStream_t main = as_stream(main_s);
  int rc = lbm_slab_ibm_block_compute(sl, dst, src, rg->send_prev, rg->send_next, main_s);
  if (rc) return rc;
  const size_t msg = (size_t)lbm_slab_ibm_msg_doubles(sl);
  LBM_REQUIRE(msg <= rg->bufsz, "lbm_ring_bgk_block_ibm: messages of %zu doubles, ring buffers of %zu", msg, rg->bufsz);
  rc = ring_sendrecv(rg, rg->send_prev, msg, rg->recv_prev, msg, rg->send_next, msg, rg->recv_next, msg, main);
  if (rc) return rc;
  return lbm_slab_ibm_block_finish(sl, dst, rg->recv_prev, rg->recv_next, main_s);
}

// ---- pressure-periodic rows over the (periodic) ring: capi_slab_pressure.hip holds the engine ---------------------
static int ring_pressure_start(lbm_ring* rg, lbm_slab_pressure* sl, double* post, double* pre, const double* m0, const double* m1,
                               lbm_stream_t main_s);
int lbm_ring_pressure_start(lbm_ring* rg, lbm_slab_pressure* sl, double* post, double* pre, lbm_stream_t main_s) {
  return ring_pressure_start(rg, sl, post, pre, nullptr, nullptr, main_s);
}
int lbm_ring_pressure_start_kbc(lbm_ring* rg, lbm_slab_pressure* sl, double* post, double* pre, const double* m0,
                                const double* m1, lbm_stream_t main_s) {
  LBM_REQUIRE(m0 && m1, "lbm_ring_pressure_start_kbc: NULL moments");
  return ring_pressure_start(rg, sl, post, pre, m0, m1, main_s);
}
static int ring_pressure_start(lbm_ring* rg, lbm_slab_pressure* sl, double* post, double* pre, const double* m0, const double* m1,
                               lbm_stream_t main_s) {
  LBM_REQUIRE(rg && sl && post && pre, "lbm_ring_pressure_start: NULL argument");
  LBM_REQUIRE(rg->prev >= 0 && rg->next >= 0 && rg->nranks >= 2, "lbm_ring_pressure_start: needs a periodic ring of at least 2 slabs");
  {
    int R = 0, C = 0, ghost = 0, D = 0;
    int rc = lbm_slab_pressure_info(sl, &R, &C, &ghost, &D);
    if (rc) return rc;
    LBM_REQUIRE(rg->g.R == R && rg->g.C == C && rg->g.ghost == ghost && ghost >= D,
                "lbm_ring_pressure_start: ring (%d x %d, %d ghost rows) and slab (%d x %d, %d ghost rows, blocks of %d) differ",
                rg->g.R, rg->g.C, rg->g.ghost, R, C, ghost, D);
  }
  const size_t n_prev = (size_t)lbm_slab_pressure_msg_doubles(sl, 0, 1), n_next = (size_t)lbm_slab_pressure_msg_doubles(sl, 1, 1);
  // (the partner of a seam sends what this side receives: the start-up messages across the pressure seam are both 2 D rows)
  double* buf[4] = {nullptr, nullptr, nullptr, nullptr};
  const size_t cnt[4] = {n_prev, n_prev, n_next, n_next};  // send_prev, recv_prev, send_next, recv_next
  hipError_t e = hipSuccess;
  for (int k = 0; k < 4 && e == hipSuccess; ++k) e = hipMalloc(&buf[k], cnt[k] * sizeof(double));
  int rc = LBM_OK;
  if (e != hipSuccess) {
    set_error("lbm_ring_pressure_start: %s", hipGetErrorString(e));
    rc = LBM_ERR_HIP;
  }
  hipStream_t main = as_stream(main_s);
  if (!rc) rc = m0 ? lbm_slab_pressure_start_pack_kbc(sl, pre, m0, m1, buf[0], buf[2], main_s) : lbm_slab_pressure_start_pack(sl, pre, buf[0], buf[2], main_s);
  if (!rc) rc = ring_sendrecv(rg, buf[0], n_prev, buf[1], n_prev, buf[2], n_next, buf[3], n_next, main);
  if (!rc) rc = m0 ? lbm_slab_pressure_start_finish_kbc(sl, post, pre, m0, m1, buf[1], buf[3], main_s)
                   : lbm_slab_pressure_start_finish(sl, post, pre, buf[1], buf[3], main_s);
  if (!rc && m0) {  // KBC: the collision on held moments left the ghost rows of `post` behind: complete halos over every seam
    rc = ring_exchange(rg, post, nullptr, main, true);
    if (!rc) {  // (errors become rc: the common tail below still syncs and frees the four start-up buffers)
      hipError_t ej = hipEventRecord(rg->edge_done, rg->edge);
      if (ej == hipSuccess) ej = hipStreamWaitEvent(main, rg->edge_done, 0);
      if (ej != hipSuccess) {
        set_error("lbm_ring_pressure_start: %s", hipGetErrorString(ej));
        rc = LBM_ERR_HIP;
      }
    }
  }
  // the start-up buffers are freed below: nothing enqueued on them may still be running -- on the error paths too
  hipError_t es = hipStreamSynchronize(main);
  if (es == hipSuccess && rg->edge) es = hipStreamSynchronize(rg->edge);
  if (!rc && es != hipSuccess) {
    set_error("lbm_ring_pressure_start: stream synchronisation failed");
    rc = LBM_ERR_HIP;
  }
  for (double* b : buf)
    if (b) (void)hipFree(b);
  return rc;
}

int lbm_ring_bgk_block_pressure(lbm_ring* rg, lbm_slab_pressure* sl, double* dst, const double* src, lbm_stream_t main_s) {
  LBM_REQUIRE(rg && sl && dst && src, "lbm_ring_bgk_block_pressure: NULL argument");
  LBM_REQUIRE(rg->prev >= 0 && rg->next >= 0, "lbm_ring_bgk_block_pressure: needs a periodic ring");
  const size_t msg = (size_t)lbm_slab_pressure_msg_doubles(sl, 0, 0);
  {
    int R = 0, C = 0, ghost = 0, D = 0;
    int rc = lbm_slab_pressure_info(sl, &R, &C, &ghost, &D);
    if (rc) return rc;
    LBM_REQUIRE(rg->g.R == R && rg->g.C == C && rg->g.ghost == ghost && ghost >= D && msg <= rg->bufsz,
                "lbm_ring_bgk_block_pressure: ring (%d x %d, %d ghost rows) and slab (%d x %d, %d ghost rows, blocks of %d) differ",
                rg->g.R, rg->g.C, rg->g.ghost, R, C, ghost, D);
  }
  hipStream_t main = as_stream(main_s);
  int rc = lbm_slab_pressure_block_compute(sl, dst, src, rg->send_prev, rg->send_next, main_s);
  if (rc) return rc;
  rc = ring_sendrecv(rg, rg->send_prev, msg, rg->recv_prev, msg, rg->send_next, msg, rg->recv_next, msg, main);
  if (rc) return rc;
  return lbm_slab_pressure_block_finish(sl, dst, rg->recv_prev, rg->recv_next, main_s);
}

// make `main` wait for an exchange enqueued with lbm_ring_exchange (initial ghost fill)
int lbm_ring_join(lbm_ring* rg, lbm_stream_t main_s) {
  LBM_REQUIRE(rg, "lbm_ring_join: NULL ring");
  LBM_CHECK_HIP(hipEventRecord(rg->edge_done, rg->edge));
  LBM_CHECK_HIP(hipStreamWaitEvent(as_stream(main_s), rg->edge_done, 0));
  return LBM_OK;
}

}  // extern "C"
