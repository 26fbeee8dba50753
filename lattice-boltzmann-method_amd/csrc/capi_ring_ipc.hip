// The peer-mapped transport of the slab ring (ring_ipc.hpp): one process per GPU -- or several on one GPU --,
// every rank owns a receive WINDOW in device memory that its two neighbours map through hipIpcMemHandle_t.
//
//   window = 4 sequence words (128 bytes apart) + [side][2 slots][slot_doubles]
//     data_seq[side]: written by the neighbour on that side -- "message k lies complete in slot k % 2"
//     ack_seq[side] : written by the neighbour on that side -- "I have copied YOUR message k out of my window"
//
// A message k to the neighbour on side s: (wait until it has acknowledged k - 2: that slot is free) -> k_ipc_put
// stores the bytes into ITS window's slot [opposite s][k % 2] and, from the last workgroup to finish, k into its
// data_seq[opposite s].  Receiving message k from side s: k_ipc_wait polls my data_seq[s] (ONE wave, bounded by a
// wall-clock limit: a neighbour that never delivers raises the status word and the chain still drains), k_ipc_get
// copies the slot into the caller's buffer and writes k into the neighbour's ack_seq[opposite s].
//
// Visibility does not lean on cache maintenance: every byte that crosses a process / device boundary is stored and
// loaded at SYSTEM scope (sc0 sc1 accesses, MI355X_MICROARCH.md "Valid forms"), drained before the sequence word is
// written, and the window is allocated uncached.  The halo logic on top (what travels, when) is the ring's and is
// the same for both transports; the reference contract is the block binding of test/decompose_domain.cpp:181-187.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <functional>
#include <new>
#include <string>
#include <thread>

#include "internal.hpp"
#include "ring_ipc.hpp"

namespace lbm {

namespace {

constexpr int kMaxRanks = 64;
constexpr size_t kHeaderBytes = 1024;
constexpr size_t kSeqStride = 128 / sizeof(uint64_t);  // sequence words sit on cache lines of their own

struct Rendezvous {  // POSIX shared memory, zero-filled by the kernel when first created
  std::atomic<int> published[kMaxRanks];
  std::atomic<int> opened, closing;
  hipIpcMemHandle_t handle[kMaxRanks];
  long long pid[kMaxRanks];
  unsigned long long slot_doubles[kMaxRanks];
};

__device__ inline uint64_t load_sys(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ inline void store_sys(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

__device__ inline int status_sys(const int* status) { return __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// ONE wave; lane 0 polls until *flag >= want or `limit` wall-clock ticks have passed.  The failure is STICKY: once the
// status word is raised (by this or an earlier wait) every later wait of the queue returns at once, so a queue of N
// launches behind a dead neighbour drains in ONE time limit, not in 4 N of them.
__global__ __launch_bounds__(64) void k_ipc_wait(const uint64_t* flag, uint64_t want, unsigned long long limit, int* status, int code) {
  if (threadIdx.x == 0) {
    if (status_sys(status) != 0) return;
    const unsigned long long t0 = wall_clock64();
    while (load_sys(flag) < want) {
      __builtin_amdgcn_s_sleep(16);
      if (wall_clock64() - t0 > limit) {
        int none = 0;  // keep the FIRST failure
        __hip_atomic_compare_exchange_strong(status, &none, code, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
  }
}

// the last workgroup to arrive publishes `seq` (every store / load of this launch has completed by then) -- unless the
// ring has failed: then nothing is announced (a slot that was not filled must not be read as a message)
__device__ inline void publish_when_all_done(uint64_t* word, uint64_t seq, unsigned* arrived, const int* status) {
  __atomic_thread_fence(__ATOMIC_RELEASE);  // system scope: this wave's accesses have drained
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned n = __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (n + 1 == gridDim.x) {
      __hip_atomic_store(arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // the status word never returns to 0: if it is 0 now, every workgroup of this launch saw 0 and copied
      if (status_sys(status) == 0) __hip_atomic_store(word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// after a failed wait the copies are skipped too (the data is void either way; the queue only has to drain)
__global__ __launch_bounds__(256) void k_ipc_put(uint64_t* __restrict__ remote, const uint64_t* __restrict__ local, size_t n,
                                                 uint64_t* remote_seq, uint64_t seq, unsigned* arrived, const int* status) {
  if (status_sys(status) == 0)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) store_sys(remote + i, local[i]);
  publish_when_all_done(remote_seq, seq, arrived, status);
}

__global__ __launch_bounds__(256) void k_ipc_get(uint64_t* __restrict__ local, const uint64_t* __restrict__ slot, size_t n,
                                                 uint64_t* remote_ack, uint64_t seq, unsigned* arrived, const int* status) {
  if (status_sys(status) == 0)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) local[i] = load_sys(slot + i);
  publish_when_all_done(remote_ack, seq, arrived, status);
}

}  // namespace

struct IpcTransport {
  int rank, nranks, nbr[2];            // nbr[0] = previous, nbr[1] = next (-1: none)
  size_t slot;                         // doubles per slot
  char* window;                        // my receive window (device)
  char* peer[2];                       // the neighbours' windows as mapped here
  bool peer_opened[2];                 // mapped by hipIpcOpenMemHandle (not my own, not shared with the other side)
  uint64_t tx[2], rx[2];               // messages sent to / received from each side so far
  unsigned* arrived;                   // [4] arrival counters of the put / get launches (device)
  bool window_cached;                  // the uncached allocation was refused and the caller accepted a cached window
  int* status_h;                       // pinned, mapped: first wait that gave up
  int* status_d;
  unsigned long long limit_ticks;
  Rendezvous* rv;
  std::string shm_name;
};

namespace {

uint64_t* seq_word(char* win, int which /* 0 data, 1 ack */, int side) { return reinterpret_cast<uint64_t*>(win) + (which * 2 + side) * kSeqStride; }
uint64_t* slot_ptr(char* win, size_t slot, int side, uint64_t k) {
  return reinterpret_cast<uint64_t*>(win + kHeaderBytes) + ((size_t)side * 2 + (size_t)(k & 1)) * slot;
}

bool wait_for(const std::function<bool()>& done, double timeout_s) {
  const auto t0 = std::chrono::steady_clock::now();
  while (!done()) {
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
  return true;
}

}  // namespace

void ipc_destroy(IpcTransport* t) {
  if (!t) return;
  (void)hipDeviceSynchronize();
  if (t->rv) {
    // nobody unmaps or frees while a neighbour's last acknowledgement may still be on its way
    t->rv->closing.fetch_add(1);
    (void)wait_for([&] { return t->rv->closing.load() >= t->nranks; }, 10.0);
  }
  for (int s = 0; s < 2; ++s)
    if (t->peer_opened[s] && t->peer[s]) (void)hipIpcCloseMemHandle(t->peer[s]);
  if (t->window) (void)hipFree(t->window);
  if (t->arrived) (void)hipFree(t->arrived);
  if (t->status_h) (void)hipHostFree(t->status_h);
  if (t->rv) munmap(t->rv, sizeof(Rendezvous));
  if (!t->shm_name.empty()) shm_unlink(t->shm_name.c_str());
  delete t;
}

int ipc_create(IpcTransport** out, const unsigned char* id128, int rank, int nranks, int prev, int next, size_t slot_doubles) {
  LBM_REQUIRE(nranks <= kMaxRanks, "lbm_ring (peer-mapped transport): at most %d ranks (got %d)", kMaxRanks, nranks);
  IpcTransport* t = new (std::nothrow) IpcTransport();
  LBM_REQUIRE(t, "lbm_ring: out of host memory");
  t->rank = rank;
  t->nranks = nranks;
  t->nbr[0] = prev;
  t->nbr[1] = next;
  t->slot = (slot_doubles + 31) / 32 * 32;
  t->window = nullptr;
  t->peer[0] = t->peer[1] = nullptr;
  t->peer_opened[0] = t->peer_opened[1] = false;
  t->window_cached = false;
  t->tx[0] = t->tx[1] = t->rx[0] = t->rx[1] = 0;
  t->arrived = nullptr;
  t->status_h = t->status_d = nullptr;
  t->rv = nullptr;
  auto fail = [&](int rc) {
    ipc_destroy(t);
    return rc;
  };
  const size_t bytes = kHeaderBytes + 4 * t->slot * sizeof(double);
  // uncached: nothing of a window may linger in this GPU's L2 while a neighbour rewrites it.  A runtime that refuses
  // the uncached allocation fails the creation -- unless the caller has said that a cached window will do
  // ("ring_ipc_cached_ok" = 1: every access to it is a system-scope access anyway; the uncached attribute is the second
  // line of defence), and then the ring says so (lbm_ring_window_cached).  "ring_ipc_force_cached" = 1 skips the
  // uncached attempt (tests).
  hipError_t e = tuning("ring_ipc_force_cached", 0) ? hipErrorOutOfMemory
                                                    : hipExtMallocWithFlags((void**)&t->window, bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    if (!tuning("ring_ipc_cached_ok", 0)) {
      set_error("lbm_ring (peer-mapped transport): the uncached allocation of the receive window was refused (%s); "
                "lbm_set_tuning(\"ring_ipc_cached_ok\", 1) accepts a cached window", hipGetErrorString(e));
      t->window = nullptr;
      return fail(LBM_ERR_HIP);
    }
    t->window_cached = true;
    e = hipMalloc((void**)&t->window, bytes);
  }
  if (e == hipSuccess) e = hipMemset(t->window, 0, kHeaderBytes);
  if (e == hipSuccess) e = hipMalloc((void**)&t->arrived, 4 * sizeof(unsigned));
  if (e == hipSuccess) e = hipMemset(t->arrived, 0, 4 * sizeof(unsigned));
  if (e == hipSuccess) e = hipHostMalloc((void**)&t->status_h, sizeof(int), hipHostMallocMapped);
  if (e == hipSuccess) {
    *t->status_h = 0;
    e = hipHostGetDevicePointer((void**)&t->status_d, t->status_h, 0);
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    set_error("lbm_ring (peer-mapped transport): %s", hipGetErrorString(e));
    return fail(LBM_ERR_HIP);
  }
  int dev = 0, khz = 0;
  (void)hipGetDevice(&dev);
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;
  const int ms = tuning("ring_ipc_timeout_ms", 20000);
  t->limit_ticks = (unsigned long long)khz * (unsigned long long)(ms > 0 ? ms : 20000);

  bool remote = false;
  for (int s = 0; s < 2; ++s)
    if (t->nbr[s] >= 0 && t->nbr[s] != rank) remote = true;
  for (int s = 0; s < 2; ++s)
    if (t->nbr[s] == rank) t->peer[s] = t->window;  // self ring: my own window
  if (!remote) {
    *out = t;
    return LBM_OK;
  }

  // ---- host rendezvous: publish my handle, map the neighbours' ----
  char name[64] = "/lbm_ring_";
  for (int i = 0; i < 16; ++i) std::snprintf(name + 10 + 2 * i, 3, "%02x", id128[i] ^ id128[i + 16] ^ id128[i + 32] ^ id128[i + 48]);
  t->shm_name = name;
  const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, sizeof(Rendezvous)) != 0) {
    set_error("lbm_ring (peer-mapped transport): shm_open(%s) failed", name);
    if (fd >= 0) close(fd);
    return fail(LBM_ERR_HIP);
  }
  void* m = mmap(nullptr, sizeof(Rendezvous), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) {
    set_error("lbm_ring (peer-mapped transport): mmap of the rendezvous segment failed");
    return fail(LBM_ERR_HIP);
  }
  t->rv = static_cast<Rendezvous*>(m);
  e = hipIpcGetMemHandle(&t->rv->handle[rank], t->window);
  if (e != hipSuccess) {
    set_error("hipIpcGetMemHandle failed: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 must be set on this driver)", hipGetErrorString(e));
    return fail(LBM_ERR_HIP);
  }
  t->rv->pid[rank] = (long long)getpid();
  t->rv->slot_doubles[rank] = t->slot;
  t->rv->published[rank].store(1, std::memory_order_release);
  for (int s = 0; s < 2; ++s) {
    const int p = t->nbr[s];
    if (p < 0 || p == rank) continue;
    if (s == 1 && p == t->nbr[0]) {  // two ranks: both neighbours are the same peer, one mapping
      t->peer[1] = t->peer[0];
      continue;
    }
    if (!wait_for([&] { return t->rv->published[p].load(std::memory_order_acquire) != 0; }, 120.0)) {
      set_error("lbm_ring (peer-mapped transport): rank %d never published its window", p);
      return fail(LBM_ERR_HIP);
    }
    if (t->rv->pid[p] == (long long)getpid()) {
      set_error("lbm_ring (peer-mapped transport): ranks %d and %d live in one process (one process per rank)", rank, p);
      return fail(LBM_ERR_INVALID);
    }
    if (t->rv->slot_doubles[p] != t->slot) {
      set_error("lbm_ring (peer-mapped transport): rank %d has slots of %llu doubles, rank %d of %zu (same columns and ghost rows everywhere)",
                p, t->rv->slot_doubles[p], rank, t->slot);
      return fail(LBM_ERR_INVALID);
    }
    void* ptr = nullptr;
    e = hipIpcOpenMemHandle(&ptr, t->rv->handle[p], hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      set_error("hipIpcOpenMemHandle(window of rank %d) failed: %s", p, hipGetErrorString(e));
      return fail(LBM_ERR_HIP);
    }
    t->peer[s] = static_cast<char*>(ptr);
    t->peer_opened[s] = true;
  }
  // everybody has mapped what it needs: the name can go (the mapping stays for the closing barrier)
  t->rv->opened.fetch_add(1);
  if (!wait_for([&] { return t->rv->opened.load() >= nranks; }, 120.0)) {
    set_error("lbm_ring (peer-mapped transport): only %d of %d ranks arrived", t->rv->opened.load(), nranks);
    return fail(LBM_ERR_HIP);
  }
  shm_unlink(name);
  t->shm_name.clear();
  *out = t;
  return LBM_OK;
}

int ipc_status(const IpcTransport* t) { return t && t->status_h ? *(volatile int*)t->status_h : 0; }
int ipc_window_cached(const IpcTransport* t) { return t && t->window_cached ? 1 : 0; }

int ipc_sendrecv(IpcTransport* t, const double* send_prev, size_t n_send_prev, double* recv_prev, size_t n_recv_prev,
                 const double* send_next, size_t n_send_next, double* recv_next, size_t n_recv_next, hipStream_t st) {
  const double* sbuf[2] = {send_prev, send_next};
  double* rbuf[2] = {recv_prev, recv_next};
  size_t ns[2] = {n_send_prev, n_send_next}, nr[2] = {n_recv_prev, n_recv_next};
  size_t pieces = 0;
  for (int s = 0; s < 2; ++s) {
    if (t->nbr[s] < 0) ns[s] = nr[s] = 0;
    pieces = std::max(pieces, std::max((ns[s] + t->slot - 1) / t->slot, (nr[s] + t->slot - 1) / t->slot));
  }
  auto grid = [](size_t n) { return (unsigned)std::min<size_t>(256, (n + 1023) / 1024); };
  // piece by piece, puts in front of gets: a put waits only for an acknowledgement, an acknowledgement only for a get,
  // a get only for the matching put of the neighbour -- no cycle however the counts differ per direction
  for (size_t c = 0; c < pieces; ++c) {
    for (int s = 1; s >= 0; --s) {
      const size_t off = c * t->slot;
      if (off >= ns[s]) continue;
      const size_t len = std::min(t->slot, ns[s] - off);
      const uint64_t k = ++t->tx[s];
      const int o = 1 - s;  // I am on the neighbour's opposite side
      if (k > 2) LBM_KLAUNCH(k_ipc_wait, dim3(1), dim3(64), 0, st, seq_word(t->window, 1, s), k - 2, t->limit_ticks, t->status_d, 2);
      LBM_KLAUNCH(k_ipc_put, dim3(grid(len)), dim3(256), 0, st, slot_ptr(t->peer[s], t->slot, o, k),
                  reinterpret_cast<const uint64_t*>(sbuf[s] + off), len, seq_word(t->peer[s], 0, o), k, t->arrived + s, t->status_d);
    }
    for (int s = 0; s < 2; ++s) {
      const size_t off = c * t->slot;
      if (off >= nr[s]) continue;
      const size_t len = std::min(t->slot, nr[s] - off);
      const uint64_t k = ++t->rx[s];
      LBM_KLAUNCH(k_ipc_wait, dim3(1), dim3(64), 0, st, seq_word(t->window, 0, s), k, t->limit_ticks, t->status_d, 1);
      LBM_KLAUNCH(k_ipc_get, dim3(grid(len)), dim3(256), 0, st, reinterpret_cast<uint64_t*>(rbuf[s] + off),
                  slot_ptr(t->window, t->slot, s, k), len, seq_word(t->peer[s], 1, 1 - s), k, t->arrived + 2 + s, t->status_d);
    }
  }
  LBM_CHECK_LAUNCH();
  return LBM_OK;
}

}  // namespace lbm
