// api.cpp -- the extern "C" entry points of include/xmpi.h.  Compiled by hipcc as host code.
#include <signal.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "comm.h"
#include "kernels.h"
#include "sched_steps.h"

namespace xmpi {

static thread_local std::string g_last_error;
thread_local uint64_t t_api_call = 0;  // the number of the public call this thread is in (XMPI_ENTER)

void set_last_error(const std::string& s) { g_last_error = s; }

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  const char* base = strrchr(file, '/');
  g_last_error = std::string(what) + " failed: " + hipGetErrorString(e) + " (" + (base ? base + 1 : file) + ":" +
                 std::to_string(line) + ")";
  (void)hipGetLastError();
  return XMPI_ERR_HIP;
}

// XMPI_TRACE=1: one line per bootstrap step on stderr (where does a rank that hangs in Init hang?)
static bool trace_on() {
  static const bool on = getenv("XMPI_TRACE") && atoi(getenv("XMPI_TRACE")) != 0;
  return on;
}
#define XMPI_TRACE_STEP(rank, what)                                                                          \
  do {                                                                                                       \
    if (::xmpi::trace_on()) fprintf(stderr, "[xmpi %d %.6f] %s\n", (rank), ::xmpi::now_seconds(), (what)); \
  } while (0)

static long env_long(const char* name, long dflt) {
  const char* v = getenv(name);
  if (!v || !*v) return dflt;
  char* end = nullptr;
  long x = strtol(v, &end, 0);
  return (end && end != v) ? x : dflt;
}

static int use_device(const xmpi_comm* c) {
  // HIP's current device is per OS thread and cgo moves goroutines between threads
  XMPI_HIP(hipSetDevice(c->device));
  return XMPI_OK;
}

#define XMPI_ENTER(c)                          \
  do {                                         \
    if (!(c) || (c)->finalized) {              \
      ::xmpi::set_last_error("communicator not initialised"); \
      return XMPI_ERR_STATE;                   \
    }                                          \
    int _rc = ::xmpi::use_device(c);           \
    if (_rc) return _rc;                       \
    ::xmpi::t_api_call = (c)->api_calls.fetch_add(1, std::memory_order_relaxed) + 1; \
  } while (0)

// no-progress limit of a steady-state wait: XMPI_TIMEOUT_S, or for ever
static double wait_limit(const xmpi_comm* c) { return c->timeout_s > 0 ? (double)c->timeout_s : 1e18; }

static size_t choose_piece(const xmpi_comm* c, size_t bytes_per_rank_chunk) {
  if (c->piece_bytes > 0) return std::min<size_t>((size_t)c->piece_bytes, c->slot_bytes);
  // ranks sharing one in-order stream have nothing to overlap: one launch per chunk
  bool all_coloc = c->shared_stream;
  for (int p = 0; p < c->size && all_coloc; p++)
    if (p != c->rank && !c->peer_coloc[p]) all_coloc = false;
  if (all_coloc) return c->slot_bytes;
  // ~4 pieces per chunk so a chunk's transfer overlaps its reduction, but never below 1 MiB (a piece
  // costs a launch and an event: ~10 us, i.e. ~0.5 MiB of link time), within [1 MiB, slot]
  size_t p = 1u << 20;
  while (p * 4 < bytes_per_rank_chunk && p < c->slot_bytes) p <<= 1;
  return std::min(p, c->slot_bytes);
}

// ---- windows and their mappings outlive communicators (shared with dsync.cpp) -----------------------------------------------
// Measured on MI355X / ROCm 7.2: memory that was exported with hipIpcGetMemHandle and mapped by another process
// is NOT given back by hipFree + hipIpcCloseMemHandle while both processes live -- 8 ranks that create and
// finalise a communicator in a loop lost 10 GiB of HBM per lifetime (8 windows of 1.25 GiB) and ran out after 27.
// So nothing of that kind is freed or unmapped per communicator any more: a finalised communicator's window (and
// flag page) goes into a per-process pool and the next communicator of that size takes it from there; a peer's
// mapping of it stays open and is found again by {owner pid, address, handle}.  This also removes the one moment
// where a stray write could meet an unmapped page ("the round-1 fault").
struct IpcMapping {
  int owner_pid;
  uint64_t owner_addr;
  uint8_t handle[64];
  void* ptr;
  int refs;
};
static std::mutex g_ipc_mu;
static std::vector<IpcMapping> g_ipc_map;

hipError_t ipc_open_shared(int owner_pid, uint64_t owner_addr, const void* handle_bytes, void** out) {
  std::lock_guard<std::mutex> g(g_ipc_mu);
  for (size_t i = 0; i < g_ipc_map.size(); i++) {
    IpcMapping& m = g_ipc_map[i];
    if (m.owner_pid != owner_pid || m.owner_addr != owner_addr) continue;
    if (memcmp(m.handle, handle_bytes, sizeof(hipIpcMemHandle_t)) == 0) {
      m.refs++;
      *out = m.ptr;
      return hipSuccess;
    }
    if (m.refs == 0) {  // the owner has put another allocation at that address: the old mapping is dead
      (void)hipIpcCloseMemHandle(m.ptr);
      (void)hipGetLastError();
      g_ipc_map.erase(g_ipc_map.begin() + (long)i);
    }
    break;
  }
  hipIpcMemHandle_t h;
  memcpy(&h, handle_bytes, sizeof h);
  void* ptr = nullptr;
  hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) return e;
  IpcMapping m;
  m.owner_pid = owner_pid;
  m.owner_addr = owner_addr;
  memset(m.handle, 0, sizeof m.handle);
  memcpy(m.handle, handle_bytes, sizeof h);
  m.ptr = ptr;
  m.refs = 1;
  g_ipc_map.push_back(m);
  *out = ptr;
  return hipSuccess;
}

// the mapping stays open (see above); mappings of processes that no longer exist are closed
void ipc_close_shared(void* ptr) {
  std::lock_guard<std::mutex> g(g_ipc_mu);
  for (IpcMapping& m : g_ipc_map)
    if (m.ptr == ptr && m.refs > 0) {
      m.refs--;
      break;
    }
  for (size_t i = 0; i < g_ipc_map.size();) {
    IpcMapping& m = g_ipc_map[i];
    if (m.refs == 0 && kill(m.owner_pid, 0) != 0 && errno == ESRCH) {
      (void)hipIpcCloseMemHandle(m.ptr);
      (void)hipGetLastError();
      g_ipc_map.erase(g_ipc_map.begin() + (long)i);
    } else {
      i++;
    }
  }
}

struct PooledBlock {
  int device;
  size_t bytes;
  int kind;  // 0 = window (hipMalloc), 1 = flag page (uncached)
  void* ptr;
  bool in_use;
  bool have_handle;
  hipIpcMemHandle_t handle;
  uint64_t mark;  // what the last user left behind for the next one (flag pages: the last epoch written into it)
};
static std::mutex g_pool_mu;
static std::vector<PooledBlock> g_pool;

// HBM that peers map: taken from the pool of blocks earlier communicators of this process left behind, or
// allocated (kind 1: uncached / fine-grained, for flag words polled by kernels)
void* pool_acquire(int device, size_t bytes, int kind, bool* fresh, uint64_t* mark) {
  std::lock_guard<std::mutex> g(g_pool_mu);
  if (fresh) *fresh = false;
  if (mark) *mark = 0;
  for (PooledBlock& b : g_pool)
    if (!b.in_use && b.device == device && b.bytes == bytes && b.kind == kind) {
      b.in_use = true;
      if (mark) *mark = b.mark;
      return b.ptr;
    }
  if (fresh) *fresh = true;
  void* p = nullptr;
  hipError_t e;
  if (kind == 1) {
    e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
    }
  } else {
    e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {  // memory is tight: give idle blocks of other sizes back first
      (void)hipGetLastError();
      for (size_t i = 0; i < g_pool.size();)
        if (!g_pool[i].in_use && g_pool[i].device == device) {
          (void)hipFree(g_pool[i].ptr);
          g_pool.erase(g_pool.begin() + (long)i);
        } else {
          i++;
        }
      e = hipMalloc(&p, bytes);
    }
  }
  if (e != hipSuccess) return nullptr;
  PooledBlock nb;
  memset(&nb, 0, sizeof nb);
  nb.device = device;
  nb.bytes = bytes;
  nb.kind = kind;
  nb.ptr = p;
  nb.in_use = true;
  g_pool.push_back(nb);
  return p;
}

// The hipIpc handle of a pooled block: exported ONCE, so that a peer recognises the block when a later
// communicator offers it again and keeps using the mapping it has (closing and re-opening mappings while other
// processes do the same is what fails with "invalid device pointer" on this stack).
hipError_t pool_handle(void* ptr, void* handle_out) {
  std::lock_guard<std::mutex> g(g_pool_mu);
  for (PooledBlock& b : g_pool)
    if (b.ptr == ptr) {
      if (!b.have_handle) {
        hipError_t e = hipIpcGetMemHandle(&b.handle, ptr);
        if (e != hipSuccess) return e;
        b.have_handle = true;
      }
      memcpy(handle_out, &b.handle, sizeof b.handle);
      return hipSuccess;
    }
  return hipErrorInvalidValue;
}

// Streams outlive communicators too.  Creating a stream's hardware queue while the GPU's queues are
// oversubscribed (several processes on one GPU, a test runner holding a context of its own) made the FIRST operation
// on a new stream take 17-32 SECONDS (lifecycle trace, profiles/README.md r02): a finalised communicator's streams go
// back to a per-process pool instead of being destroyed.
static std::mutex g_stream_mu;
static std::vector<std::pair<int, hipStream_t>> g_stream_pool;

hipStream_t stream_acquire(int device) {
  {
    std::lock_guard<std::mutex> g(g_stream_mu);
    for (size_t i = 0; i < g_stream_pool.size(); i++)
      if (g_stream_pool[i].first == device) {
        hipStream_t s = g_stream_pool[i].second;
        g_stream_pool.erase(g_stream_pool.begin() + (long)i);
        return s;
      }
  }
  hipStream_t s = nullptr;
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
  return s;
}

void stream_release(int device, hipStream_t s) {
  if (!s) return;
  std::lock_guard<std::mutex> g(g_stream_mu);
  g_stream_pool.push_back({device, s});
}

void pool_release(void* ptr, uint64_t mark) {
  std::lock_guard<std::mutex> g(g_pool_mu);
  for (PooledBlock& b : g_pool)
    if (b.ptr == ptr) {
      b.in_use = false;
      b.mark = mark;
    }
}

// the per-peer / batch streams of the staged schedules (engine.cpp), created on first use
int ensure_streams(xmpi_comm* c) {
  if (c->shared_stream || c->staged_streams) return XMPI_OK;
  bool ok = (c->batch_send_stream = stream_acquire(c->device)) && (c->batch_recv_stream = stream_acquire(c->device));
  for (int p = 0; p < c->size && ok; p++) {
    if (p == c->rank) continue;
    ok = (c->send_stream[p] = stream_acquire(c->device)) && (c->recv_stream[p] = stream_acquire(c->device));
  }
  if (!ok) return hip_fail(hipGetLastError(), "hipStreamCreate", __FILE__, __LINE__);
  c->staged_streams = true;
  return XMPI_OK;
}

static hipStream_t shared_stream_for(int device) {
  static std::mutex mu;
  static std::vector<std::pair<int, hipStream_t>> streams;
  std::lock_guard<std::mutex> g(mu);
  for (auto& kv : streams)
    if (kv.first == device) return kv.second;
  hipStream_t s = nullptr;
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
  streams.push_back({device, s});
  return s;
}

// ---- non-blocking collectives: an ordered worker per communicator --------------------------------
static thread_local bool t_in_worker = false;

static void worker_main(xmpi_comm* c) {
  t_in_worker = true;
  (void)hipSetDevice(c->device);
  for (;;) {
    std::pair<std::function<int()>, xmpi_request*> job;
    {
      std::unique_lock<std::mutex> l(c->wq_mu);
      c->wq_cv.wait(l, [c] { return c->wq_stop || !c->wq.empty(); });
      if (c->wq.empty()) return;  // stop requested and nothing left to run
      job = std::move(c->wq.front());
      c->wq.pop_front();
    }
    g_last_error.clear();
    t_api_call = c->api_calls.fetch_add(1, std::memory_order_relaxed) + 1;  // (before the job takes coll_mu: whoever holds it next has seen this -- dsync.cpp dsync_ll)
    const int rc = job.first();
    {
      std::lock_guard<std::mutex> l(job.second->mu);
      job.second->rc = rc;
      job.second->err = g_last_error;
      job.second->done = true;
      job.second->cv.notify_all();  // under the lock: the waiter frees the request as soon as it sees `done`
    }
    {
      std::lock_guard<std::mutex> l(c->wq_mu);
      c->wq_busy--;
    }
    c->wq_cv.notify_all();
  }
}

static xmpi_request* submit(xmpi_comm* c, std::function<int()> fn) {
  xmpi_request* r = new xmpi_request;
  {
    std::lock_guard<std::mutex> l(c->wq_mu);
    if (!c->worker_started) {
      c->worker = std::thread(worker_main, c);
      c->worker_started = true;
    }
    c->wq.emplace_back(std::move(fn), r);
    c->wq_busy++;
  }
  c->wq_cv.notify_all();
  return r;
}

// a blocking collective issued after non-blocking ones runs after them (same order on every rank)
static void drain_worker(xmpi_comm* c) {
  if (t_in_worker || !c->worker_started) return;
  std::unique_lock<std::mutex> l(c->wq_mu);
  c->wq_cv.wait(l, [c] { return c->wq_busy == 0; });
}

static void stop_worker(xmpi_comm* c) {
  if (!c->worker_started) return;
  drain_worker(c);
  {
    std::lock_guard<std::mutex> l(c->wq_mu);
    c->wq_stop = true;
  }
  c->wq_cv.notify_all();
  c->worker.join();
  c->worker_started = false;
}

static int collective_on_host_meeting_ranks(xmpi_comm* c, int coll, int algo, int root, const void* sendbuf, void* recvbuf, size_t count, int dtype,
                                            int op);

static int collective(xmpi_comm* c, int coll, int algo, int root, const void* sendbuf, void* recvbuf, size_t count,
                      int dtype, int op) {
  drain_worker(c);
  const size_t es = xmpi_dtype_size((xmpi_dtype)dtype);
  if (es == 0 || op < 0 || op >= XMPI_OP_COUNT || root < 0 || root >= c->size || algo < 0 || algo >= XMPI_ALGO_COUNT) {
    set_last_error("bad dtype / op / root / algo");
    return XMPI_ERR_ARG;
  }
  if (count == 0) return XMPI_OK;
  if (!recvbuf || !sendbuf) {
    set_last_error("null buffer");
    return XMPI_ERR_ARG;
  }
  RoctxRange range("xmpi:%s algo=%s bytes=%zu rank=%d/%d", coll_name(coll), algo_name(algo), count * es, c->rank, c->size);
  std::lock_guard<std::mutex> g(c->coll_mu);
  // One process per GPU (the production layout): the ranks meet on the device (dsync.cpp) -- one kernel per
  // rank, enqueued on this communicator's stream, no host barrier.  Whether this path is taken depends on the
  // job's layout and the arguments only, so every rank decides alike; buffers the peers cannot map are stood in
  // for by registered arena blocks inside.
  // RING / RHD (allreduce), RING (allgather) and TREE (bcast) name the stepped kernels there (sched.hip): every step of
  // the schedule inside one kernel per rank; with ranks that meet on the host they name the staged schedules below.
  if (dsync_takes(c, coll, algo))
    return dsync_collective(c, coll, root, sendbuf, recvbuf, count, dtype, op, c->local_stream, /*blocking=*/true, algo);
  // Ranks that meet on the host, HOST slices (what a program written against the reference passes, helloworld.go:53-81): stood in
  // for by blocks of the registered arenas -- the heap keeps them from call to call, so the zero-copy fold applies and nothing is
  // hipMalloc'ed / hipFree'd per call (35 ms per 256 MiB buffer with eight rank threads at it: scripts/r06_hostleg_probe.py) --, one copy up,
  // one copy down on the communicator's stream.  No arena memory left: the staged path's own temporary buffers (below).
  const bool send_host = !is_device_pointer(sendbuf), recv_host = !is_device_pointer(recvbuf);
  if (send_host || recv_host) {
    const size_t send_bytes = count * es, recv_bytes = coll == COLL_ALLGATHER ? send_bytes * (size_t)c->size : send_bytes;
    const bool in_place = sendbuf == recvbuf && coll != COLL_ALLGATHER;
    void* up_recv = recv_host ? heap_alloc(c->device, recv_bytes) : nullptr;
    void* up_send = send_host && !(in_place && recv_host) ? heap_alloc(c->device, send_bytes) : nullptr;
    if ((recv_host && !up_recv) || (send_host && !(in_place && recv_host) && !up_send)) {
      if (up_recv) (void)heap_free(up_recv);
      if (up_send) (void)heap_free(up_send);
      (void)hipGetLastError();
      return collective_on_host_meeting_ranks(c, coll, algo, root, sendbuf, recvbuf, count, dtype, op);
    }
    void* drecv = recv_host ? up_recv : recvbuf;
    const void* dsend = !send_host ? sendbuf : (in_place && recv_host) ? drecv : up_send;
    auto done = [&](int rc) {
      if (up_recv) (void)heap_free(up_recv);
      if (up_send) (void)heap_free(up_send);
      return rc;
    };
    hipStream_t s = c->local_stream;
    // (what goes up: the operand; a broadcast's message at its root -- in `recvbuf` --; nothing else has a say in the result)
    const void* src_up = coll == COLL_BCAST ? (c->rank == root && recv_host ? recvbuf : nullptr) : (send_host ? sendbuf : nullptr);
    void* dst_up = coll == COLL_BCAST ? drecv : const_cast<void*>(dsend);
    if (src_up) {
      if (hipMemcpyAsync(dst_up, src_up, send_bytes, hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return done(hip_fail(hipGetLastError(), "hipMemcpyAsync(host slice -> its stand-in)", __FILE__, __LINE__));
    }
    int rc = collective_on_host_meeting_ranks(c, coll, algo, root, dsend, drecv, count, dtype, op);
    if (rc == XMPI_OK && recv_host && (coll != COLL_REDUCE || c->rank == root)) {
      if (hipMemcpyAsync(recvbuf, drecv, recv_bytes, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        rc = hip_fail(hipGetLastError(), "hipMemcpyAsync(stand-in -> host slice)", __FILE__, __LINE__);
    }
    return done(rc);
  }
  return collective_on_host_meeting_ranks(c, coll, algo, root, sendbuf, recvbuf, count, dtype, op);
}

// (called with c->coll_mu held; ranks that meet on the host)
static int collective_on_host_meeting_ranks(xmpi_comm* c, int coll, int algo, int root, const void* sendbuf, void* recvbuf, size_t count, int dtype,
                                            int op) {
  const size_t es = xmpi_dtype_size((xmpi_dtype)dtype);
  // Zero-copy first (zcopy.cpp): when every rank's buffers are registered HBM one kernel per rank does
  // the whole collective in place.  Whether that holds is decided collectively, so either every rank
  // returns from here or every rank goes on to the staged schedule below.
  const bool zc_algo = algo == XMPI_ALGO_ZCOPY || algo == XMPI_ALGO_ZPUSH || algo == XMPI_ALGO_LL;
  if (c->size > 1 && (zc_algo || (algo == XMPI_ALGO_AUTO && c->zero_copy))) {
    bool done = false;
    int zrc = zero_copy_collective(c, coll, root, sendbuf, recvbuf, count, dtype, op, algo == XMPI_ALGO_ZPUSH, &done);
    if (zrc != XMPI_OK || done) return zrc;
  }
  if (zc_algo) algo = XMPI_ALGO_AUTO;  // (LL lines need ranks that meet on the device: with the others it names the fold)
  // (the push forms are forms of the stepped KERNELS: the host-driven step tables have one form, which pushes into the windows)
  if (algo == XMPI_ALGO_RING_PUSH) algo = XMPI_ALGO_RING;
  if (algo == XMPI_ALGO_RHD_PUSH) algo = XMPI_ALGO_RHD;
  if (algo == XMPI_ALGO_TREE_PUSH) algo = XMPI_ALGO_TREE;
  PlanParams pp;
  pp.coll = coll;
  pp.algo = algo;
  pp.size = c->size;
  pp.rank = c->rank;
  pp.root = root;
  pp.count = count;
  pp.elem_size = es;
  pp.channels = c->channels > 0 ? (int)c->channels : ring_channel_count(c->size);  // 0 = every link
  pp.lanes = c->lanes;
  const size_t total = count * es;
  size_t chunk = total;
  if (coll == COLL_ALLREDUCE) chunk = total / (size_t)c->size / (size_t)std::max(1, pp.channels);
  pp.piece_bytes = choose_piece(c, chunk);
  pp.fuse = 1;  // ring: receive-reduce-send / receive-copy-send as one kernel
  pp.fifo_depth = c->fifo_depth;
  pp.oneshot_bytes = (size_t)std::max<long>(0, c->oneshot_bytes);
  Plan plan;
  int rc = build_plan(pp, &plan);
  if (rc != XMPI_OK) {
    set_last_error("no schedule for this collective / algorithm / size");
    return rc;
  }

  // host-resident buffers: stage through this rank's HBM (convenience path; the hot path is HBM)
  const bool send_dev = is_device_pointer(sendbuf), recv_dev = is_device_pointer(recvbuf);
  const size_t send_bytes = total;
  const size_t recv_bytes = (coll == COLL_ALLGATHER) ? total * (size_t)c->size : total;
  void *dsend = const_cast<void*>(sendbuf), *drecv = recvbuf;
  void *tmp_send = nullptr, *tmp_recv = nullptr;
  if (!recv_dev) {
    XMPI_HIP(hipMalloc(&tmp_recv, recv_bytes));
    drecv = tmp_recv;
    if (coll == COLL_BCAST) XMPI_HIP(hipMemcpy(tmp_recv, recvbuf, recv_bytes, hipMemcpyHostToDevice));
  }
  if (!send_dev) {
    if (sendbuf == recvbuf && coll != COLL_ALLGATHER) {
      if (coll != COLL_BCAST) XMPI_HIP(hipMemcpy(drecv, sendbuf, send_bytes, hipMemcpyHostToDevice));
      dsend = drecv;
    } else {
      XMPI_HIP(hipMalloc(&tmp_send, send_bytes));
      XMPI_HIP(hipMemcpy(tmp_send, sendbuf, send_bytes, hipMemcpyHostToDevice));
      dsend = tmp_send;
    }
  }
  rc = run_plan(c, plan, dsend, drecv, dtype, op);
  if (rc == XMPI_OK && !recv_dev) {
    const bool significant = (coll != COLL_REDUCE) || c->rank == root;
    if (significant) XMPI_HIP(hipMemcpy(recvbuf, drecv, recv_bytes, hipMemcpyDeviceToHost));
  }
  if (tmp_send) (void)hipFree(tmp_send);
  if (tmp_recv) (void)hipFree(tmp_recv);
  return rc;
}


// ---- the library checks its schedules' ANSWERS on the machine it runs on ------------------------------------------------------
// The reference's own benchmark verifies every echo before it reports a time (examples/bounce/bounce.go:103-112,131-136), its
// handshake checks what came back (network.go:343-351).  Here: whoever times a schedule (xmpi_tune) or is about to rely on one
// untuned (xmpi_init's self-check) first runs it ONCE on patterned inputs -- rank r's send buffer = the counter-based pattern with
// seed kCheckSeed + r (kernels.hip fill_kernel, pattern 3: signed multiples of 2^-12 below 4 in magnitude: the float32 sum of
// sixteen of them is exact in EVERY association, so every schedule must produce the same bits) --, compares the receive buffer with
// the result computed locally (N fills folded with the two-operand kernel: no communication), and votes through the control block.
const char* const kCandName[xmpi_comm::CAND_COUNT] = {"fold (one kernel)", "fold (one kernel, 2 packets in flight)", "split (meet / body / done)",
                                                      "push-only", "ring kernel", "halving kernel", "LL lines", "ring kernel, push form",
                                                      "halving kernel, push form", "tree kernel", "tree kernel, push form"};
constexpr size_t kSecondPassBytes = (size_t)1 << 20;  // per rank: eight ranks' buffers of that size sit in the L2s (8 x 4 MiB) together from one run to the next
constexpr long kTuneTimeoutS = 20;  // no-progress limit of a candidate run in a job that otherwise waits for ever
constexpr uint64_t kCheckSeed = 0x7A11D;
constexpr int kCheckPattern = 3;

static int job_barrier(xmpi_comm* c) {
  dsync_service(c);
  Backoff bo;
  arm(bo, c);
  return c->ctl->barrier(wait_limit(c), &bo);
}

// every rank publishes a row, all meet, everybody reads the same maxima
static int vote_max(xmpi_comm* c, const double* us, const uint64_t* bad, int n, double* us_max, uint64_t* bad_max) {
  TuneVote* mine = c->ctl->vote(c->rank);
  for (int k = 0; k < kTuneCands; k++) {
    mine->us[k] = (us && k < n) ? us[k] : 0.0;
    mine->bad[k] = (bad && k < n) ? bad[k] : 0;
  }
  int rc = job_barrier(c);
  if (rc != XMPI_OK) return rc;
  for (int k = 0; k < n; k++) {
    double u = 0;
    uint64_t b = 0;
    for (int p = 0; p < c->size; p++) {
      const TuneVote* v = c->ctl->vote(p);
      u = std::max(u, v->us[k]);
      b = std::max(b, v->bad[k]);
    }
    if (us_max) us_max[k] = u;
    if (bad_max) bad_max[k] = b;
  }
  return job_barrier(c);  // nobody writes its next row before everybody has read this one
}

struct AnswerCheck {
  xmpi_comm* c = nullptr;
  char *send = nullptr, *recv = nullptr, *expect = nullptr, *expect2 = nullptr;
  size_t cap = 0, cap2 = 0;  // bytes of each (expect2: the second pass is for messages a cache could still hold)
  int have_coll = -1;      // what `expect` holds
  size_t have_bytes = 0;
  double spent_s = 0;
  uint64_t* host_word = nullptr;      // pinned: where a count reaches the host without a device-to-host copy (kernels.hip word_to_host_kernel)
  uint64_t* host_word_dev = nullptr;
  bool twice = true;       // the caller's say on the second pass (xmpi_tune: every other size class)

  int open(xmpi_comm* comm, size_t max_bytes) {
    c = comm;
    cap = max_bytes;
    send = (char*)heap_alloc(c->device, cap);
    recv = (char*)heap_alloc(c->device, cap);
    // (the expected results are this rank's own business: plain device memory -- a block of a registered arena is exported and mapped
    // by every peer, and four 256 MiB blocks per rank grew the arenas by a GiB each: 8 .. 33 s of mapping on a fresh box)
    cap2 = std::min(cap, kSecondPassBytes * (size_t)c->size);
    if (hipMalloc((void**)&expect, cap) != hipSuccess) expect = nullptr;
    if (hipMalloc((void**)&expect2, cap2) != hipSuccess) expect2 = nullptr;
    if (hipHostMalloc((void**)&host_word, 64, hipHostMallocMapped) == hipSuccess) {
      void* dev = nullptr;
      if (hipHostGetDevicePointer(&dev, host_word, 0) == hipSuccess) host_word_dev = (uint64_t*)dev;
    } else {
      host_word = nullptr;
    }
    (void)hipGetLastError();
    if (!send || !recv || !expect || !expect2 || !host_word_dev) {
      close();
      set_last_error("xmpi_tune: out of device memory");
      return XMPI_ERR_NOMEM;
    }
    XMPI_HIP(launch_fill(send, cap / 4, XMPI_F32, kCheckPattern, kCheckSeed + (uint64_t)c->rank, c->local_stream));
    XMPI_HIP(hipStreamSynchronize(c->local_stream));
    return XMPI_OK;
  }
  void close() {
    if (c) {
      (void)hipStreamSynchronize(c->local_stream);
      (void)hipGetLastError();
    }
    if (send) (void)heap_free(send);
    if (recv) (void)heap_free(recv);
    if (expect) (void)hipFree(expect);
    if (expect2) (void)hipFree(expect2);
    if (host_word) (void)hipHostFree(host_word);
    host_word = host_word_dev = nullptr;
    (void)hipGetLastError();
    send = recv = expect = expect2 = nullptr;
  }
  size_t recv_bytes(int coll, size_t per_rank) const { return coll == COLL_ALLGATHER ? per_rank * (size_t)c->size : per_rank; }
  // `expect` = what `coll` over `per_rank` bytes per rank (root 0) must leave in the receive buffer.  The pattern is a function of
  // the element's index: a shorter message is a prefix of a longer one's, so the sum and the broadcast are computed once, at `cap`.
  int expect_for(int coll, size_t per_rank) {
    const double t0 = now_seconds();
    hipStream_t s = c->local_stream;
    const bool sum = coll == COLL_ALLREDUCE || coll == COLL_REDUCE;
    if (sum && !(have_coll == COLL_ALLREDUCE || have_coll == COLL_REDUCE)) {
      XMPI_HIP(launch_fill(expect, cap / 4, XMPI_F32, kCheckPattern, kCheckSeed, s));
      for (int r = 1; r < c->size; r++) {  // (the receive buffer is free between two candidates: the other ranks' inputs pass through it)
        XMPI_HIP(launch_fill(recv, cap / 4, XMPI_F32, kCheckPattern, kCheckSeed + (uint64_t)r, s));
        XMPI_HIP(launch_reduce2(expect, expect, recv, cap / 4, XMPI_F32, XMPI_SUM, s));
      }
    } else if (coll == COLL_ALLGATHER && !(have_coll == coll && have_bytes == per_rank)) {
      for (int r = 0; r < c->size; r++)
        XMPI_HIP(launch_fill(expect + (size_t)r * per_rank, per_rank / 4, XMPI_F32, kCheckPattern, kCheckSeed + (uint64_t)r, s));
    } else if (coll == COLL_BCAST && have_coll != coll) {
      XMPI_HIP(launch_fill(expect, cap / 4, XMPI_F32, kCheckPattern, kCheckSeed, s));
      if (c->rank == 0) XMPI_HIP(launch_fill(recv, cap / 4, XMPI_F32, kCheckPattern, kCheckSeed, s));  // the root's buffer IS the message
    }
    have_coll = coll;
    have_bytes = per_rank;
    spent_s += now_seconds() - t0;
    return XMPI_OK;
  }
  // before the checked run: whatever an earlier candidate left in the receive buffer must not pass for this one's answer
  int arm(int coll, size_t per_rank) {
    const double t0 = now_seconds();
    // (bcast: the root's buffer is the input; reduce: only the root's is written)
    const bool untouched = (coll == COLL_BCAST && c->rank == 0) || (coll == COLL_REDUCE && c->rank != 0);
    // (a kernel of the library's own on the rank's stream -- the constant 166.0, which no sum of sixteen pattern values can be --, not
    // hipMemsetAsync: the runtime's fills do not run on the stream's queue alone, see count_to_host)
    if (!untouched) XMPI_HIP(launch_fill(recv, recv_bytes(coll, per_rank) / 4, XMPI_F32, /*pattern=*/2, /*seed=*/165, c->local_stream));
    spent_s += now_seconds() - t0;
    return XMPI_OK;
  }
  // The SECOND pass: the inputs change IN PLACE between two runs (every rank's buffer := 2 x itself, one local kernel; the expected
  // result doubles with it, exactly) -- what a caller's buffers do from one step to the next.  A reader that still holds lines of a
  // peer's buffer from the run before -- an L2 the schedule's acquire did not reach: the split form's once-per-XCD acquire is
  // exactly that bet -- folds OLD data, and only a changed input shows it: the first run of a fresh buffer never can.  For messages a
  // cache could still hold whole (kSecondPassBytes per rank); afterwards the inputs are what they were (refilled).
  bool second_pass(int coll, size_t per_rank) const {
    static const bool on = env_long("XMPI_CHECK_PASSES", 2) >= 2;  // (1: the first pass only -- A/B of what the second one costs)
    return on && per_rank <= kSecondPassBytes && recv_bytes(coll, per_rank) <= cap2;
  }
  int change_inputs(int coll, size_t per_rank) {
    const double t0 = now_seconds();
    hipStream_t s = c->local_stream;
    if (coll != COLL_BCAST) XMPI_HIP(launch_reduce2(send, send, send, per_rank / 4, XMPI_F32, XMPI_SUM, s));
    else if (c->rank == 0) XMPI_HIP(launch_reduce2(recv, recv, recv, per_rank / 4, XMPI_F32, XMPI_SUM, s));
    XMPI_HIP(launch_reduce2(expect2, expect, expect, recv_bytes(coll, per_rank) / 4, XMPI_F32, XMPI_SUM, s));
    spent_s += now_seconds() - t0;
    return XMPI_OK;
  }
  int restore_inputs(int coll, size_t per_rank) {
    const double t0 = now_seconds();
    hipStream_t s = c->local_stream;
    if (coll != COLL_BCAST) XMPI_HIP(launch_fill(send, per_rank / 4, XMPI_F32, kCheckPattern, kCheckSeed + (uint64_t)c->rank, s));
    else if (c->rank == 0) XMPI_HIP(launch_fill(recv, per_rank / 4, XMPI_F32, kCheckPattern, kCheckSeed, s));
    spent_s += now_seconds() - t0;
    return XMPI_OK;
  }
  // c->dev_words[0] -> *out, behind everything on the stream; no copy engine involved
  hipError_t count_to_host(uint64_t* out) {
    hipStream_t s = c->local_stream;
    __atomic_store_n(host_word, ~0ull, __ATOMIC_RELAXED);
    hipError_t e = launch_word_to_host(host_word_dev, c->dev_words, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    *out = __atomic_load_n(host_word, __ATOMIC_ACQUIRE);
    return e;
  }
  int verdict(int coll, size_t per_rank, uint64_t* bad, bool second = false) {
    *bad = 0;
    if (coll == COLL_REDUCE && c->rank != 0) return XMPI_OK;
    const double t0 = now_seconds();
    hipStream_t s = c->local_stream;
    XMPI_HIP(launch_signal(c->dev_words, 0, s));
    XMPI_HIP(launch_count_mismatch(recv, second ? expect2 : expect, recv_bytes(coll, per_rank), c->dev_words, s));
    XMPI_HIP(count_to_host(bad));
    spent_s += now_seconds() - t0;
    return XMPI_OK;
  }
};

struct TuneCand {
  int algo, split, unroll;
};
static std::vector<TuneCand> tune_candidates(const xmpi_comm* c) {
  const int u0 = (int)std::max<long>(1, std::min<long>(2, c->dsync_unroll));
  // the default comes first (xmpi_tune_decide keeps it on a tie); the order is xmpi_comm::CAND_*
  return {{XMPI_ALGO_ZCOPY, 0, u0},      {XMPI_ALGO_ZCOPY, 0, 3 - u0}, {XMPI_ALGO_ZCOPY, 1, u0},    {XMPI_ALGO_ZPUSH, 0, u0},
          {XMPI_ALGO_RING, 0, u0},       {XMPI_ALGO_RHD, 0, u0},       {XMPI_ALGO_LL, 0, u0},
          // the push forms of the stepped kernels (sched_steps.h): the same schedules with every payload byte STORED over its link
          // instead of loaded -- which of the two a link moves faster is the machine's to say
          {XMPI_ALGO_RING_PUSH, 0, u0},  {XMPI_ALGO_RHD_PUSH, 0, u0},
          // bcast and reduce: the tree kernels in both forms against the fold's two halves
          {XMPI_ALGO_TREE, 0, u0},       {XMPI_ALGO_TREE_PUSH, 0, u0}};
}
// which candidates a collective has: (the fold, LL lines) all four; allreduce every form of the fold and ring / halving in both
// forms; allgather the ring; bcast the tree (its fold is one kernel whatever the size); reduce what allreduce has of the fold, and
// the tree
static bool tune_offered(const xmpi_comm* c, int coll, const TuneCand& cd) {
  const int u0 = (int)std::max<long>(1, std::min<long>(2, c->dsync_unroll));
  const bool tree = cd.algo == XMPI_ALGO_TREE || cd.algo == XMPI_ALGO_TREE_PUSH;
  const bool ring = cd.algo == XMPI_ALGO_RING || cd.algo == XMPI_ALGO_RING_PUSH;
  const bool rhd = cd.algo == XMPI_ALGO_RHD || cd.algo == XMPI_ALGO_RHD_PUSH;
  switch (coll) {
    case COLL_ALLREDUCE: return !tree;
    case COLL_ALLGATHER: return !tree && !rhd && cd.algo != XMPI_ALGO_ZPUSH && cd.unroll == u0;
    case COLL_BCAST: return !ring && !rhd && cd.algo != XMPI_ALGO_ZPUSH && cd.unroll == u0 && cd.split == 0;
    default: return !ring && !rhd && cd.unroll == u0;  // COLL_REDUCE
  }
}

// Candidates `ks` of `coll` at `per_rank` bytes per rank: each runs once CHECKED (check: the warm-up that also maps whatever is
// new), then `iters` times against the clock.  us[k]: mean microseconds (min with what it held when keep_min); bad[k]: bytes of
// this rank's receive buffer that differ from the expected result.  Collective: every rank passes the same arguments.
static int tune_measure(xmpi_comm* c, AnswerCheck& chk, const std::vector<TuneCand>& cands, int coll, size_t per_rank, const std::vector<int>& ks,
                        int iters, bool check, bool keep_min, double* us, uint64_t* bad) {
  const long keep_split = c->dsync_split_bytes, keep_unroll = c->dsync_unroll, keep_timeout = c->timeout_s;
  // A candidate that HANGS on this machine (flag words that never arrive over a link, say) must not hang the job that merely asked
  // which schedule is fastest: with the default "wait for ever" every wait of a candidate run -- host loops and waiting kernels -- has
  // a no-progress limit of its own; the error names the candidate (leave it out with tune_mask, or set XMPI_TIMEOUT_S).
  if (keep_timeout == 0) c->timeout_s = kTuneTimeoutS;
  int rc = XMPI_OK;
  for (size_t j = 0; j < ks.size() && rc == XMPI_OK; j++) {
    const int k = ks[j];
    const TuneCand& cd = cands[(size_t)k];
    rc = job_barrier(c);  // (everybody has left the previous candidate: its receive buffer is this rank's again)
    if (rc != XMPI_OK) break;
    // (all candidates run under ONE call number -- the caller's XMPI_ENTER: dsync_ll takes "the previous call was an agent's
    // collective and this is the next call" for "nothing was enqueued since", which the poison enqueued here would belie)
    std::lock_guard<std::mutex> g(c->coll_mu);
    c->dsync_split_bytes = cd.split ? 1 : 0;
    c->dsync_unroll = cd.unroll;
    const bool tr = trace_on() && per_rank >= ((size_t)64 << 20);
    const double tb = now_seconds();
    double t_arm = 0, t_run = 0, t_verdict = 0;
    // The check's rank-local kernels (poison, compare, refill) and a collective's WAITING kernels must not share the GPU: which ranks
    // poison or compare depends on the collective (bcast: everybody but the root; reduce: the root alone), so some ranks would be
    // spinning in the next collective's kernel while another still streams 256 MiB through a local one -- and with eight processes on
    // ONE GPU that mix stalled the tree kernels for 8 .. 60 s at a time (round-6 profiles/r06/tune_stall).  So every rank's local
    // kernels have ended, on every rank, before any rank launches a kernel that waits for a peer: stream sync + the job's barrier.
    auto settle = [&]() -> int {
      if (hipStreamSynchronize(c->local_stream) != hipSuccess) return hip_fail(hipGetLastError(), "hipStreamSynchronize", __FILE__, __LINE__);
      return job_barrier(c);
    };
    if (check) {
      rc = chk.arm(coll, per_rank);
      if (rc == XMPI_OK) rc = settle();
    }
    t_arm = now_seconds() - tb;
    double t0 = now_seconds();
    for (int i = -1; i < iters && rc == XMPI_OK; i++) {
      if (i == 0) t0 = now_seconds();
      rc = dsync_collective(c, coll, 0, coll == COLL_BCAST ? chk.recv : chk.send, chk.recv, per_rank / 4, XMPI_F32, XMPI_SUM, c->local_stream,
                            /*blocking=*/i == -1 || i == iters - 1, cd.algo);
      if (i == -1) t_run = now_seconds() - tb - t_arm;
      if (i == -1 && check && rc == XMPI_OK) {
        rc = chk.verdict(coll, per_rank, &bad[k]);
        t_verdict = now_seconds() - tb - t_arm - t_run;
        // ... and once more with the inputs changed in place -- whatever THIS rank's first verdict was: the ranks see different
        // verdicts (a wrong byte lands in one rank's buffer), and a run only some of them make is a hang
        if (rc == XMPI_OK && chk.twice && chk.second_pass(coll, per_rank)) {
          uint64_t bad2 = 0;
          rc = chk.change_inputs(coll, per_rank);
          if (rc == XMPI_OK) rc = chk.arm(coll, per_rank);
          if (rc == XMPI_OK) rc = settle();
          if (rc == XMPI_OK)
            rc = dsync_collective(c, coll, 0, coll == COLL_BCAST ? chk.recv : chk.send, chk.recv, per_rank / 4, XMPI_F32, XMPI_SUM, c->local_stream, true, cd.algo);
          if (rc == XMPI_OK) rc = chk.verdict(coll, per_rank, &bad2, /*second=*/true);
          // (every rank has left the run -- a blocking collective ends behind every peer's reads of this rank's buffers -- : refill)
          if (rc == XMPI_OK) rc = chk.restore_inputs(coll, per_rank);
          bad[k] = std::max(bad[k], bad2);
        }
        if (rc == XMPI_OK && iters > 0) rc = settle();  // ... before the timed runs
      }
    }
    if (iters > 0) {
      const double t_us = (now_seconds() - t0) / iters * 1e6;
      us[k] = keep_min && us[k] > 0 ? std::min(us[k], t_us) : t_us;
    }
    c->dsync_split_bytes = keep_split;
    c->dsync_unroll = keep_unroll;
    if (tr)
      fprintf(stderr, "[xmpi %d %.6f] tune:   %s %zu B by %s: %.0f us%s; arm %.1f ms, first run %.1f ms, verdict %.1f ms, all %.1f ms\n", c->rank, now_seconds(),
              coll_name(coll), per_rank, kCandName[k], iters > 0 ? us[k] : 0.0, check ? " (checked)" : "", t_arm * 1e3, t_run * 1e3, t_verdict * 1e3,
              (now_seconds() - tb) * 1e3);
    if (rc == XMPI_ERR_TIMEOUT && keep_timeout == 0)
      set_last_error(std::string(coll_name(coll)) + " by " + kCandName[k] + " at " + std::to_string(per_rank) + " B per rank did not complete within " +
                     std::to_string(kTuneTimeoutS) + " s while the library was checking / timing it on this machine (" + xmpi_last_error() +
                     "): leave it out (xmpi_set_param \"tune_mask\") or give the job a no-progress limit (XMPI_TIMEOUT_S)");
  }
  c->timeout_s = keep_timeout;
  return rc;
}

// What follows from a rejected schedule beyond "AUTO's table leaves it out": the untuned rules must not lead to it either.
static void apply_rejections(xmpi_comm* c) {
  uint32_t any = 0;
  for (int k = 0; k < 4; k++) any |= c->tune_rejected[k];
  if (any & (1u << xmpi_comm::CAND_LL)) {  // untuned AUTO sends short messages as LL lines
    c->ll_bytes = 0;
    c->agent_ll = 0;
    // (one mechanism -- 8-byte lines stored into the peers' flag allocations -- under all four collectives: wrong for one, trusted for none)
    for (int k = 0; k < 4; k++) c->tune_rejected[k] |= 1u << xmpi_comm::CAND_LL;
  }
  if (any & (1u << xmpi_comm::CAND_SPLIT)) c->dsync_split_bytes = 0;  // ... and large ones as meet / body / done
}

// The ladder's last rung: no device-synchronised schedule is right for some call on this machine -- the ranks meet on the host from
// now on (zcopy.cpp's rendezvous through the control block, the staged step tables), as after a flag page that could not be mapped.
// Collective (the caller's decision came out of a vote).
static void demote_to_host(xmpi_comm* c, const std::string& reason) {
  ll_agent_stop(c);
  (void)hipStreamSynchronize(c->local_stream);
  (void)hipGetLastError();
  c->dsync_ok = false;
  c->tuned = false;
  c->degraded_why += std::string(c->degraded_why.empty() ? "" : "; ") + "the ranks meet on the host (no device-synchronised collectives): " + reason;
}

// what a check found: remembered, said (xmpi_degraded, xmpi_last_error, one line on stderr), and acted upon
static void note_rejections(xmpi_comm* c, const char* who, const std::string& why, bool none_right) {
  apply_rejections(c);
  if (why.empty() && !none_right) return;
  const std::string text = std::string(who) + ": " + why + (why.empty() ? "" : "; ") +
                           (none_right ? "no schedule left that is right for every call" : "left out of AUTO, refused by name (tune_rejected_<collective>)");
  c->rejected_why += std::string(c->rejected_why.empty() ? "" : "; ") + text;
  c->degraded_why += std::string(c->degraded_why.empty() ? "" : "; ") + text;
  if (none_right) demote_to_host(c, std::string(who) + " found no right schedule for some call");
  set_last_error(text);
  if (c->rank == 0) fprintf(stderr, "xmpi: degraded: %s\n", text.c_str());
}

// Send / Receive out of registered HBM straight into HBM -- the receiver's kernel LOADS the payload out of the sender's memory (the
// lingering receive agent up to 512 KiB, the pull kernel above: engine.cpp p2p_recv) --: every rank sends `bytes` of its pattern to its
// right neighbour and counts what differs in what its left one sent (even ranks send first, odd ranks receive first: the blocking
// pair is a rendezvous, network.go:569).  Collective.
static int p2p_check_round(xmpi_comm* c, AnswerCheck& chk, size_t bytes, uint64_t* bad) {
  const int N = c->size, right = (c->rank + 1) % N, left = (c->rank + N - 1) % N;
  const int tag = 0x7fff5c5c;
  hipStream_t s = c->local_stream;
  *bad = 0;
  int rc = job_barrier(c);
  if (rc != XMPI_OK) return rc;
  XMPI_HIP(launch_fill(chk.expect, bytes / 4, XMPI_F32, kCheckPattern, kCheckSeed + (uint64_t)left, s));
  XMPI_HIP(launch_fill(chk.recv, bytes / 4, XMPI_F32, 2, 165, s));
  XMPI_HIP(hipStreamSynchronize(s));
  chk.have_coll = -1;
  size_t got = 0;
  const long keep_timeout = c->timeout_s;
  if (keep_timeout == 0) c->timeout_s = kTuneTimeoutS;  // (as tune_measure: a message that never arrives is an error of the check, not a hang of Init)
  if (c->rank % 2 == 0) {
    rc = p2p_send(c, chk.send, bytes, XMPI_F32, right, tag);
    if (rc == XMPI_OK) rc = p2p_recv(c, chk.recv, bytes, XMPI_F32, left, tag, &got);
  } else {
    rc = p2p_recv(c, chk.recv, bytes, XMPI_F32, left, tag, &got);
    if (rc == XMPI_OK) rc = p2p_send(c, chk.send, bytes, XMPI_F32, right, tag);
  }
  c->timeout_s = keep_timeout;
  if (rc != XMPI_OK) return rc;
  if (got != bytes) {
    *bad = bytes;
    return XMPI_OK;
  }
  XMPI_HIP(launch_signal(c->dev_words, 0, s));
  XMPI_HIP(launch_count_mismatch(chk.recv, chk.expect, bytes, c->dev_words, s));
  XMPI_HIP(chk.count_to_host(bad));
  return XMPI_OK;
}

// xmpi_init's self-check (XMPI_SELFCHECK; default: on when the ranks sit on different GPUs): what UNTUNED AUTO can reach -- LL lines
// up to ll_bytes, the one-kernel fold, meet / body / done -- runs once, multi-tile, on patterned inputs before the first caller's
// data does; the other three collectives' folds ride along.  A job that tunes (xmpi_tune, XMPI_AUTOTUNE_BYTES) checks every
// candidate at every size anyway.  Collective.
static int init_selfcheck(xmpi_comm* c) {
  const double t_begin = now_seconds();
  t_api_call = c->api_calls.fetch_add(1, std::memory_order_relaxed) + 1;  // (as a public call: XMPI_ENTER)
  // the diagnostic counters count the CALLER's traffic (tests and benchmarks read them as such): what the check itself moves is taken out again
  // (the receive agent's launches are NUMBERED by a counter of their own -- p2p_agent_launch_no, engine.cpp agent_submit -- which goes on counting)
  struct Counters {
    uint64_t v[13];
  };
  auto counters = [&]() {
    return Counters{{c->p2p_direct_count, c->p2p_staged_count, c->p2p_lane_count, c->p2p_agent_served, c->p2p_agent_launches, c->dsync_launches,
                     c->dsync_ll_launches, c->dsync_ll_agent, c->ll_agent_launches, c->dsync_split_launches, c->dsync_sched_launches, c->dsync_bounced,
                     c->host_bounce_calls}};
  };
  const Counters before = counters();
  auto restore = [&]() {
    uint64_t* const at[13] = {&c->p2p_direct_count, &c->p2p_staged_count, &c->p2p_lane_count, &c->p2p_agent_served, &c->p2p_agent_launches, &c->dsync_launches,
                              &c->dsync_ll_launches, &c->dsync_ll_agent, &c->ll_agent_launches, &c->dsync_split_launches, &c->dsync_sched_launches,
                              &c->dsync_bounced, &c->host_bounce_calls};
    for (int k = 0; k < 13; k++) *at[k] = before.v[k];
  };
  // several 4 KiB tiles per rank's chunk at 8 ranks (fold: 4; split: 8 one-tile blocks, one per XCD); bcast just above
  // zc_bcast_push_bytes, where every rank forwards its chunk
  const size_t kFold = (size_t)128 << 10, kSplit = (size_t)256 << 10;
  const size_t kBcast = (size_t)std::max<long>(0, c->zc_bcast_push_bytes) + 16384 <= kSplit * 2 ? (size_t)std::max<long>(0, c->zc_bcast_push_bytes) + 16384 : kSplit;
  const size_t kP2PShort = (size_t)64 << 10, kP2PLong = (size_t)768 << 10;  // the receive agent's side of its 512 KiB limit, and the pull kernel's
  AnswerCheck chk;
  int rc;
  {
    std::lock_guard<std::mutex> g(c->coll_mu);
    rc = chk.open(c, std::max(std::max(kSplit, kBcast), kP2PLong));
  }
  if (rc != XMPI_OK) return rc;
  XMPI_TRACE_STEP(c->rank, "self-check: buffers ready");
  // (what the job's first xmpi_malloc and first kernel pay anyway -- the first arena allocated, exported, mapped by every peer;
  // the code object loaded -- reported apart from the checks themselves)
  c->selfcheck_setup_ms = (now_seconds() - t_begin) * 1e3;
  const std::vector<TuneCand> cands = tune_candidates(c);
  c->tune_running = true;
  std::string why;
  bool fold_wrong = false;
  auto run = [&](int coll, size_t per_rank, std::vector<int> ks, uint64_t* worst_bad) -> int {
    std::vector<uint64_t> bad(cands.size(), 0);
    {
      std::lock_guard<std::mutex> g(c->coll_mu);
      rc = chk.expect_for(coll, per_rank);
    }
    if (rc == XMPI_OK) rc = tune_measure(c, chk, cands, coll, per_rank, ks, 0, true, false, nullptr, bad.data());
    if (rc == XMPI_OK) rc = vote_max(c, nullptr, bad.data(), (int)cands.size(), nullptr, worst_bad);
    return rc;
  };
  auto reject = [&](int coll, int k, size_t per_rank, uint64_t nbad) {
    c->tune_rejected[coll] |= 1u << k;
    char t[200];
    snprintf(t, sizeof t, "%s: %s gives wrong answers on this machine (%zu B per rank: %llu bytes differ on the worst rank)", coll_name(coll), kCandName[k],
             per_rank, (unsigned long long)nbad);
    why += std::string(why.empty() ? "" : "; ") + t;
  };
  std::vector<uint64_t> wb(cands.size(), 0);
  do {
    // allreduce: LL lines, the one-kernel fold, meet / body / done
    const size_t ll = (size_t)std::min<long>(c->ll_bytes, 4096) / 16 * 16;
    if (ll >= 16) {
      if ((rc = run(COLL_ALLREDUCE, ll, {xmpi_comm::CAND_LL}, wb.data())) != XMPI_OK) break;
      if (wb[xmpi_comm::CAND_LL]) reject(COLL_ALLREDUCE, xmpi_comm::CAND_LL, ll, wb[xmpi_comm::CAND_LL]);
    }
    XMPI_TRACE_STEP(c->rank, "self-check: LL lines done");
    if ((rc = run(COLL_ALLREDUCE, kFold, {xmpi_comm::CAND_FOLD}, wb.data())) != XMPI_OK) break;
    XMPI_TRACE_STEP(c->rank, "self-check: fold done");
    if (wb[xmpi_comm::CAND_FOLD]) {
      reject(COLL_ALLREDUCE, xmpi_comm::CAND_FOLD, kFold, wb[xmpi_comm::CAND_FOLD]);
      fold_wrong = true;
    }
    if (c->dsync_split_bytes > 0) {
      if ((rc = run(COLL_ALLREDUCE, kSplit, {xmpi_comm::CAND_SPLIT}, wb.data())) != XMPI_OK) break;
      if (wb[xmpi_comm::CAND_SPLIT] && !c->body_sys) {  // the ladder's first rung (see xmpi_tune)
        const uint64_t first = wb[xmpi_comm::CAND_SPLIT];
        c->body_sys = 1;
        if ((rc = run(COLL_ALLREDUCE, kSplit, {xmpi_comm::CAND_SPLIT}, wb.data())) != XMPI_OK) break;
        char t[200];
        snprintf(t, sizeof t, "allreduce: split gave wrong answers at %zu B per rank (%llu bytes differ on the worst rank); its system-scope data kernel %s",
                 kSplit, (unsigned long long)first, wb[xmpi_comm::CAND_SPLIT] ? "does too" : "is right and takes over (body_sys)");
        why += std::string(why.empty() ? "" : "; ") + t;
        if (wb[xmpi_comm::CAND_SPLIT]) c->body_sys = 0;
      }
      if (wb[xmpi_comm::CAND_SPLIT]) reject(COLL_ALLREDUCE, xmpi_comm::CAND_SPLIT, kSplit, wb[xmpi_comm::CAND_SPLIT]);
    }
    XMPI_TRACE_STEP(c->rank, "self-check: split done");
    // the other collectives' folds (other segment tables of the same kernel; bcast above zc_bcast_push_bytes: scatter + allgather)
    for (int coll : {(int)COLL_REDUCE, (int)COLL_ALLGATHER, (int)COLL_BCAST}) {
      const size_t per_rank = coll == COLL_ALLGATHER ? kFold / (size_t)c->size / 16 * 16 : coll == COLL_BCAST ? kBcast : kFold;
      if ((rc = run(coll, per_rank, {xmpi_comm::CAND_FOLD}, wb.data())) != XMPI_OK) break;
      if (wb[xmpi_comm::CAND_FOLD]) {
        reject(coll, xmpi_comm::CAND_FOLD, per_rank, wb[xmpi_comm::CAND_FOLD]);
        fold_wrong = true;
      }
    }
    if (rc != XMPI_OK) break;
    // Send / Receive: the receiver's direct pull out of the sender's registered memory, short (agent) and long (pull kernel).  Wrong:
    // the messages travel through the mail slots of the windows instead (p2p_direct_bytes < 0: pushed by the sender's copy engine,
    // drained locally -- two copies, no load over a link), checked in turn; wrong again, or no windows: xmpi_init fails on every rank.
    for (int attempt = 0; attempt < 2 && rc == XMPI_OK; attempt++) {
      uint64_t mine[2] = {0, 0}, worst[2] = {0, 0};
      if ((rc = p2p_check_round(c, chk, kP2PShort, &mine[0])) != XMPI_OK) break;
      if ((rc = p2p_check_round(c, chk, kP2PLong, &mine[1])) != XMPI_OK) break;
      if ((rc = vote_max(c, nullptr, mine, 2, nullptr, worst)) != XMPI_OK) break;
      if (!worst[0] && !worst[1]) break;
      char t[240];
      snprintf(t, sizeof t, "Send / Receive: %s gives wrong answers on this machine (%llu of %zu / %llu of %zu bytes differ on the worst rank)",
               attempt == 0 ? "the receiver's direct pull out of the sender's registered memory" : "the mail slots too", (unsigned long long)worst[0], kP2PShort,
               (unsigned long long)worst[1], kP2PLong);
      why += std::string(why.empty() ? "" : "; ") + t;
      if (attempt == 0 && c->windows_ok && c->p2p_direct_bytes >= 0) {
        c->p2p_direct_bytes = -1;
        c->p2p_rejected |= 1u;
        why += ": messages travel through the mail slots";
        continue;
      }
      c->p2p_rejected |= 2u;
      set_last_error("xmpi_init self-check: " + why + ": no way left to move a message between GPUs that gives right answers");
      rc = XMPI_ERR_HIP;
    }
  } while (false);
  c->tune_running = false;
  {
    std::lock_guard<std::mutex> g(c->coll_mu);
    chk.close();
  }
  if (rc != XMPI_OK) {
    if (c->rank == 0 && (c->p2p_rejected & 2u)) fprintf(stderr, "xmpi: %s\n", xmpi_last_error());
    if (!(c->p2p_rejected & 2u)) c->ctl->set_abort(rc);  // (a verdict every rank reached together needs no abort; a failure of this rank alone does)
    return rc;
  }
  // an untuned job has no table to route round a wrong fold: the one-kernel fold is what every collective's AUTO comes down to
  note_rejections(c, "xmpi_init self-check", why, fold_wrong);
  restore();
  c->selfcheck_ms = (now_seconds() - t_begin) * 1e3;
  return job_barrier(c);
}
}  // namespace xmpi

using namespace xmpi;

// XMPI_ERR_PEER out of a call that waited for a peer: say which peer, if the watchdog knows (ctl.cpp check_peers)
static int why_peer(xmpi_comm* c, int rc, const char* what) {
  if (rc == XMPI_ERR_PEER && c->ctl) set_last_error(std::string(what) + ": " + c->ctl->abort_reason());
  return rc;
}

extern "C" {

size_t xmpi_dtype_size(xmpi_dtype dtype) {
  switch (dtype) {
    case XMPI_U8: return 1;
    case XMPI_I32: return 4;
    case XMPI_I64: return 8;
    case XMPI_F16: return 2;
    case XMPI_F32: return 4;
    case XMPI_F64: return 8;
    case XMPI_BF16: return 2;
    default: return 0;
  }
}

const char* xmpi_version(void) { return "xmpi 0.1 (gfx950, HIP)"; }

const char* xmpi_degraded(const xmpi_comm* c) { return (c && !c->finalized) ? c->degraded_why.c_str() : ""; }

const char* xmpi_last_error(void) { return g_last_error.c_str(); }

const char* xmpi_strerror(int code) {
  switch (code) {
    case XMPI_OK: return "ok";
    case XMPI_ERR_ARG: return "invalid argument";
    case XMPI_ERR_HIP: return "HIP runtime error";
    case XMPI_ERR_BOOTSTRAP: return "bootstrap (shared control block) failed";
    case XMPI_ERR_TIMEOUT: return "timed out waiting for a peer";
    case XMPI_ERR_TAG_EXISTS: return "tag already in use";
    case XMPI_ERR_TRUNCATE: return "message larger than the receive buffer";
    case XMPI_ERR_NOMEM: return "out of memory";
    case XMPI_ERR_STATE: return "communicator not initialised";
    case XMPI_ERR_UNSUPPORTED: return "unsupported";
    case XMPI_ERR_NOGPU: return "no usable HIP device (there is no CPU fallback)";
    case XMPI_ERR_PEER: return "a peer rank failed";
    default: return "unknown xmpi error";
  }
}

int xmpi_init(int rank, int size, int device, const char* job_key, xmpi_comm** out) {
  if (!out || size < 1 || size > kMaxRanks || rank < 0 || rank >= size) {
    set_last_error("xmpi_init: bad rank/size/out");
    return XMPI_ERR_ARG;
  }
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev < 1) {
    (void)hipGetLastError();
    set_last_error("xmpi_init: no HIP device is visible; xmpi has no CPU fallback");
    return XMPI_ERR_NOGPU;
  }
  if (device < 0) device = rank % ndev;
  if (device >= ndev) {
    set_last_error("xmpi_init: device " + std::to_string(device) + " does not exist (" + std::to_string(ndev) + " visible)");
    return XMPI_ERR_ARG;
  }
  XMPI_HIP(hipSetDevice(device));

  CtlConfig cfg;
  cfg.lanes = (int32_t)std::min<long>(kMaxLanes, std::max<long>(1, env_long("XMPI_LANES", 2)));
  cfg.fifo_depth = (int32_t)std::min<long>(64, std::max<long>(2, env_long("XMPI_FIFO_DEPTH", 8)));
  cfg.slot_bytes = (uint64_t)std::max<long>(4096, env_long("XMPI_SLOT_BYTES", 8l << 20)) / 256 * 256;
  cfg.p2p_depth = (int32_t)std::min<long>(16, std::max<long>(2, env_long("XMPI_P2P_DEPTH", 2)));
  cfg.p2p_slot_bytes = (uint64_t)std::max<long>(4096, env_long("XMPI_P2P_SLOT_BYTES", 4l << 20)) / 256 * 256;
  cfg.host_lane_bytes = env_long("XMPI_HOST_LANES", 1) ? 1 : 0;  // a request: the creator of the block sizes and reserves them
  // Two different clocks.  XMPI_INIT_TIMEOUT_S (default 60) bounds the bootstrap only -- the reference's
  // -mpi-inittimeout (network.go:223-234,307-312).  XMPI_TIMEOUT_S is the no-progress limit of Send / Receive and
  // the collectives afterwards: default 0 = wait for ever, as the reference's blocking calls do (a receiver may
  // compute for minutes before it posts its Receive); tests set it so a bug shows up as an error, not a hang.
  const double timeout = (double)env_long("XMPI_INIT_TIMEOUT_S", 60);

  XMPI_TRACE_STEP(rank, "init: joining the control block");
  std::string key = (job_key && *job_key) ? job_key : "default";
  std::string err;
  Ctl* ctl = nullptr;
  int rc = Ctl::join(key, rank, size, cfg, timeout > 0 ? timeout : 3600.0, &ctl, &err);
  if (rc != XMPI_OK) {
    set_last_error("xmpi_init: " + err);
    return rc;
  }
  XMPI_TRACE_STEP(rank, "init: joined");
  xmpi_comm* c = new xmpi_comm;
  c->rank = rank;
  c->size = size;
  c->device = device;
  c->ctl = ctl;
  c->timeout_s = std::max<long>(0, env_long("XMPI_TIMEOUT_S", 0));
  const CtlConfig& g = ctl->cfg();  // rank 0's values are the job's
  c->lanes = g.lanes;
  c->fifo_depth = g.fifo_depth;
  c->slot_bytes = g.slot_bytes;
  c->p2p_depth = g.p2p_depth;
  c->p2p_slot_bytes = g.p2p_slot_bytes;
  c->channels = env_long("XMPI_CHANNELS", 0);  // 0 = one ring channel per available link direction
  c->piece_bytes = env_long("XMPI_PIECE_BYTES", 0);
  c->copy_engine = env_long("XMPI_COPY_ENGINE", 0);
  c->batch_copies = env_long("XMPI_BATCH_COPIES", 1) ? 1 : 0;
  c->oneshot_bytes = std::max<long>(0, env_long("XMPI_ONESHOT_BYTES", 1 << 20));
  c->zero_copy = env_long("XMPI_ZERO_COPY", 1) ? 1 : 0;
  c->zc_bcast_push_bytes = std::max<long>(0, env_long("XMPI_ZC_BCAST_PUSH_BYTES", 256 << 10));
  c->zc_group_launch = env_long("XMPI_ZC_GROUP_LAUNCH", 1) ? 1 : 0;
  c->p2p_direct_bytes = env_long("XMPI_P2P_DIRECT_BYTES", 1);
  if (getenv("XMPI_KERNEL_MODE")) set_kernel_mode((int)env_long("XMPI_KERNEL_MODE", -1));
  if (getenv("XMPI_GRID_CAP")) set_grid_cap((int)env_long("XMPI_GRID_CAP", 0));
  c->coll_region_bytes = (size_t)size * c->lanes * c->fifo_depth * c->slot_bytes;
  c->window_bytes = c->coll_region_bytes + (size_t)size * kMailEntries * c->p2p_depth * c->p2p_slot_bytes;

  auto fail = [&](int code) {
    ctl->set_abort(code);
    // what this attempt took from the per-process pools goes back (a later xmpi_init in this process finds it);
    // dsync_finalize stops the helper (it reads the control block), closes the peers' flag pages, frees the pinned
    // tables and gives the page back
    dsync_finalize(c);
    if (c->ctl_registered) (void)hipHostUnregister(ctl->base());
    if (c->p2p_tickets) (void)hipFree(c->p2p_tickets);
    if (c->p2p_done) (void)hipHostFree(c->p2p_done);
    if (c->p2p_cmd) (void)hipHostFree(c->p2p_cmd);
    if (c->p2p_rec) (void)hipFree(c->p2p_rec);
    if (c->dev_words) (void)hipFree(c->dev_words);
    (void)hipGetLastError();
    if (c->window) pool_release(c->window);
    if (c->local_stream && !c->shared_stream) stream_release(c->device, c->local_stream);
    delete ctl;
    delete c;
    return code;
  };
  XMPI_TRACE_STEP(rank, "init: window");
  c->window = (char*)pool_acquire(device, c->window_bytes, 0, nullptr, nullptr);
  if (!c->window) {
    hip_fail(hipGetLastError(), "hipMalloc(window)", __FILE__, __LINE__);
    return fail(XMPI_ERR_NOMEM);
  }
  RankInfo* me = ctl->info(rank);
  me->device = device;
  me->maps = 0;
  me->maps_why[0] = 0;
  me->window_addr = (uint64_t)(uintptr_t)c->window;
  me->window_bytes = c->window_bytes;
  (void)hipDeviceGetPCIBusId(me->busid, (int)sizeof me->busid, device);
  if (size > 1) {
    hipIpcMemHandle_t h;
    e = pool_handle(c->window, &h);
    if (e != hipSuccess) {
      hip_fail(e, "hipIpcGetMemHandle", __FILE__, __LINE__);
      return fail(XMPI_ERR_HIP);
    }
    static_assert(sizeof(h) <= sizeof(me->ipc_handle), "ipc handle size");
    memcpy(me->ipc_handle, &h, sizeof h);
  }
  XMPI_TRACE_STEP(rank, "init: stream");
  // this rank's stream, before anything is enqueued anywhere (the null stream would cost a second hardware queue)
  c->local_stream = stream_acquire(device);
  if (!c->local_stream) {
    hip_fail(hipGetLastError(), "hipStreamCreate", __FILE__, __LINE__);
    return fail(XMPI_ERR_HIP);
  }
  c->dsync = env_long("XMPI_DSYNC", 1) ? 1 : 0;
  c->dsync_split_bytes = std::max<long>(0, env_long("XMPI_DSYNC_SPLIT_BYTES", 4 << 20));
  c->xcd_check = env_long("XMPI_XCD_CHECK", 1) ? 1 : 0;
  c->body_sys = env_long("XMPI_BODY_SYS", -1);  // -1: decided by the XCD probe (dsync_prepare)
  c->ll_bytes = env_long("XMPI_LL_BYTES", -1);  // -1: decided when the job's layout is known (dsync_connect)
  c->agent_ll = std::max<long>(0, std::min<long>(env_long("XMPI_AGENT_LL", 1), 2));
  c->agent_ll_bytes = std::max<long>(0, std::min<long>((long)kLLMaxPayload, env_long("XMPI_AGENT_LL_BYTES", 8192)));
  c->sched_channels = std::max<long>(0, env_long("XMPI_SCHED_CHANNELS", 0));
  c->sched_grid = std::max<long>(0, env_long("XMPI_SCHED_GRID", 0));
  c->tree_piece_bytes = std::max<long>(4096, env_long("XMPI_TREE_PIECE_BYTES", 256 << 10));
  memset(c->tune_algo, -1, sizeof c->tune_algo);
  memset(c->tune_split, -1, sizeof c->tune_split);
  memset(c->tune_unroll, 0, sizeof c->tune_unroll);
  c->dsync_grid_cap = std::max<long>(0, env_long("XMPI_DSYNC_GRID", 0));
  XMPI_TRACE_STEP(rank, "init: flag page");
  (void)dsync_prepare(c);  // this rank's flag page (device-synchronised collectives), published with the window
  XMPI_TRACE_STEP(rank, "init: published, waiting for the peers' windows");
  me->state.store(2, std::memory_order_release);
  rc = ctl->wait_all_state(2, timeout > 0 ? timeout : 3600.0);
  if (rc != XMPI_OK) {
    set_last_error("xmpi_init: a peer did not publish its HBM window");
    return fail(rc);
  }
  XMPI_TRACE_STEP(rank, "init: mapping the peers' windows");
  const int mypid = (int)getpid();
  for (int p = 0; p < size; p++) {
    if (p == rank) {
      c->peer_window[p] = c->window;
      continue;
    }
    RankInfo* pi = ctl->info(p);
    if (pi->pid == mypid) {  // rank hosted by a thread of this process
      c->peer_window[p] = (char*)(uintptr_t)pi->window_addr;
      if (pi->device != device) {
        e = hipDeviceEnablePeerAccess(pi->device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
          hip_fail(e, "hipDeviceEnablePeerAccess", __FILE__, __LINE__);
          return fail(XMPI_ERR_HIP);
        }
        (void)hipGetLastError();
      }
    } else {
      void* ptr = nullptr;
      e = ipc_open_shared(pi->pid, pi->window_addr, pi->ipc_handle, &ptr);
      if (e != hipSuccess) {
        // not the end of the job: this rank says so in the vote below (dsync_connect) and every rank keeps to what does not need
        // the windows -- the device-synchronised collectives on registered buffers, Send / Receive out of registered buffers
        // and through the host lanes
        (void)hipGetLastError();
        if (!c->window_map_failed)
          snprintf(me->maps_why, sizeof me->maps_why, "rank %d: hipIpcOpenMemHandle(window of rank %d): %s", rank, p, hipGetErrorString(e));
        c->window_map_failed = true;
        continue;
      }
      c->peer_window[p] = (char*)ptr;
      c->peer_opened[p] = true;
    }
  }
  // Ranks hosted by threads of one process on one GPU share a single in-order stream: HBM is their
  // only shared resource, so concurrent streams would only make every kernel slower, while one
  // stream lets each kernel run at full-chip bandwidth (and needs no cross-stream events).
  bool colocated = false;
  for (int p = 0; p < size; p++)
    if (p != rank && ctl->info(p)->pid == mypid && ctl->info(p)->device == device) {
      colocated = true;
      c->peer_coloc[p] = true;
    }
  const long shared_env = env_long("XMPI_SHARED_STREAM", -1);
  c->shared_stream = shared_env < 0 ? colocated : shared_env != 0;
  if (c->shared_stream) {
    hipStream_t s = shared_stream_for(device);
    if (!s) {
      hip_fail(hipGetLastError(), "hipStreamCreate(shared)", __FILE__, __LINE__);
      return fail(XMPI_ERR_HIP);
    }
    for (int p = 0; p < size; p++) c->send_stream[p] = c->recv_stream[p] = (p == rank) ? nullptr : s;
    (void)hipStreamSynchronize(c->local_stream);
    stream_release(device, c->local_stream);
    c->local_stream = c->batch_send_stream = c->batch_recv_stream = s;
  }
  // Otherwise ONE stream (made above); the per-peer streams of the staged schedules are made when a staged schedule
  // first runs (ensure_streams).  Every stream costs the process a hardware queue (the runtime multiplexes streams
  // over GPU_MAX_HW_QUEUES of them), and a GPU runs only a few dozen queues at once: 8 processes x 4 queues on one
  // GPU were time-sliced by the scheduler -- 22 ms per collective instead of 40 us (profiles/r02).
  // the control block as the GPU sees it: kernels read the job's abort flag there and write the ack of a
  // point-to-point message straight into its mail entry (engine.cpp)
  // (with the host lanes behind it, so that a lane's piece is copied to a device destination by DMA; the control
  // structures alone if the runtime will not pin that much)
  if (ctl->host_lane_bytes() > 0 && hipHostRegister(ctl->base(), ctl->bytes(), hipHostRegisterMapped) == hipSuccess) c->lanes_dev_ok = true;
  if (c->lanes_dev_ok ||
      ((void)hipGetLastError(), hipHostRegister(ctl->base(), Ctl::layout_bytes(size), hipHostRegisterMapped) == hipSuccess)) {
    c->ctl_registered = true;
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, ctl->base(), 0) == hipSuccess) c->ctl_dev = (char*)dev;
  }
  if (!c->ctl_dev) c->lanes_dev_ok = false;
  (void)hipGetLastError();
  // (two more for the copy kernels that carry a host slice into / out of a collective's stand-in, dsync.cpp)
  if (hipMalloc((void**)&c->p2p_tickets, (xmpi_comm::kP2PDoneSlots + 2) * sizeof(uint32_t)) == hipSuccess)
    (void)hipMemsetAsync(c->p2p_tickets, 0, (xmpi_comm::kP2PDoneSlots + 2) * sizeof(uint32_t), c->local_stream);
  else c->p2p_tickets = nullptr;
  (void)hipGetLastError();
  // completion words the GPU writes and a host thread polls (stream-ordered Send / Receive: slots 0..63; the pull kernels
  // of the blocking Receive: 64..127)
  if (hipHostMalloc((void**)&c->p2p_done, sizeof(uint64_t) * 4 * 2 * xmpi_comm::kP2PDoneSlots, hipHostMallocMapped) == hipSuccess) {
    memset(c->p2p_done, 0, sizeof(uint64_t) * 4 * 2 * xmpi_comm::kP2PDoneSlots);
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, c->p2p_done, 0) == hipSuccess) c->p2p_done_dev = (uint64_t*)dev;
  } else {
    c->p2p_done = nullptr;
  }
  (void)hipGetLastError();
  c->p2p_kernel_ack = env_long("XMPI_P2P_KERNEL_ACK", 1) ? 1 : 0;
  c->p2p_agent_us = std::max<long>(0, env_long("XMPI_P2P_AGENT_US", 40));
  c->ll_agent_us = std::max<long>(0, env_long("XMPI_LL_AGENT_US", c->p2p_agent_us));
  c->p2p_grid_cap = std::max<long>(0, std::min<long>(env_long("XMPI_P2P_GRID_CAP", 0), 4096));
  if (hipHostMalloc((void**)&c->p2p_cmd, 128, hipHostMallocMapped) == hipSuccess) {  // (two records: the receive agent's, the LL agent's)
    memset(c->p2p_cmd, 0, 128);
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, c->p2p_cmd, 0) == hipSuccess) {
      c->p2p_cmd_dev = (uint64_t*)dev;
      c->ll_cmd = c->p2p_cmd + 8;
      c->ll_cmd_dev = c->p2p_cmd_dev + 8;
    }
  } else {
    c->p2p_cmd = nullptr;
  }
  if (hipMalloc((void**)&c->p2p_rec, 64) == hipSuccess)
    (void)hipMemsetAsync(c->p2p_rec, 0, 64, c->local_stream);
  else c->p2p_rec = nullptr;
  (void)hipGetLastError();
  XMPI_TRACE_STEP(rank, "init: connecting flag pages");
  rc = dsync_connect(c, timeout > 0 ? timeout : 3600.0);
  if (rc != XMPI_OK) return fail(rc);
  // the helper thread: maps what peers register, and watches over their processes -- unless one of them cannot be seen from here
  // even now, when it certainly lives (ranks in different pid namespaces sharing /dev/shm: no way to ask, so nobody asks)
  c->watchdog_ms = std::max<long>(0, env_long("XMPI_WATCHDOG_MS", 50));
  for (int p = 0; p < size && c->watchdog_ms > 0; p++)
    if (p != rank && ctl->peer_gone(p)) c->watchdog_ms = 0;
  ctl->set_watch(c->watchdog_ms > 0);
  dsync_start_helper(c);
  XMPI_TRACE_STEP(rank, "init: final barrier");
  c->dsync_unroll = env_long("XMPI_DSYNC_UNROLL", c->dsync_sharers > 1 ? 1 : 2);
  // ranks sharing a GPU: fewer, longer blocks (8 processes on one MI355X: 4 MiB 170 -> 90 us, 16 MiB 231 -> 169 us);
  // a rank with a GPU to itself keeps one tile per block -- over links more waves in flight is what hides latency
  c->dsync_tiles = std::max<long>(1, env_long("XMPI_DSYNC_TILES", c->dsync_sharers > 1 ? 8 : 1));
  if (hipMalloc((void**)&c->dev_words, 4 * sizeof(uint64_t)) != hipSuccess) {
    hip_fail(hipGetLastError(), "hipStreamCreate/hipMalloc", __FILE__, __LINE__);
    return fail(XMPI_ERR_HIP);
  }
  {  // before the barrier: no rank of this process can allocate before every one of them has said so
    bool shared = false;
    for (int p = 0; p < size; p++) shared = shared || (p != rank && ctl->info(p)->pid == (int32_t)getpid());
    heap_colour_seed(rank, shared);
  }
  rc = ctl->barrier(timeout > 0 ? timeout : 3600.0);
  if (rc != XMPI_OK) {
    set_last_error("xmpi_init: barrier failed");
    return fail(rc);
  }
  heap_comm_created();
  // The ranks sit on different GPUs (or XMPI_SELFCHECK=1): what untuned AUTO can reach is tried on patterned inputs before the
  // first caller's data goes through it (init_selfcheck above).  A job that tunes right here checks every candidate anyway.
  const long tune_bytes = env_long("XMPI_AUTOTUNE_BYTES", 0);
  c->selfcheck = env_long("XMPI_SELFCHECK", -1);
  if (c->selfcheck < 0) {
    bool spread = false;
    for (int p = 0; p < size; p++) spread = spread || strncmp(ctl->info(p)->busid, me->busid, sizeof me->busid) != 0;
    c->selfcheck = spread ? 1 : 0;
  }
  if (c->selfcheck && dsync_usable(c) && !(tune_bytes > 0)) {
    XMPI_TRACE_STEP(rank, "init: self-check");
    rc = init_selfcheck(c);
    if (rc != XMPI_OK) {
      (void)xmpi_finalize(c);
      return rc;
    }
  }
  // XMPI_AUTOTUNE_BYTES=N: the library times its schedules for messages up to N bytes right here (xmpi_tune), so that a
  // program that knows nothing about tuning gets the schedule a benchmark would pick on this node; every rank sees the
  // same environment, so it is collective.  Default: off (a few hundred milliseconds and 2 x N bytes of HBM per rank).
  if (tune_bytes > 0 && size > 1) {
    XMPI_TRACE_STEP(rank, "init: tuning");
    rc = xmpi_tune(c, (size_t)tune_bytes);
    if (rc != XMPI_OK) {
      (void)xmpi_finalize(c);
      return rc;
    }
  }
  XMPI_TRACE_STEP(rank, "init: done");
  *out = c;
  return XMPI_OK;
}

int xmpi_finalize(xmpi_comm* c) {
  if (!c) return XMPI_ERR_STATE;
  if (c->finalized) return XMPI_OK;
  stop_worker(c);  // outstanding non-blocking collectives complete first
  p2p_agent_stop(c);  // the receive agent (if it still lingers) is told to go
  ll_agent_stop(c);   // ... and the LL agent
  XMPI_TRACE_STEP(c->rank, "finalize: device sync");
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  XMPI_TRACE_STEP(c->rank, "finalize: barrier");
  // nobody may still be writing into a window that is about to be unmapped
  if (!c->ctl->aborted()) {
    Backoff bo;
    arm(bo, c);
    (void)c->ctl->barrier(wait_limit(c), &bo);
  }
  XMPI_TRACE_STEP(c->rank, "finalize: closing");
  zc_close_peers(c);
  dsync_finalize(c);
  for (int p = 0; p < c->size; p++) {
    if (c->peer_opened[p]) ipc_close_shared(c->peer_window[p]);
    if (c->shared_stream) continue;  // the per-device shared stream outlives communicators
    stream_release(c->device, c->send_stream[p]);
    stream_release(c->device, c->recv_stream[p]);
  }
  if (!c->shared_stream) stream_release(c->device, c->local_stream);
  if (!c->shared_stream) {
    stream_release(c->device, c->batch_send_stream);
    stream_release(c->device, c->batch_recv_stream);
  }
  for (hipStream_t s : c->p2p_streams) stream_release(c->device, s);
  if (c->agent_stream) stream_release(c->device, c->agent_stream);
  if (c->ll_agent_stream) stream_release(c->device, c->ll_agent_stream);
  for (hipEvent_t e : c->ev_free) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->ev_timed_free) (void)hipEventDestroy(e);
  if (!c->ctl->aborted()) (void)c->ctl->barrier(wait_limit(c));
  if (c->ctl_registered) (void)hipHostUnregister(c->ctl->base());
  if (c->p2p_tickets) (void)hipFree(c->p2p_tickets);
  if (c->p2p_done) (void)hipHostFree(c->p2p_done);
  if (c->p2p_cmd) (void)hipHostFree(c->p2p_cmd);
  if (c->p2p_bounce) (void)hipHostFree(c->p2p_bounce);  // (engine.cpp p2p_recv: device -> host slice through pinned memory)
  c->p2p_bounce = c->p2p_bounce_dev = nullptr;
  if (c->p2p_rec) (void)hipFree(c->p2p_rec);
  if (c->window) pool_release(c->window);  // exported memory is never given back by the runtime: the next communicator reuses it
  if (c->temp) (void)hipFree(c->temp);
  if (c->host_stage) (void)hipFree(c->host_stage);
  if (c->dev_words) (void)hipFree(c->dev_words);
  heap_comm_destroyed(c);  // last communicator of the process: empty arenas go back to the device
  XMPI_TRACE_STEP(c->rank, "finalize: done");
  c->ctl->info(c->rank)->state.store(3, std::memory_order_release);
  delete c->ctl;
  c->ctl = nullptr;
  c->finalized = true;
  delete c;
  return XMPI_OK;
}

int xmpi_rank(const xmpi_comm* c) { return (c && !c->finalized && c->size > 0) ? c->rank : -1; }
int xmpi_size(const xmpi_comm* c) { return (c && !c->finalized) ? c->size : 0; }
int xmpi_device(const xmpi_comm* c) { return (c && !c->finalized) ? c->device : -1; }

int xmpi_barrier(xmpi_comm* c) {
  XMPI_ENTER(c);
  drain_worker(c);
  dsync_service(c);
  Backoff bo;
  arm(bo, c);
  int rc = c->ctl->barrier(wait_limit(c), &bo);
  if (rc != XMPI_OK) set_last_error("barrier: a peer did not arrive");
  return rc;
}

void* xmpi_malloc(xmpi_comm* c, size_t bytes) {
  if (!c || c->finalized || use_device(c) != XMPI_OK) return nullptr;
  void* p = heap_alloc(c->device, bytes);  // a block of a registered arena (heap.cpp)
  if (!p) hip_fail(hipGetLastError(), "hipMalloc(arena)", __FILE__, __LINE__);
  return p;
}

int xmpi_free(xmpi_comm* c, void* p) {
  XMPI_ENTER(c);
  if (p && !heap_free(p)) {
    set_last_error("free: not a live buffer of xmpi_malloc");
    return XMPI_ERR_ARG;
  }
  return XMPI_OK;
}

int xmpi_register(xmpi_comm* c, void* p, size_t bytes) {
  XMPI_ENTER(c);
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  if (!p || !is_device_pointer(p) || hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p) != hipSuccess) {
    (void)hipGetLastError();
    set_last_error("register: not a device allocation");
    return XMPI_ERR_ARG;
  }
  if ((char*)p + bytes > (char*)base + size) {
    set_last_error("register: the range runs past its allocation");
    return XMPI_ERR_ARG;
  }
  if (heap_owns(p)) return XMPI_OK;  // memory of xmpi_malloc is registered as it is
  return registry_add((void*)base, size, c->device);
}

int xmpi_deregister(xmpi_comm* c, void* p) {
  XMPI_ENTER(c);
  if (heap_owns(p)) {
    set_last_error("deregister: memory of xmpi_malloc stays registered until xmpi_free");
    return XMPI_ERR_ARG;
  }
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  if (!p || hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p) != hipSuccess) {
    (void)hipGetLastError();
    registry_remove(c, p);
    return XMPI_OK;
  }
  registry_remove(c, (void*)base);
  return XMPI_OK;
}

int xmpi_memcpy(xmpi_comm* c, void* dst, const void* src, size_t bytes) {
  XMPI_ENTER(c);
  if (bytes) {  // on the communicator's stream (the null stream would cost the process another hardware queue)
    XMPI_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, c->local_stream));
    XMPI_HIP(hipStreamSynchronize(c->local_stream));
  }
  return XMPI_OK;
}

int xmpi_memset(xmpi_comm* c, void* dst, int byte, size_t bytes) {
  XMPI_ENTER(c);
  if (bytes) {
    XMPI_HIP(hipMemsetAsync(dst, byte, bytes, c->local_stream));
    XMPI_HIP(hipStreamSynchronize(c->local_stream));
  }
  return XMPI_OK;
}

int xmpi_sync(xmpi_comm* c) {
  XMPI_ENTER(c);
  XMPI_HIP(hipDeviceSynchronize());
  return XMPI_OK;
}

int xmpi_send(xmpi_comm* c, const void* buf, size_t count, xmpi_dtype dtype, int dest, int tag) {
  XMPI_ENTER(c);
  const size_t es = xmpi_dtype_size(dtype);
  if (!es || dest < 0 || dest >= c->size || (count && !buf)) {
    set_last_error("send: bad dtype / destination / buffer");
    return XMPI_ERR_ARG;
  }
  return why_peer(c, p2p_send(c, buf, count * es, (int)dtype, dest, tag), "send");
}

int xmpi_send_nowait(xmpi_comm* c, const void* buf, size_t count, xmpi_dtype dtype, int dest, int tag) {
  XMPI_ENTER(c);
  const size_t es = xmpi_dtype_size(dtype);
  if (!es || dest < 0 || dest >= c->size || (count && !buf)) {
    set_last_error("send: bad dtype / destination / buffer");
    return XMPI_ERR_ARG;
  }
  return why_peer(c, p2p_send(c, buf, count * es, (int)dtype, dest, tag, /*wait_ack=*/false), "send");
}

int xmpi_wait(xmpi_comm* c, int dest, int tag) {
  XMPI_ENTER(c);
  if (dest < 0 || dest >= c->size) return XMPI_ERR_ARG;
  return why_peer(c, p2p_wait(c, dest, tag), "wait");
}

int xmpi_recv(xmpi_comm* c, void* buf, size_t capacity, xmpi_dtype dtype, int src, int tag, size_t* got) {
  XMPI_ENTER(c);
  const size_t es = xmpi_dtype_size(dtype);
  if (!es || src < 0 || src >= c->size || (capacity && !buf)) {
    set_last_error("receive: bad dtype / source / buffer");
    return XMPI_ERR_ARG;
  }
  size_t got_bytes = 0;
  int rc = p2p_recv(c, buf, capacity * es, (int)dtype, src, tag, &got_bytes);
  if (got) *got = got_bytes / es;
  return why_peer(c, rc, "receive");
}

int xmpi_probe(xmpi_comm* c, int src, int tag, size_t* count, xmpi_dtype* dtype) {
  XMPI_ENTER(c);
  if (src < 0 || src >= c->size) return XMPI_ERR_ARG;
  size_t bytes = 0;
  int dt = 0;
  int rc = p2p_probe(c, src, tag, &bytes, &dt);
  if (rc != XMPI_OK) return rc;
  const size_t es = xmpi_dtype_size((xmpi_dtype)dt);
  if (count) *count = es ? bytes / es : 0;
  if (dtype) *dtype = (xmpi_dtype)dt;
  return XMPI_OK;
}

int xmpi_bcast(xmpi_comm* c, void* buf, size_t count, xmpi_dtype dtype, int root, int algo) {
  XMPI_ENTER(c);
  return collective(c, COLL_BCAST, algo, root, buf, buf, count, (int)dtype, XMPI_SUM);
}

int xmpi_reduce(xmpi_comm* c, const void* sendbuf, void* recvbuf, size_t count, xmpi_dtype dtype, xmpi_op op, int root,
                int algo) {
  XMPI_ENTER(c);
  // recvbuf is only significant at the root; other ranks may pass NULL
  void* rb = recvbuf ? recvbuf : const_cast<void*>(sendbuf);
  if (c->rank == root && !recvbuf) {
    set_last_error("reduce: root needs a receive buffer");
    return XMPI_ERR_ARG;
  }
  return collective(c, COLL_REDUCE, algo, root, sendbuf, rb, count, (int)dtype, (int)op);
}

int xmpi_allreduce(xmpi_comm* c, const void* sendbuf, void* recvbuf, size_t count, xmpi_dtype dtype, xmpi_op op,
                   int algo) {
  XMPI_ENTER(c);
  return collective(c, COLL_ALLREDUCE, algo, 0, sendbuf, recvbuf, count, (int)dtype, (int)op);
}

int xmpi_allgather(xmpi_comm* c, const void* sendbuf, void* recvbuf, size_t count, xmpi_dtype dtype, int algo) {
  XMPI_ENTER(c);
  return collective(c, COLL_ALLGATHER, algo, 0, sendbuf, recvbuf, count, (int)dtype, XMPI_SUM);
}

// ---- stream-ordered forms --------------------------------------------------------------------------
static int on_stream(xmpi_comm* c, int coll, int root, const void* sendbuf, void* recvbuf, size_t count, int dtype, int op,
                     void* stream) {
  const size_t es = xmpi_dtype_size((xmpi_dtype)dtype);
  if (es == 0 || op < 0 || op >= XMPI_OP_COUNT || root < 0 || root >= c->size) {
    set_last_error("bad dtype / op / root");
    return XMPI_ERR_ARG;
  }
  if (count == 0) return XMPI_OK;
  if (!sendbuf || !recvbuf) {
    set_last_error("null buffer");
    return XMPI_ERR_ARG;
  }
  drain_worker(c);
  std::lock_guard<std::mutex> g(c->coll_mu);
  hipStream_t s = stream ? (hipStream_t)stream : c->local_stream;
  if (c->size == 1) {  // a job of one: the result is the input
    if (coll != COLL_BCAST && sendbuf != recvbuf) XMPI_HIP(hipMemcpyAsync(recvbuf, sendbuf, count * es, hipMemcpyDefault, s));
    return XMPI_OK;
  }
  if (!dsync_usable(c)) {
    // ranks sharing a (process, GPU) pair meet on the host (see dsync.cpp): order the call after the stream's
    // work, run it blocking.  Correct, but the host waits -- the layout this API is for is one process per GPU.
    XMPI_HIP(hipStreamSynchronize(s));
    bool done = false;
    int rc = zero_copy_collective(c, coll, root, sendbuf, recvbuf, count, dtype, op, false, &done);
    if (rc != XMPI_OK || done) return rc;
    set_last_error("stream-ordered collectives need buffers of xmpi_malloc / xmpi_register when ranks share a process and a GPU");
    return XMPI_ERR_UNSUPPORTED;
  }
  // device memory only: a copy to or from pageable host memory would block this call until the kernel before it
  // has ended, i.e. until every peer has arrived -- the opposite of what the stream-ordered forms are for
  if (!is_device_pointer(sendbuf) || !is_device_pointer(recvbuf)) {
    set_last_error("stream-ordered collectives take device memory (use the blocking forms for host buffers)");
    return XMPI_ERR_ARG;
  }
  return dsync_collective(c, coll, root, sendbuf, recvbuf, count, dtype, op, s, /*blocking=*/false, XMPI_ALGO_AUTO);
}

int xmpi_allreduce_on_stream(xmpi_comm* c, const void* sendbuf, void* recvbuf, size_t count, xmpi_dtype dtype, xmpi_op op,
                             void* stream) {
  XMPI_ENTER(c);
  return on_stream(c, COLL_ALLREDUCE, 0, sendbuf, recvbuf, count, (int)dtype, (int)op, stream);
}

int xmpi_allgather_on_stream(xmpi_comm* c, const void* sendbuf, void* recvbuf, size_t count, xmpi_dtype dtype, void* stream) {
  XMPI_ENTER(c);
  return on_stream(c, COLL_ALLGATHER, 0, sendbuf, recvbuf, count, (int)dtype, XMPI_SUM, stream);
}

int xmpi_bcast_on_stream(xmpi_comm* c, void* buf, size_t count, xmpi_dtype dtype, int root, void* stream) {
  XMPI_ENTER(c);
  return on_stream(c, COLL_BCAST, root, buf, buf, count, (int)dtype, XMPI_SUM, stream);
}

int xmpi_reduce_on_stream(xmpi_comm* c, const void* sendbuf, void* recvbuf, size_t count, xmpi_dtype dtype, xmpi_op op,
                          int root, void* stream) {
  XMPI_ENTER(c);
  void* rb = recvbuf ? recvbuf : const_cast<void*>(sendbuf);
  if (c->rank == root && !recvbuf) {
    set_last_error("reduce: root needs a receive buffer");
    return XMPI_ERR_ARG;
  }
  return on_stream(c, COLL_REDUCE, root, sendbuf, rb, count, (int)dtype, (int)op, stream);
}

void* xmpi_stream_create(xmpi_comm* c) {
  if (!c || c->finalized || use_device(c) != XMPI_OK) return nullptr;
  hipStream_t s = nullptr;
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
    hip_fail(hipGetLastError(), "hipStreamCreate", __FILE__, __LINE__);
    return nullptr;
  }
  return (void*)s;
}

int xmpi_stream_destroy(xmpi_comm* c, void* stream) {
  XMPI_ENTER(c);
  if (stream) {
    std::lock_guard<std::mutex> g(c->coll_mu);
    if (c->dsync_last_stream == (hipStream_t)stream) {  // the next device-synchronised launch would order itself behind it
      (void)hipStreamSynchronize((hipStream_t)stream);
      c->dsync_last_stream = nullptr;
    }
    XMPI_HIP(hipStreamDestroy((hipStream_t)stream));
  }
  return XMPI_OK;
}

// ---- hipGraph capture of stream-ordered collectives ---------------------------------------------------------------
// A device-synchronised collective is one kernel launch whose arguments do not change from call to call on the same
// buffers (its epoch is counted on the device, DsyncPage::epoch_now), so a sequence of them -- with the caller's own
// kernels in between -- can be captured once and replayed: the launch-bound inner loop of an iterative solver becomes one
// hipGraphLaunch per iteration.  Every rank captures the same sequence and replays it the same number of times.
int xmpi_graph_begin(xmpi_comm* c, void* stream) {
  XMPI_ENTER(c);
  if (!stream) {
    set_last_error("graph capture needs a stream of the caller's (xmpi_stream_create)");
    return XMPI_ERR_ARG;
  }
  if (!dsync_usable(c)) {
    set_last_error("graph capture needs ranks that meet on the device (one process per GPU)");
    return XMPI_ERR_UNSUPPORTED;
  }
  // relaxed: other threads of this process (the helper that maps peers' buffers) may keep calling the runtime
  XMPI_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeRelaxed));
  return XMPI_OK;
}

int xmpi_graph_end(xmpi_comm* c, void* stream, void** graph_out) {
  XMPI_ENTER(c);
  if (!stream || !graph_out) return XMPI_ERR_ARG;
  hipGraph_t g = nullptr;
  XMPI_HIP(hipStreamEndCapture((hipStream_t)stream, &g));
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) return hip_fail(e, "hipGraphInstantiate", __FILE__, __LINE__);
  *graph_out = (void*)exec;
  return XMPI_OK;
}

int xmpi_graph_launch(xmpi_comm* c, void* graph, void* stream) {
  XMPI_ENTER(c);
  if (!graph) return XMPI_ERR_ARG;
  hipStream_t s = stream ? (hipStream_t)stream : c->local_stream;
  std::lock_guard<std::mutex> g(c->coll_mu);
  dsync_graph_launched(c, s, /*before=*/true);
  XMPI_HIP(hipGraphLaunch((hipGraphExec_t)graph, s));
  dsync_graph_launched(c, s, /*before=*/false);
  return XMPI_OK;
}

int xmpi_graph_destroy(xmpi_comm* c, void* graph) {
  XMPI_ENTER(c);
  if (graph) XMPI_HIP(hipGraphExecDestroy((hipGraphExec_t)graph));
  return XMPI_OK;
}

int xmpi_stream_sync(xmpi_comm* c, void* stream) {
  XMPI_ENTER(c);
  hipStream_t s = stream ? (hipStream_t)stream : c->local_stream;
  // poll, serving the peers meanwhile: a peer may be waiting for this rank to map a buffer it just registered
  hipEvent_t fin = ev_get(c, false);
  if (!fin) return XMPI_ERR_HIP;
  XMPI_HIP(hipEventRecord(fin, s));
  Backoff bo;
  arm(bo, c);
  for (;;) {
    const hipError_t e = hipEventQuery(fin);
    if (e == hipSuccess) break;
    if (e != hipErrorNotReady) return hip_fail(e, "hipEventQuery", __FILE__, __LINE__);
    (void)hipGetLastError();
    bo.pause();
  }
  ev_put(c, fin, false);
  std::lock_guard<std::mutex> g(c->coll_mu);
  const int prc = c->dsync_ok ? dsync_p2p_reap(c) : XMPI_OK;  // the stream-ordered sends / receives that have completed
  const int crc = dsync_check(c);
  if (crc == XMPI_OK && prc == XMPI_ERR_PEER) set_last_error("send / receive: the job was aborted while the kernel waited: " + c->ctl->abort_reason());
  return crc != XMPI_OK ? crc : prc;
}

// ---- stream-ordered Send / Receive ----------------------------------------------------------------------------------
static int p2p_on_stream_ok(xmpi_comm* c, const void* buf, size_t count, xmpi_dtype dtype, int peer) {
  const size_t es = xmpi_dtype_size(dtype);
  if (!es || peer < 0 || peer >= c->size || (count && !buf)) {
    set_last_error("send / receive: bad dtype / peer / buffer");
    return XMPI_ERR_ARG;
  }
  if (!c->dsync_ok || !c->dpage) {
    set_last_error("stream-ordered send / receive need ranks that meet on the device (one process per GPU)");
    return XMPI_ERR_UNSUPPORTED;
  }
  if (count && !is_device_pointer(buf)) {
    set_last_error("stream-ordered send / receive take device memory (use the blocking forms for host buffers)");
    return XMPI_ERR_ARG;
  }
  return XMPI_OK;
}

int xmpi_send_on_stream(xmpi_comm* c, const void* buf, size_t count, xmpi_dtype dtype, int dest, int tag, void* stream) {
  XMPI_ENTER(c);
  const int rc = p2p_on_stream_ok(c, buf, count, dtype, dest);
  if (rc != XMPI_OK) return rc;
  std::lock_guard<std::mutex> g(c->coll_mu);
  return dsync_send(c, buf, count * xmpi_dtype_size(dtype), (int)dtype, dest, tag, (hipStream_t)stream);
}

int xmpi_recv_on_stream(xmpi_comm* c, void* buf, size_t capacity, xmpi_dtype dtype, int src, int tag, void* stream) {
  XMPI_ENTER(c);
  const int rc = p2p_on_stream_ok(c, buf, capacity, dtype, src);
  if (rc != XMPI_OK) return rc;
  std::lock_guard<std::mutex> g(c->coll_mu);
  return dsync_recv(c, buf, capacity * xmpi_dtype_size(dtype), (int)dtype, src, tag, (hipStream_t)stream);
}

int xmpi_iallreduce(xmpi_comm* c, const void* sendbuf, void* recvbuf, size_t count, xmpi_dtype dtype, xmpi_op op, int algo,
                    xmpi_request** req) {
  XMPI_ENTER(c);
  if (!req) return XMPI_ERR_ARG;
  *req = submit(c, [=] { return xmpi_allreduce(c, sendbuf, recvbuf, count, dtype, op, algo); });
  return XMPI_OK;
}

int xmpi_iallgather(xmpi_comm* c, const void* sendbuf, void* recvbuf, size_t count, xmpi_dtype dtype, int algo,
                    xmpi_request** req) {
  XMPI_ENTER(c);
  if (!req) return XMPI_ERR_ARG;
  *req = submit(c, [=] { return xmpi_allgather(c, sendbuf, recvbuf, count, dtype, algo); });
  return XMPI_OK;
}

int xmpi_ibcast(xmpi_comm* c, void* buf, size_t count, xmpi_dtype dtype, int root, int algo, xmpi_request** req) {
  XMPI_ENTER(c);
  if (!req) return XMPI_ERR_ARG;
  *req = submit(c, [=] { return xmpi_bcast(c, buf, count, dtype, root, algo); });
  return XMPI_OK;
}

int xmpi_ireduce(xmpi_comm* c, const void* sendbuf, void* recvbuf, size_t count, xmpi_dtype dtype, xmpi_op op, int root,
                 int algo, xmpi_request** req) {
  XMPI_ENTER(c);
  if (!req) return XMPI_ERR_ARG;
  *req = submit(c, [=] { return xmpi_reduce(c, sendbuf, recvbuf, count, dtype, op, root, algo); });
  return XMPI_OK;
}

int xmpi_request_test(xmpi_request* r, int* done) {
  if (!r || !done) return XMPI_ERR_ARG;
  std::lock_guard<std::mutex> l(r->mu);
  *done = r->done ? 1 : 0;
  return XMPI_OK;
}

int xmpi_request_wait(xmpi_request* r) {
  if (!r) return XMPI_ERR_ARG;
  int rc;
  {
    std::unique_lock<std::mutex> l(r->mu);
    r->cv.wait(l, [r] { return r->done; });
    rc = r->rc;
    if (rc != XMPI_OK) set_last_error(r->err);
  }
  delete r;
  return rc;
}

int xmpi_allreduce_repeat(xmpi_comm* c, const void* sendbuf, void* recvbuf, size_t count, xmpi_dtype dtype, xmpi_op op,
                          int algo, int iters) {
  XMPI_ENTER(c);
  // Ranks that meet on the device: the steps are ENQUEUED back to back on the communicator's stream and waited for
  // once -- what a stream-ordered caller does, and what the device rendezvous is for (no host round trip per
  // step).  Sampled launches carry their own events, read when the last step has been waited for.
  const bool on_device = count > 0 && sendbuf && recvbuf && xmpi_dtype_size(dtype) && algo >= 0 && algo < XMPI_ALGO_COUNT &&
                         dsync_takes(c, COLL_ALLREDUCE, algo) && is_device_pointer(sendbuf) && is_device_pointer(recvbuf);
  if (on_device) {
    drain_worker(c);
    std::lock_guard<std::mutex> g(c->coll_mu);
    for (int i = 0; i < iters; i++) {
      const int rc = dsync_collective(c, COLL_ALLREDUCE, 0, sendbuf, recvbuf, count, (int)dtype, (int)op, c->local_stream,
                                      /*blocking=*/i == iters - 1, algo);
      if (rc != XMPI_OK) return rc;
    }
    return XMPI_OK;
  }
  // Every rank of the job a thread of this process on this GPU (bench.py on a 1-GPU box): they share one in-order
  // stream and one launch folds everybody's chunks, so K steps are K launches enqueued back to back between two
  // rendezvous instead of K x (rendezvous, launch, wait, rendezvous).  Anything the zero-copy path cannot take
  // (unregistered buffers) falls through to the step-by-step loop -- on every rank alike, the decision is collective.
  bool all_coloc = c->shared_stream && c->zc_group_launch && c->size > 1 && iters > 1;
  for (int p = 0; p < c->size && all_coloc; p++)
    if (p != c->rank && !c->peer_coloc[p]) all_coloc = false;
  int first = 0;
  if (all_coloc && count > 0 && sendbuf && recvbuf && xmpi_dtype_size(dtype) && op >= 0 && op < XMPI_OP_COUNT &&
      (algo == XMPI_ALGO_ZCOPY || (algo == XMPI_ALGO_AUTO && c->zero_copy))) {
    drain_worker(c);
    std::lock_guard<std::mutex> g(c->coll_mu);
    bool done = false;
    const int rc = zero_copy_collective(c, COLL_ALLREDUCE, 0, sendbuf, recvbuf, count, (int)dtype, (int)op, false, &done, iters);
    if (rc != XMPI_OK || done) return rc;
    first = 0;  // went staged (collectively): run the steps one by one
  }
  for (int i = first; i < iters; i++) {
    const int rc = collective(c, COLL_ALLREDUCE, algo, 0, sendbuf, recvbuf, count, (int)dtype, (int)op);
    if (rc != XMPI_OK) return rc;
  }
  return XMPI_OK;
}

// ---- local kernels -----------------------------------------------------------------------------

static int timed_launch(xmpi_comm* c, int kind, size_t bytes, hipError_t (*launch)(void*, hipEvent_t, hipEvent_t),
                        void* ctx) {
  hipStream_t s = c->local_stream;
  if (!c->prof_on) {
    XMPI_HIP(launch(ctx, nullptr, nullptr));
    XMPI_HIP(hipStreamSynchronize(s));
    return XMPI_OK;
  }
  // the events ride on the dispatch itself: they carry the kernel's own begin / end timestamps
  hipEvent_t a = ev_get(c, true), b = ev_get(c, true);
  if (!a || !b) return XMPI_ERR_HIP;
  XMPI_HIP(launch(ctx, a, b));
  XMPI_HIP(hipStreamSynchronize(s));
  float ms = 0.f;
  XMPI_HIP(hipEventElapsedTime(&ms, a, b));
  c->prof[kind].add(ms, bytes);
  ev_put(c, a, true);
  ev_put(c, b, true);
  return XMPI_OK;
}

int xmpi_reduce_local(xmpi_comm* c, void* dst, const void* a, const void* b, size_t count, xmpi_dtype dtype, xmpi_op op) {
  XMPI_ENTER(c);
  const size_t es = xmpi_dtype_size(dtype);
  if (!es || op < 0 || op >= XMPI_OP_COUNT) return XMPI_ERR_ARG;
  struct Ctx { xmpi_comm* c; void* dst; const void *a, *b; size_t n; int dt, op; } ctx{c, dst, a, b, count, (int)dtype, (int)op};
  return timed_launch(c, PROF_REDUCE2, 3 * count * es,
                      [](void* p, hipEvent_t es, hipEvent_t ee) {
                        Ctx* x = (Ctx*)p;
                        return launch_reduce2(x->dst, x->a, x->b, x->n, x->dt, x->op, x->c->local_stream, es, ee);
                      },
                      &ctx);
}

int xmpi_reduce_local_n(xmpi_comm* c, void* dst, const void* const* srcs, int nsrc, size_t count, xmpi_dtype dtype,
                        xmpi_op op) {
  XMPI_ENTER(c);
  const size_t es = xmpi_dtype_size(dtype);
  if (!es || op < 0 || op >= XMPI_OP_COUNT || nsrc < 1 || nsrc > kMaxReduceSrcs) return XMPI_ERR_ARG;
  struct Ctx { xmpi_comm* c; void* dst; const void* const* s; int ns; size_t n; int dt, op; } ctx{c, dst, srcs, nsrc, count, (int)dtype, (int)op};
  return timed_launch(c, PROF_REDUCEN, (size_t)(nsrc + 1) * count * es,
                      [](void* p, hipEvent_t es, hipEvent_t ee) {
                        Ctx* x = (Ctx*)p;
                        return launch_reduce_n(x->dst, x->s, x->ns, x->n, x->dt, x->op, x->c->local_stream, es, ee);
                      },
                      &ctx);
}

int xmpi_copy_local(xmpi_comm* c, void* dst, const void* src, size_t bytes) {
  XMPI_ENTER(c);
  struct Ctx { xmpi_comm* c; void* dst; const void* src; size_t n; } ctx{c, dst, src, bytes};
  return timed_launch(c, PROF_COPY, 2 * bytes,
                      [](void* p, hipEvent_t es, hipEvent_t ee) {
                        Ctx* x = (Ctx*)p;
                        return launch_copy(x->dst, x->src, x->n, x->c->local_stream, es, ee);
                      },
                      &ctx);
}

int xmpi_reduce_local_multi(xmpi_comm* c, void* const* dsts, int ndst, const void* const* srcs, int nsrc, size_t count,
                            xmpi_dtype dtype, xmpi_op op) {
  XMPI_ENTER(c);
  const size_t es = xmpi_dtype_size(dtype);
  if (!es || op < 0 || op >= XMPI_OP_COUNT || nsrc < 1 || nsrc > kMaxReduceSrcs || ndst < 1 || ndst > kMaxReduceSrcs)
    return XMPI_ERR_ARG;
  struct Ctx { xmpi_comm* c; void* const* d; int nd; const void* const* s; int ns; size_t n; int dt, op; }
      ctx{c, dsts, ndst, srcs, nsrc, count, (int)dtype, (int)op};
  return timed_launch(c, PROF_ZCOPY, (size_t)(nsrc + ndst) * count * es,
                      [](void* p, hipEvent_t es, hipEvent_t ee) {
                        Ctx* x = (Ctx*)p;
                        return launch_reduce_n_multi(x->d, x->nd, x->s, x->ns, x->n, x->dt, x->op, x->c->local_stream,
                                                     es, ee);
                      },
                      &ctx);
}

int xmpi_copy_local_pairs(xmpi_comm* c, void* const* dsts, const void* const* srcs, int n, size_t bytes) {
  XMPI_ENTER(c);
  if (n < 1 || n > kMaxReduceSrcs || !dsts || !srcs) return XMPI_ERR_ARG;
  struct Ctx { xmpi_comm* c; void* const* d; const void* const* s; int n; size_t bytes; } ctx{c, dsts, srcs, n, bytes};
  return timed_launch(c, PROF_ZCOPY, (size_t)(2 * n) * bytes,
                      [](void* p, hipEvent_t es, hipEvent_t ee) {
                        Ctx* x = (Ctx*)p;
                        return launch_copy_pairs(x->d, x->s, x->n, x->bytes, x->c->local_stream, es, ee);
                      },
                      &ctx);
}

int xmpi_copy_local_multi(xmpi_comm* c, void* const* dsts, int ndst, const void* src, size_t bytes) {
  XMPI_ENTER(c);
  if (ndst < 1 || ndst > kMaxReduceSrcs) return XMPI_ERR_ARG;
  struct Ctx { xmpi_comm* c; void* const* d; int nd; const void* src; size_t n; } ctx{c, dsts, ndst, src, bytes};
  return timed_launch(c, PROF_ZCOPY, (size_t)(1 + ndst) * bytes,
                      [](void* p, hipEvent_t es, hipEvent_t ee) {
                        Ctx* x = (Ctx*)p;
                        return launch_copy_multi(x->d, x->nd, x->src, x->n, x->c->local_stream, es, ee);
                      },
                      &ctx);
}

int xmpi_heap_selftest(uint64_t seed, int rounds) { return heap_selftest(seed, rounds); }

int xmpi_zc_chunk(size_t count, size_t elem_size, int size, int j, size_t* elem_off, size_t* elem_cnt) {
  if (!elem_off || !elem_cnt || size < 1 || j < 0 || j >= size || elem_size < 1) return XMPI_ERR_ARG;
  zc_chunk(count, elem_size, size, j, elem_off, elem_cnt);
  return XMPI_OK;
}

int xmpi_count_mismatch(xmpi_comm* c, const void* a, const void* b, size_t bytes, uint64_t* out) {
  XMPI_ENTER(c);
  if (!out) return XMPI_ERR_ARG;
  std::lock_guard<std::mutex> g(c->coll_mu);
  hipStream_t s = c->local_stream;
  XMPI_HIP(hipMemsetAsync(c->dev_words, 0, 32, s));
  XMPI_HIP(launch_count_mismatch(a, b, bytes, c->dev_words, s));
  XMPI_HIP(hipMemcpyAsync(out, c->dev_words, 8, hipMemcpyDeviceToHost, s));
  XMPI_HIP(hipStreamSynchronize(s));
  return XMPI_OK;
}

int xmpi_checksum(xmpi_comm* c, const void* buf, size_t bytes, uint64_t* out) {
  XMPI_ENTER(c);
  if (!out) return XMPI_ERR_ARG;
  std::lock_guard<std::mutex> g(c->coll_mu);
  hipStream_t s = c->local_stream;
  XMPI_HIP(hipMemsetAsync(c->dev_words, 0, 32, s));
  XMPI_HIP(launch_checksum(buf, bytes, c->dev_words, s));
  XMPI_HIP(hipMemcpyAsync(out, c->dev_words, 8, hipMemcpyDeviceToHost, s));
  XMPI_HIP(hipStreamSynchronize(s));
  return XMPI_OK;
}

int xmpi_diff_stats(xmpi_comm* c, const void* a, const void* b, size_t count, xmpi_dtype dtype, double stats[3]) {
  XMPI_ENTER(c);
  if (!stats) return XMPI_ERR_ARG;
  if (dtype != XMPI_F16 && dtype != XMPI_BF16 && dtype != XMPI_F32 && dtype != XMPI_F64) return XMPI_ERR_ARG;
  std::lock_guard<std::mutex> g(c->coll_mu);
  hipStream_t s = c->local_stream;
  uint64_t w[4] = {0, 0, 0, 0};
  XMPI_HIP(hipMemsetAsync(c->dev_words, 0, 32, s));
  XMPI_HIP(launch_diff_stats(a, b, count, (int)dtype, c->dev_words, s));
  XMPI_HIP(hipMemcpyAsync(w, c->dev_words, 24, hipMemcpyDeviceToHost, s));
  XMPI_HIP(hipStreamSynchronize(s));
  memcpy(&stats[0], &w[0], 8);
  memcpy(&stats[1], &w[1], 8);
  stats[2] = (double)w[2];
  return XMPI_OK;
}

int xmpi_diff_rel(xmpi_comm* c, const void* a, const void* b, size_t count, xmpi_dtype dtype, double* max_rel) {
  XMPI_ENTER(c);
  if (!max_rel) return XMPI_ERR_ARG;
  if (dtype != XMPI_F16 && dtype != XMPI_BF16 && dtype != XMPI_F32 && dtype != XMPI_F64) return XMPI_ERR_ARG;
  std::lock_guard<std::mutex> g(c->coll_mu);
  hipStream_t s = c->local_stream;
  uint64_t w[4] = {0, 0, 0, 0};
  XMPI_HIP(hipMemsetAsync(c->dev_words, 0, 32, s));
  XMPI_HIP(launch_diff_stats(a, b, count, (int)dtype, c->dev_words, s));
  XMPI_HIP(hipMemcpyAsync(w, c->dev_words, 32, hipMemcpyDeviceToHost, s));
  XMPI_HIP(hipStreamSynchronize(s));
  memcpy(max_rel, &w[3], 8);
  if (w[2]) *max_rel = 1.0 / 0.0;  // a NaN on one side only
  return XMPI_OK;
}

int xmpi_fill_pattern(xmpi_comm* c, void* buf, size_t count, xmpi_dtype dtype, int pattern, uint64_t seed) {
  XMPI_ENTER(c);
  if (!xmpi_dtype_size(dtype) || pattern < 0 || pattern > 3) return XMPI_ERR_ARG;
  XMPI_HIP(launch_fill(buf, count, (int)dtype, pattern, seed, c->local_stream));
  XMPI_HIP(hipStreamSynchronize(c->local_stream));
  return XMPI_OK;
}

// ---- tuning / introspection ----------------------------------------------------------------------

int xmpi_set_param(xmpi_comm* c, const char* name, long value) {
  if (!c || c->finalized || !name) return XMPI_ERR_STATE;
  std::lock_guard<std::mutex> g(c->coll_mu);
  const std::string n = name;
  if (n == "channels") c->channels = std::max<long>(0, value);
  else if (n == "piece_bytes") c->piece_bytes = std::max<long>(0, value);
  else if (n == "copy_engine") c->copy_engine = value ? 1 : 0;
  else if (n == "timeout_s") c->timeout_s = value;
  else if (n == "prof_every") c->prof_every = std::max<long>(1, value);
  else if (n == "batch_copies") c->batch_copies = value ? 1 : 0;
  else if (n == "oneshot_bytes") c->oneshot_bytes = std::max<long>(0, value);
  else if (n == "zero_copy") c->zero_copy = value ? 1 : 0;
  else if (n == "zc_bcast_push_bytes") c->zc_bcast_push_bytes = std::max<long>(0, value);
  else if (n == "zc_group_launch") c->zc_group_launch = value ? 1 : 0;
  else if (n == "p2p_direct_bytes") c->p2p_direct_bytes = value;  // < 0: always through the mail slots
  else if (n == "p2p_kernel_ack") c->p2p_kernel_ack = value ? 1 : 0;
  else if (n == "p2p_agent_us") c->p2p_agent_us = std::max<long>(0, value);
  else if (n == "dsync") c->dsync = value ? 1 : 0;
  else if (n == "dsync_grid") c->dsync_grid_cap = std::max<long>(0, value);
  else if (n == "dsync_unroll") c->dsync_unroll = std::max<long>(1, std::min<long>(2, value));
  else if (n == "dsync_tiles") c->dsync_tiles = std::max<long>(1, value);
  else if (n == "p2p_grid_cap") c->p2p_grid_cap = std::max<long>(0, std::min<long>(value, 4096));
  else if (n == "ll_bytes") c->ll_bytes = std::max<long>(0, std::min<long>((long)kLLMaxPayload, value));  // untuned AUTO: LL lines up to here
  else if (n == "agent_ll") c->agent_ll = value < 0 ? 0 : std::min<long>(value, 2);  // blocking LL collectives by the lingering agent (no launch); (2 = 1)
  else if (n == "ll_agent_us") c->ll_agent_us = std::max<long>(0, value);
  else if (n == "agent_ll_bytes") c->agent_ll_bytes = std::max<long>(0, std::min<long>((long)kLLMaxPayload, value));
  else if (n == "dsync_split_bytes") c->dsync_split_bytes = std::max<long>(0, value);  // 0: always one kernel
  else if (n == "xcd_check") c->xcd_check = value ? 1 : 0;
  else if (n == "body_sys") c->body_sys = value ? 1 : 0;
  else if (n == "xcds") c->xcds = (int)std::max<long>(0, std::min<long>(value, 32));  // (tests: a GPU with more XCDs than it has)
  else if (n == "sched_channels") c->sched_channels = std::max<long>(0, value);
  else if (n == "sched_grid") c->sched_grid = std::max<long>(0, value);
  else if (n == "tree_piece_bytes") c->tree_piece_bytes = std::max<long>(4096, value);
  else if (n == "tuned") c->tuned = value != 0;  // 0: AUTO forgets the table of xmpi_tune
  else if (n == "tune_mask") c->tune_mask = value;  // bit k = 0: xmpi_tune leaves candidate k out (1 other unroll, 2 meet / body / done,
                                                    // 3 push-only, 4 ring kernel, 5 halving kernel, 6 LL lines, 7 ring kernel push form,
                                                    // 8 halving kernel push form, 9 tree kernel, 10 tree kernel push form); the default
                                                    // form always runs
  else if (n.rfind("tune_", 0) == 0) {  // tune_<algo|split|unroll>_<collective 0..3>_<size class>: a row of the table AUTO follows once
                                        // "tuned" is 1 -- xmpi_tune's to write; a benchmark or a test standing in for it may
    int coll = -1, cls = -1;
    char what[16] = {0};
    if (sscanf(name, "tune_%15[a-z]_%d_%d", what, &coll, &cls) != 3 || coll < 0 || coll >= 4 || cls < 0 || cls >= xmpi_comm::kTuneClasses ||
        value < -1 || value >= XMPI_ALGO_COUNT)
      return XMPI_ERR_ARG;
    const std::string w = what;
    if (w == "algo") c->tune_algo[coll][cls] = (int8_t)value;
    else if (w == "split") c->tune_split[coll][cls] = (int8_t)value;
    else if (w == "unroll") c->tune_unroll[coll][cls] = (int8_t)value;
    else return XMPI_ERR_ARG;
  }
  else if (n == "kernel_mode") set_kernel_mode((int)value);  // process-wide
  else if (n == "grid_cap") set_grid_cap((int)value);        // process-wide
  else return XMPI_ERR_ARG;
  return XMPI_OK;
}

long xmpi_get_param(const xmpi_comm* c, const char* name) {
  if (!c || c->finalized || !name) return -1;
  const std::string n = name;
  if (n == "channels") return c->channels;
  if (n == "piece_bytes") return c->piece_bytes;
  if (n == "copy_engine") return c->copy_engine;
  if (n == "timeout_s") return c->timeout_s;
  if (n == "watchdog_ms") return c->watchdog_ms;
  if (n == "dead_rank") return c->ctl->dead_rank();  // the first rank whose process the watchdog found gone; -1 = none
  if (n == "zero_copy") return c->zero_copy;
  if (n == "heap_arenas" || n == "heap_reserved" || n == "heap_in_use") {
    size_t a = 0, r = 0, u = 0;
    heap_stats(c->device, &a, &r, &u);
    return (long)(n == "heap_arenas" ? a : n == "heap_reserved" ? r : u);
  }
  if (n == "hbm_free_mib" || n == "hbm_total_mib") {  // of this rank's GPU, as the runtime reports it
    size_t fr = 0, tot = 0;
    if (hipSetDevice(c->device) != hipSuccess || hipMemGetInfo(&fr, &tot) != hipSuccess) return -1;
    return (long)((n == "hbm_free_mib" ? fr : tot) >> 20);
  }
  if (n == "p2p_direct_bytes") return c->p2p_direct_bytes;
  if (n == "p2p_kernel_ack") return c->p2p_kernel_ack;
  if (n == "p2p_agent_us") return c->p2p_agent_us;
  if (n == "p2p_agent_served") return (long)c->p2p_agent_served;
  if (n == "p2p_agent_launches") return (long)c->p2p_agent_launches;
  if (n == "dsync") return dsync_usable(c) ? 1 : 0;
  if (n == "dsync_epoch") return (long)c->dsync_epoch;
  if (n == "dsync_launches") return (long)c->dsync_launches;
  if (n == "dsync_bounced") return (long)c->dsync_bounced;
  if (n == "dsync_sharers") return c->dsync_sharers;
  if (n == "dsync_sharers_job") return c->dsync_sharers_job;
  if (n == "dsync_unroll") return c->dsync_unroll;
  if (n == "dsync_tiles") return c->dsync_tiles;
  if (n == "p2p_grid_cap") return c->p2p_grid_cap;
  if (n == "ll_bytes") return c->ll_bytes;
  if (n == "ll_max_bytes") return (long)kLLMaxPayload;
  if (n == "dsync_ll_launches") return (long)c->dsync_ll_launches;
  if (n == "agent_ll") return c->agent_ll;
  if (n == "agent_ll_bytes") return c->agent_ll_bytes;
  if (n == "ll_agent_us") return c->ll_agent_us;
  if (n == "dsync_ll_agent") return (long)c->dsync_ll_agent;
  if (n == "ll_agent_launches") return (long)c->ll_agent_launches;
  if (n == "agent_ll_wait_ns") return (long)c->agent_ll_wait_ns;  // command written -> answer seen, summed over dsync_ll_agent calls
  if (n == "dsync_grid") return c->dsync_grid_cap;
  if (n == "dsync_split_bytes") return c->dsync_split_bytes;
  if (n == "roctx") return roctx_enabled() ? 1 : 0;  // named ranges for the profilers are on (trace.h)
  if (n == "xcd_check") return c->xcd_check;
  if (n == "body_sys") return c->body_sys;
  if (n == "xcds") return c->xcds;
  if (n == "xcd_probe_mask") return (long)c->xcd_probe_mask;
  if (n == "xcd_short") return (long)c->xcd_short;
  if (n == "xcd_meet_mask") return c->dsync_status ? (long)__atomic_load_n(c->dsync_status + 6, __ATOMIC_RELAXED) : -1;
  if (n == "xcd_done_mask") return c->dsync_status ? (long)__atomic_load_n(c->dsync_status + 7, __ATOMIC_RELAXED) : -1;
  if (n == "dsync_split_launches") return (long)c->dsync_split_launches;
  if (n == "dsync_sched_launches") return (long)c->dsync_sched_launches;
  if (n == "dsync_land_bytes") return (long)c->dsync_land_bytes;
  // what xmpi_init's vote left the job with: 0 = everything; bit 0 (1): the split form's data kernel runs at system scope (the XCD
  // probe of THIS rank's GPU); bit 1 (2): the ranks meet on the host (some flag page could not be allocated, exported or mapped);
  // bit 2 (4): no windows (some window could not be mapped: no staged step tables, no mail slots); bit 3 (8): a schedule gave wrong
  // answers when the library checked it here and was taken out (tune_rejected).  xmpi_degraded() says why.
  if (n == "degraded") {
    bool shared_gpu = false;  // (ranks hosted by threads of one process on one GPU meet on the host by design, not by degradation)
    for (int p = 0; p < c->size; p++) shared_gpu = shared_gpu || c->peer_coloc[p];
    // (... and so do more ranks than the device side is sized for)
    return (c->body_sys == 1 && c->dsync_ok ? 1 : 0) | ((c->size > 1 && c->size <= kDsyncRanks && c->dsync && !c->dsync_ok && !shared_gpu) ? 2 : 0) | (c->windows_ok ? 0 : 4) |
           (c->rejected_why.empty() ? 0 : 8);
  }
  if (n == "windows_ok") return c->windows_ok ? 1 : 0;
  if (n == "sched_channels") return c->sched_channels;
  if (n == "sched_grid") return c->sched_grid;
  if (n == "tree_piece_bytes") return c->tree_piece_bytes;
  if (n == "tuned") return c->tuned ? 1 : 0;
  if (n == "tune_mask") return c->tune_mask;
  // schedules whose answers the library found wrong on this machine (xmpi_tune, xmpi_init's self-check): how many in all, and per
  // collective (0 allreduce, 1 allgather, 2 bcast, 3 reduce) a bit per candidate -- the numbering of tune_mask
  if (n == "tune_rejected") return __builtin_popcount(c->tune_rejected[0]) + __builtin_popcount(c->tune_rejected[1]) + __builtin_popcount(c->tune_rejected[2]) + __builtin_popcount(c->tune_rejected[3]);
  if (n.rfind("tune_rejected_", 0) == 0) {
    const int coll = atoi(name + 14);
    return (coll >= 0 && coll < 4 && name[14] >= '0' && name[14] <= '3' && !name[15]) ? (long)c->tune_rejected[coll] : -1;
  }
  if (n == "selfcheck") return c->selfcheck;
  if (n == "p2p_rejected") return (long)c->p2p_rejected;  // the self-check's verdict on Send / Receive: bit 0 the direct pull (-> mail slots), bit 1 everything
  if (n == "init_selfcheck_ms") return c->selfcheck_ms < 0 ? -1 : (long)(c->selfcheck_ms + 0.999);  // xmpi_init's self-check: -1 = did not run
  if (n == "init_selfcheck_us") return c->selfcheck_ms < 0 ? -1 : (long)(c->selfcheck_ms * 1e3);
  if (n == "init_selfcheck_setup_us") return c->selfcheck_ms < 0 ? -1 : (long)(c->selfcheck_setup_ms * 1e3);  // ... of which: its buffers (the job's first arena, the first kernel's code object)
  if (n == "tune_us") return (long)(c->tune_ms * 1e3);              // the last xmpi_tune, all of it
  if (n == "tune_check_us") return (long)(c->tune_check_ms * 1e3);  // ... the part spent checking answers (expected results, poison, compare)
  if (n.rfind("tune_", 0) == 0) {  // tune_<algo|split|unroll>_<collective 0..3>_<size class>: the table of xmpi_tune
    int coll = -1, cls = -1;
    char what[16] = {0};
    if (sscanf(name, "tune_%15[a-z]_%d_%d", what, &coll, &cls) == 3 && coll >= 0 && coll < 4 && cls >= 0 && cls < xmpi_comm::kTuneClasses) {
      const std::string w = what;
      if (w == "algo") return c->tune_algo[coll][cls];
      if (w == "split") return c->tune_split[coll][cls];
      if (w == "unroll") return c->tune_unroll[coll][cls];
    }
    return -1;
  }
  if (n.rfind("prof_min_ns_", 0) == 0 || n.rfind("prof_max_ns_", 0) == 0) {  // the shortest / longest sampled launch of kind 0..4 since xmpi_prof_reset
    const int kind = name[12] - '0';
    if (kind < 0 || kind >= PROF_KINDS || name[13]) return -1;
    return (long)((n[6] == 'i' ? c->prof[kind].min_ms : c->prof[kind].max_ms) * 1e6);
  }
  if (n == "p2p_direct_count") return (long)c->p2p_direct_count;
  if (n == "p2p_staged_count") return (long)c->p2p_staged_count;
  if (n == "p2p_lane_count") return (long)c->p2p_lane_count;
  if (n == "host_bounce_calls") return (long)c->host_bounce_calls;
  if (n == "host_lane_bytes") return (long)c->ctl->host_lane_bytes();
  if (n == "zc_seq") return (long)c->zc_seq;
  if (n == "zc_fallbacks_unregistered") return (long)c->zc_fallbacks_unregistered;
  if (n == "zc_fallbacks_unmappable") return (long)c->zc_fallbacks_unmappable;
  if (n == "shared_stream") return c->shared_stream ? 1 : 0;
  if (n == "kernel_mode") return get_kernel_mode();
  if (n == "last_run_us") return (long)c->last_run_us;
  if (n == "last_sync_us") return (long)c->last_sync_us;
  if (n == "lanes") return c->lanes;
  if (n == "fifo_depth") return c->fifo_depth;
  if (n == "slot_bytes") return (long)c->slot_bytes;
  if (n == "p2p_slot_bytes") return (long)c->p2p_slot_bytes;
  if (n == "window_bytes") return (long)c->window_bytes;
  if (n == "ring_channels_max") return ring_channel_count(c->size) * c->lanes;
  if (n == "ring_channels") return ring_channel_count(c->size);
  return -1;
}

int xmpi_prof_enable(xmpi_comm* c, int on) {
  if (!c || c->finalized) return XMPI_ERR_STATE;
  std::lock_guard<std::mutex> g(c->coll_mu);
  c->prof_on = on != 0;
  return XMPI_OK;
}

int xmpi_prof_reset(xmpi_comm* c) {
  if (!c || c->finalized) return XMPI_ERR_STATE;
  std::lock_guard<std::mutex> g(c->coll_mu);
  for (auto& p : c->prof) p = ProfCounter();
  return XMPI_OK;
}

int xmpi_prof_get(xmpi_comm* c, int kind, uint64_t* launches, double* total_ms, uint64_t* bytes) {
  if (!c || c->finalized) return XMPI_ERR_STATE;
  if (kind < 0 || kind >= PROF_KINDS) return XMPI_ERR_ARG;
  std::lock_guard<std::mutex> g(c->coll_mu);
  if (kind == PROF_ZCOPY && !c->dsync_prof_pending.empty()) {  // launches whose events nobody has read yet
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->local_stream);
    dsync_prof_harvest(c);
  }
  if (launches) *launches = c->prof[kind].launches;
  if (total_ms) *total_ms = c->prof[kind].total_ms;
  if (bytes) *bytes = c->prof[kind].bytes;
  return XMPI_OK;
}

int xmpi_link_probe(xmpi_comm* c, int peer, size_t bytes, int engine, int iters, int direction, double* gbps) {
  XMPI_ENTER(c);
  if (peer < 0 || peer >= c->size || iters < 1 || !gbps) return XMPI_ERR_ARG;
  std::lock_guard<std::mutex> g(c->coll_mu);
  if (!c->windows_ok) {
    set_last_error("link probe: it copies between the HBM windows, which this job could not map (xmpi_degraded)");
    return XMPI_ERR_UNSUPPORTED;
  }
  {
    const int src = ensure_streams(c);
    if (src != XMPI_OK) return src;
  }
  // the FIFO slots this rank owns in the peer's window (idle between collectives) are the remote
  // end; the slots the peer owns in this rank's window are the local end
  const size_t span = (size_t)c->lanes * c->fifo_depth * c->slot_bytes;
  bytes = std::min(bytes, span);
  char* remote = c->peer_window[peer] + c->coll_slot_off(c->rank, 0, 0);
  char* local = c->window + c->coll_slot_off(peer, 0, 0);
  char* dst = direction == 0 ? remote : local;  // 0 = write to the peer, 1 = read from the peer
  char* src = direction == 0 ? local : remote;
  hipStream_t s = c->send_stream[peer] ? c->send_stream[peer] : c->local_stream;
  hipEvent_t a = ev_get(c, true), b = ev_get(c, true);
  if (!a || !b) return XMPI_ERR_HIP;
  // engine 2: the stepped kernels' own accesses -- system-scope loads, written-through stores -- with as many workers as they run
  const int sys_grid = (int)std::max<long>(1, std::min<long>((long)((bytes + kSchedTileBytes - 1) / kSchedTileBytes), 1024 / std::max(1, c->dsync_sharers)));
  auto once = [&]() -> hipError_t {
    if (engine == 2) return launch_sys_copy(dst, src, bytes, sys_grid, s);
    if (engine == 1) return launch_copy(dst, src, bytes, s);
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s);
  };
  for (int w = 0; w < 2; w++) XMPI_HIP(once());
  XMPI_HIP(hipStreamSynchronize(s));
  XMPI_HIP(hipEventRecord(a, s));
  for (int i = 0; i < iters; i++) XMPI_HIP(once());
  XMPI_HIP(hipEventRecord(b, s));
  XMPI_HIP(hipStreamSynchronize(s));
  float ms = 0.f;
  XMPI_HIP(hipEventElapsedTime(&ms, a, b));
  ev_put(c, a, true);
  ev_put(c, b, true);
  *gbps = ms > 0 ? (double)bytes * iters / (ms * 1e-3) / 1e9 : 0.0;
  if (bytes >= ((size_t)1 << 20)) c->link_gbps[peer] = std::max(c->link_gbps[peer], *gbps);  // (sizes the pull kernel's grid)
  return XMPI_OK;
}

// Host-only exercise of the control plane (no HIP call): join, barriers, a token passed round the
// ring through the pipe counters and a mail-entry handshake with the next rank.  Lets the N > 1
// bootstrap / rendezvous logic be tested with plain OS processes on a machine without a GPU.
int xmpi_ctl_selftest(const char* job_key, int rank, int size, int rounds) {
  CtlConfig cfg{2, 8, 8u << 20, 2, 4u << 20, 1};  // (host lanes requested)
  std::string err;
  Ctl* ctl = nullptr;
  int rc = Ctl::join(job_key ? job_key : "selftest", rank, size, cfg, (double)env_long("XMPI_INIT_TIMEOUT_S", 30), &ctl, &err);
  if (rc != XMPI_OK) {
    set_last_error("ctl selftest: " + err);
    return rc;
  }
  auto wait_for = [&](auto pred) {
    const double t0 = now_seconds();
    Backoff bo;
    while (!pred()) {
      if (ctl->aborted()) return XMPI_ERR_PEER;
      if (now_seconds() - t0 > 30.0) return XMPI_ERR_TIMEOUT;
      bo.pause();
    }
    return XMPI_OK;
  };
  const int next = (rank + 1) % size, prev = (rank + size - 1) % size;
  for (int k = 1; k <= rounds && rc == XMPI_OK; k++) {
    rc = ctl->barrier(30.0);
    if (rc != XMPI_OK || size == 1) continue;
    // token round the ring through the head counters
    if (rank == 0) {
      ctl->pipe(0, next, 0)->head.v.store((uint64_t)k, std::memory_order_release);
      rc = wait_for([&] { return ctl->pipe(prev, 0, 0)->head.v.load(std::memory_order_acquire) == (uint64_t)k; });
    } else {
      rc = wait_for([&] { return ctl->pipe(prev, rank, 0)->head.v.load(std::memory_order_acquire) == (uint64_t)k; });
      ctl->pipe(rank, next, 0)->head.v.store((uint64_t)k, std::memory_order_release);
    }
    if (rc != XMPI_OK) break;
    // tagged rendezvous with the next rank (the states of a Send / Receive pair, without payload)
    MailEntry* out = ctl->mail(rank, next, k % kMailEntries);
    uint32_t expect = MAIL_FREE;
    if (!out->state.compare_exchange_strong(expect, MAIL_CLAIMED)) {
      rc = XMPI_ERR_STATE;
      break;
    }
    out->tag = k;
    out->bytes = (uint64_t)k * 10u + (uint64_t)rank;
    memset(&out->src, 0, sizeof out->src);  // the direct-pull offer of a registered payload travels with the header
    out->src.base = 0x1000u * (uint64_t)(rank + 1);
    out->src.gen = (uint64_t)k;
    out->src.offset = (uint64_t)rank;
    out->direct.store(DIRECT_OFFERED, std::memory_order_relaxed);
    out->state.store(MAIL_POSTED, std::memory_order_release);
    MailEntry* in = ctl->mail(prev, rank, k % kMailEntries);
    rc = wait_for([&] { return in->state.load(std::memory_order_acquire) == MAIL_POSTED && in->tag == k; });
    if (rc != XMPI_OK) break;
    if (in->bytes != (uint64_t)k * 10u + (uint64_t)prev || in->direct.load(std::memory_order_acquire) != DIRECT_OFFERED ||
        in->src.base != 0x1000u * (uint64_t)(prev + 1) || in->src.gen != (uint64_t)k || in->src.offset != (uint64_t)prev) {
      rc = XMPI_ERR_STATE;
      break;
    }
    in->direct.store(k % 2 ? DIRECT_ACCEPTED : DIRECT_DECLINED, std::memory_order_release);
    in->state.store(MAIL_DONE, std::memory_order_release);
    rc = wait_for([&] { return out->state.load(std::memory_order_acquire) == MAIL_DONE; });
    if (rc == XMPI_OK && out->direct.load(std::memory_order_acquire) != (k % 2 ? DIRECT_ACCEPTED : DIRECT_DECLINED))
      rc = XMPI_ERR_STATE;
    out->state.store(MAIL_FREE, std::memory_order_release);
    if (rc != XMPI_OK) break;
    // a host-resident payload through the entry's host lane (engine.cpp p2p_send / p2p_recv, DIRECT_HOST): a ring of
    // kHostLaneSlots pieces, head written by the sender, tail by the receiver; every rank sends to the next and receives from
    // the one before at once, lengths from one byte to three times the ring
    if (ctl->host_lane_bytes() > 0) {
      const size_t lane_bytes = ctl->host_lane_bytes(), piece = lane_bytes / kHostLaneSlots;
      auto length = [&](int r) { return (size_t)(((uint64_t)k * 7919u + (uint64_t)r * 104729u) % (3u * lane_bytes)) + 1; };
      auto byte_at = [&](int r, size_t i) { return (uint8_t)(i * 31u + (size_t)k + (size_t)r * 7u); };
      const int e = k % kMailEntries;
      const size_t out_bytes = length(rank), in_bytes = length(prev);
      const uint64_t out_np = (out_bytes + piece - 1) / piece, in_np = (in_bytes + piece - 1) / piece;
      char* lane_out = ctl->host_lane(rank, next, e);
      const char* lane_in = ctl->host_lane(prev, rank, e);
      PipeCtl* po = &ctl->mail(rank, next, e)->pipe;
      PipeCtl* pin = &ctl->mail(prev, rank, e)->pipe;
      std::vector<uint8_t> got(in_bytes);
      uint64_t filled = 0, taken = 0;
      const double t0 = now_seconds();
      Backoff bo;
      while ((filled < out_np || taken < in_np) && rc == XMPI_OK) {
        bool moved = false;
        if (filled < out_np && filled - po->tail.v.load(std::memory_order_acquire) < (uint64_t)kHostLaneSlots) {
          const size_t off = (size_t)filled * piece, n = std::min(piece, out_bytes - off);
          char* slot = lane_out + (size_t)(filled % kHostLaneSlots) * piece;
          for (size_t i = 0; i < n; i++) slot[i] = (char)byte_at(rank, off + i);
          po->head.v.store(++filled, std::memory_order_release);
          moved = true;
        }
        if (taken < in_np && pin->head.v.load(std::memory_order_acquire) > taken) {
          const size_t off = (size_t)taken * piece, n = std::min(piece, in_bytes - off);
          memcpy(got.data() + off, lane_in + (size_t)(taken % kHostLaneSlots) * piece, n);
          pin->tail.v.store(++taken, std::memory_order_release);
          moved = true;
        }
        if (!moved) {
          if (ctl->aborted()) rc = XMPI_ERR_PEER;
          else if (now_seconds() - t0 > 30.0) rc = XMPI_ERR_TIMEOUT;
          bo.pause();
        }
      }
      for (size_t i = 0; i < in_bytes && rc == XMPI_OK; i++)
        if (got[i] != byte_at(prev, i)) {
          set_last_error("ctl selftest: host lane payload differs at byte " + std::to_string(i) + " of " + std::to_string(in_bytes));
          rc = XMPI_ERR_STATE;
        }
      if (rc != XMPI_OK) break;
      rc = ctl->barrier(30.0);  // everybody has drained its lane: the counters start the next message at zero
      if (rc != XMPI_OK) break;
      po->head.v.store(0, std::memory_order_relaxed);
      pin->tail.v.store(0, std::memory_order_relaxed);
      rc = ctl->barrier(30.0);
      if (rc != XMPI_OK) break;
    }
    // zero-copy collective k: descriptors are double-buffered by sequence parity; everybody reads
    // everybody's after the barrier, a rank that freed buffers says so in its retire log
    BufDesc* mine = ctl->desc(rank, (uint64_t)k);
    mine->ok = 1;
    mine->fresh = (k + rank) % 3 == 0;
    mine->send.base = 0x100000u * (uint64_t)(rank + 1) + (uint64_t)k;
    mine->send.gen = (uint64_t)k * 100u + (uint64_t)rank;
    mine->recv = mine->send;
    mine->recv.offset = 64u * (uint64_t)k;
    for (size_t b = 0; b < sizeof mine->send.handle; b++) mine->send.handle[b] = (uint8_t)(b + (size_t)rank + (size_t)k);
    mine->seq.store((uint64_t)k, std::memory_order_release);
    RetireLog* log = ctl->retired(rank);
    for (int j = 0; j < rank + 1; j++) {  // rank r retires r+1 allocations per round
      const uint64_t n = log->count.load(std::memory_order_relaxed);
      log->gen[n % kRetireRing] = ((uint64_t)rank << 32) | n;
      log->count.store(n + 1, std::memory_order_release);
    }
    rc = ctl->barrier(30.0);
    if (rc != XMPI_OK) break;
    for (int p = 0; p < size && rc == XMPI_OK; p++) {
      const BufDesc* d = ctl->desc(p, (uint64_t)k);
      const RetireLog* lp = ctl->retired(p);
      const uint64_t n = lp->count.load(std::memory_order_acquire);
      bool good = d->seq.load(std::memory_order_acquire) == (uint64_t)k && d->ok == 1 && d->fresh == ((k + p) % 3 == 0) &&
                  d->send.base == 0x100000u * (uint64_t)(p + 1) + (uint64_t)k &&
                  d->send.gen == (uint64_t)k * 100u + (uint64_t)p && d->recv.offset == 64u * (uint64_t)k &&
                  d->send.handle[5] == (uint8_t)(5 + p + k) && n == (uint64_t)k * (uint64_t)(p + 1);
      for (uint64_t j = n > (uint64_t)kRetireRing ? n - kRetireRing : 0; j < n && good; j++)
        good = lp->gen[j % kRetireRing] == (((uint64_t)p << 32) | j);
      if (!good) rc = XMPI_ERR_STATE;
    }
    if (rc != XMPI_OK) break;
    // device-synchronised collectives: a rank publishes a registration (slot k % 4), every peer reads it and
    // acknowledges, and the owner goes on only when all have (dsync.cpp `publish` / `dsync_service` / `await_acks`)
    PubTable* pt = ctl->published(rank);
    const uint64_t n = pt->count.load(std::memory_order_relaxed);
    PubEntry& pe = pt->e[n % kPubRing];
    pe.gen = (uint64_t)k * 1000u + (uint64_t)rank;
    pe.base = 0x200000u * (uint64_t)(rank + 1);
    pe.bytes = (uint64_t)k << 20;
    pe.reserved = (uint64_t)(k % 4);
    for (size_t b = 0; b < sizeof pe.handle; b++) pe.handle[b] = (uint8_t)(b ^ (size_t)rank ^ (size_t)k);
    pt->count.store(n + 1, std::memory_order_release);
    bool mine_acked = false;
    std::vector<uint64_t> seen((size_t)size, (uint64_t)(k - 1));
    rc = wait_for([&] {
      for (int p = 0; p < size; p++) {  // serve the peers while waiting for them, like every wait loop of the library
        if (p == rank) continue;
        PubTable* pp = ctl->published(p);
        const uint64_t np = pp->count.load(std::memory_order_acquire);
        while (seen[(size_t)p] < np) {
          const PubEntry& e = pp->e[seen[(size_t)p] % kPubRing];
          const uint64_t kk = seen[(size_t)p] + 1;  // entry number == round it was published in
          if (e.gen != kk * 1000u + (uint64_t)p || e.base != 0x200000u * (uint64_t)(p + 1) || e.bytes != (kk << 20) ||
              e.reserved != kk % 4 || e.handle[7] != (uint8_t)(7 ^ (size_t)p ^ (size_t)kk))
            return true;  // corrupt entry: leave the wait, the check below fails
          seen[(size_t)p]++;
        }
        ctl->acked(rank, p)->store(seen[(size_t)p], std::memory_order_release);
      }
      mine_acked = true;
      bool served_all = true;  // (the library keeps serving from inside its barrier; here: stay until every peer's entry of this round is acknowledged)
      for (int p = 0; p < size; p++) {
        if (p == rank) continue;
        if (ctl->acked(p, rank)->load(std::memory_order_acquire) < n + 1) mine_acked = false;
        if (seen[(size_t)p] < n + 1) served_all = false;
      }
      return mine_acked && served_all;
    });
    if (rc == XMPI_OK && !mine_acked) rc = XMPI_ERR_STATE;
    if (rc != XMPI_OK) break;
    rc = ctl->barrier(30.0);  // nobody starts the next round's publication before everybody has checked this one
  }
  if (rc != XMPI_OK) ctl->set_abort(rc);
  else rc = ctl->barrier(30.0);
  delete ctl;
  return rc;
}

// ---- the library's own schedule table ---------------------------------------------------------------------------------

// Which of n candidates (mean times in microseconds; <= 0 = did not run) AUTO should take: the fastest -- but the default
// (index 0) stays unless another one beats it by more than `margin` (a fraction: noise must not flip the schedule).
int xmpi_tune_decide(const double* us, int n, double margin) {
  if (!us || n < 1) return -1;
  int best = -1;
  for (int i = 0; i < n; i++)
    if (us[i] > 0 && (best < 0 || us[i] < us[best])) best = i;
  if (best < 0) return -1;
  if (best != 0 && us[0] > 0 && us[best] >= us[0] * (1.0 - (margin > 0 ? margin : 0.0))) return 0;
  return best;
}

// Times the schedules this job's layout offers for allreduce-sum f32, allgather, bcast and reduce on the real buffers, size class by size
// class -- after CHECKING each one's answer at that size (above) --, lets every rank see the same (max over ranks) figures and fills the
// table AUTO consults (dsync.cpp tuned_choice).  A candidate that was wrong on ANY rank at ANY size leaves the collective's table on
// EVERY rank (xmpi_get_param "tune_rejected_<collective>", xmpi_degraded(), xmpi_last_error()); a wrong DEFAULT walks the ladder the
// mapping vote walks: split -> its system-scope data kernel -> the one-kernel fold -> (no right schedule left for some size) the ranks
// meet on the host.  Collective: every rank calls it with the same max_bytes.  With ranks that meet on the host there is nothing to choose.
int xmpi_tune(xmpi_comm* c, size_t max_bytes) {
  XMPI_ENTER(c);
  drain_worker(c);
  if (!dsync_usable(c) || c->size < 2) return XMPI_OK;
  const double t_begin = now_seconds();
  max_bytes = std::min<size_t>(std::max<size_t>(max_bytes, 1024), (size_t)1 << 30);
  AnswerCheck chk;
  int rc;
  {
    std::lock_guard<std::mutex> g(c->coll_mu);
    rc = chk.open(c, max_bytes);
  }
  if (rc != XMPI_OK) return rc;
  const std::vector<TuneCand> cands = tune_candidates(c);
  const long keep_split = c->dsync_split_bytes;
  const bool keep_tuned = c->tuned;
  c->tuned = false;
  c->tune_running = true;
  memset(c->tune_algo, -1, sizeof c->tune_algo);
  memset(c->tune_split, -1, sizeof c->tune_split);
  memset(c->tune_unroll, 0, sizeof c->tune_unroll);
  memset(c->tune_rejected, 0, sizeof c->tune_rejected);
  std::string why;
  bool none_right = false;
  struct Row {
    size_t per_rank;
    std::vector<double> worst;
  };
  std::vector<Row> all_rows[4];
  uint32_t ll_out = 0;  // LL lines are ONE mechanism under all four collectives: wrong for one, trusted for none (apply_rejections)
  for (int coll : {(int)COLL_ALLREDUCE, (int)COLL_REDUCE, (int)COLL_ALLGATHER, (int)COLL_BCAST}) {  // (the two that share the expected sum side by side)
    std::vector<Row>& rows = all_rows[coll];
    uint32_t rejected = ll_out;
    for (size_t bytes = 1024; bytes <= max_bytes && rc == XMPI_OK; bytes *= 4) {
      const size_t per_rank = coll == COLL_ALLGATHER ? bytes / (size_t)c->size / 16 * 16 : bytes;
      if (per_rank < 16) continue;
      // (XMPI_TUNE_ITERS: a cap on the timed runs per candidate -- the rehearsals on virtual devices, where a "kernel" is a host thread and
      // a time means nothing, keep every candidate's CHECKED run and pay for two timed ones)
      static const long iters_cap = env_long("XMPI_TUNE_ITERS", 0);
      const int by_size = bytes <= ((size_t)1 << 20) ? 20 : (bytes <= ((size_t)32 << 20) ? 6 : 3);
      const int iters = iters_cap > 0 ? (int)std::min<long>(by_size, iters_cap) : by_size;
      std::vector<double> us(cands.size(), 0.0), worst(cands.size(), 0.0);
      std::vector<uint64_t> bad(cands.size(), 0), worst_bad(cands.size(), 0);
      std::vector<int> ks;
      for (size_t k = 0; k < cands.size(); k++) {
        if (!tune_offered(c, coll, cands[k])) continue;
        if (cands[k].algo == XMPI_ALGO_LL && per_rank > kLLMaxPayload) continue;
        if (k > 0 && !((c->tune_mask >> k) & 1)) continue;  // a schedule the caller has ruled out on this machine (never the default)
        if ((rejected >> k) & 1u) continue;                 // ... or a smaller size has (the vote: the same on every rank)
        ks.push_back((int)k);
      }
      {
        std::lock_guard<std::mutex> g(c->coll_mu);
        rc = chk.expect_for(coll, per_rank);
      }
      // few iterations fit a large message into a tuning budget, and a few iterations are noisy (eight processes on one GPU:
      // +-6 % between two runs of one schedule): large sizes are measured twice, the candidates interleaved, and the better
      // figure of each counts
      const int rounds = bytes > ((size_t)1 << 20) ? 2 : 1;
      // (the check's second pass -- inputs changed in place -- at every other size up to 1 MiB: 4 KiB, 64 KiB, 1 MiB; halves what it costs)
      chk.twice = bytes == ((size_t)4 << 10) || bytes == ((size_t)64 << 10) || bytes == ((size_t)1 << 20);
      for (int round = 0; round < rounds && rc == XMPI_OK; round++)
        rc = tune_measure(c, chk, cands, coll, per_rank, ks, iters, /*check=*/round == 0, /*keep_min=*/round > 0, us.data(), bad.data());
      if (rc != XMPI_OK) break;
      // every rank must read the same figures: the slowest rank's times, the worst rank's answers
      rc = vote_max(c, us.data(), bad.data(), (int)cands.size(), worst.data(), worst_bad.data());
      if (rc != XMPI_OK) break;
      // the ladder's first rung: meet / body / done gave wrong answers with the data kernel that relies on the meet and done
      // kernels' acquire / release once per XCD -- its system-scope form relies on nothing (what the XCD probe would have chosen)
      if (worst_bad[xmpi_comm::CAND_SPLIT] && !c->body_sys) {
        c->body_sys = 1;
        us[xmpi_comm::CAND_SPLIT] = 0;
        bad[xmpi_comm::CAND_SPLIT] = 0;
        rc = tune_measure(c, chk, cands, coll, per_rank, {xmpi_comm::CAND_SPLIT}, iters, true, false, us.data(), bad.data());
        std::vector<double> w2(cands.size(), 0.0);
        std::vector<uint64_t> b2(cands.size(), 0);
        if (rc == XMPI_OK) rc = vote_max(c, us.data(), bad.data(), (int)cands.size(), w2.data(), b2.data());
        if (rc != XMPI_OK) break;
        char t[200];
        snprintf(t, sizeof t, "%s: split gave wrong answers at %zu B per rank (%llu bytes differ on the worst rank); its system-scope data kernel %s",
                 coll_name(coll), per_rank, (unsigned long long)worst_bad[xmpi_comm::CAND_SPLIT], b2[xmpi_comm::CAND_SPLIT] ? "does too" : "is right and takes over (body_sys)");
        why += std::string(why.empty() ? "" : "; ") + t;
        worst[xmpi_comm::CAND_SPLIT] = w2[xmpi_comm::CAND_SPLIT];
        worst_bad[xmpi_comm::CAND_SPLIT] = b2[xmpi_comm::CAND_SPLIT];
        if (b2[xmpi_comm::CAND_SPLIT]) c->body_sys = 0;
      }
      for (size_t k = 0; k < cands.size(); k++)
        if (worst_bad[k] && !((rejected >> k) & 1u)) {
          rejected |= 1u << k;
          char t[200];
          snprintf(t, sizeof t, "%s: %s gives wrong answers on this machine (first at %zu B per rank: %llu bytes differ on the worst rank)", coll_name(coll),
                   kCandName[k], per_rank, (unsigned long long)worst_bad[k]);
          why += std::string(why.empty() ? "" : "; ") + t;
        }
      rows.push_back({per_rank, worst});
      if (trace_on()) fprintf(stderr, "[xmpi %d %.6f] tune: %s %zu B per rank done\n", c->rank, now_seconds(), coll_name(coll), per_rank);
    }
    if (rc != XMPI_OK) break;
    c->tune_rejected[coll] = rejected;
    ll_out |= rejected & (1u << xmpi_comm::CAND_LL);
  }
  // the tables, once the rejected sets are known: what was wrong at one size is not trusted at another
  for (int coll = 0; coll < 4 && rc == XMPI_OK; coll++) {
    std::vector<Row>& rows = all_rows[coll];
    const uint32_t rejected = c->tune_rejected[coll] | ll_out;
    for (size_t ri = 0; ri < rows.size(); ri++) {
      const size_t per_rank = rows[ri].per_rank;
      std::vector<double>& worst = rows[ri].worst;
      bool ran = false;
      for (size_t k = 0; k < cands.size(); k++) {
        ran = ran || worst[k] > 0;
        if ((rejected >> k) & 1u) worst[k] = 0;
      }
      // "the default stays on a tie" -- and the untuned library already runs meet / body / done from dsync_split_bytes on: there
      // candidate 2 is the default, so it is decided with the two swapped
      // (what dsync_split_bytes is compared with: the bytes one rank's kernel moves -- dsync.cpp launch)
      const size_t moved = coll == COLL_ALLREDUCE ? 2 * per_rank : coll == COLL_REDUCE ? per_rank / (size_t)c->size * (size_t)(c->size + 1)
                           : coll == COLL_ALLGATHER ? per_rank * (size_t)(c->size + 1) : 0;
      const bool split_is_default = keep_split > 0 && moved >= (size_t)keep_split && worst[2] > 0;
      if (split_is_default) std::swap(worst[0], worst[2]);
      int best = xmpi_tune_decide(worst.data(), (int)worst.size(), 0.03);
      if (split_is_default && (best == 0 || best == 2)) best = 2 - best;
      if (best < 0) {
        if (ran) none_right = true;  // every schedule this collective has at this size is wrong here
        continue;
      }
      int k = 0;
      while (k + 1 < xmpi_comm::kTuneClasses && (per_rank >> (k + 9)) != 0) k++;
      for (int kk = k; kk < xmpi_comm::kTuneClasses && kk < k + 2; kk++) {  // this class and the one to the next measured size
        c->tune_algo[coll][kk] = (int8_t)cands[(size_t)best].algo;
        c->tune_split[coll][kk] = (int8_t)cands[(size_t)best].split;
        c->tune_unroll[coll][kk] = (int8_t)cands[(size_t)best].unroll;
      }
      if (ri == 0)  // below the smallest measured size: what won there
        for (int kk = 0; kk < k; kk++) {
          c->tune_algo[coll][kk] = c->tune_algo[coll][k];
          c->tune_split[coll][kk] = c->tune_split[coll][k];
          c->tune_unroll[coll][kk] = c->tune_unroll[coll][k];
        }
      for (int kk = k + 2; kk < xmpi_comm::kTuneClasses; kk++) {  // beyond the largest measured size: what won there
        c->tune_algo[coll][kk] = c->tune_algo[coll][k];
        c->tune_split[coll][kk] = c->tune_split[coll][k];
        c->tune_unroll[coll][kk] = c->tune_unroll[coll][k];
      }
    }
  }
  c->tune_running = false;
  {
    std::lock_guard<std::mutex> g(c->coll_mu);
    chk.close();
  }
  c->tune_check_ms = chk.spent_s * 1e3;
  if (rc != XMPI_OK) {
    c->tuned = keep_tuned;
    c->ctl->set_abort(rc);  // (the other ranks are in, or on their way to, a barrier of this very call: they must not wait for this one)
    return rc;
  }
  c->tuned = true;
  note_rejections(c, "xmpi_tune", why, none_right);
  rc = xmpi_barrier(c);
  c->tune_ms = (now_seconds() - t_begin) * 1e3;
  return rc;
}

// The step program a stepped kernel (sched.hip) runs on `rank` for ring channel `channel`, as text -- produced by the very
// function the kernel calls (sched_steps.h).  One line per step:
//   g wait=<rank>:<value> sig=<rank>,<rank>:<value> nmv=<moves> then per move
//   | ns=<1|2|3> D=<ref> D2=<ref> A=<ref> B=<ref> C=<ref> lo=<byte> hi=<byte>        ref = <rank>.<s|r|l><+offset> or -
// (s = send buffer, r = receive buffer, l = landing block).  form: 0 = pull, 1 = push; in_place: every rank's send buffer is its
// receive buffer (what decides whether a push-form ring lands in the receive buffers or in landing blocks).
// (host logic only; tests/sched_sim.py executes all ranks' programs on the CPU).  Returns the needed length.
int xmpi_sched_dump(int sched, int form, int in_place, int size, int rank, int root, int pieces, size_t count, size_t elem_size, int nchan,
                    int channel, char* out, size_t cap) {
  if (size < 1 || size > kDsyncRanks || rank < 0 || rank >= size || root < 0 || root >= size || elem_size < 1 || nchan < 1 ||
      nchan > kMaxSchedChannels || channel < 0 || channel >= nchan || sched < SCHED_RING_ALLREDUCE || sched > SCHED_TREE_REDUCE ||
      form < 0 || form > 1 || (sched == SCHED_TREE_REDUCE && pieces > 127))  // (a step number must fit the low byte of a flag word)
    return XMPI_ERR_ARG;
  DsyncSchedArgs a;
  memset(&a, 0, sizeof a);
  a.d.me = rank;
  a.d.n = size;
  a.sched = sched;
  a.push = (uint32_t)form;
  a.nchan = nchan;
  a.root = root;
  a.pieces = std::max(1, pieces);
  a.count = count;
  a.elem_size = (uint32_t)elem_size;
  for (int ch = 0; ch < nchan; ch++) {
    std::vector<int> ord;
    ring_order(size, ch, &ord);
    for (int i = 0; i < size; i++) a.order[ch][i] = (uint8_t)ord[(size_t)i];
  }
  // recognisable addresses: rank r's send / receive buffer / landing block = ((r+1) << 44) | (kind << 42) | 2^41 (+ a signed offset)
  uint64_t send[kDsyncRanks], recv[kDsyncRanks], land[kDsyncRanks];
  auto fake = [](int r, int kind) { return ((uint64_t)(r + 1) << 44) | ((uint64_t)kind << 42) | (1ull << 41); };
  for (int r = 0; r < size; r++) {
    // in place: the send buffer IS the receive buffer (tree reduce: at the root only -- nobody else has one; allgather: the
    // rank's block of it)
    recv[r] = fake(r, 1);
    send[r] = !in_place || sched == SCHED_TREE_BCAST || (sched == SCHED_TREE_REDUCE && r != root) ? fake(r, 0)
              : sched == SCHED_RING_ALLGATHER           ? recv[r] + (uint64_t)r * count * elem_size
                                                        : recv[r];
    DsyncSchedArgs ar = a;
    ar.d.me = r;
    land[r] = sched_land_bytes(ar, in_place != 0) ? fake(r, 2) : 0;  // (as dsync.cpp lends them)
  }
  auto show = [&](uint64_t base, char* buf, size_t n) {
    if (!base) {
      snprintf(buf, n, "-");
      return;
    }
    const int r = (int)(base >> 44) - 1, kind = (int)((base >> 42) & 3);
    const long long off = (long long)(base - fake(r, kind));
    snprintf(buf, n, "%d.%c%+lld", r, "srl?"[kind], off);
  };
  std::string t;
  const int ns = sched_nsteps(a);
  for (int g = 1; g <= ns; g++) {
    SchedStep st;
    sched_step(a, send, recv, land, g, channel, &st);
    char line[160];
    snprintf(line, sizeof line, "%d wait=%d:%u sig=%d,%d:%u nmv=%d", g, st.wait_rank, st.wait_val, st.sig[0], st.sig[1], st.sig_val, st.nmv);
    t += line;
    for (int k = 0; k < st.nmv; k++) {
      const SchedMove& m = st.mv[k];
      char d[48], d2[48], x[48], y[48], z[48], mv[400];
      show(m.D, d, sizeof d);
      show(m.D2, d2, sizeof d2);
      show(m.A, x, sizeof x);
      show(m.ns >= 2 ? m.B : 0, y, sizeof y);
      show(m.ns >= 3 ? m.C : 0, z, sizeof z);
      snprintf(mv, sizeof mv, " | ns=%d D=%s D2=%s A=%s B=%s C=%s lo=%llu hi=%llu", m.ns, d, d2, x, y, z, (unsigned long long)m.lo,
               (unsigned long long)m.hi);
      t += mv;
    }
    t += "\n";
  }
  if (out && cap) {
    const size_t n = std::min(cap - 1, t.size());
    memcpy(out, t.data(), n);
    out[n] = 0;
  }
  return (int)std::min<size_t>(t.size() + 1, 0x7fffffff);
}

size_t xmpi_sched_land_bytes(int sched, int in_place, int size, int rank, int root, size_t count, size_t elem_size) {
  if (size < 1 || size > kDsyncRanks || rank < 0 || rank >= size || root < 0 || root >= size) return 0;
  DsyncSchedArgs a;
  memset(&a, 0, sizeof a);
  a.d.me = rank;
  a.d.n = size;
  a.sched = sched;
  a.push = 1;
  a.root = root;
  a.count = count;
  a.elem_size = (uint32_t)elem_size;
  return (size_t)sched_land_bytes(a, in_place != 0);
}

int xmpi_plan_dump(int coll, int algo, int size, int rank, int root, size_t count, size_t elem_size, int channels,
                   size_t piece_elems, int fifo_depth, size_t oneshot_bytes, char* out, size_t cap) {
  PlanParams pp;
  pp.coll = coll;
  pp.algo = (algo == XMPI_ALGO_ZCOPY || algo == XMPI_ALGO_ZPUSH) ? (int)XMPI_ALGO_AUTO : algo;  // the staged fallback
  pp.size = size;
  pp.rank = rank;
  pp.root = root;
  pp.count = count;
  pp.elem_size = elem_size;
  pp.channels = channels;
  pp.lanes = 2;
  pp.piece_bytes = piece_elems * elem_size;
  pp.fuse = 1;
  pp.fifo_depth = fifo_depth > 0 ? fifo_depth : 8;  // (0: the library's defaults)
  pp.oneshot_bytes = oneshot_bytes != (size_t)-1 ? oneshot_bytes : (size_t)1 << 20;
  Plan plan;
  int rc = build_plan(pp, &plan);
  if (rc != XMPI_OK) return rc;
  const std::string t = plan_to_text(plan);
  if (out && cap) {
    const size_t n = std::min(cap - 1, t.size());
    memcpy(out, t.data(), n);
    out[n] = 0;
  }
  return (int)std::min<size_t>(t.size() + 1, 0x7fffffff);
}

}  // extern "C"
