// zcopy.cpp -- zero-copy collectives: kernels that read the peers' user buffers and write the
// peers' user buffers directly (xGMI loads / stores), no staging through the receive windows.
//
// The reference moves every payload through three host copies and a TCP socket
// (network.go:539,563; mpi.go:77-81); the staged schedules of plan.cpp already replace that by one
// push into the receiver's HBM window plus one pass out of it.  For buffers the peers can map
// (xmpi_malloc / xmpi_register), even that staging goes away:
//
//   allreduce   rank j folds chunk j of ALL ranks' send buffers in rank order 0..N-1 (one local
//               read + N-1 reads over N-1 different links) and stores the result into chunk j of
//               ALL ranks' receive buffers (one local write + N-1 writes over the same links):
//               ONE kernel per rank, N reads + N writes per element, 2(N-1)/N * S on the wire per
//               rank -- the same wire bytes as a ring, spread over every link of the mesh at once,
//               and the result is bit-identical to the rank-order oracle for every dtype.
//   reduce      the same fold, stored only into the root's receive buffer.
//   allgather   rank j stores its block into block j of every receive buffer (one read, N writes).
//   bcast       small: the root stores into every buffer; large: the root scatters chunk j to
//               rank j, then every rank forwards its chunk to the others (two kernels, each link
//               carries S/N twice instead of one link carrying S).
//
// Synchronisation is host-side through the control block: publish buffer descriptors -> barrier
// (everyone's input is ready, everyone's output may be overwritten) -> kernel -> barrier (all
// remote reads of my input and writes of my output have completed).
#include <dirent.h>
#include <signal.h>
#include <unistd.h>

#include <cerrno>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "comm.h"
#include "kernels.h"

namespace xmpi {
namespace {

// ---- registry of exportable allocations of this process ---------------------------------------
struct Alloc {
  uint64_t base;
  size_t bytes;
  int device;
  uint64_t gen;
  bool have_handle;
  hipIpcMemHandle_t handle;
};
std::mutex g_reg_mu;
std::map<uint64_t, Alloc> g_reg;  // by base address
uint64_t g_next_gen = 1;

// the registered allocation that holds [p, p+need) on `device`; fills the IPC handle on first use
bool registry_lookup(const void* p, size_t need, int device, Alloc* out) {
  std::lock_guard<std::mutex> g(g_reg_mu);
  const uint64_t a = (uint64_t)(uintptr_t)p;
  auto it = g_reg.upper_bound(a);
  if (it == g_reg.begin()) return false;
  --it;
  Alloc& al = it->second;
  if (a < al.base || a + need > al.base + al.bytes || al.device != device) return false;
  if (!al.have_handle) {
    if (hipIpcGetMemHandle(&al.handle, (void*)(uintptr_t)al.base) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    al.have_handle = true;
  }
  *out = al;
  return true;
}

// ---- mappings of the peers' allocations (one per process: ranks hosted by threads share them) ---
struct PeerMap {
  int pid;
  uint64_t base, gen;
  void* ptr;
};
std::mutex g_map_mu;
std::vector<PeerMap> g_maps;

hipError_t map_peer(int pid, const BufRef& r, void** out) {
  std::lock_guard<std::mutex> g(g_map_mu);
  for (size_t i = 0; i < g_maps.size(); i++) {
    PeerMap& m = g_maps[i];
    if (m.pid != pid || m.base != r.base) continue;
    if (m.gen == r.gen) {
      *out = m.ptr;
      return hipSuccess;
    }
    (void)hipIpcCloseMemHandle(m.ptr);  // the owner freed that allocation and re-used the address
    g_maps.erase(g_maps.begin() + (long)i);
    break;
  }
  hipIpcMemHandle_t h;
  static_assert(sizeof h <= sizeof r.handle, "ipc handle size");
  memcpy(&h, r.handle, sizeof h);
  void* ptr = nullptr;
  hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return e;
  }
  g_maps.push_back({pid, r.base, r.gen, ptr});
  *out = ptr;
  return hipSuccess;
}

bool fill_ref(xmpi_comm* c, const void* p, size_t need, BufRef* ref, bool* fresh) {
  memset(ref, 0, sizeof *ref);
  if (!p) return false;
  Alloc al;
  if (!registry_lookup(p, need, c->device, &al)) return false;
  ref->base = al.base;
  ref->gen = al.gen;
  ref->offset = (uint64_t)(uintptr_t)p - al.base;
  ref->bytes = al.bytes;
  memcpy(ref->handle, &al.handle, sizeof al.handle);
  if (c->zc_announced.insert({al.base, al.gen}).second) *fresh = true;
  return true;
}

double tmo(const xmpi_comm* c) { return c->timeout_s > 0 ? (double)c->timeout_s : 1e18; }

int count_open_fds() {
  int n = 0;
  if (DIR* d = opendir("/proc/self/fd")) {
    while (readdir(d)) n++;
    closedir(d);
  }
  return n;
}

// launch + wait; a sampled launch carries its own begin / end events (kind PROF_ZCOPY)
struct Launcher {
  xmpi_comm* c;
  hipEvent_t start = nullptr, stop = nullptr;
  size_t bytes = 0;
  explicit Launcher(xmpi_comm* comm) : c(comm) {}
  int begin(size_t traffic) {
    bytes = traffic;
    if (c->prof_on && (c->prof_seq[PROF_ZCOPY]++ % (uint64_t)std::max<long>(1, c->prof_every)) == 0) {
      start = ev_get(c, true);
      stop = ev_get(c, true);
      if (!start || !stop) return XMPI_ERR_HIP;
    }
    return XMPI_OK;
  }
  int finish() {
    hipEvent_t fin = ev_get(c, false);
    if (!fin) return XMPI_ERR_HIP;
    XMPI_HIP(hipEventRecord(fin, c->local_stream));
    // poll instead of hipEventSynchronize: its wake-up latency is a visible share of a small collective
    Backoff bo;
    for (;;) {
      const hipError_t e = hipEventQuery(fin);
      if (e == hipSuccess) break;
      if (e != hipErrorNotReady) return hip_fail(e, "hipEventQuery", __FILE__, __LINE__);
      (void)hipGetLastError();
      bo.pause();
    }
    ev_put(c, fin, false);
    if (start) {
      float ms = 0.f;
      XMPI_HIP(hipEventElapsedTime(&ms, start, stop));
      ProfCounter& pc = c->prof[PROF_ZCOPY];
      pc.add(ms, bytes);
      ev_put(c, start, true);
      ev_put(c, stop, true);
      start = stop = nullptr;
    }
    return XMPI_OK;
  }
};

}  // namespace

int registry_add(void* base, size_t bytes, int device) {
  std::lock_guard<std::mutex> g(g_reg_mu);
  Alloc al;
  memset(&al, 0, sizeof al);
  al.base = (uint64_t)(uintptr_t)base;
  al.bytes = bytes;
  al.device = device;
  al.gen = g_next_gen++;
  al.have_handle = false;
  g_reg[al.base] = al;
  return XMPI_OK;
}

// Forget an allocation that is about to be freed.  If peers may have mapped it (a handle was
// exported), its generation goes into this rank's retire log so they unmap it before they map anything
// newer of this rank.
void registry_remove(xmpi_comm* c, void* base) {
  std::lock_guard<std::mutex> g(g_reg_mu);
  auto it = g_reg.find((uint64_t)(uintptr_t)base);
  if (it == g_reg.end()) return;
  if (it->second.have_handle && c && c->ctl) {
    RetireLog* log = c->ctl->retired(c->rank);
    const uint64_t n = log->count.load(std::memory_order_relaxed);
    log->gen[n % kRetireRing] = it->second.gen;
    log->count.store(n + 1, std::memory_order_release);
  }
  g_reg.erase(it);
}

bool registry_alive(uint64_t gen) {
  std::lock_guard<std::mutex> g(g_reg_mu);
  for (auto& kv : g_reg)
    if (kv.second.gen == gen) return true;
  return false;
}

namespace {
// unmap what the peers have freed since this rank last looked (before anything new is mapped)
void drop_retired(xmpi_comm* c) {
  const int mypid = (int)getpid();
  std::lock_guard<std::mutex> gz(c->zc_mu);  // collectives and point-to-point calls both come here
  for (int p = 0; p < c->size; p++) {
    const int pid = c->ctl->info(p)->pid;
    if (p == c->rank || pid == mypid) continue;
    RetireLog* log = c->ctl->retired(p);
    const uint64_t n = log->count.load(std::memory_order_acquire);
    uint64_t k = c->zc_retired_seen[p];
    if (k == n) continue;
    const bool all = n - k > (uint64_t)kRetireRing;  // the ring lapped this reader: drop everything of that rank
    std::lock_guard<std::mutex> g(g_map_mu);
    for (size_t i = 0; i < g_maps.size();) {
      bool dead = false;
      if (g_maps[i].pid == pid) {
        dead = all;
        for (uint64_t j = k; j < n && !dead; j++) dead = log->gen[j % kRetireRing] == g_maps[i].gen;
      }
      if (dead) {
        (void)hipIpcCloseMemHandle(g_maps[i].ptr);
        g_maps.erase(g_maps.begin() + (long)i);
      } else {
        i++;
      }
    }
    c->zc_retired_seen[p] = n;
  }
  (void)hipGetLastError();
}
}  // namespace

// point-to-point rendezvous (engine.cpp): where `p` lives, for the peer to copy straight out of it
bool zc_export(xmpi_comm* c, const void* p, size_t need, BufRef* ref) {
  memset(ref, 0, sizeof *ref);
  Alloc al;
  if (!p || !registry_lookup(p, need, c->device, &al)) return false;
  ref->base = al.base;
  ref->gen = al.gen;
  ref->offset = (uint64_t)(uintptr_t)p - al.base;
  ref->bytes = al.bytes;
  memcpy(ref->handle, &al.handle, sizeof al.handle);
  return true;
}

// ... and the peer's side: a pointer to the same bytes in this process (mapping cached per allocation)
bool zc_import(xmpi_comm* c, int peer, const BufRef& ref, void** out) {
  const int pid = c->ctl->info(peer)->pid;
  if (pid == (int)getpid()) {  // a thread of this process (or this rank itself)
    *out = (void*)(uintptr_t)(ref.base + ref.offset);
    return true;
  }
  drop_retired(c);
  void* base = nullptr;
  if (map_peer(pid, ref, &base) != hipSuccess) return false;
  *out = (char*)base + ref.offset;
  return true;
}

// A communicator is being finalised.  The mappings of its peers' allocations are NOT closed: other communicators
// of this process may address the same peers through them (arenas outlive communicators), and closing and
// re-opening mappings while other processes do the same is exactly what fails sporadically on this stack
// ("invalid device pointer").  They go when their allocation is retired (drop_retired) or their owner is gone.
void zc_close_peers(const xmpi_comm* c) {
  (void)c;
  std::lock_guard<std::mutex> g(g_map_mu);
  for (size_t i = 0; i < g_maps.size();) {
    if (kill(g_maps[i].pid, 0) != 0 && errno == ESRCH) {
      (void)hipIpcCloseMemHandle(g_maps[i].ptr);
      g_maps.erase(g_maps.begin() + (long)i);
    } else {
      i++;
    }
  }
  (void)hipGetLastError();
}

static int zc_run(xmpi_comm* c, int coll, int root, const void* sendbuf, void* recvbuf, size_t count, int dtype,
                  int op, bool push, bool* done, int iters) {
  *done = false;
  const int N = c->size, me = c->rank;
  const size_t es = xmpi_dtype_size((xmpi_dtype)dtype);
  const size_t send_bytes = count * es;
  const size_t recv_bytes = (coll == COLL_ALLGATHER) ? send_bytes * (size_t)N : send_bytes;
  const bool recv_significant = (coll != COLL_REDUCE) || me == root;

  // 1. publish what the peers need to reach my buffers
  const uint64_t seq = ++c->zc_seq;
  BufDesc* mine = c->ctl->desc(me, seq);
  bool fresh = false;
  bool ok = fill_ref(c, sendbuf, send_bytes, &mine->send, &fresh);
  if (recv_significant) ok = fill_ref(c, recvbuf, recv_bytes, &mine->recv, &fresh) && ok;
  else mine->recv = mine->send;
  mine->ok = ok ? 1 : 0;
  mine->fresh = fresh ? 1 : 0;
  mine->in_place = (sendbuf == recvbuf) ? 1 : 0;
  {
    const bool reduces = coll == COLL_ALLREDUCE || coll == COLL_REDUCE, rooted = coll == COLL_BCAST || coll == COLL_REDUCE;
    uint64_t h = 0x9E3779B97F4A7C15ull;
    for (uint64_t v : {(uint64_t)coll + 1, (uint64_t)(push ? 2 : 1), (uint64_t)send_bytes, reduces ? (uint64_t)dtype + 1 : 0,
                       reduces ? (uint64_t)op + 1 : 0, rooted ? (uint64_t)root + 1 : 0, (uint64_t)iters}) {
      h = (h ^ v) * 0x100000001B3ull;
      h ^= h >> 29;
    }
    mine->sig = (uint32_t)(h ^ (h >> 32));
  }
  mine->verdict.store(0, std::memory_order_relaxed);
  mine->seq.store(seq, std::memory_order_release);
  int rc = c->ctl->barrier(tmo(c));
  if (rc != XMPI_OK) {
    set_last_error("zero-copy collective: a peer did not arrive");
    return rc;
  }

  drop_retired(c);

  // 2. every rank reads the same descriptors and reaches the same decision
  bool all_ok = true, any_fresh = false, any_in_place = false;
  for (int p = 0; p < N; p++) {
    BufDesc* d = c->ctl->desc(p, seq);
    if (d->seq.load(std::memory_order_acquire) != seq) {
      set_last_error("zero-copy collective: ranks disagree on the collective sequence (mismatched calls?)");
      c->ctl->set_abort(XMPI_ERR_STATE);
      return XMPI_ERR_STATE;
    }
    if (d->sig != mine->sig) {  // (every rank reads the same descriptors: every rank returns this)
      set_last_error("collective: the ranks are not in the same call (collective, length, dtype, operation or root differ between "
                     "this rank and a peer); nothing was moved");
      c->ctl->set_abort(XMPI_ERR_ARG);
      return XMPI_ERR_ARG;
    }
    all_ok = all_ok && d->ok == 1;
    any_fresh = any_fresh || d->fresh == 1;
    any_in_place = any_in_place || d->in_place == 1;
  }
  if (!all_ok) {  // staged path, on every rank
    c->zc_fallbacks_unregistered++;
    return XMPI_OK;
  }

  // 3. map the peers' buffers (cached per allocation)
  const int mypid = (int)getpid();
  char* psend[kMaxRanks];
  char* precv[kMaxRanks];
  bool mapped = true;
  for (int p = 0; p < N && mapped; p++) {
    if (p == me) {
      psend[p] = (char*)const_cast<void*>(sendbuf);
      precv[p] = (char*)recvbuf;
      continue;
    }
    const BufDesc* d = c->ctl->desc(p, seq);
    const int pid = c->ctl->info(p)->pid;
    if (pid == mypid) {  // a thread of this process: its pointers are mine
      psend[p] = (char*)(uintptr_t)(d->send.base + d->send.offset);
      precv[p] = (char*)(uintptr_t)(d->recv.base + d->recv.offset);
      continue;
    }
    void *bs = nullptr, *br = nullptr;
    hipError_t me_err = map_peer(pid, d->send, &bs);
    if (me_err == hipSuccess) {
      if (d->recv.base == d->send.base && d->recv.gen == d->send.gen) br = bs;
      else me_err = map_peer(pid, d->recv, &br);
    }
    if (me_err != hipSuccess) {
      mapped = false;
      static std::atomic<int> said{0};
      if (said.fetch_add(1) < 3)  // a mapping failure costs the zero-copy path: say why, a few times
        fprintf(stderr, "xmpi: rank %d cannot map a buffer of rank %d (pid %d): %s; %zu mappings open, %d fds open\n",
                me, p, pid, hipGetErrorString(me_err), g_maps.size(), count_open_fds());
    }
    if (mapped) {
      psend[p] = (char*)bs + d->send.offset;
      precv[p] = (char*)br + d->recv.offset;
    }
  }
  if (any_fresh) {  // new allocations were opened somewhere: agree that everybody could
    mine->verdict.store(mapped ? 1 : -1, std::memory_order_release);
    rc = c->ctl->barrier(tmo(c));
    if (rc != XMPI_OK) return rc;
    for (int p = 0; p < N; p++)
      if (c->ctl->desc(p, seq)->verdict.load(std::memory_order_acquire) != 1) {  // staged, everywhere
        c->zc_fallbacks_unmappable++;
        return XMPI_OK;
      }
  } else if (!mapped) {
    set_last_error("zero-copy collective: a peer buffer that was mapped before can no longer be mapped");
    c->ctl->set_abort(XMPI_ERR_HIP);
    return XMPI_ERR_HIP;
  }

  // 4. the data movement: one kernel (bcast of a large buffer: two, with a barrier in between)
  hipStream_t s = c->local_stream;
  Launcher L(c);
  const size_t al = std::max<size_t>(1, 16 / es);
  const bool use_push = push && coll == COLL_ALLREDUCE && !any_in_place && N > 1 && count % ((size_t)N * al) == 0;
  if (use_push) {
    // Write-only variant (XMPI_ALGO_ZPUSH): nothing is READ over xGMI.  The receive buffer of rank q is
    // its own staging area -- region p (p != q) receives rank p's contribution to chunk q, region q is
    // where q folds them (rank order) -- and then every rank pushes its folded chunk to everybody.
    // Three kernels and two more barriers than the read-based form, the same bytes on the wire, all of
    // them posted writes.  Needs out-of-place buffers and equal chunks; otherwise the read-based form runs.
    const size_t C = count / (size_t)N, cb = C * es;
    {  // 1. my contribution to chunk q -> region `me` of rank q's receive buffer
      void* dst[kMaxBatch];
      const void* src[kMaxBatch];
      size_t bytes[kMaxBatch];
      int n = 0;
      for (int d = 1; d < N; d++) {
        const int q = (me + d) % N;
        dst[n] = precv[q] + (size_t)me * cb;
        src[n] = psend[me] + (size_t)q * cb;
        bytes[n] = cb;
        if (++n == kMaxBatch || d == N - 1) {
          rc = L.begin(2 * (size_t)n * cb);
          if (rc) return rc;
          XMPI_HIP(launch_copy_batch(dst, nullptr, src, bytes, n, s, L.start, L.stop));
          rc = L.finish();
          if (rc) return rc;
          n = 0;
        }
      }
    }
    rc = c->ctl->barrier(tmo(c));  // every contribution to my chunk has landed in my receive buffer
    if (rc != XMPI_OK) return rc;
    {  // 2. fold chunk `me` in rank order: all operands are local now
      const void* srcs[kMaxRanks];
      for (int p = 0; p < N; p++) srcs[p] = (p == me) ? psend[me] + (size_t)me * cb : precv[me] + (size_t)p * cb;
      void* d1[1] = {precv[me] + (size_t)me * cb};
      rc = L.begin((size_t)(N + 1) * cb);
      if (rc) return rc;
      XMPI_HIP(launch_reduce_n_multi(d1, 1, srcs, N, C, dtype, op, s, L.start, L.stop));
      rc = L.finish();
      if (rc) return rc;
    }
    rc = c->ctl->barrier(tmo(c));  // nobody still reads the staging regions my result is about to overwrite
    if (rc != XMPI_OK) return rc;
    {  // 3. my folded chunk -> region `me` of everybody's receive buffer
      void* dsts[kMaxRanks];
      int nd = 0;
      for (int d = 1; d < N; d++) dsts[nd++] = precv[(me + d) % N] + (size_t)me * cb;
      rc = L.begin((size_t)(1 + nd) * cb);
      if (rc) return rc;
      XMPI_HIP(launch_copy_multi(dsts, nd, precv[me] + (size_t)me * cb, cb, s, L.start, L.stop));
      rc = L.finish();
      if (rc) return rc;
    }
  } else if (coll == COLL_ALLREDUCE || coll == COLL_REDUCE) {
    // Ranks hosted by threads of this process on this GPU share one stream and one HBM: their chunks
    // are adjacent, so the lowest of them folds the whole run in ONE launch (the others only take part
    // in the barriers) instead of one launch each queueing behind the same runtime lock.
    int lo = me, hi = me;  // the maximal run of consecutive co-located ranks around me
    if (c->zc_group_launch) {
      while (lo > 0 && c->peer_coloc[lo - 1]) lo--;
      while (hi + 1 < N && c->peer_coloc[hi + 1]) hi++;
    }
    size_t off = 0, cnt = 0, off_hi = 0, cnt_hi = 0;
    zc_chunk(count, es, N, lo, &off, &cnt);
    zc_chunk(count, es, N, hi, &off_hi, &cnt_hi);
    cnt = off_hi + cnt_hi - off;
    if (me == lo && cnt > 0) {
      const void* srcs[kMaxRanks];
      void* dsts[kMaxRanks];
      for (int p = 0; p < N; p++) srcs[p] = psend[p] + off * es;
      int nd = 0;
      if (coll == COLL_REDUCE) {
        dsts[nd++] = precv[root] + off * es;
      } else {
        dsts[nd++] = precv[me] + off * es;  // local store first, then one store per link
        for (int d = 1; d < N; d++) dsts[nd++] = precv[(me + d) % N] + off * es;
      }
      // iters > 1 (xmpi_allreduce_repeat, every rank of the job a thread of this process on this GPU): the same
      // allreduce `iters` times, enqueued back to back on the one in-order stream all ranks share and waited for
      // once -- stream order is all the synchronisation K identical steps need.  Sampled launches carry their own
      // begin / end events, read after the wait.
      std::vector<std::pair<hipEvent_t, hipEvent_t>> sampled;
      for (int it = 0; it < iters; it++) {
        rc = L.begin((size_t)(N + nd) * cnt * es);
        if (rc) return rc;
        XMPI_HIP(launch_reduce_n_multi(dsts, nd, srcs, N, cnt, dtype, op, s, L.start, L.stop));
        if (it + 1 < iters && L.start) {
          sampled.push_back({L.start, L.stop});
          L.start = L.stop = nullptr;
        }
      }
      rc = L.finish();
      if (rc) return rc;
      for (auto& ev : sampled) {
        float ms = 0.f;
        XMPI_HIP(hipEventElapsedTime(&ms, ev.first, ev.second));
        ProfCounter& pc = c->prof[PROF_ZCOPY];
        pc.add(ms, (size_t)(N + nd) * cnt * es);
        ev_put(c, ev.first, true);
        ev_put(c, ev.second, true);
      }
    }
  } else if (coll == COLL_ALLGATHER) {
    void* dsts[kMaxRanks];
    int nd = 0;
    for (int d = 0; d < N; d++) dsts[nd++] = precv[(me + d) % N] + (size_t)me * send_bytes;
    rc = L.begin((size_t)(1 + nd) * send_bytes);
    if (rc) return rc;
    XMPI_HIP(launch_copy_multi(dsts, nd, psend[me], send_bytes, s, L.start, L.stop));
    rc = L.finish();
    if (rc) return rc;
  } else {  // COLL_BCAST: the buffer is both input (root) and output (everyone else)
    const bool push = N <= 2 || send_bytes <= (size_t)std::max<long>(0, c->zc_bcast_push_bytes);
    if (push) {
      if (me == root) {
        void* dsts[kMaxRanks];
        int nd = 0;
        for (int d = 1; d < N; d++) dsts[nd++] = precv[(me + d) % N];
        rc = L.begin((size_t)(1 + nd) * send_bytes);
        if (rc) return rc;
        XMPI_HIP(launch_copy_multi(dsts, nd, psend[me], send_bytes, s, L.start, L.stop));
        rc = L.finish();
        if (rc) return rc;
      }
    } else {
      if (me == root) {  // scatter: chunk j goes to rank j only
        void* dst[kMaxBatch];
        const void* src[kMaxBatch];
        size_t bytes[kMaxBatch];
        int n = 0;
        size_t total = 0;
        auto flush = [&]() -> int {
          if (n == 0) return XMPI_OK;
          int r2 = L.begin(2 * total);
          if (r2) return r2;
          XMPI_HIP(launch_copy_batch(dst, nullptr, src, bytes, n, s, L.start, L.stop));
          n = 0;
          total = 0;
          return L.finish();
        };
        for (int d = 1; d < N; d++) {
          const int j = (me + d) % N;
          size_t off = 0, cnt = 0;
          zc_chunk(count, es, N, j, &off, &cnt);
          if (cnt == 0) continue;
          dst[n] = precv[j] + off * es;
          src[n] = psend[me] + off * es;
          bytes[n] = cnt * es;
          total += cnt * es;
          if (++n == kMaxBatch && (rc = flush()) != XMPI_OK) return rc;
        }
        if ((rc = flush()) != XMPI_OK) return rc;
      }
      rc = c->ctl->barrier(tmo(c));  // chunk j has landed on rank j
      if (rc != XMPI_OK) return rc;
      {  // allgather of the chunks: rank j forwards chunk j (the root its own) to everyone who lacks it
        size_t off = 0, cnt = 0;
        zc_chunk(count, es, N, me, &off, &cnt);
        void* dsts[kMaxRanks];
        int nd = 0;
        for (int d = 1; d < N; d++) {
          const int q = (me + d) % N;
          if (q != root) dsts[nd++] = precv[q] + off * es;
        }
        if (cnt > 0 && nd > 0) {
          rc = L.begin((size_t)(1 + nd) * cnt * es);
          if (rc) return rc;
          XMPI_HIP(launch_copy_multi(dsts, nd, precv[me] + off * es, cnt * es, s, L.start, L.stop));
          rc = L.finish();
          if (rc) return rc;
        }
      }
    }
  }

  // 5. nobody leaves while a peer may still be reading its input or writing its output
  rc = c->ctl->barrier(tmo(c));
  if (rc != XMPI_OK) {
    set_last_error("zero-copy collective: a peer did not finish");
    return rc;
  }
  *done = true;
  return XMPI_OK;
}

int zero_copy_collective(xmpi_comm* c, int coll, int root, const void* sendbuf, void* recvbuf, size_t count,
                         int dtype, int op, bool push, bool* done, int iters) {
  RoctxRange range("xmpi:zcopy(host rendezvous) %s count=%zu x%d", coll_name(coll), count, iters);
  const int rc = zc_run(c, coll, root, sendbuf, recvbuf, count, dtype, op, push, done, iters);
  if (rc != XMPI_OK) c->ctl->set_abort(rc);  // peers waiting in a barrier stop waiting
  return rc;
}

}  // namespace xmpi
