// engine.cpp -- executes a step table (plan.h) over the HBM pipes, and the tagged
// point-to-point rendezvous.  Compiled by hipcc as host code.
//
// Data path of one SEND/RECV pair (replaces gob-encode -> net.Conn -> gob-decode of
// network.go:518-625):
//   sender    : peer copy  local HBM -> slot in the RECEIVER's window (xGMI write, SDMA or copy
//               kernel) on send_stream[peer]; when its event completes the host publishes
//               head++ in the shared control block.
//   receiver  : sees head > consumed, launches the reduction / copy-out kernel on
//               recv_stream[peer] reading the slot from its own HBM; when that completes it
//               publishes tail++ (the ack of network.go:616-624, one counter instead of a message).
// Nothing on the GPU ever blocks on another process: a step is enqueued only once its
// cross-process precondition already holds, so streams cannot deadlock however HIP maps them
// to hardware queues.  Same-process dependencies are chained with events (no host round trip).
#include <algorithm>
#include <cstddef>
#include <cstring>

#include "comm.h"
#include "kernels.h"

namespace xmpi {

hipEvent_t ev_get(xmpi_comm* c, bool timed) {
  std::lock_guard<std::mutex> g(c->ev_mu);
  std::vector<hipEvent_t>& pool = timed ? c->ev_timed_free : c->ev_free;
  if (!pool.empty()) {
    hipEvent_t e = pool.back();
    pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreateWithFlags(&e, timed ? hipEventDefault : hipEventDisableTiming) != hipSuccess) return nullptr;
  return e;
}

void ev_put(xmpi_comm* c, hipEvent_t e, bool timed) {
  std::lock_guard<std::mutex> g(c->ev_mu);
  if (e) (timed ? c->ev_timed_free : c->ev_free).push_back(e);
}

bool is_device_pointer(const void* p) {
  hipPointerAttribute_t a;
  memset(&a, 0, sizeof a);
  hipError_t e = hipPointerGetAttributes(&a, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();  // unregistered host memory: clear the sticky error
    return false;
  }
  return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged || a.type == hipMemoryTypeArray;
}

namespace {

// Ranks hosted by threads of one process enqueue onto one stream: serialise each step's enqueue so a
// profiled launch is bracketed by ITS events only.
std::mutex g_shared_stream_mu;
struct SharedStreamLock {
  std::unique_lock<std::mutex> l;
  explicit SharedStreamLock(const xmpi_comm* c) {
    if (c->shared_stream) l = std::unique_lock<std::mutex>(g_shared_stream_mu);
  }
};

struct InFlight {
  int step;
  hipEvent_t done;         // completion tracking (not needed for stream-ordered steps)
  hipEvent_t start, stop;  // only for profiled launches: attached to the dispatch itself
  int prof_kind;
  size_t prof_bytes;
  std::vector<int> more;   // further steps completed by the same launch (batched copies)
};

int peer_copy(xmpi_comm* c, void* dst, const void* src, size_t bytes, hipStream_t s, hipEvent_t es, hipEvent_t ee) {
  if (c->copy_engine == 1) {
    XMPI_HIP(launch_copy(dst, src, bytes, s, es, ee));
  } else {
    if (es) XMPI_HIP(hipEventRecord(es, s));
    XMPI_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s));
    if (ee) XMPI_HIP(hipEventRecord(ee, s));
  }
  return XMPI_OK;
}

struct Exec {
  xmpi_comm* c;
  const Plan& plan;
  char* bufs[3];
  int dtype, op;
  int N, L;
  size_t es;

  std::vector<std::deque<int>> sq, rq;  // per (peer, lane): pending SEND / RECV steps in FIFO order
  std::deque<int> lq;                   // LOCAL_COPY / REDUCE_N
  std::vector<uint8_t> state;           // 0 pending, 1 issued, 2 complete
  std::vector<hipEvent_t> ev;
  std::vector<int> stream_of;
  std::vector<char*> slot_ptr;          // RECV_*: the slot this step reads
  std::vector<uint64_t> slot_seq;
  std::vector<uint64_t> out_seq;        // fused steps: sequence number of the slot they push
  std::vector<std::deque<InFlight>> fl;  // per stream, completion is in issue order
  std::vector<std::deque<char>> unreleased;  // per (peer, lane): popped slots not yet released
  std::vector<std::deque<char>> unpublished;  // per (peer, lane): pushed slots whose head is not yet out
  size_t remaining;
  double last_progress;
  bool eager_used = false;
  std::vector<InFlight> prof_pending;  // sampled eager steps: timed after the final sync

  Exec(xmpi_comm* comm, const Plan& p, const void* sb, void* rb, int dt, int o)
      : c(comm), plan(p), dtype(dt), op(o), N(comm->size), L(comm->lanes) {
    bufs[BUF_SEND] = (char*)const_cast<void*>(sb);
    bufs[BUF_RECV] = (char*)rb;
    bufs[BUF_TEMP] = (char*)comm->temp;
    es = xmpi_dtype_size((xmpi_dtype)dt);
    const size_t n = p.steps.size();
    sq.resize((size_t)N * L);
    rq.resize((size_t)N * L);
    state.assign(n, 0);
    ev.assign(n, nullptr);
    stream_of.assign(n, -1);
    slot_ptr.assign(n, nullptr);
    slot_seq.assign(n, 0);
    out_seq.assign(n, 0);
    fl.resize((size_t)2 * N + 3);
    unreleased.resize((size_t)N * L);
    unpublished.resize((size_t)N * L);
    remaining = n;
    for (size_t i = 0; i < n; i++) {
      const Step& s = p.steps[i];
      if (s.kind == STEP_SEND) sq[(size_t)s.peer * L + s.lane].push_back((int)i);
      else if (s.kind == STEP_RECV_REDUCE || s.kind == STEP_RECV_COPY || s.kind == STEP_RECV_HOLD)
        rq[(size_t)s.peer * L + s.lane].push_back((int)i);
      else if (is_fused(s)) {  // pops one pipe and pushes another: in FIFO order on both
        rq[(size_t)s.peer * L + s.lane].push_back((int)i);
        sq[(size_t)s.peer2 * L + s.lane2].push_back((int)i);
      } else lq.push_back((int)i);
    }
    last_progress = now_seconds();
  }

  static bool is_fused(const Step& s) { return s.kind == STEP_RECV_REDUCE_SEND || s.kind == STEP_RECV_COPY_SEND; }

  hipStream_t stream(int sid) const {
    if (sid < N) return c->send_stream[sid];
    if (sid < 2 * N) return c->recv_stream[sid - N];
    if (sid == 2 * N + 1) return c->batch_send_stream;
    if (sid == 2 * N + 2) return c->batch_recv_stream;
    return c->local_stream;
  }

  // a step may be enqueued once its dependencies are enqueued (chained with hipStreamWaitEvent)
  bool deps_issued(const Step& s) const {
    for (int d = 0; d < s.ndeps; d++)
      if (state[(size_t)s.deps[d]] < 1) return false;
    return true;
  }

  int chain_deps(const Step& s, int sid) {
    for (int d = 0; d < s.ndeps; d++) {
      const int j = s.deps[d];
      if (state[(size_t)j] == 1 && stream(stream_of[(size_t)j]) != stream(sid) && ev[(size_t)j])
        XMPI_HIP(hipStreamWaitEvent(stream(sid), ev[(size_t)j], 0));
    }
    return XMPI_OK;
  }

  int begin_op(int i, InFlight* f, int prof_kind, size_t prof_bytes) {
    f->step = i;
    f->done = f->start = f->stop = nullptr;
    f->prof_kind = prof_kind;
    f->prof_bytes = prof_bytes;
    if (c->prof_on && prof_kind >= 0 &&
        (c->prof_seq[prof_kind]++ % (uint64_t)std::max<long>(1, c->prof_every)) == 0) {
      f->start = ev_get(c, true);
      f->stop = ev_get(c, true);
      if (!f->start || !f->stop) return XMPI_ERR_HIP;
    }
    return XMPI_OK;
  }

  // eager = the consumer of this step's effect runs on the SAME in-order stream (ranks hosted by
  // one process on one GPU): stream order already guarantees what the completion event would, so
  // the counters are published at enqueue time and no event is recorded at all.
  int end_op(int i, int sid, InFlight* f, bool eager) {
    stream_of[(size_t)i] = sid;
    for (int j : f->more) stream_of[(size_t)j] = sid;
    if (eager) {
      if (f->start) prof_pending.push_back(*f);
      eager_used = true;
      state[(size_t)i] = 2;
      remaining--;
      publish(plan.steps[(size_t)i], i);
      for (int j : f->more) {
        state[(size_t)j] = 2;
        remaining--;
        publish(plan.steps[(size_t)j], j);
      }
      return XMPI_OK;
    }
    f->done = ev_get(c, false);
    if (!f->done) return XMPI_ERR_HIP;
    XMPI_HIP(hipEventRecord(f->done, stream(sid)));
    ev[(size_t)i] = f->done;
    state[(size_t)i] = 1;
    for (int j : f->more) {
      ev[(size_t)j] = f->done;
      state[(size_t)j] = 1;
    }
    fl[(size_t)sid].push_back(*f);
    return XMPI_OK;
  }

  bool coloc(int peer) const { return c->shared_stream && c->peer_coloc[peer]; }

  void release_slot(int peer, int lane, uint64_t seq) {
    std::deque<char>& u = unreleased[(size_t)peer * L + lane];
    uint64_t& base = c->recvd_released[peer][lane];
    u[(size_t)(seq - base)] = 1;
    while (!u.empty() && u.front()) {
      u.pop_front();
      base++;
    }
    c->ctl->pipe(peer, c->rank, lane)->tail.v.store(base, std::memory_order_release);
  }

  int account(const InFlight& f) {
    if (f.start) {
      float ms = 0.f;
      XMPI_HIP(hipEventElapsedTime(&ms, f.start, f.stop));
      ProfCounter& pc = c->prof[f.prof_kind];
      pc.add(ms, f.prof_bytes);
      ev_put(c, f.start, true);
      ev_put(c, f.stop, true);
    }
    if (f.done) ev_put(c, f.done, false);
    return XMPI_OK;
  }

  // pushes of one pipe may complete out of order (batched and single launches run on different
  // streams): head only advances over a gap-free prefix of filled slots
  void publish_push(int peer, int lane, uint64_t seq) {
    std::deque<char>& u = unpublished[(size_t)peer * L + lane];
    uint64_t& base = c->sent_done[peer][lane];
    u[(size_t)(seq - base)] = 1;
    while (!u.empty() && u.front()) {
      u.pop_front();
      base++;
    }
    c->ctl->pipe(c->rank, peer, lane)->head.v.store(base, std::memory_order_release);
  }

  // make the effect of a finished (or stream-ordered) step visible to the peer
  void publish(const Step& s, int step) {
    switch (s.kind) {
      case STEP_SEND:
        publish_push(s.peer, s.lane, slot_seq[(size_t)step]);
        break;
      case STEP_RECV_REDUCE:
      case STEP_RECV_COPY:
        release_slot(s.peer, s.lane, slot_seq[(size_t)step]);
        break;
      case STEP_RECV_REDUCE_SEND:
      case STEP_RECV_COPY_SEND:
        publish_push(s.peer2, s.lane2, out_seq[(size_t)step]);
        release_slot(s.peer, s.lane, slot_seq[(size_t)step]);
        break;
      case STEP_REDUCE_N:
        for (int k = 0; k < s.nsrcs; k++)
          if (s.srcs[k] >= 0) {
            const Step& h = plan.steps[(size_t)s.srcs[k]];
            release_slot(h.peer, h.lane, slot_seq[(size_t)s.srcs[k]]);
          }
        break;
      default:
        break;
    }
  }

  int complete(const InFlight& f) {
    int rc = account(f);
    if (rc) return rc;
    ev[(size_t)f.step] = nullptr;
    state[(size_t)f.step] = 2;
    remaining--;
    publish(plan.steps[(size_t)f.step], f.step);
    for (int j : f.more) {
      ev[(size_t)j] = nullptr;
      state[(size_t)j] = 2;
      remaining--;
      publish(plan.steps[(size_t)j], j);
    }
    return XMPI_OK;
  }

  // ---- batched launches: everything of one kind that is ready right now goes out in ONE launch ------
  enum BatchKind { BATCH_SEND = 0, BATCH_RECV_COPY, BATCH_RECV_REDUCE, BATCH_RRS, BATCH_RCS, BATCH_KINDS };

  static int kind_of_batch(int bk) {
    switch (bk) {
      case BATCH_SEND: return STEP_SEND;
      case BATCH_RECV_COPY: return STEP_RECV_COPY;
      case BATCH_RECV_REDUCE: return STEP_RECV_REDUCE;
      case BATCH_RRS: return STEP_RECV_REDUCE_SEND;
      default: return STEP_RECV_COPY_SEND;
    }
  }

  bool can_pop(const Step& s) const {
    return c->ctl->pipe(s.peer, c->rank, s.lane)->head.v.load(std::memory_order_acquire) > c->recvd[s.peer][s.lane];
  }
  bool can_push(int peer, int lane) const {
    const uint64_t tail = c->ctl->pipe(c->rank, peer, lane)->tail.v.load(std::memory_order_acquire);
    return c->sent[peer][lane] - tail < (uint64_t)c->fifo_depth;
  }

  // is step i (the front of a queue) of batch kind bk and ready to be launched right now?
  bool batch_ready(int i, int bk) const {
    const Step& s = plan.steps[(size_t)i];
    if (s.kind != kind_of_batch(bk) || !deps_issued(s) || s.bytes > c->slot_bytes) return false;
    switch (bk) {
      case BATCH_SEND: return can_push(s.peer, s.lane);
      case BATCH_RECV_COPY:
      case BATCH_RECV_REDUCE: return can_pop(s);
      default: {  // fused: next in line on BOTH of its pipes, data in, room out
        const std::deque<int>& in = rq[(size_t)s.peer * L + s.lane];
        const std::deque<int>& out = sq[(size_t)s.peer2 * L + s.lane2];
        return !in.empty() && in.front() == i && !out.empty() && out.front() == i && can_pop(s) &&
               can_push(s.peer2, s.lane2);
      }
    }
  }

  int issue_batch(const std::vector<int>& steps, int bk) {
    void *dst[kMaxBatch], *dst2[kMaxBatch];
    const void *src[kMaxBatch], *opa[kMaxBatch];
    size_t bytes[kMaxBatch], counts[kMaxBatch];
    const int n = (int)steps.size();
    const int sid = bk == BATCH_SEND ? 2 * N + 1 : 2 * N + 2;
    size_t total = 0, written = 0;
    bool eager = c->shared_stream;
    for (int k = 0; k < n; k++) {
      const int i = steps[(size_t)k];
      const Step& s = plan.steps[(size_t)i];
      int rc = chain_deps(s, sid);
      if (rc) return rc;
      dst[k] = dst2[k] = nullptr;
      src[k] = opa[k] = nullptr;
      if (bk != BATCH_SEND) {  // pop the incoming slot
        const uint64_t seq = c->recvd[s.peer][s.lane];
        char* slot = c->window + c->coll_slot_off(s.peer, s.lane, seq);
        slot_ptr[(size_t)i] = slot;
        slot_seq[(size_t)i] = seq;
        c->recvd[s.peer][s.lane] = seq + 1;
        unreleased[(size_t)s.peer * L + s.lane].push_back(0);
        src[k] = slot;
        opa[k] = bufs[s.src_buf] + s.src_off;  // reductions: the local operand
        if (bk != BATCH_RRS || s.keep_local) dst[k] = bufs[s.dst_buf] + s.dst_off;
        if (!coloc(s.peer)) eager = false;
      }
      if (bk == BATCH_SEND || bk == BATCH_RRS || bk == BATCH_RCS) {  // claim the outgoing slot
        const int peer = bk == BATCH_SEND ? s.peer : s.peer2, lane = bk == BATCH_SEND ? s.lane : s.lane2;
        const uint64_t seq = c->sent[peer][lane];
        char* remote = c->peer_window[peer] + c->coll_slot_off(c->rank, lane, seq);
        c->sent[peer][lane] = seq + 1;
        unpublished[(size_t)peer * L + lane].push_back(0);
        if (bk == BATCH_SEND) {
          slot_seq[(size_t)i] = seq;
          dst[k] = remote;
          src[k] = bufs[s.src_buf] + s.src_off;
        } else {
          out_seq[(size_t)i] = seq;
          dst2[k] = remote;
        }
        if (!coloc(peer)) eager = false;
      }
      bytes[k] = s.bytes;
      counts[k] = s.bytes / es;
      total += s.bytes;
      written += s.bytes * (size_t)((dst[k] ? 1 : 0) + (dst2[k] ? 1 : 0));
    }
    InFlight f;
    SharedStreamLock lk(c);
    const bool reduce = bk == BATCH_RECV_REDUCE || bk == BATCH_RRS;
    const int pk = bk == BATCH_SEND ? PROF_PEER : (reduce ? PROF_REDUCE2 : PROF_COPY);
    const size_t pb = bk == BATCH_SEND ? total : (reduce ? 2 * total + written : total + written);
    int rc = begin_op(steps[0], &f, pk, pb);
    if (rc) return rc;
    for (int k = 1; k < n; k++) f.more.push_back(steps[(size_t)k]);
    if (reduce)
      XMPI_HIP(launch_reduce2_batch(dst, dst2, opa, src, counts, n, dtype, op, stream(sid), f.start, f.stop));
    else
      XMPI_HIP(launch_copy_batch(dst, dst2, src, bytes, n, stream(sid), f.start, f.stop));
    return end_op(steps[0], sid, &f, eager);
  }

  // Returns the number of steps issued, <0 on error.  Plain sends and slot drains batch only when
  // they are kernels (copy_engine 1) and at least two are ready; the fused ring steps exist only as
  // kernels and always go through here, alone if need be.
  int try_batches() {
    int issued = 0;
    for (int bk = 0; bk < BATCH_KINDS; bk++) {
      const bool fused = bk == BATCH_RRS || bk == BATCH_RCS;
      if (!fused && !c->batch_copies) continue;
      if ((bk == BATCH_SEND || bk == BATCH_RECV_COPY) && c->copy_engine != 1) continue;
      std::vector<std::deque<int>>& qs = bk == BATCH_SEND ? sq : rq;
      const size_t max_group = c->batch_copies ? (size_t)kMaxBatch : 1;
      for (;;) {
        std::vector<int> ready;
        for (auto& q : qs) {
          if (q.empty() || ready.size() >= max_group) continue;
          if (batch_ready(q.front(), bk)) ready.push_back(q.front());
        }
        if (ready.size() < (fused ? 1u : 2u)) break;  // a single plain step takes the ordinary path
        int rc = issue_batch(ready, bk);
        if (rc) return rc;
        for (int i : ready) {
          const Step& s = plan.steps[(size_t)i];
          if (bk == BATCH_SEND) sq[(size_t)s.peer * L + s.lane].pop_front();
          else rq[(size_t)s.peer * L + s.lane].pop_front();
          if (fused) sq[(size_t)s.peer2 * L + s.lane2].pop_front();
        }
        issued += (int)ready.size();
      }
    }
    return issued;
  }

  // returns 1 if issued, 0 if not ready, <0 on error
  int try_send(int i) {
    const Step& s = plan.steps[(size_t)i];
    if (is_fused(s) || !deps_issued(s)) return 0;  // fused steps are launched by try_batches
    const uint64_t seq = c->sent[s.peer][s.lane];
    const uint64_t tail = c->ctl->pipe(c->rank, s.peer, s.lane)->tail.v.load(std::memory_order_acquire);
    if (seq - tail >= (uint64_t)c->fifo_depth) return 0;
    if (s.bytes > c->slot_bytes) return XMPI_ERR_ARG;
    const int sid = s.peer;
    int rc = chain_deps(s, sid);
    if (rc) return rc;
    InFlight f;
    SharedStreamLock lk(c);  // [start marker, launch, done marker] stay contiguous on a shared stream
    rc = begin_op(i, &f, PROF_PEER, s.bytes);
    if (rc) return rc;
    char* dst = c->peer_window[s.peer] + c->coll_slot_off(c->rank, s.lane, seq);
    rc = peer_copy(c, dst, bufs[s.src_buf] + s.src_off, s.bytes, stream(sid), f.start, f.stop);
    if (rc) return rc;
    c->sent[s.peer][s.lane] = seq + 1;
    slot_seq[(size_t)i] = seq;
    unpublished[(size_t)s.peer * L + s.lane].push_back(0);
    rc = end_op(i, sid, &f, coloc(s.peer));
    if (rc) return rc;
    return 1;
  }

  int try_recv(int i) {
    const Step& s = plan.steps[(size_t)i];
    if (is_fused(s) || !deps_issued(s)) return 0;  // fused steps are launched by try_batches
    const uint64_t seq = c->recvd[s.peer][s.lane];
    const uint64_t head = c->ctl->pipe(s.peer, c->rank, s.lane)->head.v.load(std::memory_order_acquire);
    if (head <= seq) return 0;
    char* slot = c->window + c->coll_slot_off(s.peer, s.lane, seq);
    slot_ptr[(size_t)i] = slot;
    slot_seq[(size_t)i] = seq;
    c->recvd[s.peer][s.lane] = seq + 1;
    unreleased[(size_t)s.peer * L + s.lane].push_back(0);
    if (s.kind == STEP_RECV_HOLD) {
      state[(size_t)i] = 2;
      remaining--;
      return 1;
    }
    const int sid = N + s.peer;
    int rc = chain_deps(s, sid);
    if (rc) return rc;
    InFlight f;
    SharedStreamLock lk(c);  // [start marker, launch, done marker] stay contiguous on a shared stream
    if (s.kind == STEP_RECV_REDUCE) {
      rc = begin_op(i, &f, PROF_REDUCE2, 3 * s.bytes);
      if (rc) return rc;
      XMPI_HIP(launch_reduce2(bufs[s.dst_buf] + s.dst_off, bufs[s.src_buf] + s.src_off, slot, s.bytes / es, dtype,
                              op, stream(sid), f.start, f.stop));
    } else {
      rc = begin_op(i, &f, PROF_COPY, 2 * s.bytes);
      if (rc) return rc;
      XMPI_HIP(launch_copy(bufs[s.dst_buf] + s.dst_off, slot, s.bytes, stream(sid), f.start, f.stop));
    }
    return end_op(i, sid, &f, coloc(s.peer)) ? XMPI_ERR_HIP : 1;
  }

  int try_local(int i) {
    const Step& s = plan.steps[(size_t)i];
    if (!deps_issued(s)) return 0;
    const int sid = 2 * N;
    if (s.kind == STEP_REDUCE_N) {
      for (int k = 0; k < s.nsrcs; k++)
        if (s.srcs[k] >= 0 && state[(size_t)s.srcs[k]] == 0) return 0;
    }
    int rc = chain_deps(s, sid);
    if (rc) return rc;
    InFlight f;
    SharedStreamLock lk(c);  // [start marker, launch, done marker] stay contiguous on a shared stream
    if (s.kind == STEP_REDUCE_N) {
      const void* srcs[kMaxSrcs];
      for (int k = 0; k < s.nsrcs; k++)
        srcs[k] = (s.srcs[k] < 0) ? (const void*)(bufs[s.src_buf] + s.src_off) : (const void*)slot_ptr[(size_t)s.srcs[k]];
      rc = begin_op(i, &f, PROF_REDUCEN, (size_t)(s.nsrcs + 1) * s.bytes);
      if (rc) return rc;
      XMPI_HIP(launch_reduce_n(bufs[s.dst_buf] + s.dst_off, srcs, s.nsrcs, s.bytes / es, dtype, op, stream(sid),
                               f.start, f.stop));
    } else {
      rc = begin_op(i, &f, PROF_COPY, 2 * s.bytes);
      if (rc) return rc;
      const char* src = bufs[s.src_buf] + s.src_off;
      char* dst = bufs[s.dst_buf] + s.dst_off;
      XMPI_HIP(launch_copy(dst, src, s.bytes, stream(sid), f.start, f.stop));
    }
    bool eager = c->shared_stream;
    if (s.kind == STEP_REDUCE_N)
      for (int k = 0; k < s.nsrcs; k++)
        if (s.srcs[k] >= 0 && !coloc(plan.steps[(size_t)s.srcs[k]].peer)) eager = false;
    return end_op(i, sid, &f, eager) ? XMPI_ERR_HIP : 1;
  }

  int run() {
    Backoff bo;
    arm(bo, c);
    while (remaining > 0) {
      bool progressed = false;
      {
        const int nb = try_batches();
        if (nb < 0) return nb;
        progressed = nb > 0;
      }
      for (auto& q : sq)
        while (!q.empty()) {
          int r = try_send(q.front());
          if (r < 0) return r;
          if (r == 0) break;
          q.pop_front();
          progressed = true;
        }
      for (auto& q : rq)
        while (!q.empty()) {
          int r = try_recv(q.front());
          if (r < 0) return r;
          if (r == 0) break;
          q.pop_front();
          progressed = true;
        }
      // local steps may become ready out of order (different pieces wait on different pipes)
      for (size_t k = 0; k < lq.size();) {
        int r = try_local(lq[k]);
        if (r < 0) return r;
        if (r == 1) {
          lq.erase(lq.begin() + (long)k);
          progressed = true;
        } else {
          k++;
          if (k >= 8) break;  // look a few steps ahead only
        }
      }
      for (auto& q : fl)
        while (!q.empty()) {
          hipError_t e = hipEventQuery(q.front().done);
          if (e == hipErrorNotReady) {
            (void)hipGetLastError();
            break;
          }
          if (e != hipSuccess) return hip_fail(e, "hipEventQuery", __FILE__, __LINE__);
          int rc = complete(q.front());
          if (rc) return rc;
          q.pop_front();
          progressed = true;
        }
      if (progressed) {
        last_progress = now_seconds();
        bo.n = 0;
        continue;
      }
      if (c->ctl->aborted()) {
        set_last_error(c->ctl->abort_reason());
        return XMPI_ERR_PEER;
      }
      if (c->timeout_s > 0 && now_seconds() - last_progress > (double)c->timeout_s) {
        set_last_error("collective made no progress for " + std::to_string(c->timeout_s) + " s");
        return XMPI_ERR_TIMEOUT;
      }
      bo.pause();
    }
    if (eager_used) {  // the call is blocking: this rank's stream-ordered work must have finished
      hipEvent_t fin = ev_get(c, false);
      if (!fin) return XMPI_ERR_HIP;
      XMPI_HIP(hipEventRecord(fin, c->local_stream));
      const double ts = now_seconds();
      XMPI_HIP(hipEventSynchronize(fin));
      c->last_sync_us = (now_seconds() - ts) * 1e6;
      ev_put(c, fin, false);
      for (const InFlight& f : prof_pending) {
        int rc = account(f);
        if (rc) return rc;
      }
      prof_pending.clear();
    }
    return XMPI_OK;
  }
};

}  // namespace

int run_plan(xmpi_comm* c, const Plan& plan, const void* sendbuf, void* recvbuf, int dtype, int op) {
  if (plan.steps.empty()) return XMPI_OK;
  if (!c->windows_ok) {  // (api.cpp collective() sends every call of such a job to the device-synchronised path: not reached)
    set_last_error("the staged step tables need the HBM windows, which this job could not map (xmpi_degraded)");
    return XMPI_ERR_UNSUPPORTED;
  }
  RoctxRange range("xmpi:run_plan steps=%zu", plan.steps.size());
  {
    const int src = ensure_streams(c);
    if (src != XMPI_OK) return src;
  }
  if (plan.temp_bytes > c->temp_bytes) {
    if (c->temp) XMPI_HIP(hipFree(c->temp));
    c->temp = nullptr;
    c->temp_bytes = 0;
    XMPI_HIP(hipMalloc(&c->temp, plan.temp_bytes));
    c->temp_bytes = plan.temp_bytes;
  }
  const double t0 = now_seconds();
  Exec ex(c, plan, sendbuf, recvbuf, dtype, op);
  c->last_sync_us = 0;
  int rc = ex.run();
  c->last_run_us = (now_seconds() - t0) * 1e6;
  if (rc != XMPI_OK) {
    c->ctl->set_abort(rc);
    // leave no work behind that still references pooled events
    (void)hipDeviceSynchronize();
  }
  return rc;
}

// ---- tagged point-to-point --------------------------------------------------------------------

namespace {

struct P2PStream {
  hipStream_t s = nullptr;
};

hipStream_t p2p_stream_get(xmpi_comm* c) {
  std::lock_guard<std::mutex> g(c->p2p_mu);
  if (!c->p2p_streams.empty()) {
    hipStream_t s = c->p2p_streams.back();
    c->p2p_streams.pop_back();
    return s;
  }
  return stream_acquire(c->device);
}

void p2p_stream_put(xmpi_comm* c, hipStream_t s) {
  std::lock_guard<std::mutex> g(c->p2p_mu);
  c->p2p_streams.push_back(s);
}

struct TagGuard {
  xmpi_comm* c;
  std::set<std::pair<int, int>>* reg;
  std::pair<int, int> key;
  bool held = false;
  TagGuard(xmpi_comm* comm, std::set<std::pair<int, int>>* r, int peer, int tag) : c(comm), reg(r), key(peer, tag) {
    std::lock_guard<std::mutex> g(c->p2p_mu);
    held = reg->insert(key).second;
  }
  ~TagGuard() {
    if (held) {
      std::lock_guard<std::mutex> g(c->p2p_mu);
      reg->erase(key);
    }
  }
};

struct StreamLease {
  xmpi_comm* c;
  hipStream_t s;
  explicit StreamLease(xmpi_comm* comm) : c(comm), s(p2p_stream_get(comm)) {}
  ~StreamLease() {
    if (s) p2p_stream_put(c, s);
  }
};

bool timed_out(xmpi_comm* c, double t0) { return c->timeout_s > 0 && now_seconds() - t0 > (double)c->timeout_s; }

// A send that found no matching receive within XMPI_TIMEOUT_S takes its message back: the entry goes from POSTED to
// FREE and the call returns XMPI_ERR_TIMEOUT with the job intact (the reference would block for ever,
// network.go:569; a test harness prefers an error).  false = a receive matched it in the meantime: keep waiting.
bool withdraw(MailEntry* m) {
  uint32_t expect = MAIL_POSTED;
  if (!m->state.compare_exchange_strong(expect, MAIL_CLAIMED, std::memory_order_acq_rel)) return false;
  m->pipe.head.v.store(0, std::memory_order_relaxed);
  m->pipe.tail.v.store(0, std::memory_order_relaxed);
  m->state.store(MAIL_FREE, std::memory_order_release);
  return true;
}

// an entry this rank has claimed but not posted goes back
static bool withdraw_claimed(MailEntry* m) {
  m->state.store(MAIL_FREE, std::memory_order_release);
  return true;
}

// rendezvous: wait for the receiver's verdict (network.go:569 waits for the ack message), free the entry
int await_ack(xmpi_comm* c, MailEntry* m, int dest, int tag) {
  int rc = XMPI_OK;
  double tp = now_seconds();
  Backoff bo;
  arm(bo, c);
  while (rc == XMPI_OK && m->state.load(std::memory_order_acquire) != MAIL_DONE) {
    if (c->ctl->aborted()) rc = XMPI_ERR_PEER;
    else if (c->timeout_s > 0 && now_seconds() - tp > (double)c->timeout_s) {
      if (withdraw(m)) {
        set_last_error("send to rank " + std::to_string(dest) + " tag " + std::to_string(tag) + ": no matching receive");
        return XMPI_ERR_TIMEOUT;
      }
      tp = now_seconds();  // matched a moment ago: the receiver is copying
    }
    bo.pause();
  }
  if (rc == XMPI_OK) {
    rc = m->status.load(std::memory_order_acquire);
    m->pipe.head.v.store(0, std::memory_order_relaxed);
    m->pipe.tail.v.store(0, std::memory_order_relaxed);
    m->state.store(MAIL_FREE, std::memory_order_release);
  } else {
    c->ctl->set_abort(rc);  // a peer failed: the entry is in an unknown state
  }
  return rc;
}

}  // namespace

// Blocks of the pull kernel (sched.hip p2p_pull_kernel: 16 KiB in flight per block; every block acquires before and releases
// after its share -- a cache operation each).  What bounds a message is the memory it comes out of.  Sender and receiver on ONE
// GPU: the sender's HBM, but the per-block cache operations cost more than extra blocks bring -- 16 MiB, half round trip, r04:
// 32 blocks 28.9 us, 64 23.6, 128 25.2, 256 34.0, 512 50.8, 1024 92.2 (1 MiB: 11.0 with 32, 12.2 with 64 and more) -- so one
// block per 256 KiB, at least 16, at most 512.  Different GPUs: ONE link; blocks beyond bandwidth x latency in flight add
// nothing but contention, so the cap follows the rate xmpi_link_probe measured for this pair (link_gbps; the nominal 64 GB/s
// per direction until it has run) at ~4 us of round trip, with a margin of 2.  "p2p_grid_cap" / XMPI_P2P_GRID_CAP override.
static long p2p_pull_cap(const xmpi_comm* c, int peer, size_t bytes) {
  if (c->p2p_grid_cap > 0) return c->p2p_grid_cap;
  const RankInfo* a = c->ctl->info(c->rank);
  const RankInfo* b = c->ctl->info(peer);
  if (strncmp(a->busid, b->busid, sizeof a->busid) == 0) return std::max<long>(16, std::min<long>((long)(bytes >> 18), 512));
  const double gbps = c->link_gbps[peer] > 0 ? c->link_gbps[peer] : 64.0;
  const long blocks = (long)(2.0 * gbps * 1e9 * 4e-6 / 16384.0) + 1;
  return std::max<long>(16, std::min<long>(blocks, 256));
}

constexpr size_t kP2PBounceBytes = (size_t)256 << 10;  // device -> host slice through pinned memory up to this length

// ---- the receive agent: a copy-and-ack kernel that lingers (sched.hip p2p_agent_kernel) ------------------------------------
// One command at a time per communicator.  Returns true when the agent copied the message and wrote both acks; false: the
// caller launches the ordinary kernel (the agent could not be started).
static bool agent_submit(xmpi_comm* c, void* dst, const void* from, size_t bytes, MailEntry* m) {
  if (c->p2p_agent_us <= 0 || !c->p2p_cmd_dev || !c->p2p_rec || !c->ctl_dev || bytes == 0 || bytes > ((size_t)512 << 10)) return false;  // (longer messages: the ordinary kernel's wide grid)
  // One command at a time.  The reference promises concurrent Receives on different {peer, tag} (mpi.go:121-125): a Receive
  // that finds the agent busy with somebody else's message does not queue up behind it -- the launch-per-message kernel on
  // this call's own stream serves it in parallel.
  std::unique_lock<std::mutex> g(c->agent_mu, std::try_to_lock);
  if (!g.owns_lock()) return false;
  volatile uint64_t* cmd = c->p2p_cmd;
  const uint64_t seq = ++c->agent_seq;
  const uint64_t mail_off = (uint64_t)((char*)&m->state - (char*)c->ctl->base());
  // (the agent polls all four words while they are written and takes them only when [0] and [3] both carry this number:
  // every word is an atomic store, so that the words it reads early are merely old, never torn)
  __atomic_store_n((uint64_t*)&cmd[1], (uint64_t)(uintptr_t)from, __ATOMIC_RELAXED);
  __atomic_store_n((uint64_t*)&cmd[2], (uint64_t)(uintptr_t)dst, __ATOMIC_RELAXED);
  __atomic_store_n((uint64_t*)&cmd[3], (mail_off & 0xffffffffull) | (seq << 32), __ATOMIC_RELEASE);
  __atomic_store_n((uint64_t*)&cmd[0], 1ull | ((uint64_t)bytes << 2) | (seq << 24), __ATOMIC_RELEASE);  // the doorbell last
  auto launch = [&]() -> bool {
    if (!c->agent_stream) c->agent_stream = stream_acquire(c->device);
    if (!c->agent_stream) return false;
    __atomic_store_n((uint64_t*)&cmd[7], 0, __ATOMIC_RELEASE);
    P2PAgentArgs a;
    memset(&a, 0, sizeof a);
    a.cmd = c->p2p_cmd_dev;
    a.rec = c->p2p_rec;
    a.ctl_dev = (uint64_t)(uintptr_t)c->ctl_dev;
    a.seq0 = seq;
    a.launch = (c->p2p_agent_launch_no + 1) & 0x7fffff;
    a.alone_bytes = 64 << 10;
    a.patience_ticks = (uint64_t)c->p2p_agent_us * 100;  // wall_clock64 runs at 100 MHz
    a.mail_done_value = MAIL_DONE;
    if (launch_p2p_agent(a, 8, c->agent_stream) != hipSuccess) {  // (32 blocks: the 31 that watch a word in device memory slowed block 0 down -- 6.2 us instead of 4.5)
      (void)hipGetLastError();
      return false;
    }
    c->agent_running = true;
    c->p2p_agent_launch_no++;  // numbers the launches (never reset); p2p_agent_launches beside it is the caller's diagnostic count
    c->p2p_agent_launches++;
    return true;
  };
  if (!c->agent_running && !launch()) {
    __atomic_store_n((uint64_t*)&cmd[0], 0, __ATOMIC_RELEASE);  // nobody will read it
    --c->agent_seq;
    return false;
  }
  Backoff bo;
  const double t0 = now_seconds();
  for (unsigned spins = 1;; spins++) {
    if (__atomic_load_n((const uint64_t*)&cmd[6], __ATOMIC_ACQUIRE) == seq) break;  // copied and acknowledged
    if (__atomic_load_n((const uint64_t*)&cmd[7], __ATOMIC_ACQUIRE) != 0) {  // the agent had gone (its patience ran out)
      if (__atomic_load_n((const uint64_t*)&cmd[6], __ATOMIC_ACQUIRE) == seq) break;
      c->agent_running = false;
      if (!launch()) {  // (cannot happen after a launch that worked; give the message to the ordinary kernel)
        __atomic_store_n((uint64_t*)&cmd[0], 0, __ATOMIC_RELEASE);
        --c->agent_seq;
        return false;
      }
    }
    // Off the fast path, now and then: an agent that faulted (an unmapped payload), a queue that was torn down or a job that
    // was aborted must not leave this thread spinning with the lock held.  The stream is idle only when the agent has ended:
    // if it ended without serving this command and without saying "gone", it is broken -- the ordinary kernel takes over
    // (and reports whatever is wrong with the payload through its own error path).
    if ((spins & 0xfff) == 0) {
      bool give_up = c->ctl->aborted() || (c->timeout_s > 0 && now_seconds() - t0 > (double)c->timeout_s);
      if (!give_up) {
        const hipError_t e = hipStreamQuery(c->agent_stream);
        if (e != hipErrorNotReady) {
          (void)hipGetLastError();
          give_up = __atomic_load_n((const uint64_t*)&cmd[6], __ATOMIC_ACQUIRE) != seq &&
                    __atomic_load_n((const uint64_t*)&cmd[7], __ATOMIC_ACQUIRE) == 0;
        } else {
          (void)hipGetLastError();
        }
      }
      if (give_up) {
        if (__atomic_load_n((const uint64_t*)&cmd[6], __ATOMIC_ACQUIRE) == seq) break;
        __atomic_store_n((uint64_t*)&cmd[0], 0, __ATOMIC_RELEASE);  // the doorbell is withdrawn: nobody may act on it any more
        c->agent_running = false;
        return false;
      }
    }
    bo.pause();
  }
  c->p2p_agent_served++;
  return true;
}

// a lingering agent is told to go and waited for (finalize; nothing else needs it: it goes by itself)
void p2p_agent_stop(xmpi_comm* c) {
  std::lock_guard<std::mutex> g(c->agent_mu);
  if (!c->agent_running || !c->p2p_cmd) return;
  volatile uint64_t* cmd = c->p2p_cmd;
  const uint64_t seq = ++c->agent_seq;
  __atomic_store_n((uint64_t*)&cmd[3], seq << 32, __ATOMIC_RELEASE);  // (polled by the agent while it is written: atomic, like agent_submit's)
  __atomic_store_n((uint64_t*)&cmd[0], 2ull | (seq << 24), __ATOMIC_RELEASE);
  Backoff bo;
  const double t0 = now_seconds();
  while (__atomic_load_n((const uint64_t*)&cmd[7], __ATOMIC_ACQUIRE) == 0 && now_seconds() - t0 < 5.0) bo.pause();
  c->agent_running = false;
}

// ---- the LL agent: a one-block kernel that lingers behind a blocking small collective (ll.hip ll_agent_kernel) -------------
// The command record and the conversation are the receive agent's (above); the caller holds coll_mu (dsync.cpp dsync_ll), so
// there is one command at a time by construction.  1: the agent ran the collective and everything it wrote is visible;
// 0: not taken (no agent, broken, job aborted) -- nothing has happened that a launched LL kernel of the same epoch would
// not repeat line for line; -1: the agent took it and never answered within the no-progress limit (+ 5 s) -- the collective has
// FAILED (the job's abort flag is set): the caller must not launch anything for this epoch beside an agent that may still run.
int agent_submit_ll(xmpi_comm* c, const void* send, void* recv, size_t bytes, int ll_coll, int root, int dtype, int op, bool consecutive) {
  if (c->ll_agent_us <= 0 || !c->ll_cmd_dev || !c->dsync_ok || !c->dpage || c->size < 2 || c->size > kDsyncRanks || bytes == 0 ||
      bytes > kLLMaxPayload)
    return 0;
  volatile uint64_t* cmd = c->ll_cmd;
  if (c->ll_agent_running && __atomic_load_n((const uint64_t*)&cmd[7], __ATOMIC_ACQUIRE) != 0) c->ll_agent_running = false;  // it said it went: started again below
  const uint64_t seq = ++c->ll_agent_seq;
  const uint64_t meta = (uint64_t)(ll_coll & 3) | ((uint64_t)(root & 15) << kAgentLLRootShift) |
                        ((uint64_t)(dtype & 7) << kAgentLLDtypeShift) | ((uint64_t)(op & 3) << kAgentLLOpShift) |
                        ((uint64_t)(consecutive ? 1 : 0) << kAgentLLConsecutiveShift);
  // (the agent polls all four words while they are written and takes them only when [0] and [3] both carry this number: every
  // word is an atomic store, so that the words it reads early are merely old, never torn)
  __atomic_store_n((uint64_t*)&cmd[1], (uint64_t)(uintptr_t)send, __ATOMIC_RELAXED);
  __atomic_store_n((uint64_t*)&cmd[2], (uint64_t)(uintptr_t)recv, __ATOMIC_RELAXED);
  __atomic_store_n((uint64_t*)&cmd[3], meta | (seq << 32), __ATOMIC_RELEASE);
  __atomic_store_n((uint64_t*)&cmd[0], 1ull | ((uint64_t)bytes << 2) | (seq << 24), __ATOMIC_RELEASE);  // the doorbell last
  auto launch = [&]() -> bool {
    if (!c->ll_agent_stream) c->ll_agent_stream = stream_acquire(c->device);
    if (!c->ll_agent_stream) return false;
    __atomic_store_n((uint64_t*)&cmd[7], 0, __ATOMIC_RELEASE);
    LLAgentArgs a;
    memset(&a, 0, sizeof a);
    a.cmd = c->ll_cmd_dev;
    a.seq0 = seq;
    a.patience_ticks = (uint64_t)c->ll_agent_us * 100;  // wall_clock64 runs at 100 MHz
    for (int p = 0; p < c->size; p++) a.ll.page[p] = c->peer_page[p];
    a.ll.me = c->rank;
    a.ll.n = c->size;
    a.ll.epoch_floor = c->dsync_base;
    a.ll.host_epoch = c->dsync_status_dev ? (uint64_t*)(c->dsync_status_dev + 2) : nullptr;
    a.ll.abort_word = c->dsync_abort_dev;
    a.ll.status = c->dsync_status_dev;
    a.ll.spin_limit = c->timeout_s > 0 ? (uint64_t)c->timeout_s * 100000000ull : 0;
    if (launch_ll_agent(a, c->ll_agent_stream) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    c->ll_agent_running = true;
    c->ll_agent_launches++;
    return true;
  };
  auto withdraw = [&]() {
    __atomic_store_n((uint64_t*)&cmd[0], 0, __ATOMIC_RELEASE);  // nobody may act on it any more
    --c->ll_agent_seq;
    return 0;
  };
  if (!c->ll_agent_running && !launch()) return withdraw();
  Backoff bo;
  bo.idle = [](void* p) { dsync_service((xmpi_comm*)p); };  // (a peer may be waiting for this rank to map a buffer before it can start)
  bo.idle_arg = c;
  const double t0 = now_seconds();
  for (unsigned spins = 1;; spins++) {
    if (__atomic_load_n((const uint64_t*)&cmd[6], __ATOMIC_ACQUIRE) == seq) break;  // done
    if (__atomic_load_n((const uint64_t*)&cmd[7], __ATOMIC_ACQUIRE) != 0) {  // the agent had gone (its patience ran out)
      if (__atomic_load_n((const uint64_t*)&cmd[6], __ATOMIC_ACQUIRE) == seq) break;
      c->ll_agent_running = false;
      if (!launch()) return withdraw();
    }
    // Off the fast path, now and then: an agent that faulted, a queue that was torn down -- the stream is idle only when the
    // agent has ended; if it ended without serving this command and without saying "gone", it is broken and the launched kernel
    // takes over.  A dead PEER is the agent's own business (ll_gather gives up within the no-progress limit and says why);
    // this thread allows it that limit and a little more.
    if ((spins & 0xfff) == 0) {
      const hipError_t e = hipStreamQuery(c->ll_agent_stream);
      (void)hipGetLastError();
      if (e != hipErrorNotReady && __atomic_load_n((const uint64_t*)&cmd[6], __ATOMIC_ACQUIRE) != seq &&
          __atomic_load_n((const uint64_t*)&cmd[7], __ATOMIC_ACQUIRE) == 0) {
        // the stream is idle: the agent has ENDED, without serving this command and without saying "gone" -- broken.  Nothing of it
        // runs any more, so the launched kernel may take the epoch over.
        c->ll_agent_running = false;
        __atomic_store_n((uint64_t*)&cmd[0], 0, __ATOMIC_RELEASE);
        return 0;  // (the number stays consumed: a relaunch starts at the next one)
      }
      if (c->timeout_s > 0 && now_seconds() - t0 > (double)c->timeout_s + 5.0) {
        if (__atomic_load_n((const uint64_t*)&cmd[6], __ATOMIC_ACQUIRE) == seq) break;
        // The agent may still be INSIDE the collective (its own clock should have cut its waits short by now): a second kernel for
        // the same epoch beside it would store into the same slots and answer the same record.  The collective has failed: the
        // job's abort flag makes the agent's waits end, and nothing reuses the record before it has gone or its stream is idle.
        c->ctl->set_abort(XMPI_ERR_TIMEOUT);
        __atomic_store_n((uint64_t*)&cmd[0], 0, __ATOMIC_RELEASE);
        const double t1 = now_seconds();
        while (__atomic_load_n((const uint64_t*)&cmd[7], __ATOMIC_ACQUIRE) == 0 && hipStreamQuery(c->ll_agent_stream) == hipErrorNotReady &&
               now_seconds() - t1 < 10.0)
          bo.pause();
        (void)hipGetLastError();
        c->ll_agent_running = false;
        return -1;
      }
    }
    bo.pause();
  }
  return 1;
}

// a lingering LL agent is told to go and waited for (finalize; nothing else needs it: it goes by itself)
void ll_agent_stop(xmpi_comm* c) {
  if (!c->ll_agent_running || !c->ll_cmd) return;
  volatile uint64_t* cmd = c->ll_cmd;
  const uint64_t seq = ++c->ll_agent_seq;
  __atomic_store_n((uint64_t*)&cmd[3], seq << 32, __ATOMIC_RELEASE);
  __atomic_store_n((uint64_t*)&cmd[0], 2ull | (seq << 24), __ATOMIC_RELEASE);
  Backoff bo;
  const double t0 = now_seconds();
  while (__atomic_load_n((const uint64_t*)&cmd[7], __ATOMIC_ACQUIRE) == 0 && now_seconds() - t0 < 5.0) bo.pause();
  c->ll_agent_running = false;
}

// wait_ack = false is the reference author's intended Send (commented out at mpi.go:132-152): return
// once the payload has left the caller's buffer; p2p_wait() later collects the receiver's confirmation
// and frees the {dest, tag} pair.
int p2p_send(xmpi_comm* c, const void* buf, size_t bytes, int dtype, int dest, int tag, bool wait_ack) {
  // {dest,tag} unique among concurrent sends (mpi.go:121-125; the reference panics at
  // network.go:469, here it is an error code the Go shim turns into mpi.TagExists)
  RoctxRange range("xmpi:send dest=%d tag=%d bytes=%zu", dest, tag, bytes);
  TagGuard tg(c, &c->send_tags, dest, tag);
  if (!tg.held) {
    set_last_error("tag " + std::to_string(tag) + " already in use sending to " + std::to_string(dest));
    return XMPI_ERR_TAG_EXISTS;
  }
  StreamLease lease(c);
  if (!lease.s) return hip_fail(hipGetLastError(), "hipStreamCreate", __FILE__, __LINE__);
  const bool dev_src = bytes == 0 || heap_owns(buf) || is_device_pointer(buf);  // (the arena lookup is the cheap answer)
  const double t0 = now_seconds();
  Backoff bo;
  arm(bo, c);

  // claim a mail entry of the ordered pair (me -> dest)
  MailEntry* m = nullptr;
  int entry = -1;
  while (!m) {
    for (int e = 0; e < kMailEntries && !m; e++) {
      MailEntry* cand = c->ctl->mail(c->rank, dest, e);
      uint32_t expect = MAIL_FREE;
      if (cand->state.compare_exchange_strong(expect, MAIL_CLAIMED, std::memory_order_acq_rel)) {
        m = cand;
        entry = e;
      }
    }
    if (m) break;
    if (c->ctl->aborted()) return XMPI_ERR_PEER;
    if (timed_out(c, t0)) {
      set_last_error("send: no free mail entry towards rank " + std::to_string(dest));
      return XMPI_ERR_TIMEOUT;
    }
    bo.pause();
  }
  m->tag = tag;
  m->dtype = dtype;
  m->bytes = bytes;
  m->status.store(XMPI_OK, std::memory_order_relaxed);
  // A payload in HOST memory -- what the reference's callers pass: Go slices (network.go:518, bounce.go:96) -- travels
  // through the entry's host lane in the shared segment: the first pieces are in place before the message is posted, the
  // rest follows as the receiver drains.  (Staging it through both GPUs' HBM took 3 PCIe crossings, a hipMalloc and two
  // events per message: 40 us one way for 8 bytes -- the reference's loopback TCP takes 8.)
  const size_t lane_bytes = c->ctl->host_lane_bytes();
  if (!dev_src && bytes > 0 && lane_bytes > 0) {
    const size_t piece = lane_bytes / kHostLaneSlots;
    const uint64_t np = (bytes + piece - 1) / piece;
    char* lane = c->ctl->host_lane(c->rank, dest, entry);
    uint64_t filled = 0;
    auto fill = [&]() {
      const size_t off = (size_t)filled * piece;
      memcpy(lane + (size_t)(filled % kHostLaneSlots) * piece, (const char*)buf + off, std::min(piece, bytes - off));
      filled++;
    };
    while (filled < np && filled < (uint64_t)kHostLaneSlots) fill();
    m->pipe.head.v.store(filled, std::memory_order_relaxed);
    m->direct.store(DIRECT_HOST, std::memory_order_relaxed);
    m->state.store(MAIL_POSTED, std::memory_order_release);
    int rc = XMPI_OK;
    double tp = now_seconds();
    bo.n = 0;
    while (filled < np) {
      if (filled - m->pipe.tail.v.load(std::memory_order_acquire) < (uint64_t)kHostLaneSlots) {
        fill();
        m->pipe.head.v.store(filled, std::memory_order_release);
        tp = now_seconds();
        bo.n = 0;
        continue;
      }
      if (m->state.load(std::memory_order_acquire) == MAIL_DONE) break;  // the receiver gave up (truncate ...)
      if (c->ctl->aborted()) {
        rc = XMPI_ERR_PEER;
        break;
      }
      if (c->timeout_s > 0 && now_seconds() - tp > (double)c->timeout_s) {
        if (withdraw(m)) {
          set_last_error("send to rank " + std::to_string(dest) + " tag " + std::to_string(tag) + ": no matching receive");
          return XMPI_ERR_TIMEOUT;
        }
        tp = now_seconds();
      }
      bo.pause();
    }
    if (rc != XMPI_OK) {
      c->ctl->set_abort(rc);
      return rc;
    }
    if (!wait_ack) {
      std::lock_guard<std::mutex> g(c->p2p_mu);
      c->pending_sends[{dest, tag}] = m;
      tg.held = false;
      return XMPI_OK;
    }
    return await_ack(c, m, dest, tag);
  }
  // A job that voted its windows away (xmpi_init: a rank could not map one) has no mail slots: a payload the receiver cannot
  // pull as it lies -- unregistered device memory, a host slice with the host lanes off -- first goes into a registered block of
  // this rank (one local copy), and THAT is offered.  xmpi_send_nowait needs the slots: not in this mode.
  struct StandIn {
    void* p = nullptr;
    ~StandIn() {
      if (p) (void)heap_free(p);
    }
  } standin;
  bool dev_now = dev_src;
  if (!c->windows_ok && bytes > 0) {
    BufRef probe;
    if (!wait_ack) {
      (void)withdraw_claimed(m);
      set_last_error("send_nowait: this job runs without windows (xmpi_degraded): no mail slots to leave the payload in");
      return XMPI_ERR_UNSUPPORTED;
    }
    if (!(dev_src && zc_export(c, buf, bytes, &probe))) {
      standin.p = heap_alloc(c->device, bytes);
      if (!standin.p || hipMemcpyAsync(standin.p, buf, bytes, hipMemcpyDefault, lease.s) != hipSuccess || hipStreamSynchronize(lease.s) != hipSuccess) {
        (void)hipGetLastError();
        (void)withdraw_claimed(m);
        set_last_error("send: no registered block for the payload (this job runs without windows)");
        return XMPI_ERR_NOMEM;
      }
      buf = standin.p;
      dev_now = true;
    }
  }
  // A registered source (xmpi_malloc / xmpi_register) is offered to the receiver, which then copies
  // straight out of it: one pass over the data and one xGMI crossing instead of slot-in + slot-out.
  // (A job without windows has no mail slots: the offer is the ONLY way a device payload travels there, whatever p2p_direct_bytes
  // says -- "always through the mail slots" (< 0) or a threshold above this message would leave the receiver waiting for slots
  // nobody fills, for ever by default.)
  const bool must_offer = !c->windows_ok && bytes > 0;
  const bool offered = wait_ack && dev_now && (must_offer || (c->p2p_direct_bytes >= 0 && bytes >= (size_t)std::max<long>(1, c->p2p_direct_bytes))) &&
                       zc_export(c, buf, bytes, &m->src);
  m->direct.store(offered ? DIRECT_OFFERED : DIRECT_NONE, std::memory_order_relaxed);
  m->state.store(MAIL_POSTED, std::memory_order_release);

  int rc = XMPI_OK;
  double tp = now_seconds();
  if (offered) {  // rendezvous first: the matching receive decides how the payload travels
    bo.n = 0;
    while (m->direct.load(std::memory_order_acquire) == DIRECT_OFFERED &&
           m->state.load(std::memory_order_acquire) != MAIL_DONE) {
      if (c->ctl->aborted()) {
        rc = XMPI_ERR_PEER;
        break;
      }
      if (c->timeout_s > 0 && now_seconds() - tp > (double)c->timeout_s) {
        if (withdraw(m)) {
          set_last_error("send to rank " + std::to_string(dest) + " tag " + std::to_string(tag) + ": no matching receive");
          return XMPI_ERR_TIMEOUT;  // nothing was pushed, the entry is free again, the job goes on
        }
        tp = now_seconds();
      }
      bo.pause();
    }
  }
  const bool push = rc == XMPI_OK && m->direct.load(std::memory_order_acquire) != DIRECT_ACCEPTED &&
                    m->state.load(std::memory_order_acquire) != MAIL_DONE && (c->windows_ok || bytes == 0);
  // (no windows: a receiver that could not take the offer has answered with its error -- MAIL_DONE -- and nothing is pushed)

  const size_t slot = c->p2p_slot_bytes;
  const uint64_t npieces = push ? (bytes + slot - 1) / slot : 0;
  const uint64_t depth = (uint64_t)c->p2p_depth;
  std::deque<hipEvent_t> inflight;
  uint64_t issued = 0, published = 0;
  bool withdrawn = false;
  void* stage = nullptr;
  if (!dev_src && npieces > 0) {  // host payload: bounce through this rank's HBM
    if (hipMalloc(&stage, std::min<size_t>(bytes, depth * slot)) != hipSuccess)
      rc = hip_fail(hipGetLastError(), "hipMalloc(stage)", __FILE__, __LINE__);
  }
  bo.n = 0;
  tp = now_seconds();
  while (rc == XMPI_OK && published < npieces) {
    bool progressed = false;
    if (issued < npieces) {
      const uint64_t tail = m->pipe.tail.v.load(std::memory_order_acquire);
      if (issued - tail < depth) {
        const size_t off = (size_t)issued * slot, n = std::min(slot, bytes - off);
        char* dst = c->peer_window[dest] + c->p2p_slot_off(c->rank, entry, issued);
        hipError_t e;
        if (dev_src) {
          e = c->copy_engine == 1 ? launch_copy(dst, (const char*)buf + off, n, lease.s)
                                  : hipMemcpyAsync(dst, (const char*)buf + off, n, hipMemcpyDeviceToDevice, lease.s);
        } else {
          char* st = (char*)stage + (size_t)(issued % depth) * slot;
          e = hipMemcpyAsync(st, (const char*)buf + off, n, hipMemcpyHostToDevice, lease.s);
          if (e == hipSuccess) e = hipMemcpyAsync(dst, st, n, hipMemcpyDeviceToDevice, lease.s);
        }
        hipEvent_t ev = (e == hipSuccess) ? ev_get(c, false) : nullptr;
        if (e == hipSuccess && ev) e = hipEventRecord(ev, lease.s);
        if (e != hipSuccess || !ev) {
          rc = hip_fail(e, "p2p send copy", __FILE__, __LINE__);
          break;
        }
        inflight.push_back(ev);
        issued++;
        progressed = true;
      }
    }
    while (!inflight.empty()) {
      hipError_t e = hipEventQuery(inflight.front());
      if (e == hipErrorNotReady) {
        (void)hipGetLastError();
        break;
      }
      if (e != hipSuccess) {
        rc = hip_fail(e, "hipEventQuery", __FILE__, __LINE__);
        break;
      }
      ev_put(c, inflight.front(), false);
      inflight.pop_front();
      m->pipe.head.v.store(++published, std::memory_order_release);
      progressed = true;
    }
    if (progressed) {
      tp = now_seconds();
      bo.n = 0;
      continue;
    }
    if (m->state.load(std::memory_order_acquire) == MAIL_DONE) break;  // receiver gave up (truncate...)
    if (c->ctl->aborted()) rc = XMPI_ERR_PEER;
    else if (c->timeout_s > 0 && now_seconds() - tp > (double)c->timeout_s) {
      if (inflight.empty() && withdraw(m)) {  // the slots are full and nobody drains them: take the message back
        set_last_error("send to rank " + std::to_string(dest) + " tag " + std::to_string(tag) + ": no matching receive");
        withdrawn = true;
        break;
      }
      tp = now_seconds();
    }
    bo.pause();
  }
  if (!inflight.empty()) {
    (void)hipStreamSynchronize(lease.s);
    while (!inflight.empty()) {
      ev_put(c, inflight.front(), false);
      inflight.pop_front();
    }
  }
  if (stage) (void)hipFree(stage);
  if (withdrawn) return XMPI_ERR_TIMEOUT;
  if (rc != XMPI_OK) {
    c->ctl->set_abort(rc);  // the entry is in an unknown state: the job cannot continue
    return rc;
  }
  if (!wait_ack) {  // the payload sits in the receiver's window: the caller's buffer is free again
    std::lock_guard<std::mutex> g(c->p2p_mu);
    c->pending_sends[{dest, tag}] = m;
    tg.held = false;  // {dest, tag} stays reserved until p2p_wait
    return XMPI_OK;
  }
  return await_ack(c, m, dest, tag);
}

int p2p_wait(xmpi_comm* c, int dest, int tag) {
  MailEntry* m = nullptr;
  {
    std::lock_guard<std::mutex> g(c->p2p_mu);
    auto it = c->pending_sends.find({dest, tag});
    if (it == c->pending_sends.end()) {
      set_last_error("wait: no send to rank " + std::to_string(dest) + " with tag " + std::to_string(tag) + " is outstanding");
      return XMPI_ERR_ARG;
    }
    m = it->second;
    c->pending_sends.erase(it);
  }
  const int rc = await_ack(c, m, dest, tag);
  std::lock_guard<std::mutex> g(c->p2p_mu);
  c->send_tags.erase({dest, tag});
  return rc;
}

// Wait for a message {src, tag} to be posted and report its size without consuming it (lets a
// host-language binding size the destination the way gob's in-place decode does, network.go:597).
int p2p_probe(xmpi_comm* c, int src, int tag, size_t* bytes, int* dtype) {
  const double t0 = now_seconds();
  Backoff bo;
  arm(bo, c);
  for (;;) {
    for (int e = 0; e < kMailEntries; e++) {
      MailEntry* m = c->ctl->mail(src, c->rank, e);
      if (m->state.load(std::memory_order_acquire) == MAIL_POSTED && m->tag == tag) {
        if (bytes) *bytes = m->bytes;
        if (dtype) *dtype = m->dtype;
        return XMPI_OK;
      }
    }
    if (c->ctl->aborted()) return XMPI_ERR_PEER;
    if (timed_out(c, t0)) {
      set_last_error("probe from rank " + std::to_string(src) + " tag " + std::to_string(tag) + ": no matching send");
      return XMPI_ERR_TIMEOUT;
    }
    bo.pause();
  }
}

int p2p_recv(xmpi_comm* c, void* buf, size_t cap_bytes, int dtype, int src, int tag, size_t* got_bytes) {
  RoctxRange range("xmpi:recv src=%d tag=%d capacity=%zu", src, tag, cap_bytes);
  TagGuard tg(c, &c->recv_tags, src, tag);
  if (!tg.held) {
    set_last_error("tag " + std::to_string(tag) + " already in use receiving from " + std::to_string(src));
    return XMPI_ERR_TAG_EXISTS;
  }
  StreamLease lease(c);
  if (!lease.s) return hip_fail(hipGetLastError(), "hipStreamCreate", __FILE__, __LINE__);
  double t0 = now_seconds();
  Backoff bo;
  arm(bo, c);
  MailEntry* m = nullptr;
  int entry = -1;
  while (!m) {
    for (int e = 0; e < kMailEntries && !m; e++) {
      MailEntry* cand = c->ctl->mail(src, c->rank, e);
      if (cand->state.load(std::memory_order_acquire) == MAIL_POSTED && cand->tag == tag) {
        uint32_t expect = MAIL_POSTED;
        if (cand->state.compare_exchange_strong(expect, MAIL_MATCHED, std::memory_order_acq_rel)) {
          if (cand->tag != tag) {  // withdrawn and re-posted with another tag between the look and the claim
            cand->state.store(MAIL_POSTED, std::memory_order_release);
            continue;
          }
          m = cand;
          entry = e;
        }
      }
    }
    if (m) break;
    if (c->ctl->aborted()) return XMPI_ERR_PEER;
    if (timed_out(c, t0)) {
      set_last_error("receive from rank " + std::to_string(src) + " tag " + std::to_string(tag) + ": no matching send");
      return XMPI_ERR_TIMEOUT;
    }
    bo.pause();
  }
  const size_t bytes = m->bytes;
  if (got_bytes) *got_bytes = bytes;
  int verdict = XMPI_OK;
  if (m->dtype != dtype) {
    set_last_error("receive: dtype differs from the sender's");
    verdict = XMPI_ERR_ARG;
  } else if (bytes > cap_bytes) {
    set_last_error("receive: message of " + std::to_string(bytes) + " bytes does not fit " + std::to_string(cap_bytes));
    verdict = XMPI_ERR_TRUNCATE;
  }
  if (verdict != XMPI_OK) {
    m->status.store(verdict, std::memory_order_release);
    m->state.store(MAIL_DONE, std::memory_order_release);
    return verdict;
  }
  const bool dev_dst = bytes == 0 || heap_owns(buf) || is_device_pointer(buf);
  int rc = XMPI_OK;
  double tp = now_seconds();
  if (m->direct.load(std::memory_order_acquire) == DIRECT_HOST) {
    // the payload comes through the entry's host lane (p2p_send): a host destination takes it with memcpy, piece by
    // piece; a device destination by DMA out of the (registered) lane, as many pieces at a time as have arrived
    const size_t piece = c->ctl->host_lane_bytes() / kHostLaneSlots;
    const uint64_t np = (bytes + piece - 1) / piece;
    const char* lane = c->ctl->host_lane(src, c->rank, entry);
    uint64_t taken = 0;
    bo.n = 0;
    auto stalled = [&]() {
      if (c->ctl->aborted()) rc = XMPI_ERR_PEER;
      else if (c->timeout_s > 0 && now_seconds() - tp > (double)c->timeout_s) {
        set_last_error("receive: sender stalled");
        rc = XMPI_ERR_TIMEOUT;
      }
      bo.pause();
    };
    if (dev_dst && c->lanes_dev_ok && c->p2p_kernel_ack && np <= (uint64_t)kHostLaneSlots) {
      // a message that fits the ring lies there in one piece (it was complete before it was posted): the receive agent pulls it
      // out of the pinned lane like it pulls a message out of a peer's HBM, and writes the ack -- no DMA call, no event
      while (rc == XMPI_OK && m->pipe.head.v.load(std::memory_order_acquire) < np) stalled();
      if (rc == XMPI_OK && agent_submit(c, buf, c->ctl_dev + (lane - (const char*)c->ctl->base()), bytes, m)) {
        __atomic_fetch_add(&c->p2p_lane_count, 1, __ATOMIC_RELAXED);
        return XMPI_OK;
      }
    }
    // Longer messages stream through the ring.  A host destination takes the pieces with memcpy.  A device destination has a
    // kernel pull every run of pieces that has arrived (the GPU reads the pinned lane itself; its last block writes a completion
    // word this thread polls) -- one DMA call + event per 64 KiB piece took twice as long (r03 session 14: 1 MiB 357 us per round
    // trip instead of 240), and so does the runtime's staged copy when the lane could not be pinned (the fallback below).
    const bool pull = dev_dst && c->lanes_dev_ok && c->p2p_kernel_ack && c->p2p_done_dev && c->p2p_tickets;
    uint64_t pending = 0, pending_id = 0;
    volatile uint64_t* pending_word = nullptr;
    while (rc == XMPI_OK && taken < np) {
      bool progressed = false;
      const uint64_t head = m->pipe.head.v.load(std::memory_order_acquire);
      if (head > taken && !dev_dst) {
        for (uint64_t k = taken; k < head; k++) {
          const size_t off = (size_t)k * piece;
          memcpy((char*)buf + off, lane + (size_t)(k % kHostLaneSlots) * piece, std::min(piece, bytes - off));
        }
        taken = head;
        m->pipe.tail.v.store(taken, std::memory_order_release);
        progressed = true;
      } else if (head > taken && pull && !pending) {
        const uint64_t first = taken % kHostLaneSlots, run = std::min<uint64_t>(head - taken, kHostLaneSlots - first);  // contiguous in the lane
        const size_t off = (size_t)taken * piece;
        pending_id = c->p2p_pull_next.fetch_add(1, std::memory_order_relaxed) + 1;
        const int slot = (int)(pending_id % (uint64_t)xmpi_comm::kP2PDoneSlots);
        pending_word = c->p2p_done + 4 * (xmpi_comm::kP2PDoneSlots + slot);
        P2PPullArgs pa;
        memset(&pa, 0, sizeof pa);
        pa.dst = (char*)buf + off;
        pa.src = c->ctl_dev + ((lane + (size_t)first * piece) - (const char*)c->ctl->base());
        pa.bytes = std::min((size_t)run * piece, bytes - off);
        pa.ticket = c->p2p_tickets + slot;
        pa.host_done = c->p2p_done_dev + 4 * (xmpi_comm::kP2PDoneSlots + slot);
        pa.done_value = pending_id;
        const long gx = std::max<long>(1, std::min<long>(16, (long)((pa.bytes + 16383) >> 14)));
        if (launch_p2p_pull(pa, (int)gx, lease.s) != hipSuccess) {
          rc = hip_fail(hipGetLastError(), "p2p pull out of the host lane", __FILE__, __LINE__);
          break;
        }
        pending = run;
        progressed = true;
      } else if (head > taken && dev_dst && !pull) {
        for (uint64_t k = taken; k < head && rc == XMPI_OK; k++) {
          const size_t off = (size_t)k * piece;
          if (hipMemcpyAsync((char*)buf + off, lane + (size_t)(k % kHostLaneSlots) * piece, std::min(piece, bytes - off), hipMemcpyHostToDevice,
                             lease.s) != hipSuccess)
            rc = hip_fail(hipGetLastError(), "p2p receive from the host lane", __FILE__, __LINE__);
        }
        if (rc == XMPI_OK && hipStreamSynchronize(lease.s) != hipSuccess) rc = hip_fail(hipGetLastError(), "hipStreamSynchronize", __FILE__, __LINE__);
        taken = head;
        m->pipe.tail.v.store(taken, std::memory_order_release);
        progressed = true;
      }
      if (pending && __atomic_load_n((const uint64_t*)pending_word, __ATOMIC_ACQUIRE) == pending_id) {
        taken += pending;
        pending = 0;
        m->pipe.tail.v.store(taken, std::memory_order_release);
        progressed = true;
      }
      if (progressed) {
        tp = now_seconds();
        bo.n = 0;
        continue;
      }
      stalled();
    }
    if (pending) (void)hipStreamSynchronize(lease.s);  // (an error above: the kernel in flight must not outlive the call)
    if (rc != XMPI_OK) {
      c->ctl->set_abort(rc);
      return rc;
    }
    __atomic_fetch_add(&c->p2p_lane_count, 1, __ATOMIC_RELAXED);
    m->status.store(XMPI_OK, std::memory_order_release);
    m->state.store(MAIL_DONE, std::memory_order_release);  // the ack (network.go:616-624)
    return XMPI_OK;
  }
  if (m->direct.load(std::memory_order_acquire) == DIRECT_OFFERED) {
    // the sender's buffer is registered: copy straight out of it (mapped once per allocation)
    void* from = nullptr;
    if (!dev_dst && zc_import(c, src, m->src, &from)) {
      // ... into HOST memory (the caller handed a slice): one copy device -> host, no slots in between
      m->direct.store(DIRECT_ACCEPTED, std::memory_order_release);
      if (c->p2p_kernel_ack && bytes <= kP2PBounceBytes) {
        // short: the receive agent copies into a pinned block and acks; the slice gets it with memcpy (the runtime's copy into
        // pageable memory is a staged, synchronous affair of 20 us)
        std::lock_guard<std::mutex> g(c->p2p_bounce_mu);
        if (!c->p2p_bounce) {
          void* p = nullptr;
          void* dev = nullptr;
          if (hipHostMalloc(&p, kP2PBounceBytes, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&dev, p, 0) == hipSuccess) {
            c->p2p_bounce = (char*)p;
            c->p2p_bounce_dev = (char*)dev;
          } else {
            (void)hipGetLastError();
            if (p) (void)hipHostFree(p);
          }
        }
        if (c->p2p_bounce_dev && agent_submit(c, c->p2p_bounce_dev, from, bytes, m)) {
          memcpy(buf, c->p2p_bounce, bytes);
          __atomic_fetch_add(&c->p2p_direct_count, 1, __ATOMIC_RELAXED);
          return XMPI_OK;
        }
      }
      if (hipMemcpyAsync(buf, from, bytes, hipMemcpyDeviceToHost, lease.s) != hipSuccess || hipStreamSynchronize(lease.s) != hipSuccess) {
        rc = hip_fail(hipGetLastError(), "p2p direct copy to the host", __FILE__, __LINE__);
        c->ctl->set_abort(rc);
        return rc;
      }
      __atomic_fetch_add(&c->p2p_direct_count, 1, __ATOMIC_RELAXED);
      m->status.store(XMPI_OK, std::memory_order_release);
      m->state.store(MAIL_DONE, std::memory_order_release);
      return XMPI_OK;
    }
    if (dev_dst && zc_import(c, src, m->src, &from)) {
      m->direct.store(DIRECT_ACCEPTED, std::memory_order_release);
      if (c->p2p_kernel_ack && agent_submit(c, buf, from, bytes, m)) {  // the lingering agent took it: no launch at all
        __atomic_fetch_add(&c->p2p_direct_count, 1, __ATOMIC_RELAXED);
        return XMPI_OK;
      }
      if (c->p2p_kernel_ack && c->ctl_dev && c->p2p_done_dev && c->p2p_tickets) {
        // ONE kernel copies and acks: its last block writes DONE into the message's mail entry (the sender's host thread
        // polls it: the ack of network.go:616-624, without this rank's host in between) and the completion word this
        // thread polls.  Nothing in it waits for anybody.
        const uint64_t id = c->p2p_pull_next.fetch_add(1, std::memory_order_relaxed) + 1;
        const int slot = (int)(id % (uint64_t)xmpi_comm::kP2PDoneSlots);
        volatile uint64_t* done = c->p2p_done + 4 * (xmpi_comm::kP2PDoneSlots + slot);
        P2PPullArgs pa;
        memset(&pa, 0, sizeof pa);
        pa.dst = buf;
        pa.src = from;
        pa.bytes = bytes;
        pa.ticket = c->p2p_tickets + slot;
        pa.host_done = c->p2p_done_dev + 4 * (xmpi_comm::kP2PDoneSlots + slot);
        pa.done_value = id;
        char* mdev = c->ctl_dev + ((char*)m - (char*)c->ctl->base());
        pa.mail_state = (uint32_t*)(mdev + ((char*)&m->state - (char*)m));
        pa.mail_status = (int32_t*)(mdev + ((char*)&m->status - (char*)m));
        pa.mail_done_value = MAIL_DONE;
        long gx = (long)((bytes + 16383) >> 14);  // a 16 KiB tile per block and pass; how many blocks: p2p_pull_cap
        gx = std::max<long>(1, std::min<long>(gx, p2p_pull_cap(c, src, bytes)));
        hipError_t e = launch_p2p_pull(pa, (int)gx, lease.s);
        if (e != hipSuccess) rc = hip_fail(e, "p2p pull kernel", __FILE__, __LINE__);
        bo.n = 0;
        while (rc == XMPI_OK && __atomic_load_n((const uint64_t*)done, __ATOMIC_ACQUIRE) != id) {
          if ((bo.n & 1023u) == 1023u && c->ctl->aborted()) rc = XMPI_ERR_PEER;
          bo.pause();
        }
        if (rc != XMPI_OK) {
          c->ctl->set_abort(rc);
          return rc;
        }
        __atomic_fetch_add(&c->p2p_direct_count, 1, __ATOMIC_RELAXED);
        return XMPI_OK;
      }
      hipError_t e = c->copy_engine == 1 ? launch_copy(buf, from, bytes, lease.s)
                                         : hipMemcpyAsync(buf, from, bytes, hipMemcpyDeviceToDevice, lease.s);
      hipEvent_t ev = (e == hipSuccess) ? ev_get(c, false) : nullptr;
      if (e == hipSuccess && ev) e = hipEventRecord(ev, lease.s);
      if (e != hipSuccess || !ev) rc = hip_fail(e, "p2p direct copy", __FILE__, __LINE__);
      bo.n = 0;
      while (rc == XMPI_OK) {
        e = hipEventQuery(ev);
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) {
          rc = hip_fail(e, "hipEventQuery", __FILE__, __LINE__);
          break;
        }
        (void)hipGetLastError();
        if (c->ctl->aborted()) rc = XMPI_ERR_PEER;
        bo.pause();
      }
      if (ev) ev_put(c, ev, false);
      if (rc != XMPI_OK) {
        c->ctl->set_abort(rc);
        return rc;
      }
      __atomic_fetch_add(&c->p2p_direct_count, 1, __ATOMIC_RELAXED);
      m->status.store(XMPI_OK, std::memory_order_release);
      m->state.store(MAIL_DONE, std::memory_order_release);  // the ack (network.go:616-624)
      return XMPI_OK;
    }
    if (!c->windows_ok && bytes > 0) {  // ... which this job does not have: both sides get the error, the job goes on
      set_last_error("receive: the sender's buffer cannot be mapped here and this job runs without windows (xmpi_degraded)");
      m->status.store(XMPI_ERR_UNSUPPORTED, std::memory_order_release);
      m->state.store(MAIL_DONE, std::memory_order_release);
      return XMPI_ERR_UNSUPPORTED;
    }
    m->direct.store(DIRECT_DECLINED, std::memory_order_release);  // host destination / not mappable: use the slots
  }
  if (!c->windows_ok && bytes > 0) {  // a message that was not even offered (a sender of another mind): there are no slots to wait on
    set_last_error("receive: the message was posted for the mail slots, which this job does not have (xmpi_degraded)");
    m->status.store(XMPI_ERR_UNSUPPORTED, std::memory_order_release);
    m->state.store(MAIL_DONE, std::memory_order_release);
    return XMPI_ERR_UNSUPPORTED;
  }
  __atomic_fetch_add(&c->p2p_staged_count, 1, __ATOMIC_RELAXED);
  const size_t slot = c->p2p_slot_bytes;
  const uint64_t npieces = (bytes + slot - 1) / slot;
  std::deque<hipEvent_t> inflight;
  uint64_t issued = 0, drained = 0;
  tp = now_seconds();
  bo.n = 0;
  while (rc == XMPI_OK && drained < npieces) {
    bool progressed = false;
    if (issued < npieces && m->pipe.head.v.load(std::memory_order_acquire) > issued) {
      const size_t off = (size_t)issued * slot, n = std::min(slot, bytes - off);
      const char* from = c->window + c->p2p_slot_off(src, entry, issued);
      hipError_t e = (dev_dst && c->copy_engine == 1)
                         ? launch_copy((char*)buf + off, from, n, lease.s)
                         : hipMemcpyAsync((char*)buf + off, from, n,
                                          dev_dst ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, lease.s);
      hipEvent_t ev = (e == hipSuccess) ? ev_get(c, false) : nullptr;
      if (e == hipSuccess && ev) e = hipEventRecord(ev, lease.s);
      if (e != hipSuccess || !ev) {
        rc = hip_fail(e, "p2p recv copy", __FILE__, __LINE__);
        break;
      }
      inflight.push_back(ev);
      issued++;
      progressed = true;
    }
    while (!inflight.empty()) {
      hipError_t e = hipEventQuery(inflight.front());
      if (e == hipErrorNotReady) {
        (void)hipGetLastError();
        break;
      }
      if (e != hipSuccess) {
        rc = hip_fail(e, "hipEventQuery", __FILE__, __LINE__);
        break;
      }
      ev_put(c, inflight.front(), false);
      inflight.pop_front();
      m->pipe.tail.v.store(++drained, std::memory_order_release);
      progressed = true;
    }
    if (progressed) {
      tp = now_seconds();
      bo.n = 0;
      continue;
    }
    if (c->ctl->aborted()) rc = XMPI_ERR_PEER;
    else if (c->timeout_s > 0 && now_seconds() - tp > (double)c->timeout_s) {
      set_last_error("receive: sender stalled");
      rc = XMPI_ERR_TIMEOUT;
    }
    bo.pause();
  }
  if (!inflight.empty()) {
    (void)hipStreamSynchronize(lease.s);
    while (!inflight.empty()) {
      ev_put(c, inflight.front(), false);
      inflight.pop_front();
    }
  }
  if (rc != XMPI_OK) {
    c->ctl->set_abort(rc);
    return rc;
  }
  m->status.store(XMPI_OK, std::memory_order_release);
  m->state.store(MAIL_DONE, std::memory_order_release);  // the ack (network.go:616-624)
  return XMPI_OK;
}

}  // namespace xmpi
