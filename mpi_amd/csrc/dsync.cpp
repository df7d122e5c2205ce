// dsync.cpp -- device-synchronised, stream-ordered collectives.
//
// The zero-copy collectives of zcopy.cpp meet on the HOST: two or three barriers through the shared control
// block around every kernel.  Here the ranks meet on the DEVICE: each rank owns a flag page in its HBM
// (uncached, mapped by every peer), and one kernel per rank (kernels.hip `dsync_fold_kernel`) announces this
// rank's buffers to the peers, waits for theirs, moves the data straight between the user buffers over xGMI
// and exchanges "done" before it ends.  The host only enqueues that kernel on a stream -- its own for the
// blocking calls (xmpi_allreduce ...), the caller's for xmpi_*_on_stream -- and never polls a peer: what the
// reference does with a message + ack over a net.Conn per Send (network.go:562-571) is two 8-byte stores
// over a link here.
//
// What still needs the host, rarely: a kernel can only address a peer's buffer through a mapping (hipIpc)
// this process has opened.  Every rank therefore PUBLISHES the allocations it registers (control block,
// PubTable), every peer maps them the next time it is inside the library (`dsync_service`, also called from
// every wait loop), writes {slot -> mapping} into the translation table its kernels read, and acknowledges.
// A rank uses a buffer in a device-synchronised collective only after all peers acknowledged the allocation
// it lives in: in steady state (buffers allocated once, reused) nothing of this runs.
//
// Applies when no two ranks share a (process, GPU) pair -- the production layout, one process per MI355X.
// Ranks hosted by threads of one process on one GPU (bench.py on a single-GPU box) keep the host-synchronised
// path: they share one in-order stream, and a kernel that waits for a kernel queued behind it never ends.
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <cstring>

#include "comm.h"
#include "kernels.h"
#include "sched_steps.h"

namespace xmpi {

static_assert(kDsyncRanks <= kMaxRanks, "the device side serves a subset of the jobs the host side admits");
static_assert(sizeof(DsyncPage) <= kStepOff, "flag page");
constexpr size_t kPageBytes = kDsyncPageBytes;  // the page, the step flags of the stepped kernels, the Send / Receive boxes

namespace {

double wait_limit(const xmpi_comm* c) { return c->timeout_s > 0 ? (double)c->timeout_s : 1e18; }

void idle_hook(void* arg) { dsync_service((xmpi_comm*)arg); }

}  // namespace

// Called by xmpi_init before this rank's RankInfo is published (state 2).
int dsync_prepare(xmpi_comm* c) {
  RankInfo* me = c->ctl->info(c->rank);
  me->flag_addr = 0;
  if (c->size < 2 || !c->dsync || c->size > kDsyncRanks) return XMPI_OK;  // (more ranks than the device side is sized for: they meet on the host)
  // a rank without a flag page does not fail the job: every rank sees flag_addr == 0 and keeps to the host-synchronised path;
  // it says why (xmpi_degraded)
  auto none = [&](const char* what, hipError_t e) {
    (void)hipGetLastError();
    if (!me->maps_why[0]) snprintf(me->maps_why, sizeof me->maps_why, "rank %d: %s: %s", c->rank, what, hipGetErrorString(e));
    return XMPI_OK;
  };
  // uncached HBM: the page is polled by this GPU and written by the others; it must never sit in an L2.
  // From the per-process pool (exported memory outlives communicators -- api.cpp), and NOT cleared when it is
  // re-used: clearing is GPU work on a page seven other processes have mapped, and in a crowded GPU (eight ranks
  // plus a test runner with a context of its own) that one 64 KiB fill took 10-50 s.  Instead the epochs of the new
  // communicator start above everything an earlier one may have left in any rank's page (flag_epoch, dsync_connect).
  bool fresh = false;
  uint64_t last_epoch = 0;
  void* page = pool_acquire(c->device, kPageBytes, 1, &fresh, &last_epoch);
  if (!page) return none("hipExtMallocWithFlags(uncached flag page)", hipGetLastError());
  if (fresh) {
    hipError_t e = hipMemsetAsync(page, 0, kPageBytes, c->local_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->local_stream);
    if (e != hipSuccess) {
      pool_release(page, last_epoch);
      return none("hipMemset(flag page)", e);
    }
  }
  me->flag_epoch = last_epoch;
  hipIpcMemHandle_t h;
  const hipError_t he = pool_handle(page, &h);
  if (he != hipSuccess) {
    pool_release(page, last_epoch);
    return none("hipIpcGetMemHandle(uncached flag page)", he);
  }
  c->dpage = (DsyncPage*)page;
  static_assert(sizeof h <= sizeof me->flag_handle, "ipc handle size");
  memcpy(me->flag_handle, &h, sizeof h);
  me->flag_addr = (uint64_t)(uintptr_t)page;
  return XMPI_OK;
}

// Called by xmpi_init once every rank's RankInfo is visible.  Maps the peers' flag pages, then the job VOTES: every rank
// publishes what it could map (the peers' windows: xmpi_init; their flag pages: here), all meet at a barrier, and every rank
// reads the same answers -- the best level everybody reached:
//   every flag page mapped by everybody    the ranks meet on the device (this file)
//   otherwise                              they meet on the host (zcopy.cpp) and the staged step tables serve the rest
//   some window not mapped                 no staged step tables and no mail slots: every collective takes the device-synchronised
//                                          path, Send / Receive work out of registered buffers and through the host lanes
//   neither                                xmpi_init fails -- on EVERY rank, at once, with the reason (the reference's Init returns an
//                                          error only when the mesh cannot be built, network.go:53-65)
// xmpi_get_param("degraded") / xmpi_degraded() say which and why.
int dsync_connect(xmpi_comm* c, double timeout_s) {
  c->dsync_ok = false;
  if (c->size < 2) return XMPI_OK;
  const int N = c->size, mypid = (int)getpid();
  RankInfo* me = c->ctl->info(c->rank);
  bool usable = N <= kDsyncRanks;
  int sharers = 0, sharers_job = 1;
  for (int p = 0; p < N; p++) {
    const RankInfo* a = c->ctl->info(p);
    if (a->flag_addr == 0) usable = false;
    if (strncmp(a->busid, c->ctl->info(c->rank)->busid, sizeof a->busid) == 0) sharers++;
    int on_its_gpu = 0;  // the most crowded GPU of the job: what every rank must read alike (the stepped kernels' shape)
    for (int q = 0; q < N; q++)
      if (strncmp(a->busid, c->ctl->info(q)->busid, sizeof a->busid) == 0) on_its_gpu++;
    sharers_job = std::max(sharers_job, on_its_gpu);
    for (int q = p + 1; q < N; q++) {
      const RankInfo* b = c->ctl->info(q);
      if (a->pid == b->pid && a->device == b->device) usable = false;  // two ranks on one stream: see the header
    }
  }
  // (pinned words the kernels write: word 0 first failure of a kernel, bytes 8..15 epoch of the last kernel that ended, 16..23 the
  // blocking caller's completion word, words 6..11 XCD masks and probes, word 12 the flag self-test's answer)
  if (usable && hipHostMalloc((void**)&c->dsync_status, 64, hipHostMallocMapped) == hipSuccess) {
    memset(c->dsync_status, 0, 64);
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, c->dsync_status, 0) == hipSuccess) c->dsync_status_dev = (uint32_t*)dev;
  }
  (void)hipGetLastError();
  // the job's abort flag, readable by the GPU: a kernel that waits for a dead peer gives up
  if (c->ctl_dev) c->dsync_abort_dev = (const int32_t*)(c->ctl_dev + ((char*)&c->ctl->header()->abort_code - (char*)c->ctl->base()));
  bool mapped = usable;
  for (int p = 0; p < N && mapped; p++) {
    RankInfo* pi = c->ctl->info(p);
    if (p == c->rank) {
      c->peer_page[p] = c->dpage;
    } else if (pi->pid == mypid) {  // a thread of this process on another GPU (peer access is enabled by xmpi_init)
      c->peer_page[p] = (DsyncPage*)(uintptr_t)pi->flag_addr;
    } else {
      void* ptr = nullptr;
      hipError_t e = ipc_open_shared(pi->pid, pi->flag_addr, pi->flag_handle, &ptr);
      if (e != hipSuccess) {  // (an uncached allocation of ANOTHER device: nothing promises that the runtime opens it)
        (void)hipGetLastError();
        if (!me->maps_why[0])
          snprintf(me->maps_why, sizeof me->maps_why, "rank %d: hipIpcOpenMemHandle(flag page of rank %d): %s", c->rank, p, hipGetErrorString(e));
        mapped = false;
        break;
      }
      c->peer_page[p] = (DsyncPage*)ptr;
      c->peer_page_opened[p] = true;
    }
  }
  // A mapping that OPENS is not yet one that WORKS: an uncached allocation of another device, opened through hipIpc, has to carry
  // a peer's 8-byte store to the lane that polls it here.  Try it now, with a clock (a few seconds, inside xmpi_init), rather than
  // find out in the first collective, which by default waits for ever: every rank stores a token into every peer's page and waits
  // for theirs.  A rank whose words do not arrive votes "flags: no" below, and the job meets on the host.
  // The ranks get here after unsynchronised work (seven hipIpcOpenMemHandle calls, the registration of the control block,
  // hipMallocs): the self-test's clock is to measure whether the flags carry a store, not how far apart the ranks arrive -- so
  // they meet first.  Every rank, whatever it could map: a barrier only some ranks reach is a hang.
  {
    const int brc = c->ctl->barrier(timeout_s);
    if (brc != XMPI_OK) {
      set_last_error("xmpi_init: a peer did not reach the flag self-test");
      return brc;
    }
  }
  if (mapped && c->dsync_status_dev) {
    uint64_t token = 0;
    for (int p = 0; p < N; p++) token = std::max(token, c->ctl->info(p)->flag_epoch);
    token += 1;  // (above every token an earlier communicator left in these never-cleared pages: dsync_finalize moves the mark on)
    c->dsync_selftest_token = token;  // ... this one included, whatever the vote below says (dsync_finalize)
    const double limit_s = std::min(5.0, std::max(1.0, timeout_s / 4));
    __atomic_store_n(c->dsync_status + 12, 0u, __ATOMIC_RELAXED);
    hipError_t e = launch_flag_selftest(c->peer_page, c->rank, N, token, (uint64_t)(limit_s * 1e8), c->dsync_abort_dev, c->dsync_status_dev + 12,
                                        c->local_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->local_stream);
    const uint32_t seen = __atomic_load_n(c->dsync_status + 12, __ATOMIC_ACQUIRE), all = N >= 32 ? ~0u : (1u << N) - 1u;
    if (e != hipSuccess || (seen & all) != all) {
      (void)hipGetLastError();
      if (!me->maps_why[0]) {
        if (e != hipSuccess) snprintf(me->maps_why, sizeof me->maps_why, "rank %d: flag self-test: %s", c->rank, hipGetErrorString(e));
        else snprintf(me->maps_why, sizeof me->maps_why, "rank %d: flag words of ranks %#x never arrived here (%.0f s)", c->rank, all & ~seen, limit_s);
      }
      mapped = false;
    }
  }
  // Untuned AUTO sends messages up to ll_bytes per rank as LL lines (ll.hip).  Measured with 2 processes (each kernel has the
  // GPU it runs on to itself, as on a node with one rank per GPU): 4.9 / 6.1 / 6.4 us enqueued at 1 / 4 / 16 KiB against
  // 8.9 / 8.8 / 9.4 for the fold; eight processes time-slicing ONE GPU: 45 / 56 / 65 against 47 / 47 / 42 (their polling lanes
  // compete with each other's stores for the one memory system) -- so ranks that share a GPU would keep LL to 1 KiB, were it not for the agent:
  // With the LL agent (ll.hip ll_agent_kernel) a BLOCKING call of up to 4 KiB needs no launch at all: eight processes on one GPU,
  // blocking allreduce 7.9 / 9.2 us at 1 / 4 KiB against 38 / 45 launched (no kernel, so nothing for eight processes' queues to be
  // time-sliced over) -- worth the 12 % an ENQUEUED 4 KiB LL collective loses to the fold there.  (The choice must not depend on
  // how a rank calls -- a blocking rank and an enqueueing one have to run the same protocol -- so it is one limit for both.)
  // What this rank sees -- the ranks on ITS GPU, ITS environment -- may differ from what a peer sees (5 ranks on 2 GPUs; a variable
  // set for one rank): LL or fold is a protocol choice, a rank that folds while its peer sends lines waits for ever.  So every
  // rank publishes its choice with its vote and the job takes the smallest.
  c->dsync_sharers = std::max(1, sharers);
  c->dsync_sharers_job = sharers_job;
  long mine = c->ll_bytes;
  if (mine < 0) mine = c->dsync_sharers > 2 ? ((c->agent_ll && c->ll_agent_us > 0) ? 4096 : 1024) : 8192;
  me->ll_choice = std::min<long>(mine, (long)kLLMaxPayload);
  me->maps = (c->window_map_failed ? 0 : kMapsWindows) | (mapped ? kMapsFlags : 0);
  int rc = c->ctl->barrier(timeout_s);  // ---- the vote: everything above is published, everything below is read by all alike
  if (rc != XMPI_OK) {
    set_last_error("xmpi_init: a peer did not reach the vote on what the job can map");
    return rc;
  }
  bool all_windows = true, all_flags = usable;
  std::string why_windows, why_flags;
  bool why_flags_witness = false;
  long ll = (long)kLLMaxPayload;
  for (int p = 0; p < N; p++) {
    const RankInfo* a = c->ctl->info(p);
    ll = std::min<long>(ll, (long)a->ll_choice);
    const std::string why(a->maps_why, strnlen(a->maps_why, sizeof a->maps_why));
    if (!(a->maps & kMapsWindows)) {
      all_windows = false;
      if (why_windows.empty()) why_windows = why.empty() ? "rank " + std::to_string(p) + " could not map a peer's window" : why;
    }
    if (usable && !(a->maps & kMapsFlags)) {
      all_flags = false;
      // (a rank that could not OPEN a page is the cause; the ranks whose self-test then waited for its words in vain are its witnesses)
      const bool witness = why.find("never arrived") != std::string::npos;
      if (why_flags.empty() || (why_flags_witness && !witness && !why.empty())) {
        why_flags = why.empty() ? "rank " + std::to_string(p) + " could not map a peer's flag page" : why;
        why_flags_witness = witness;
      }
    }
    if (!usable && c->dsync && a->flag_addr == 0 && why_flags.empty() && !why.empty()) why_flags = why;  // (it has no page to offer)
  }
  c->ll_bytes = std::max<long>(0, ll);
  c->windows_ok = all_windows;
  if (!all_flags)  // what this rank did map is of no use: nobody will write there
    for (int p = 0; p < N; p++) {
      if (c->peer_page_opened[p]) ipc_close_shared(c->peer_page[p]);
      c->peer_page_opened[p] = false;
      c->peer_page[p] = nullptr;
    }
  if (!all_windows && !all_flags) {
    set_last_error("xmpi_init: the job has no transport left: " + why_windows + (why_flags.empty() ? "" : "; " + why_flags) +
                   (usable ? "" : "; and the ranks cannot meet on the device (no flag pages, or ranks sharing a stream)"));
    return XMPI_ERR_HIP;
  }
  if (!all_windows)
    c->degraded_why = "no windows (no staged step tables, no mail slots; collectives: device-synchronised only): " + why_windows;
  if (!all_flags && !why_flags.empty())
    c->degraded_why += std::string(c->degraded_why.empty() ? "" : "; ") + "the ranks meet on the host (no device-synchronised collectives): " + why_flags;
  if (!c->degraded_why.empty() && c->rank == 0) fprintf(stderr, "xmpi: degraded: %s\n", c->degraded_why.c_str());
  if (!all_flags) return XMPI_OK;
  // the translation table {peer, slot} -> {registration number, where this process mapped it}: pinned host memory
  // the kernels read; the host is its only writer and needs no hardware queue to update it
  if (hipHostMalloc((void**)&c->dsync_table, sizeof(DsyncEntry) * kMaxRanks * kDsyncArenas, hipHostMallocMapped) != hipSuccess)
    return hip_fail(hipGetLastError(), "hipHostMalloc(translation table)", __FILE__, __LINE__);
  memset(c->dsync_table, 0, sizeof(DsyncEntry) * kMaxRanks * kDsyncArenas);
  {
    void* dev = nullptr;
    XMPI_HIP(hipHostGetDevicePointer(&dev, c->dsync_table, 0));
    c->dsync_table_dev = (const DsyncEntry*)dev;
  }
  // split form (sched.hip): what the meet kernel resolves for the data kernel, in ordinary device memory
  if (hipMalloc((void**)&c->dsync_res, sizeof(DsyncResolved)) != hipSuccess)
    return hip_fail(hipGetLastError(), "hipMalloc(resolved table)", __FILE__, __LINE__);
  if (hipEventCreateWithFlags(&c->dsync_order_ev, hipEventDisableTiming) != hipSuccess)
    return hip_fail(hipGetLastError(), "hipEventCreate", __FILE__, __LINE__);
  // The split form acquires / releases once per XCD from kXcdBlocks one-wave blocks and counts on the dispatcher dealing them
  // round the XCDs.  Nothing promises that, so look: how many XCDs the GPU has (a grid that fills it) and which ones a grid of
  // kXcdBlocks reaches (status words 8, 9: pinned, zeroed above).  If the small grid misses one, the data kernel runs in its
  // system-scope form from the start (body_sys) -- and the done kernel checks every launch anyway (xcd_check).
  if (c->dsync_status_dev) {
    bool ok = launch_xcc_probe(c->dsync_status_dev + 8, 1024, c->local_stream) == hipSuccess;
    for (int k = 0; ok && k < 4; k++) {  // the small grid a few times: the answer must not depend on where the dispatcher stood
      __atomic_store_n(c->dsync_status + 10, 0u, __ATOMIC_RELAXED);
      ok = launch_xcc_probe(c->dsync_status_dev + 10, kXcdBlocks, c->local_stream) == hipSuccess &&
           hipStreamSynchronize(c->local_stream) == hipSuccess;
      const uint32_t m = __atomic_load_n(c->dsync_status + 10, __ATOMIC_RELAXED);
      c->xcd_probe_mask = k == 0 ? m : (c->xcd_probe_mask & m);
    }
    if (ok) c->xcds = __builtin_popcount(__atomic_load_n(c->dsync_status + 8, __ATOMIC_RELAXED));
    (void)hipGetLastError();
    const bool covered = c->xcds > 0 && __builtin_popcount(c->xcd_probe_mask) >= c->xcds;
    if (c->body_sys < 0) c->body_sys = (c->xcds > 0 && !covered) ? 1 : 0;
    if (c->xcds > 0 && !covered)
      fprintf(stderr, "xmpi: rank %d: a %d-block grid reaches XCDs %#x of %d -- split collectives use system-scope loads / stores\n",
              c->rank, kXcdBlocks, c->xcd_probe_mask, c->xcds);
  }
  if (c->body_sys < 0) c->body_sys = 0;
  // epochs of this communicator: above whatever earlier communicators left in ANY rank's (pooled, uncleared) page;
  // the same number on every rank.  It also tags the translations this communicator's kernels cache in the page.
  uint64_t base = 0;
  for (int p = 0; p < N; p++) base = std::max(base, c->ctl->info(p)->flag_epoch);
  c->dsync_epoch = base;
  c->dsync_base = base;
  c->dsync_tag = base + 1;
  c->dsync_ok = true;
  return XMPI_OK;
}

// The rank's helper thread (xmpi_init starts it for every job of more than one process):
//  * A rank must map what its peers register even while its own threads are blocked somewhere the library cannot see (a
//    hipStreamSynchronize of the caller's, a long computation): the helper looks once a millisecond -- one load per peer when
//    there is nothing to do.  (Every wait loop of the library looks as well, so inside the library the answer comes within
//    microseconds.)
//  * A peer whose PROCESS is gone is an error at once, with default settings (XMPI_TIMEOUT_S = 0: wait for ever): every
//    watchdog_ms (50) the helper asks the kernel whether the processes that joined as the other ranks still exist (pid + start
//    time, ctl.cpp peer_gone) and raises the job's abort flag for the first one that does not -- every host wait loop and every
//    waiting kernel (kdev.h spin_until, the LL and receive agents) polls that flag and comes back with XMPI_ERR_PEER.  What the
//    reference's peers get from their sockets (network.go:555,611,623: a lost connection fails Send / Receive immediately).
void dsync_start_helper(xmpi_comm* c) {
  if (c->size < 2 || c->dsync_helper.joinable()) return;
  bool other_process = false;
  for (int p = 0; p < c->size; p++) other_process = other_process || c->ctl->info(p)->pid != (int32_t)getpid();
  const bool watch = c->watchdog_ms > 0 && other_process;
  if (!c->dsync_ok && !watch) return;
  c->dsync_helper_stop = false;
  c->dsync_helper = std::thread([c, watch] {
    (void)hipSetDevice(c->device);
    const long every = std::max<long>(1, c->watchdog_ms);
    double next_look = now_seconds() + (double)every * 1e-3;
    while (!c->dsync_helper_stop.load(std::memory_order_acquire)) {
      dsync_service(c);
      if (watch && now_seconds() >= next_look) {
        if (!c->ctl->aborted()) (void)c->ctl->check_peers();
        next_look = now_seconds() + (double)every * 1e-3;
      }
      timespec ts{0, 1000000};
      nanosleep(&ts, nullptr);
    }
  });
}

void dsync_stop_helper(xmpi_comm* c) {
  if (c->dsync_helper.joinable()) {
    c->dsync_helper_stop.store(true, std::memory_order_release);
    c->dsync_helper.join();
  }
}

void dsync_finalize(xmpi_comm* c) {
  dsync_stop_helper(c);
  // the next user of the page starts above the last epoch written into it (graph replays are counted on the device:
  // the kernels copy the page's counter into the pinned status area)
  uint64_t last = c->dsync_epoch;
  if (c->ctl) last = std::max<uint64_t>(last, c->ctl->info(c->rank)->flag_epoch);  // (a communicator that never got going)
  if (c->dsync_status) last = std::max<uint64_t>(last, __atomic_load_n((const uint64_t*)(c->dsync_status + 2), __ATOMIC_ACQUIRE));
  // the flag self-test's token lies in the peers' (never cleared) pages whatever the vote said: the next communicator's token -- and
  // epochs -- start above it, or a stale token would pass a self-test no store arrived for
  last = std::max<uint64_t>(last, c->dsync_selftest_token);
  last += 1;  // every communicator gets a number of its own (dsync_tag = base + 1), also one that never ran a collective: the
              // tag marks its translation-cache entries and its Send / Receive message numbers in the (uncleared) page
  for (auto& p : c->dsync_prof_pending) {
    (void)hipEventDestroy(p.start);
    (void)hipEventDestroy(p.stop);
  }
  c->dsync_prof_pending.clear();
  for (int p = 0; p < c->size; p++)
    if (c->peer_page_opened[p]) ipc_close_shared(c->peer_page[p]);
  if (c->host_bounce_dev) (void)hipHostFree(c->host_bounce);
  c->host_bounce = c->host_bounce_dev = nullptr;
  if (c->dsync_status) (void)hipHostFree(c->dsync_status);
  c->dsync_status = nullptr;
  if (c->dsync_res) (void)hipFree(c->dsync_res);
  c->dsync_res = nullptr;
  if (c->dsync_order_ev) (void)hipEventDestroy(c->dsync_order_ev);
  c->dsync_order_ev = nullptr;
  if (c->dsync_table) (void)hipHostFree(c->dsync_table);
  c->dsync_table = nullptr;
  for (auto& p : c->p2p_pending)
    for (void* b : p.bufs) (void)heap_free(b);
  c->p2p_pending.clear();
  for (auto& b : c->dsync_deferred) {
    (void)hipEventDestroy(b.done);
    for (void* p : b.bufs) (void)heap_free(p);
  }
  c->dsync_deferred.clear();
  for (void* p : c->dsync_leaked) (void)heap_free(p);
  c->dsync_leaked.clear();
  if (c->land_block) (void)heap_free(c->land_block);
  c->land_block = nullptr;
  c->land_block_bytes = 0;
  if (c->dpage) pool_release(c->dpage, last);
  c->dpage = nullptr;
  (void)hipGetLastError();
}

// Map what the peers have published since this rank last looked, and acknowledge.  Cheap when there is nothing
// to do (one load per peer); safe to call from any thread of the rank (a second caller just skips).
void dsync_service(xmpi_comm* c) {
  if (!c->dsync_ok) return;
  std::unique_lock<std::mutex> l(c->dsync_mu, std::try_to_lock);
  if (!l.owns_lock()) return;
  const int mypid = (int)getpid();
  for (int p = 0; p < c->size; p++) {
    if (p == c->rank) continue;
    PubTable* pt = c->ctl->published(p);
    const uint64_t n = pt->count.load(std::memory_order_acquire);
    uint64_t k = c->dsync_seen[p];
    if (k == n) continue;
    (void)hipSetDevice(c->device);  // may be the first HIP call of this thread (a wait loop of Send / Receive)
    for (; k < n; k++) {
      const PubEntry& e = pt->e[k % kPubRing];
      const int slot = (int)(e.reserved & 0xff);
      DsyncEntry ent;
      memset(&ent, 0, sizeof ent);
      ent.gen = e.gen;
      ent.bytes = e.bytes;
      if (c->ctl->info(p)->pid == mypid) {
        ent.base = e.base;  // same address space
      } else {
        BufRef ref;
        memset(&ref, 0, sizeof ref);
        ref.base = e.base;
        ref.gen = e.gen;
        ref.bytes = e.bytes;
        memcpy(ref.handle, e.handle, sizeof ref.handle);
        void* mapped = nullptr;
        if (zc_import(c, p, ref, &mapped)) ent.base = (uint64_t)(uintptr_t)mapped;
        else ent.gen = 0;  // cannot be mapped here: a kernel that meets it reports DSYNC_UNMAPPED
      }
      if (slot >= 0 && slot < kDsyncArenas) {  // the number last: a kernel that reads it (acquire) sees the rest
        DsyncEntry* t = &c->dsync_table[p * kDsyncArenas + slot];
        __atomic_store_n(&t->gen, 0, __ATOMIC_RELEASE);
        t->base = ent.base;
        t->bytes = ent.bytes;
        __atomic_store_n(&t->gen, ent.gen, __ATOMIC_RELEASE);
      }
    }
    c->dsync_seen[p] = n;
    c->ctl->acked(c->rank, p)->store(n, std::memory_order_release);
  }
}

namespace {

// the slot of my translation-table row that holds registration `gen`; publishes it if need be.
// *pub_index = how many published entries a peer must have processed to know it.
int publish(xmpi_comm* c, const BufRef& ref, int* slot_out, uint64_t* pub_index, bool capturing = false) {
  for (int s = 0; s < kDsyncArenas; s++)
    if (c->dsync_slot_gen[s] == ref.gen) {
      c->dsync_slot_used[s] = c->dsync_epoch + 1;
      *slot_out = s;
      *pub_index = c->dsync_slot_pub[s];
      return XMPI_OK;
    }
  int slot = -1;
  for (int s = 0; s < kDsyncArenas && slot < 0; s++)
    if (c->dsync_slot_gen[s] == 0) slot = s;
  if (slot < 0)  // slots of registrations that have been freed / deregistered since are free again
    for (int s = 0; s < kDsyncArenas; s++)
      if (!registry_alive(c->dsync_slot_gen[s])) {
        c->dsync_slot_gen[s] = 0;
        if (slot < 0) slot = s;
      }
  if (slot < 0) {
    // all slots hold live registrations: re-use the one that has not been used for longest.  Collectives in
    // flight may still name it -- let them finish first (this is rare: > 32 registered allocations in use).
    if (capturing) {  // a synchronisation is not allowed while a stream of the thread captures
      set_last_error("graph capture: the buffer's registration has no translation slot yet (use it in one collective before capturing)");
      return XMPI_ERR_ARG;
    }
    XMPI_HIP(hipDeviceSynchronize());
    slot = 0;
    for (int s = 1; s < kDsyncArenas; s++)
      if (c->dsync_slot_used[s] < c->dsync_slot_used[slot]) slot = s;
  }
  PubTable* pt = c->ctl->published(c->rank);
  const uint64_t n = pt->count.load(std::memory_order_relaxed);
  // the ring must not overwrite an entry a peer has not read yet
  Backoff bo;
  bo.idle = idle_hook;
  bo.idle_arg = c;
  const double t0 = now_seconds();
  for (;;) {
    uint64_t lo = n;
    for (int p = 0; p < c->size; p++)
      if (p != c->rank) lo = std::min(lo, c->ctl->acked(p, c->rank)->load(std::memory_order_acquire));
    if (n - lo < (uint64_t)kPubRing) break;
    if (c->ctl->aborted()) return XMPI_ERR_PEER;
    if (now_seconds() - t0 > wait_limit(c)) return XMPI_ERR_TIMEOUT;
    dsync_service(c);
    bo.pause();
  }
  PubEntry& e = pt->e[n % kPubRing];
  e.gen = ref.gen;
  e.base = ref.base;
  e.bytes = ref.bytes;
  e.reserved = (uint64_t)slot;
  memcpy(e.handle, ref.handle, sizeof e.handle);
  pt->count.store(n + 1, std::memory_order_release);
  c->dsync_slot_gen[slot] = ref.gen;
  c->dsync_slot_pub[slot] = n + 1;
  c->dsync_slot_used[slot] = c->dsync_epoch + 1;
  *slot_out = slot;
  *pub_index = n + 1;
  return XMPI_OK;
}

int await_acks(xmpi_comm* c, uint64_t pub_index) {
  Backoff bo;
  bo.idle = idle_hook;
  bo.idle_arg = c;
  const double t0 = now_seconds();
  for (;;) {
    bool all = true;
    for (int p = 0; p < c->size && all; p++)
      if (p != c->rank && c->ctl->acked(p, c->rank)->load(std::memory_order_acquire) < pub_index) all = false;
    if (all) return XMPI_OK;
    if (c->ctl->aborted()) {
      set_last_error(c->ctl->abort_reason());
      return XMPI_ERR_PEER;
    }
    if (now_seconds() - t0 > wait_limit(c)) {
      set_last_error("a peer did not map a newly registered buffer (is it inside the library at all?)");
      return XMPI_ERR_TIMEOUT;
    }
    dsync_service(c);
    bo.pause();
  }
}

// buffers lent to collectives on a stream (bounce copies of unregistered memory): given back once the
// stream has passed them
void reap_deferred(xmpi_comm* c, bool wait) {
  for (size_t i = 0; i < c->dsync_deferred.size();) {
    auto& b = c->dsync_deferred[i];
    hipError_t e = wait ? hipEventSynchronize(b.done) : hipEventQuery(b.done);
    if (e != hipSuccess) {  // not passed yet -- or not knowable (an error): the blocks stay lent rather than be re-used under a kernel
      (void)hipGetLastError();
      i++;
      continue;
    }
    (void)hipEventDestroy(b.done);
    for (void* p : b.bufs) (void)heap_free(p);
    c->dsync_deferred.erase(c->dsync_deferred.begin() + (long)i);
  }
}

struct Resolved {
  const void* send = nullptr;
  void* recv = nullptr;
  BufRef sref, rref;
  void* tmp_send = nullptr;  // registered stand-ins of buffers the peers cannot map
  void* tmp_recv = nullptr;
};

}  // namespace

// pinned memory for the host slices of blocking collectives; false = not available (the runtime's own staged copies serve)
static bool host_bounce_ready(xmpi_comm* c) {
  if (c->host_bounce_dev) return true;
  if (c->host_bounce) return false;  // tried before
  void* p = nullptr;
  if (hipHostMalloc(&p, 2 * xmpi_comm::kHostBounce, hipHostMallocMapped) != hipSuccess) {
    (void)hipGetLastError();
    c->host_bounce = reinterpret_cast<char*>(1);  // remember the failure
    return false;
  }
  void* dev = nullptr;
  if (hipHostGetDevicePointer(&dev, p, 0) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipHostFree(p);
    c->host_bounce = reinterpret_cast<char*>(1);
    return false;
  }
  c->host_bounce = (char*)p;
  c->host_bounce_dev = (char*)dev;
  return true;
}

// one launch of the copy-and-flag kernel (sched.hip p2p_pull_kernel) between a stand-in and the pinned memory above
static hipError_t bounce_copy(xmpi_comm* c, void* dst, const void* src, size_t bytes, int which, uint64_t* done, uint64_t done_value,
                              hipStream_t s) {
  P2PPullArgs pa;
  memset(&pa, 0, sizeof pa);
  pa.dst = dst;
  pa.src = src;
  pa.bytes = bytes;
  pa.ticket = c->p2p_tickets + xmpi_comm::kP2PDoneSlots + which;
  pa.host_done = done;
  pa.done_value = done_value;
  const long gx = std::max<long>(1, std::min<long>(16, (long)((bytes + 16383) >> 14)));
  return launch_p2p_pull(pa, (int)gx, s);
}

// A blocking call waits for its stream (polling: the wake-up latency of hipStreamSynchronize is a visible share of a small
// collective), serving the peers meanwhile.  by_word: the closing block of the (last) kernel writes the call's number into a
// pinned word (dsync_status bytes 16..23): no event between the kernel and this thread.  (The kernel itself gives up on a
// dead peer -- abort flag, XMPI_TIMEOUT_S -- and still writes the word.)
static int wait_blocking(xmpi_comm* c, hipStream_t stream, bool by_word, uint64_t done_id) {
  Backoff bo;
  bo.idle = idle_hook;
  bo.idle_arg = c;
  if (by_word) {
    const volatile uint64_t* done = (const volatile uint64_t*)(c->dsync_status + 4);
    // Safety valve, off the fast path: once the wait is long, ask the stream now and then -- a stream that is idle (or broken)
    // while the word is still missing must not hang the caller.
    unsigned spins = 0;
    while (__atomic_load_n((const uint64_t*)done, __ATOMIC_ACQUIRE) != done_id) {
      bo.pause();
      if ((++spins & 0x3fff) == 0) {
        const hipError_t e = hipStreamQuery(stream);
        if (e == hipErrorNotReady) {
          (void)hipGetLastError();
          continue;
        }
        if (e != hipSuccess) return hip_fail(e, "hipStreamQuery", __FILE__, __LINE__);
        if (__atomic_load_n((const uint64_t*)done, __ATOMIC_ACQUIRE) != done_id) {
          set_last_error("collective: the stream drained but the closing block never reported (internal)");
          return XMPI_ERR_STATE;
        }
      }
    }
    return XMPI_OK;
  }
  hipEvent_t fin = ev_get(c, false);
  if (!fin) return XMPI_ERR_HIP;
  XMPI_HIP(hipEventRecord(fin, stream));
  for (;;) {
    const hipError_t e = hipEventQuery(fin);
    if (e == hipSuccess) break;
    if (e != hipErrorNotReady) return hip_fail(e, "hipEventQuery", __FILE__, __LINE__);
    (void)hipGetLastError();
    bo.pause();
  }
  ev_put(c, fin, false);
  return XMPI_OK;
}

// the kernels of one rank share the page's epoch counter, ticket and slots: one at a time.  On one stream that is
// stream order; a launch on another stream than the previous one waits for it.
// (The event is recorded when the stream CHANGES, at the tail of the previous stream -- not behind every launch: an event
// per collective costs the queue a packet and was visible in the small-message figures.)
static int order_behind_last(xmpi_comm* c, hipStream_t stream, bool capturing) {
  if (capturing || !c->dsync_last_stream || c->dsync_last_stream == stream) return XMPI_OK;
  hipStreamCaptureStatus pc = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(c->dsync_last_stream, &pc);
  (void)hipGetLastError();
  if (pc == hipStreamCaptureStatusNone) {  // (a capturing stream runs nothing until its graph is launched: dsync_graph_launched)
    XMPI_HIP(hipEventRecord(c->dsync_order_ev, c->dsync_last_stream));
    XMPI_HIP(hipStreamWaitEvent(stream, c->dsync_order_ev, 0));
  }
  return XMPI_OK;
}

// ---- LL small collectives (ll.hip): the payload is pushed into the peers' flag allocations, nothing is registered,
// announced or translated; one kernel, one one-way hop.  Whether a call goes this way is decided by dsync_collective from the
// arguments and the job's layout only (the same on every rank); what kind of memory a rank passes is this rank's business.
static int dsync_ll(xmpi_comm* c, int coll, int root, const void* sendbuf, void* recvbuf, size_t count, int dtype, int op,
                    hipStream_t stream, bool blocking, bool capturing) {
  const int N = c->size, me = c->rank;
  const size_t unit = count * xmpi_dtype_size((xmpi_dtype)dtype);
  const size_t recv_bytes = coll == COLL_ALLGATHER ? unit * (size_t)N : unit;
  RoctxRange range("xmpi:dsync %s form=ll bytes=%zu epoch=%llu %s", coll_name(coll), unit, (unsigned long long)c->dsync_epoch + 1,
                   blocking ? "blocking" : "enqueued");
  const bool reads = coll != COLL_BCAST || me == root;           // this rank's send buffer is read
  const bool writes = (coll != COLL_REDUCE || me == root) && !(coll == COLL_BCAST && me == root);  // its receive buffer is written
  std::vector<void*> lent;
  auto fail = [&](int rc) {
    for (void* p : lent) (void)heap_free(p);
    c->ctl->set_abort(rc);
    return rc;
  };
  auto on_gpu = [&](const void* p, size_t bytes) {
    BufRef ref;
    return zc_export(c, p, bytes, &ref) || is_device_pointer(p);
  };
  const void* send = sendbuf;
  void* recv = recvbuf;
  void* out_tmp = nullptr;   // a stand-in whose contents go home to recvbuf
  bool host_out = false;     // ... or the result lies in the pinned block
  if (reads && !on_gpu(sendbuf, unit)) {  // a host slice: through pinned memory the kernel reads itself, or a stand-in
    if (capturing) {
      set_last_error("graph capture needs device buffers");
      return XMPI_ERR_ARG;
    }
    if (blocking && host_bounce_ready(c)) {
      memcpy(c->host_bounce, sendbuf, unit);
      send = c->host_bounce_dev;
      c->host_bounce_calls++;
    } else {
      void* tmp = heap_alloc(c->device, unit);
      if (!tmp) return fail(XMPI_ERR_NOMEM);
      lent.push_back(tmp);
      const hipError_t ce = hipMemcpyAsync(tmp, sendbuf, unit, hipMemcpyDefault, stream);
      if (ce != hipSuccess) return fail(hip_fail(ce, "hipMemcpyAsync(stand-in)", __FILE__, __LINE__));
      send = tmp;
    }
  }
  if (writes && !on_gpu(recvbuf, recv_bytes)) {
    if (capturing) {
      set_last_error("graph capture needs device buffers");
      return XMPI_ERR_ARG;
    }
    if (blocking && recv_bytes <= xmpi_comm::kHostBounce && c->dsync_status_dev && host_bounce_ready(c)) {
      recv = c->host_bounce_dev + xmpi_comm::kHostBounce;
      host_out = true;
      c->host_bounce_calls++;
    } else {
      out_tmp = heap_alloc(c->device, recv_bytes);
      if (!out_tmp) return fail(XMPI_ERR_NOMEM);
      lent.push_back(out_tmp);
      recv = out_tmp;
    }
  }
  if (!writes) recv = const_cast<void*>(send);  // (never written; the launcher wants a pointer)
  if (!reads) send = recv;
  const int ll_coll = coll == COLL_ALLREDUCE ? LL_ALLREDUCE : coll == COLL_REDUCE ? LL_REDUCE : coll == COLL_BCAST ? LL_BCAST : LL_ALLGATHER;
  // A blocking call -- the only kind the reference's API has (mpi.go:47-48) -- is launch + kernel + completion word, and the
  // launch is half of it.  The LL agent (ll.hip ll_agent_kernel: a one-block kernel that lingers behind the previous blocking small
  // collective, as the receive agent does behind a Receive) runs the same lines without one: the command goes through pinned
  // memory, the answer comes back the same way.  Only when the stream the call would have
  // been enqueued on is idle and nothing this communicator enqueued elsewhere is still running (the agent cannot wait for a
  // stream; the launched kernel is ordered behind both), nothing is being profiled per dispatch, and no stand-in needs a copy
  // on the stream first.
  // (the streams are not asked when the previous call into the library on this communicator was itself a collective the agent
  // ran: it found them idle, and nothing has been enqueued through the library since)
  // THIS call's number (what its XMPI_ENTER drew), not the counter as it stands now: another thread of the communicator may have
  // entered since -- it waits for coll_mu, is counted already, and what it goes on to enqueue is not this call's to vouch for
  const uint64_t calls = t_api_call;
  auto idle = [](hipStream_t s) { return hipStreamQuery(s) == hipSuccess; };
  const bool quiet = calls == c->agent_quiet_at + 1, consecutive = calls == c->agent_epoch_at + 1;
  // up to agent_ll_bytes (8 KiB: a lane's two rounds of lines are waited for together -- blocking 8.8 us against 11.7 launched,
  // 2 processes; host slices 10.0 against 13.8; at 16 KiB it is a tie, 12.1 / 11.7, and beyond the launched kernel's many blocks
  // win) -- scripts/r04_agent_limit.sh
  const size_t agent_limit = (size_t)std::max<long>(0, c->agent_ll_bytes);
  if (blocking && !capturing && lent.empty() && c->agent_ll && unit <= agent_limit && !c->prof_on &&
      (quiet ||
       (idle(stream) && (!c->dsync_last_stream || c->dsync_last_stream == stream || idle(c->dsync_last_stream))))) {
    ++c->dsync_epoch;
    const double t_cmd = now_seconds();
    // An agent that has gone is started again, whatever the caller's pace.  Tried and measured (scripts/r04_agent_patience.sh, 2
    // processes, 1 KiB, the caller's own work between two calls 100 / 500 us, patience 40): starting it only for a BURST of calls --
    // the previous one less than a patience ago -- and launching the ordinary kernel otherwise: 25.4 / 25.3 us per call, against
    // 14.4 / 21.5 with the agent started by every call (a launch into a GPU that has been idle for 100 us costs more than one into a
    // busy GPU; the agent's launch overlaps with the command already lying in its record) and 7.5 inside its patience.
    const int took = agent_submit_ll(c, send, recv, unit, ll_coll, root, dtype, op, consecutive);
    if (took < 0) {  // taken and never answered: the collective has failed, nothing may run for this epoch beside the agent
      set_last_error("collective: the LL agent did not answer within XMPI_TIMEOUT_S (a peer that never arrived?)");
      return fail(XMPI_ERR_TIMEOUT);
    }
    if (took > 0) {
      c->agent_ll_wait_ns += (uint64_t)((now_seconds() - t_cmd) * 1e9);
      // (only when no other thread has entered the library on this communicator meanwhile: what it goes on to enqueue is not
      // this call's to vouch for)
      if (c->api_calls.load(std::memory_order_relaxed) == calls) c->agent_quiet_at = c->agent_epoch_at = calls;
      c->dsync_ll_launches++;  // (an LL collective, whoever ran its lines)
      c->dsync_ll_agent++;
      if (host_out) memcpy(recvbuf, c->host_bounce + xmpi_comm::kHostBounce, recv_bytes);
      return dsync_check(c);
    }
    --c->dsync_epoch;
  }
  (void)hipGetLastError();
  int rc = order_behind_last(c, stream, capturing);
  if (rc != XMPI_OK) return fail(rc);

  DsyncLLArgs a;
  memset(&a, 0, sizeof a);
  for (int p = 0; p < N; p++) a.page[p] = c->peer_page[p];
  a.me = me;
  a.n = N;
  a.coll = ll_coll;
  a.root = root;
  a.epoch_floor = c->dsync_base;
  a.host_epoch = c->dsync_status_dev ? (uint64_t*)(c->dsync_status_dev + 2) : nullptr;
  const uint64_t done_id = ++c->dsync_done_seq;
  uint64_t* const done_dev = (blocking && c->dsync_status_dev) ? (uint64_t*)(c->dsync_status_dev + 4) : nullptr;
  a.host_done = done_dev;
  a.done_value = done_id;
  a.send = send;
  a.recv = recv;
  a.bytes = unit;
  a.abort_word = c->dsync_abort_dev;
  a.status = c->dsync_status_dev;
  a.spin_limit = c->timeout_s > 0 ? (uint64_t)c->timeout_s * 100000000ull : 0;
  hipEvent_t pstart = nullptr, pstop = nullptr;
  const bool sampled = !capturing && c->prof_on && (c->prof_seq[PROF_ZCOPY]++ % (uint64_t)std::max<long>(1, c->prof_every)) == 0;
  if (sampled) {
    pstart = ev_get(c, true);
    pstop = ev_get(c, true);
    if (!pstart || !pstop) return fail(XMPI_ERR_HIP);
  }
  ++c->dsync_epoch;
  {
    const hipError_t le = launch_dsync_ll(a, dtype, op, stream, pstart, pstop);
    if (le != hipSuccess) {  // nothing was enqueued: the host's count goes back, the lent blocks too, and the peers -- whose kernels
      --c->dsync_epoch;      // wait for this rank's lines -- are told through the job's abort flag (fail)
      return fail(hip_fail(le, "LL kernel launch", __FILE__, __LINE__));
    }
  }
  c->dsync_launches++;
  c->dsync_ll_launches++;
  if (!capturing) c->dsync_last_stream = stream;
  const size_t traffic = 2 * unit * (size_t)N;  // (own payload read, N-1 pushes of twice its size ... : a latency path, not a bandwidth one)
  if (pstart) c->dsync_prof_pending.push_back({pstart, pstop, traffic});
  if (!blocking) {
    if (out_tmp) {
      const hipError_t ce = hipMemcpyAsync(recvbuf, out_tmp, recv_bytes, hipMemcpyDefault, stream);
      if (ce != hipSuccess) return fail(hip_fail(ce, "hipMemcpyAsync(result of a stand-in)", __FILE__, __LINE__));
    }
    if (!lent.empty()) {
      xmpi_comm::DsyncDeferred d;
      if (hipEventCreateWithFlags(&d.done, hipEventDisableTiming) != hipSuccess) return fail(XMPI_ERR_HIP);
      const hipError_t re = hipEventRecord(d.done, stream);
      if (re != hipSuccess) {
        (void)hipEventDestroy(d.done);
        return fail(hip_fail(re, "hipEventRecord", __FILE__, __LINE__));
      }
      d.bufs = lent;
      c->dsync_deferred.push_back(d);
    }
    return XMPI_OK;
  }
  rc = wait_blocking(c, stream, done_dev != nullptr, done_id);
  if (rc != XMPI_OK) return fail(rc);
  // (the kernel's last act was the word this call waited for; everything before it on the streams is over -- unless another thread
  // has entered the library meanwhile)
  if (c->api_calls.load(std::memory_order_relaxed) == calls) c->agent_quiet_at = calls;
  if (host_out) {
    memcpy(recvbuf, c->host_bounce + xmpi_comm::kHostBounce, recv_bytes);
  } else if (out_tmp) {
    hipError_t ce = hipMemcpyAsync(recvbuf, out_tmp, recv_bytes, hipMemcpyDefault, stream);
    if (ce == hipSuccess) ce = hipStreamSynchronize(stream);
    if (ce != hipSuccess) return fail(hip_fail(ce, "copy of a stand-in's result", __FILE__, __LINE__));
  }
  for (void* p : lent) (void)heap_free(p);
  lent.clear();
  dsync_prof_harvest(c);
  return dsync_check(c);
}

bool dsync_usable(const xmpi_comm* c) { return c->dsync_ok && c->dsync && c->size > 1; }

// blocks a kernel of this rank may keep waiting at once: the kernels of all ranks on one GPU spin together, so with
// several ranks per GPU they must all be resident (half of its 8192 wave slots, 4 waves per block, shared)
static long dsync_block_cap(const xmpi_comm* c) {
  if (c->dsync_grid_cap > 0) return c->dsync_grid_cap;
  // (a rank alone on its GPU keeps to half of the chip too: while its blocks wait for a late peer the caller's other streams
  // still find wave slots -- scripts/overlap_probe.hip: a full-chip compute kernel next to a waiting 1024-block collective
  // runs 5 % slower, next to the meet / body / done form 1.5 %)
  // (Eight processes on one GPU, 64 instead of 128 blocks per rank: 6 % faster at 256 MiB on one box, 12 % slower on the next --
  // scripts/r03_tiles.sh; XMPI_DSYNC_GRID=256 per rank = every wave slot of the chip: the kernels wait for each other for ever.)
  return 1024 / std::max(1, c->dsync_sharers);
}

int dsync_grid(const xmpi_comm* c, size_t packets_per_segment, int nseg, int unroll) {
  long cap = std::max<long>(1, dsync_block_cap(c) / std::max(1, nseg));
  // several tiles per block: every block costs seven polling lanes on this rank's page and a ticket, which is what a
  // small collective spends its time on (8 processes, 1 MiB: 32 blocks per rank -> 84 us, see profiles/README.md)
  const size_t per_block = (size_t)256 * (size_t)std::max(1, unroll) * (size_t)std::max<long>(1, c->dsync_tiles);
  const size_t want = (packets_per_segment + per_block - 1) / per_block;
  return (int)std::max<size_t>(1, std::min<size_t>(want, (size_t)cap));
}

// which calls the device-synchronised path takes (the same answer on every rank: it depends on the job's layout
// and the arguments only)
bool dsync_takes(const xmpi_comm* c, int coll, int algo) {
  if (!dsync_usable(c)) return false;
  if (!c->windows_ok) return true;  // a job without windows has no staged step tables: DIRECT and whatever else names them is the fold
  switch (algo) {
    case XMPI_ALGO_AUTO: return c->zero_copy != 0;
    case XMPI_ALGO_ZCOPY:
    case XMPI_ALGO_ZPUSH:
    case XMPI_ALGO_LL: return true;
    case XMPI_ALGO_RING:  // the stepped kernels (sched.hip), pull and push form
    case XMPI_ALGO_RING_PUSH: return coll == COLL_ALLREDUCE || coll == COLL_ALLGATHER;
    case XMPI_ALGO_RHD:
    case XMPI_ALGO_RHD_PUSH: return coll == COLL_ALLREDUCE;
    case XMPI_ALGO_TREE:
    case XMPI_ALGO_TREE_PUSH: return coll == COLL_BCAST || coll == COLL_REDUCE;
    default: return false;
  }
}

// hipGraphLaunch of a captured sequence of collectives is a device-synchronised launch like any other: it is ordered
// against the rank's other streams (before = true: ahead of the launch; false: behind it)
void dsync_graph_launched(xmpi_comm* c, hipStream_t stream, bool before) {
  if (!c->dsync_ok || !c->dsync_order_ev) return;
  if (before) {
    if (c->dsync_last_stream && c->dsync_last_stream != stream) {
      (void)hipEventRecord(c->dsync_order_ev, c->dsync_last_stream);
      (void)hipStreamWaitEvent(stream, c->dsync_order_ev, 0);
    }
  } else {
    c->dsync_last_stream = stream;
  }
  (void)hipGetLastError();
}

// the library's schedule for an AUTO call (xmpi_tune fills the table; untuned: the zero-copy fold, split by size)
static void tuned_choice(const xmpi_comm* c, int coll, size_t bytes, int* algo, int* split, int* unroll) {
  int k = 0;
  while (k + 1 < xmpi_comm::kTuneClasses && (bytes >> (k + 9)) != 0) k++;
  if (c->tuned && coll >= 0 && coll < 4) {
    if (c->tune_algo[coll][k] >= 0) *algo = c->tune_algo[coll][k];
    if (c->tune_split[coll][k] >= 0) *split = c->tune_split[coll][k];
    if (c->tune_unroll[coll][k] > 0) *unroll = c->tune_unroll[coll][k];
  }
}

// the call signature the kernels announce and compare (DsyncArgs::sig)
static inline uint64_t sig_mix(uint64_t h, uint64_t v) {
  h = (h ^ v) * 0x100000001B3ull;
  return h ^ (h >> 29);
}
static inline uint64_t sig_fold(uint64_t h) { return std::max<uint64_t>(1, (h ^ (h >> 32)) & 0xffffffffull); }

// a HIP failure inside a collective that has already borrowed blocks or advanced the epoch: the blocks go back and the peers --
// whose kernels would wait for this rank -- are told through the job's abort flag (the function's `fail`)
#define DS_HIP(call)                                                                 \
  do {                                                                               \
    const hipError_t _e = (call);                                                    \
    if (_e != hipSuccess) return fail(::xmpi::hip_fail(_e, #call, __FILE__, __LINE__)); \
  } while (0)

// One device-synchronised collective, enqueued on `stream`.  blocking: wait for it (the xmpi_allreduce family);
// otherwise return once it is enqueued (xmpi_*_on_stream).  Every rank of the job takes this path for the same
// calls (the decision depends on the communicator and the arguments only), so the epochs agree.
int dsync_collective(xmpi_comm* c, int coll, int root, const void* sendbuf, void* recvbuf, size_t count, int dtype,
                     int op, hipStream_t stream, bool blocking, int algo) {
  const int N = c->size, me = c->rank;
  const size_t es = xmpi_dtype_size((xmpi_dtype)dtype);
  const size_t send_bytes = count * es;
  const size_t recv_bytes = (coll == COLL_ALLGATHER) ? send_bytes * (size_t)N : send_bytes;
  const bool recv_significant = (coll != COLL_REDUCE) || me == root;
  if (!stream) stream = c->local_stream;
  const uint64_t calls_at_entry = t_api_call;  // (dsync_ll's shortcut: see agent_quiet_at)
  dsync_service(c);
  reap_deferred(c, false);
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(stream, &cap);
  (void)hipGetLastError();
  const bool capturing = cap != hipStreamCaptureStatusNone;

  // the schedule: what the caller named, or the library's own table
  int split_pref = -1, unroll = (int)std::max<long>(1, std::min<long>(2, c->dsync_unroll));
  if (algo == XMPI_ALGO_AUTO) {
    tuned_choice(c, coll, send_bytes, &algo, &split_pref, &unroll);
    // untuned: short messages go as {data, flag} lines (ll.hip) -- one one-way hop instead of two round trips
    if (algo == XMPI_ALGO_AUTO && c->zero_copy && send_bytes <= (size_t)std::max<long>(0, c->ll_bytes)) algo = XMPI_ALGO_LL;
  }
  // A schedule whose ANSWERS xmpi_tune or xmpi_init's self-check found wrong on this machine (tune_rejected: the outcome of a vote,
  // the same bits on every rank) is not run for a caller: AUTO never leads here (the table leaves it out, apply_rejections the untuned
  // rules), so this is a caller -- or a table written by hand -- naming it.  Every rank refuses alike, before anything is lent,
  // announced or moved.
  const uint32_t rejected = (c->tune_running || coll < 0 || coll >= 4) ? 0u : c->tune_rejected[coll];
  auto refuse = [&](int cand) {
    static const char* const what[] = {"the one-kernel fold", "the one-kernel fold", "meet / body / done", "push-only", "the ring kernel", "the halving kernel",
                                       "LL lines", "the ring kernel's push form", "the halving kernel's push form", "the tree kernel", "the tree kernel's push form"};
    set_last_error(std::string(coll_name(coll)) + " by " + what[cand] + ": refused -- it gave wrong answers on this machine when the library checked it (xmpi_get_param "
                   "\"tune_rejected_" + std::to_string(coll) + "\"; xmpi_degraded() says where)");
    return XMPI_ERR_UNSUPPORTED;
  };
  if (algo == XMPI_ALGO_LL) {
    if (send_bytes <= kLLMaxPayload) {
      if ((rejected >> xmpi_comm::CAND_LL) & 1u) return refuse(xmpi_comm::CAND_LL);
      return dsync_ll(c, coll, root, sendbuf, recvbuf, count, dtype, op, stream, blocking, capturing);
    }
    algo = XMPI_ALGO_ZCOPY;  // named, but too long for the slots: the fold (the same decision on every rank)
  }
  // the push forms of the stepped kernels: the same schedule, the data stored into the peer instead of loaded from it
  bool sched_push = algo == XMPI_ALGO_RING_PUSH || algo == XMPI_ALGO_RHD_PUSH || algo == XMPI_ALGO_TREE_PUSH;
  int sched_algo = algo == XMPI_ALGO_RING_PUSH ? XMPI_ALGO_RING : algo == XMPI_ALGO_RHD_PUSH ? XMPI_ALGO_RHD
                   : algo == XMPI_ALGO_TREE_PUSH ? XMPI_ALGO_TREE : algo;
  if (capturing) {
    // A graph cannot hold what is lent per call -- a landing block, the tree reduce's accumulator: replays would use it after it
    // went back to the arena.  Captured, a push form runs as its pull form (the same operands, order and association: the same
    // bits) and the tree reduce as the fold -- whatever named them, the caller or the tuner's table.  Every rank captures a
    // collective or none does (as with push-only below), so every rank decides alike: by the arguments, not by what IT would lend.
    if (sched_push) algo = sched_algo;
    sched_push = false;
    if (sched_algo == XMPI_ALGO_TREE && coll == COLL_REDUCE) algo = sched_algo = XMPI_ALGO_ZCOPY;
  }
  const bool stepped = (sched_algo == XMPI_ALGO_RING && (coll == COLL_ALLREDUCE || coll == COLL_ALLGATHER)) ||
                       (sched_algo == XMPI_ALGO_RHD && coll == COLL_ALLREDUCE) ||
                       (sched_algo == XMPI_ALGO_TREE && (coll == COLL_BCAST || coll == COLL_REDUCE));
  const bool push = algo == XMPI_ALGO_ZPUSH;
  if (rejected) {
    const bool fold_no = (rejected >> xmpi_comm::CAND_FOLD) & 1u, split_no = (rejected >> xmpi_comm::CAND_SPLIT) & 1u;
    if (stepped) {
      const int cand = sched_algo == XMPI_ALGO_RING ? (sched_push ? xmpi_comm::CAND_RING_PUSH : xmpi_comm::CAND_RING)
                       : sched_algo == XMPI_ALGO_RHD ? (sched_push ? xmpi_comm::CAND_RHD_PUSH : xmpi_comm::CAND_RHD)
                                                     : (sched_push ? xmpi_comm::CAND_TREE_PUSH : xmpi_comm::CAND_TREE);
      if ((rejected >> cand) & 1u) return refuse(cand);
    } else if (push && (coll == COLL_ALLREDUCE || coll == COLL_REDUCE) && !capturing) {
      if ((rejected >> xmpi_comm::CAND_ZPUSH) & 1u) return refuse(xmpi_comm::CAND_ZPUSH);
    } else if (coll == COLL_BCAST) {  // (its fold is one kernel whatever the size)
      if (fold_no) return refuse(xmpi_comm::CAND_FOLD);
    } else {
      // one kernel or meet / body / done is each rank's own choice (the two mix: dsync_begin / the meet kernel speak one protocol):
      // a rank keeps to the one of the two that is right here
      if (fold_no && split_no) return refuse(xmpi_comm::CAND_FOLD);
      if (split_no) split_pref = 0;
      else if (fold_no && c->dsync_res) split_pref = 1;
    }
  }
  RoctxRange range("xmpi:dsync %s algo=%s bytes=%zu epoch=%llu %s", coll_name(coll), algo_name(algo), send_bytes,
                   (unsigned long long)c->dsync_epoch + 1, blocking ? "blocking" : capturing ? "captured" : "enqueued");

  // 1. buffers the peers can map.  Anything else -- host memory, device memory that was never registered --
  //    is stood in for by a block of a registered arena (one local copy in, one out); the collective itself is
  //    the same zero-copy exchange.
  Resolved r;
  r.send = sendbuf;
  r.recv = recv_significant ? recvbuf : const_cast<void*>(sendbuf);
  const bool in_place = sendbuf == recvbuf;
  std::vector<void*> lent, outgrown;
  // what may still be in use by collectives enqueued EARLIER goes back behind an event on the stream (or, failing that, at finalize)
  auto defer_free = [&](std::vector<void*>& bufs) {
    if (bufs.empty()) return;
    xmpi_comm::DsyncDeferred d;
    if (hipEventCreateWithFlags(&d.done, hipEventDisableTiming) == hipSuccess && hipEventRecord(d.done, stream) == hipSuccess) {
      d.bufs = bufs;
      c->dsync_deferred.push_back(d);
    } else {
      (void)hipGetLastError();
      for (void* p : bufs) c->dsync_leaked.push_back(p);  // given back by dsync_finalize
    }
    bufs.clear();
  };
  auto fail = [&](int rc) {
    for (void* p : lent) (void)heap_free(p);
    defer_free(outgrown);
    c->ctl->set_abort(rc);
    return rc;
  };
  // a capture bakes addresses into the graph: a stand-in would be given back to the arena while replays still use it
  auto no_standin = [&]() {
    set_last_error("graph capture needs registered device buffers (xmpi_malloc / xmpi_register)");
    return XMPI_ERR_ARG;
  };
  if (!zc_export(c, r.send, send_bytes, &r.sref)) {
    if (capturing) return no_standin();
    r.tmp_send = heap_alloc(c->device, send_bytes);
    if (!r.tmp_send) return fail(XMPI_ERR_NOMEM);
    lent.push_back(r.tmp_send);
    if (coll != COLL_BCAST || me == root) {
      // a host slice of a blocking call goes in through pinned memory the GPU reads itself (memcpy + one small kernel in
      // stream order) -- the runtime's copy out of pageable memory is a staged, synchronous affair of 10 us and more
      if (blocking && send_bytes <= xmpi_comm::kHostBounce && c->p2p_tickets && !is_device_pointer(sendbuf) && host_bounce_ready(c)) {
        memcpy(c->host_bounce, sendbuf, send_bytes);
        DS_HIP(bounce_copy(c, r.tmp_send, c->host_bounce_dev, send_bytes, 0, nullptr, 0, stream));
        c->host_bounce_calls++;
      } else {
        DS_HIP(hipMemcpyAsync(r.tmp_send, sendbuf, send_bytes, hipMemcpyDefault, stream));
      }
    }
    r.send = r.tmp_send;
    if (!zc_export(c, r.send, send_bytes, &r.sref)) return fail(XMPI_ERR_HIP);
    c->dsync_bounced++;
  }
  if (!recv_significant || (in_place && coll != COLL_ALLGATHER)) {
    r.recv = const_cast<void*>(r.send);
    r.rref = r.sref;
  } else if (!zc_export(c, r.recv, recv_bytes, &r.rref)) {
    if (capturing) return no_standin();
    r.tmp_recv = heap_alloc(c->device, recv_bytes);
    if (!r.tmp_recv) return fail(XMPI_ERR_NOMEM);
    lent.push_back(r.tmp_recv);
    r.recv = r.tmp_recv;
    if (!zc_export(c, r.recv, recv_bytes, &r.rref)) return fail(XMPI_ERR_HIP);
    c->dsync_bounced++;
  }

  // The communicator KEEPS one registered block for what a stepped collective needs beside the caller's buffers -- the push forms'
  // landing block, the pull-form tree reduce's accumulator -- and uses it again for the next one (grown when one needs more): the
  // peers touch it only between this rank's announce for a collective and its close, and its kernels run one at a time -- so
  // back-to-back enqueued collectives share one block, where a block lent per call would have each of them take a new one before
  // the stream has given the last one back (1 GiB fp16, halving push form, 5 enqueued steps: a new 1 GiB arena allocated,
  // exported and mapped by every peer per step -- 100 ms instead of 9).
  auto own_block = [&](size_t bytes) -> void* {
    if (!c->land_block || c->land_block_bytes < bytes) {
      // the outgrown block goes back once THIS collective's kernel -- behind every earlier one -- has passed.  Not through `lent`:
      // a failure path gives `lent` back at once, and earlier ENQUEUED collectives may still be landing in this block
      if (c->land_block) outgrown.push_back(c->land_block);
      c->land_block = heap_alloc(c->device, bytes);
      c->land_block_bytes = c->land_block ? bytes : 0;
    }
    return c->land_block;
  };
  // binary-tree reduce, pull form: an inner node that is not the root accumulates its subtree's partial result in a block the
  // parent can read (sched_steps.h SCHED_TREE_REDUCE); the caller's receive buffer means nothing there
  if (stepped && !sched_push && coll == COLL_REDUCE && me != root && 2 * ((me - root + N) % N) + 1 < N) {
    if (capturing) return no_standin();
    void* acc = own_block(send_bytes);
    if (!acc) return fail(XMPI_ERR_NOMEM);
    r.recv = acc;
    if (!zc_export(c, r.recv, send_bytes, &r.rref)) return fail(XMPI_ERR_HIP);
  }

  // where the result of a stand-in goes home from (step 4), and how: a host slice of a blocking call comes out through pinned
  // memory -- one more small kernel in stream order copies the stand-in there and THEN writes the completion word
  const void* out_src = r.tmp_recv ? r.tmp_recv
                        : (r.tmp_send && recv_significant && (in_place || coll == COLL_BCAST) && coll != COLL_ALLGATHER) ? r.tmp_send
                                                                                                                          : nullptr;
  const bool host_out = blocking && out_src && recv_bytes <= xmpi_comm::kHostBounce && c->p2p_tickets && c->dsync_status_dev &&
                        !is_device_pointer(recvbuf) && host_bounce_ready(c);

  // 2. the peers know the allocations
  int sslot = 0, rslot = 0;
  uint64_t need = 0, pi = 0;
  int rc = publish(c, r.sref, &sslot, &pi, capturing);
  if (rc) return fail(rc);
  need = std::max(need, pi);
  rc = publish(c, r.rref, &rslot, &pi, capturing);
  if (rc) return fail(rc);
  need = std::max(need, pi);
  rc = await_acks(c, need);
  if (rc) return fail(rc);

  rc = order_behind_last(c, stream, capturing);
  if (rc) return fail(rc);

  // 3. the kernel(s)
  DsyncArgs a;
  memset(&a, 0, sizeof a);
  for (int p = 0; p < N; p++) a.page[p] = c->peer_page[p];
  a.me = me;
  a.n = N;
  a.send_gen = r.sref.gen;
  a.send_off = r.sref.offset;
  a.send_slot = (uint64_t)sslot;
  a.recv_gen = r.rref.gen;
  a.recv_off = r.rref.offset;
  a.recv_slot = (uint64_t)rslot;
  a.table = c->dsync_table_dev;
  a.tag = c->dsync_tag;
  a.epoch_floor = c->dsync_base;
  a.host_epoch = c->dsync_status_dev ? (uint64_t*)(c->dsync_status_dev + 2) : nullptr;
  // a blocking call learns that the collective is over from a word its closing block writes (dsync_status bytes 16..23),
  // not from an event: set on the LAST kernel of the collective only (below)
  const uint64_t done_id = ++c->dsync_done_seq;
  uint64_t* const done_dev = (blocking && c->dsync_status_dev) ? (uint64_t*)(c->dsync_status_dev + 4) : nullptr;
  uint64_t* const done_k = host_out ? nullptr : done_dev;  // (host_out: the copy-out kernel behind the collective writes it)
  a.my_send = r.send;
  a.my_recv = r.recv;
  a.abort_word = c->dsync_abort_dev;
  a.status = c->dsync_status_dev;
  a.xcc_need = (c->xcd_check && !c->body_sys) ? c->xcds : 0;  // (body_sys: the data kernel needs no L2 to have been acquired)
  a.spin_limit = c->timeout_s > 0 ? (uint64_t)c->timeout_s * 100000000ull : 0;  // wall_clock64 ticks at 100 MHz
  const uint32_t everyone = N >= 32 ? 0xffffffffu : ((1u << N) - 1u);
  const size_t al = std::max<size_t>(1, 16 / es);

  hipEvent_t pstart = nullptr, pstop = nullptr;
  // sampled launches carry their own begin / end events (attached to the dispatch); a launch that is only enqueued
  // leaves them for the next blocking call to read, so sampling does not put a host wait between enqueued steps
  const bool sampled = !capturing && c->prof_on && (c->prof_seq[PROF_ZCOPY]++ % (uint64_t)std::max<long>(1, c->prof_every)) == 0;
  size_t traffic = 0;
  auto prof_events = [&]() -> bool {
    if (sampled && !pstart) {
      pstart = ev_get(c, true);
      pstop = ev_get(c, true);
      if (!pstart || !pstop) return false;
    }
    return true;
  };
  // one rendezvous + data movement + completion exchange: ONE kernel, or -- large messages -- meet / body / done
  auto launch = [&](int nsrc, int kdtype, int kop, size_t packets, size_t bytes_moved, bool last = true) -> int {
    ++c->dsync_epoch;  // the host's count (the kernels count for themselves, from the page: see epoch_floor)
    if (!prof_events()) return XMPI_ERR_HIP;
    a.host_done = last ? done_k : nullptr;
    a.done_value = done_id;
    const bool split = a.nseg > 0 && c->dsync_res &&
                       (split_pref >= 0 ? split_pref != 0 : (c->dsync_split_bytes > 0 && bytes_moved >= (size_t)c->dsync_split_bytes));
    RoctxRange lr("xmpi:launch %s nsrc=%d bytes=%zu epoch=%llu", split ? (c->body_sys ? "meet/body(sys)/done" : "meet/body/done") : "fold",
                  nsrc, bytes_moved, (unsigned long long)c->dsync_epoch);
    if (split) {
      XMPI_HIP(launch_dsync_meet(a, c->dsync_res, stream));
      XMPI_HIP(launch_dsync_body(c->dsync_res, a.nseg, packets + 1, nsrc, kdtype, kop, bytes_moved, c->body_sys != 0, stream,
                                 sampled ? pstart : nullptr, sampled ? pstop : nullptr));
      XMPI_HIP(launch_dsync_done(a, c->dsync_res, stream));
      c->dsync_launches += 3;
      c->dsync_split_launches++;
      return XMPI_OK;
    }
    const int gx = a.nseg > 0 ? dsync_grid(c, packets, a.nseg, unroll) : 1;
    XMPI_HIP(launch_dsync_fold(a, nsrc, kdtype, kop, gx, unroll, stream, sampled ? pstart : nullptr, sampled ? pstop : nullptr));
    c->dsync_launches++;
    return XMPI_OK;
  };

  // (under capture the push-only form is the fold: its staging area is a block the communicator may replace later)
  const bool use_push = push && (coll == COLL_ALLREDUCE || coll == COLL_REDUCE) && !capturing;
  // What this call is, for the peers to compare with theirs (kdev.h dsync_begin): ranks that are not in the same collective, on the
  // same schedule, over the same bytes end with an error before any of them has touched another's memory -- instead of a hang, or
  // of a fold over buffers of different lengths.  (One kernel or meet / body / done is NOT part of it: those mix.)
  {
    const bool reduces = coll == COLL_ALLREDUCE || coll == COLL_REDUCE, rooted = coll == COLL_BCAST || coll == COLL_REDUCE;
    const uint64_t form = stepped ? 16u + 2u * (uint64_t)sched_algo + (sched_push ? 1u : 0u)
                          : use_push ? 2u
                          : (coll == COLL_BCAST && !(N <= 2 || send_bytes <= (size_t)std::max<long>(0, c->zc_bcast_push_bytes))) ? 3u : 1u;
    uint64_t h = 0x9E3779B97F4A7C15ull;
    for (uint64_t v : {(uint64_t)coll + 1, form, (uint64_t)send_bytes, reduces ? (uint64_t)dtype + 1 : 0, reduces ? (uint64_t)op + 1 : 0,
                       rooted ? (uint64_t)root + 1 : 0})
      h = sig_mix(h, v);
    a.sig = sig_fold(h);
  }
  if (stepped) {
    // ring / recursive halving + doubling / binary tree: ONE kernel per rank runs every step of the schedule, the steps
    // released by flag words between the peers' kernels (sched.hip) -- the schedules north_star names, without a host
    // between their steps
    DsyncSchedArgs sa;
    memset(&sa, 0, sizeof sa);
    sa.d = a;
    sa.root = root;
    sa.count = count;
    sa.elem_size = (uint32_t)es;
    sa.pieces = 1;
    sa.push = sched_push ? 1u : 0u;
    size_t step_bytes = send_bytes;  // what the largest step of the schedule moves
    int nchan = 1;
    if (sched_algo == XMPI_ALGO_RING) {
      sa.sched = coll == COLL_ALLREDUCE ? SCHED_RING_ALLREDUCE : SCHED_RING_ALLGATHER;
      step_bytes = coll == COLL_ALLREDUCE ? (send_bytes + (size_t)N - 1) / (size_t)N : send_bytes;
      // every channel is a different cyclic order of the ranks (plan.cpp ring_order: on an even mesh N-2 directed rings
      // that share no link direction); ranks sharing a GPU have no links to spread over
      const int avail = std::min(ring_channel_count(N), kMaxSchedChannels);
      // (the shape of a stepped kernel -- channels, workers -- is protocol: worker w waits for worker w of its peer.  It follows
      // the job's most crowded GPU, which every rank reads alike, not this rank's own: 5 ranks on 2 GPUs sit 3 + 2)
      nchan = c->sched_channels > 0 ? (int)std::min<long>(c->sched_channels, avail) : (c->dsync_sharers_job > 1 ? 1 : avail);
      // (reduce-scatter 2 reads + 1 write per step, allgather 1 + 1; the push form reads its own first chunk once more)
      traffic = coll == COLL_ALLREDUCE ? (5 * (size_t)(N - 1) + (sched_push ? 1 : 0)) * step_bytes : 2 * (size_t)N * send_bytes;
    } else if (sched_algo == XMPI_ALGO_RHD) {
      sa.sched = SCHED_RHD_ALLREDUCE;
      step_bytes = send_bytes / 2;
      // halving: 3 x (S/2 + S/4 + ...), doubling: 2 x the same; push form: the landing regions are written and read -- one more
      traffic = (sched_push ? 6 : 5) * (send_bytes - send_bytes / (size_t)N);
      if ((N & (N - 1)) != 0) traffic += 3 * send_bytes;     // (no power of two: the fold-in / fold-out steps, at most)
    } else {
      sa.sched = coll == COLL_BCAST ? SCHED_TREE_BCAST : SCHED_TREE_REDUCE;
      const size_t piece = (size_t)std::max<long>(4096, c->tree_piece_bytes);
      sa.pieces = (int)std::min<size_t>(32, std::max<size_t>(1, (send_bytes + piece - 1) / piece));
      step_bytes = (send_bytes + (size_t)sa.pieces - 1) / (size_t)sa.pieces;
      const int v = (me - root + N) % N;
      const size_t children = (size_t)((2 * v + 1 < N) + (2 * v + 2 < N));
      if (coll == COLL_BCAST) {  // pull: a node reads its parent's piece and writes its own; push: it reads its own and writes each child's
        traffic = sched_push ? 2 * send_bytes * children : (me == root ? 0 : 2 * send_bytes);
      } else {  // pull: 2 reads + 1 write per child; push: own input + one slot per child read, one buffer stored (upwards, or the result)
        traffic = sched_push ? (children + 2) * send_bytes : 3 * send_bytes * children;
      }
    }
    // push form: the block the peers store into where this rank's receive buffer cannot take their data yet (sched_steps.h
    // sched_land_bytes: in-place ring allreduce -- the receive buffer still is the input; halving -- a region per level; tree
    // reduce -- a slot per child).  Lent per call from the registered arenas: the peers have mapped those long ago.
    sa.d.me = me;
    sa.d.n = N;
    const size_t land_bytes = (size_t)sched_land_bytes(sa, r.send == r.recv);
    c->dsync_land_bytes = land_bytes;
    if (land_bytes) {
      if (capturing) return no_standin();
      void* const own = own_block(land_bytes);
      if (!own) return fail(XMPI_ERR_NOMEM);
      void* land = own;
      BufRef lref;
      if (!zc_export(c, land, land_bytes, &lref)) return fail(XMPI_ERR_HIP);
      int lslot = 0;
      rc = publish(c, lref, &lslot, &pi, capturing);
      if (rc == XMPI_OK) rc = await_acks(c, pi);
      if (rc) return fail(rc);
      sa.d.land_gen = lref.gen;
      sa.d.land_off = lref.offset;
      sa.d.land_slot = (uint64_t)lslot;
      sa.d.my_land = land;
    }
    const size_t tiles = std::max<size_t>(1, (step_bytes + kSchedTileBytes - 1) / kSchedTileBytes);
    const long job_cap = c->dsync_grid_cap > 0 ? c->dsync_grid_cap : 1024 / std::max(1, c->dsync_sharers_job);
    long workers = c->sched_grid > 0 ? c->sched_grid : (long)std::min<size_t>(tiles, (size_t)job_cap);
    workers = std::max<long>(1, std::min<long>(workers, kStepSlots));
    nchan = (int)std::max<long>(1, std::min<long>(nchan, workers));
    const int gx = (int)std::max<long>(1, workers / nchan);
    sa.nchan = nchan;
    // (worker w of a rank waits for worker w of its peer, over the same channels and pieces: the shape is part of the call)
    sa.d.sig = sig_fold(sig_mix(sig_mix(sig_mix(a.sig, (uint64_t)nchan), (uint64_t)gx), (uint64_t)sa.pieces));
    for (int ch = 0; ch < nchan; ch++) {
      std::vector<int> ord;
      ring_order(N, ch, &ord);
      for (int i = 0; i < N; i++) sa.order[ch][i] = (uint8_t)ord[(size_t)i];
    }
    ++c->dsync_epoch;
    if (!prof_events()) return fail(XMPI_ERR_HIP);
    sa.d.host_done = done_k;
    sa.d.done_value = done_id;
    {
      RoctxRange lr("xmpi:launch sched=%d channels=%d workers=%d pieces=%d epoch=%llu", sa.sched, nchan, gx, sa.pieces,
                    (unsigned long long)c->dsync_epoch);
      DS_HIP(launch_dsync_sched(sa, dtype, op, gx, stream, sampled ? pstart : nullptr, sampled ? pstop : nullptr));
    }
    c->dsync_launches++;
    c->dsync_sched_launches++;
    rc = XMPI_OK;
  } else if (use_push) {
    // Push-only (XMPI_ALGO_ZPUSH): the fold with nothing READ over xGMI -- loads over a link are round trips, stores are posted.
    // Two device-synchronised kernels: every rank stores its contribution to chunk q into region `me` of rank q's own block
    // (own_block: the communicator's staging area, announced with the buffers; chunks cut as the fold cuts them, zc_chunk --
    // so in place and ragged counts work like anything else); then, all of chunk `me` being local, folds it in rank order and
    // stores the result into its place in everybody's receive buffer.  One hop each way, S / N per link direction and kernel.
    size_t maxc = 0;
    for (int q = 0; q < N; q++) {
      size_t off = 0, cnt = 0;
      zc_chunk(count, es, N, q, &off, &cnt);
      maxc = std::max(maxc, cnt * es);
    }
    const size_t region = (maxc + 255) / 256 * 256;
    void* const own = own_block(region * (size_t)N);
    if (!own) return fail(XMPI_ERR_NOMEM);
    BufRef lref;
    if (!zc_export(c, own, region * (size_t)N, &lref)) return fail(XMPI_ERR_HIP);
    int lslot = 0;
    rc = publish(c, lref, &lslot, &pi, capturing);
    if (rc == XMPI_OK) rc = await_acks(c, pi);
    if (rc) return fail(rc);
    a.land_gen = lref.gen;
    a.land_off = lref.offset;
    a.land_slot = (uint64_t)lslot;
    a.my_land = own;
    c->dsync_land_bytes = region * (size_t)N;
    split_pref = 0;
    size_t my_off = 0, my_cnt = 0, maxp = 0;
    zc_chunk(count, es, N, me, &my_off, &my_cnt);
    for (int q = 0; q < N; q++) {
      size_t off = 0, cnt = 0;
      zc_chunk(count, es, N, q, &off, &cnt);
      if (q == me || cnt == 0) continue;
      DsyncSeg& g = a.seg[a.nseg++];
      g.src_off = off * es;             // my contribution to chunk q ...
      g.dst_off = (size_t)me * region;  // ... into region `me` of rank q's block
      g.count = cnt * es;
      g.src_mask = 1u << me;
      g.dst_mask = 1u << q;
      g.dst_to_land = 1;
      maxp = std::max(maxp, cnt * es / 16);
      traffic += 2 * cnt * es;
    }
    rc = launch(1, XMPI_U8, XMPI_SUM, maxp, traffic, /*last=*/false);
    if (rc == XMPI_OK) {
      // every contribution to chunk `me` is local now (the first kernel's close: every peer's stores have landed): ONE more kernel
      // folds them in rank order and stores the result into its place in everybody's receive buffer -- the fold's own kernels
      // (one kernel, or meet / body / done by size) with local sources.  Its rendezvous doubles as "my buffers may be written";
      // nobody's block is written again before its owner's next collective has announced it.
      memset(a.seg, 0, sizeof a.seg);
      a.nseg = my_cnt > 0 ? 1 : 0;
      a.seg[0].src_off = a.seg[0].dst_off = my_off * es;
      a.seg[0].count = my_cnt;
      a.seg[0].src_mask = everyone;
      a.seg[0].src_from_recv = 2;
      a.seg[0].stage_stride = region;
      a.seg[0].dst_mask = coll == COLL_REDUCE ? (1u << root) : everyone;  // (reduce: the folded chunks meet in the root's buffer)
      split_pref = -1;  // (meet / body / done by size, like the fold it is)
      const size_t moved = (size_t)(N + (coll == COLL_REDUCE ? 1 : N)) * my_cnt * es;
      traffic += moved;
      rc = launch(N, dtype, op, my_cnt / al, moved);
    }
  } else if (coll == COLL_ALLREDUCE || coll == COLL_REDUCE) {
    size_t off = 0, cnt = 0;
    zc_chunk(count, es, N, me, &off, &cnt);
    a.nseg = cnt > 0 ? 1 : 0;
    a.seg[0].src_off = a.seg[0].dst_off = off * es;
    a.seg[0].count = cnt;
    a.seg[0].src_mask = everyone;
    a.seg[0].dst_mask = coll == COLL_REDUCE ? (1u << root) : everyone;
    traffic = (size_t)(N + (coll == COLL_REDUCE ? 1 : N)) * cnt * es;
    rc = launch(N, dtype, op, cnt / al, traffic);
  } else if (coll == COLL_ALLGATHER) {
    a.nseg = 1;
    a.seg[0].src_off = 0;
    a.seg[0].dst_off = (size_t)me * send_bytes;
    a.seg[0].count = send_bytes;
    a.seg[0].src_mask = 1u << me;
    a.seg[0].dst_mask = everyone;
    traffic = (size_t)(1 + N) * send_bytes;
    rc = launch(1, XMPI_U8, XMPI_SUM, send_bytes / 16, traffic);
  } else {  // COLL_BCAST: `send` and `recv` are the same buffer on every rank
    const bool root_pushes = N <= 2 || send_bytes <= (size_t)std::max<long>(0, c->zc_bcast_push_bytes);
    if (root_pushes) {  // the root stores into every buffer; the others only take part in the rendezvous
      a.nseg = me == root ? 1 : 0;
      a.seg[0].count = send_bytes;
      a.seg[0].src_mask = 1u << root;
      a.seg[0].dst_mask = everyone & ~(1u << root);
      traffic = me == root ? (size_t)N * send_bytes : 0;
      split_pref = 0;  // (the ranks differ in what they launch: keep to the one-kernel form, whose shape does not matter)
      rc = launch(1, XMPI_U8, XMPI_SUM, send_bytes / 16, traffic);
    } else {
      // the root scatters chunk j to rank j (one segment per destination, each over its own link), then every
      // rank forwards its chunk to the others: each link carries S/N twice instead of the root's links carrying S
      size_t maxp = 0;
      split_pref = 0;
      if (me == root) {
        for (int j = 0; j < N; j++) {
          size_t off = 0, cnt = 0;
          zc_chunk(count, es, N, j, &off, &cnt);
          if (j == root || cnt == 0) continue;
          DsyncSeg& g = a.seg[a.nseg++];
          g.src_off = g.dst_off = off * es;
          g.count = cnt * es;
          g.src_mask = 1u << root;
          g.dst_mask = 1u << j;
          maxp = std::max(maxp, cnt * es / 16);
          traffic += 2 * cnt * es;
        }
      }
      rc = launch(1, XMPI_U8, XMPI_SUM, maxp, traffic, /*last=*/false);
      if (rc == XMPI_OK) {
        memset(a.seg, 0, sizeof a.seg);
        size_t off = 0, cnt = 0;
        zc_chunk(count, es, N, me, &off, &cnt);
        a.nseg = cnt > 0 ? 1 : 0;
        a.seg[0].src_off = a.seg[0].dst_off = off * es;
        a.seg[0].count = cnt * es;
        a.seg[0].src_mask = 1u << me;
        a.seg[0].dst_mask = everyone & ~(1u << me) & ~(1u << root);
        traffic += (size_t)(N - 1) * cnt * es;
        rc = launch(1, XMPI_U8, XMPI_SUM, cnt * es / 16, (size_t)(N - 1) * cnt * es);
      }
    }
  }
  if (rc != XMPI_OK) return fail(rc);
  if (!capturing) c->dsync_last_stream = stream;

  if (host_out) {
    DS_HIP(bounce_copy(c, c->host_bounce_dev + xmpi_comm::kHostBounce, out_src, recv_bytes, 1, done_dev, done_id, stream));
    c->host_bounce_calls++;
  }

  // 4. results of a stand-in go home; stand-ins go back to the arena when the stream has passed them.
  //    (A copy into pageable host memory blocks the calling thread until the kernel before it has ended -- and the
  //    kernel ends only when every peer has arrived, which a peer may be unable to do before THIS rank has mapped a
  //    buffer it just registered.  So a blocking call copies out after its polling wait below, which serves the
  //    peers; the stream-ordered forms take device memory only, where the copy really is asynchronous.)
  if (!blocking) {
    if (pstart) c->dsync_prof_pending.push_back({pstart, pstop, traffic});
    if (out_src) DS_HIP(hipMemcpyAsync(recvbuf, out_src, recv_bytes, hipMemcpyDeviceToDevice, stream));
    lent.insert(lent.end(), outgrown.begin(), outgrown.end());
    outgrown.clear();
    if (!lent.empty()) {
      xmpi_comm::DsyncDeferred d;
      if (hipEventCreateWithFlags(&d.done, hipEventDisableTiming) != hipSuccess) return fail(XMPI_ERR_HIP);
      DS_HIP(hipEventRecord(d.done, stream));
      d.bufs = lent;
      c->dsync_deferred.push_back(d);
    }
    return XMPI_OK;
  }

  rc = wait_blocking(c, stream, done_dev != nullptr, done_id);
  if (rc != XMPI_OK) return fail(rc);
  lent.insert(lent.end(), outgrown.begin(), outgrown.end());  // (the stream has passed this collective, and with it every earlier one)
  outgrown.clear();
  if (c->api_calls.load(std::memory_order_relaxed) == calls_at_entry) c->agent_quiet_at = calls_at_entry;  // (as in dsync_ll: the next blocking small collective need not ask the streams)
  if (host_out) {
    memcpy(recvbuf, c->host_bounce + xmpi_comm::kHostBounce, recv_bytes);
  } else if (out_src) {
    DS_HIP(hipMemcpyAsync(recvbuf, out_src, recv_bytes, hipMemcpyDefault, stream));
    DS_HIP(hipStreamSynchronize(stream));
  }
  for (void* p : lent) (void)heap_free(p);
  lent.clear();
  if (pstart) c->dsync_prof_pending.push_back({pstart, pstop, traffic});
  dsync_prof_harvest(c);
  return dsync_check(c);
}

// ---- stream-ordered Send / Receive ------------------------------------------------------------------------------------
// The reference's Send is a gob message on a net.Conn and a wait for the ack message (network.go:562-571), its Receive
// reads, routes by tag, acks and decodes (network.go:575-625).  Here both are ONE kernel each, enqueued on a stream:
// the sender's writes {message number, tag, dtype, bytes, where the payload lives} into a box of the receiver's flag
// allocation (64 bytes over xGMI) and waits for the answer; the receiver's waits for a box with its tag, pulls the
// payload straight out of the sender's HBM and answers.  Nothing is polled by a host thread.  (The blocking
// xmpi_send / xmpi_recv do NOT use waiting kernels -- engine.cpp: a kernel that waits for a peer holds its hardware
// queue, and the reference's semantics let a program block in several Sends / Receives at once, in any order.)

namespace {

int p2p_done_slot(xmpi_comm* c, uint64_t* id_out) {
  const uint64_t id = ++c->p2p_done_next;
  *id_out = id;
  const int slot = (int)(id % xmpi_comm::kP2PDoneSlots);
  __atomic_store_n(&c->p2p_done[4 * slot], 0, __ATOMIC_RELEASE);
  return slot;
}

int p2p_status_to_rc(uint64_t st) {
  if (st == 0) return XMPI_OK;
  if (st >= 0x100) {
    const uint32_t why = (uint32_t)(st - 0x100);
    if (why == DSYNC_TIMEOUT) {
      set_last_error("send / receive: the peer did not arrive within XMPI_TIMEOUT_S (kernel wait cut short)");
      return XMPI_ERR_TIMEOUT;
    }
    if (why == DSYNC_UNMAPPED) {
      set_last_error("receive: the sender's buffer is not mapped here");
      return XMPI_ERR_STATE;
    }
    set_last_error("send / receive: the job was aborted while the kernel waited");  // (callers with the communicator at hand say why: abort_reason)
    return XMPI_ERR_PEER;
  }
  if (st == 6) set_last_error("receive: the message does not fit the buffer");
  else set_last_error("receive: dtype differs from the sender's");
  return -(int)st;
}

void p2p_fill(xmpi_comm* c, P2PArgs* a, int peer, int tag, int dtype) {
  memset(a, 0, sizeof *a);
  a->my_page = c->dpage;
  a->peer_page = c->peer_page[peer];
  a->me = c->rank;
  a->peer = peer;
  a->tag = tag;
  a->dtype = dtype;
  a->comm_tag = c->dsync_tag;
  a->table = c->dsync_table_dev;
  a->abort_word = c->dsync_abort_dev;
  a->spin_limit = c->timeout_s > 0 ? (uint64_t)c->timeout_s * 100000000ull : 0;
}

}  // namespace

// the operations that have completed since the last look: their stand-ins go back, the first failure is returned
int dsync_p2p_reap(xmpi_comm* c) {
  int rc = XMPI_OK;
  for (size_t i = 0; i < c->p2p_pending.size();) {
    xmpi_comm::P2PPending& p = c->p2p_pending[i];
    if (__atomic_load_n(&c->p2p_done[4 * p.slot], __ATOMIC_ACQUIRE) != p.id) {
      i++;
      continue;
    }
    const int r = p2p_status_to_rc(c->p2p_done[4 * p.slot + 1]);
    if (r != XMPI_OK && rc == XMPI_OK) rc = r;
    for (void* b : p.bufs) (void)heap_free(b);
    c->p2p_pending.erase(c->p2p_pending.begin() + (long)i);
  }
  return rc;
}

int dsync_send(xmpi_comm* c, const void* buf, size_t bytes, int dtype, int dest, int tag, hipStream_t stream) {
  RoctxRange range("xmpi:send_on_stream dest=%d tag=%d bytes=%zu", dest, tag, bytes);
  if (!stream) stream = c->local_stream;
  dsync_service(c);
  if ((int)c->p2p_pending.size() >= xmpi_comm::kP2PDoneSlots - 2) {  // completion words are a ring: do not lap it
    set_last_error("too many stream-ordered sends / receives outstanding: xmpi_stream_sync first");
    return XMPI_ERR_STATE;
  }
  BufRef ref;
  memset(&ref, 0, sizeof ref);
  std::vector<void*> lent;
  int slot = 0;
  if (bytes > 0) {
    const void* src = buf;
    if (!zc_export(c, src, bytes, &ref)) {  // memory the receiver cannot map: a registered stand-in (one local copy)
      void* tmp = heap_alloc(c->device, bytes);
      if (!tmp) return XMPI_ERR_NOMEM;
      lent.push_back(tmp);
      XMPI_HIP(hipMemcpyAsync(tmp, buf, bytes, hipMemcpyDeviceToDevice, stream));
      if (!zc_export(c, tmp, bytes, &ref)) {
        (void)heap_free(tmp);
        return XMPI_ERR_HIP;
      }
      c->dsync_bounced++;
    }
    uint64_t pi = 0;
    int rc = publish(c, ref, &slot, &pi);
    if (rc == XMPI_OK) rc = await_acks(c, pi);
    if (rc != XMPI_OK) {
      for (void* p : lent) (void)heap_free(p);
      return rc;
    }
  }
  P2PArgs a;
  p2p_fill(c, &a, dest, tag, dtype);
  a.seq = ((c->dsync_tag & 0xffffffffull) << 32) | (c->p2p_out_seq[dest] + 1);  // (consumed below, once the kernel is enqueued)
  a.bytes = bytes;
  a.gen = ref.gen;
  a.slot = (uint64_t)slot;
  a.off = ref.offset;
  uint64_t id = 0;
  const int ds = p2p_done_slot(c, &id);
  a.host_done = c->p2p_done_dev + 4 * ds;
  a.done_value = id;
  if (launch_p2p_send(a, stream) != hipSuccess) {
    // message n was never posted: its number must not be consumed (message n + 8 would wait for its ack for ever)
    const int rc = hip_fail(hipGetLastError(), "p2p send kernel", __FILE__, __LINE__);
    for (void* p : lent) (void)heap_free(p);
    return rc;
  }
  ++c->p2p_out_seq[dest];
  c->p2p_pending.push_back({ds, id, lent});
  return XMPI_OK;
}

int dsync_recv(xmpi_comm* c, void* buf, size_t cap_bytes, int dtype, int src, int tag, hipStream_t stream) {
  RoctxRange range("xmpi:recv_on_stream src=%d tag=%d capacity=%zu", src, tag, cap_bytes);
  if (!stream) stream = c->local_stream;
  dsync_service(c);
  if ((int)c->p2p_pending.size() >= xmpi_comm::kP2PDoneSlots - 2) {
    set_last_error("too many stream-ordered sends / receives outstanding: xmpi_stream_sync first");
    return XMPI_ERR_STATE;
  }
  P2PArgs a;
  p2p_fill(c, &a, src, tag, dtype);
  a.bytes = cap_bytes;
  a.buf = buf;
  // unique per PAGE, not per communicator: the page is pooled and never cleared, a later communicator's first receive must
  // not match the go record an earlier one left behind (the communicator number is what a.seq carries as well)
  a.op_id = ((c->dsync_tag & 0xffffffffull) << 32) | (++c->p2p_op_id & 0xffffffffull);
  uint64_t id = 0;
  const int ds = p2p_done_slot(c, &id);
  a.host_done = c->p2p_done_dev + 4 * ds;
  a.done_value = id;
  // a few blocks for a large message; every block but the first only waits for the first (a local word)
  long gx = (long)((cap_bytes + 16383) >> 14);
  gx = std::max<long>(1, std::min<long>(gx, c->dsync_sharers > 1 ? 16 : 128));
  XMPI_HIP(launch_p2p_recv(a, (int)gx, stream));
  c->p2p_pending.push_back({ds, id, {}});
  return XMPI_OK;
}

// the events of sampled launches that have ended (a launch that was only enqueued leaves them for a later look) go into
// the profile counters
void dsync_prof_harvest(xmpi_comm* c) {
  for (size_t i = 0; i < c->dsync_prof_pending.size();) {
    auto& p = c->dsync_prof_pending[i];
    float ms = 0.f;
    if (hipEventQuery(p.stop) != hipSuccess || hipEventElapsedTime(&ms, p.start, p.stop) != hipSuccess) {
      (void)hipGetLastError();
      i++;
      continue;
    }
    ProfCounter& pc = c->prof[PROF_ZCOPY];
    pc.add(ms, p.bytes);
    ev_put(c, p.start, true);
    ev_put(c, p.stop, true);
    c->dsync_prof_pending.erase(c->dsync_prof_pending.begin() + (long)i);
  }
}

// the first failure a kernel of this rank reported since the last look (a wait that was cut short, a buffer
// reference it could not translate); clears it
int dsync_check(xmpi_comm* c) {
  if (!c->dsync_status) return XMPI_OK;
  const uint32_t st = __atomic_exchange_n(c->dsync_status, 0u, __ATOMIC_ACQ_REL);
  if (st == DSYNC_OK) return XMPI_OK;
  int rc = XMPI_ERR_PEER;
  if (st == DSYNC_TIMEOUT) {
    set_last_error("collective: a peer did not arrive within XMPI_TIMEOUT_S (kernel wait cut short)");
    rc = XMPI_ERR_TIMEOUT;
  } else if (st == DSYNC_UNMAPPED) {
    set_last_error("collective: a peer's buffer is not mapped here (registration freed while in use?)");
    rc = XMPI_ERR_STATE;
  } else if (st == DSYNC_MISMATCH) {
    set_last_error("collective: the ranks are not in the same call (collective, schedule, length, dtype, operation or root differ "
                   "between this rank and a peer); nothing was moved");
    rc = XMPI_ERR_ARG;
  } else if (st == DSYNC_XCD) {
    c->xcd_short++;
    set_last_error("collective: the meet / done kernels of the split form did not reach every XCD's L2 (masks in xcd_meet_mask / "
                   "xcd_done_mask); set XMPI_BODY_SYS=1");
    rc = XMPI_ERR_STATE;
  } else {
    set_last_error("collective: the job was aborted while the kernel waited for a peer: " + c->ctl->abort_reason());
  }
  c->ctl->set_abort(rc);
  return rc;
}

}  // namespace xmpi
