// comm.h -- the communicator object behind the C ABI (include/xmpi.h).
#pragma once
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <thread>
#include <mutex>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "../../include/xmpi.h"
#include "../../include/xmpi_test.h"
#include "ctl.h"
#include "kernels.h"
#include "trace.h"
#include "plan.h"

namespace xmpi {

void set_last_error(const std::string& s);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define XMPI_HIP(call)                                                     \
  do {                                                                     \
    hipError_t _e = (call);                                                \
    if (_e != hipSuccess) return ::xmpi::hip_fail(_e, #call, __FILE__, __LINE__); \
  } while (0)

struct ProfCounter {
  uint64_t launches = 0;
  double total_ms = 0.0;
  uint64_t bytes = 0;
  double min_ms = 0.0, max_ms = 0.0;  // of the sampled launches (0: none yet)
  void add(double ms, uint64_t nbytes) {
    min_ms = launches ? (ms < min_ms ? ms : min_ms) : ms;
    max_ms = launches ? (ms > max_ms ? ms : max_ms) : ms;
    launches++;
    total_ms += ms;
    bytes += nbytes;
  }
};

enum ProfKind { PROF_REDUCE2 = 0, PROF_REDUCEN = 1, PROF_COPY = 2, PROF_PEER = 3, PROF_ZCOPY = 4, PROF_KINDS = 5 };

}  // namespace xmpi

// a non-blocking collective in flight (xmpi_iallreduce ...): completed by the communicator's worker
struct xmpi_request {
  std::mutex mu;
  std::condition_variable cv;
  bool done = false;
  int rc = 0;
  std::string err;  // the worker's error text, handed to the thread that waits
};

struct xmpi_comm {
  int rank = 0, size = 0, device = 0;
  xmpi::Ctl* ctl = nullptr;

  // HBM receive window of this rank and the mapped windows of the peers
  char* window = nullptr;
  size_t window_bytes = 0;
  char* peer_window[xmpi::kMaxRanks] = {nullptr};
  bool peer_opened[xmpi::kMaxRanks] = {false};
  // What the job agreed it can do (xmpi_init's vote, dsync_connect): a rank that cannot map a peer's window or flag page does not
  // fail the job -- every rank publishes what it mapped, and all of them take the best level everybody reached (network.go:53-65:
  // Init returns an error only when the mesh cannot be built).
  bool windows_ok = true;        // every rank mapped every peer's window: the staged step tables and the mail slots exist
  bool window_map_failed = false;  // ... this rank could not (before the vote)
  std::string degraded_why;      // "" = nothing degraded; otherwise which level the job runs at and the first reason a rank gave

  // window layout (identical on every rank)
  int lanes = 2, fifo_depth = 8, p2p_depth = 2;
  size_t slot_bytes = 4u << 20, p2p_slot_bytes = 4u << 20;
  size_t coll_region_bytes = 0;
  size_t coll_slot_off(int src, int lane, uint64_t seq) const {
    return (((size_t)src * lanes + lane) * fifo_depth + (size_t)(seq % (uint64_t)fifo_depth)) * slot_bytes;
  }
  size_t p2p_slot_off(int src, int entry, uint64_t seq) const {
    return coll_region_bytes +
           (((size_t)src * xmpi::kMailEntries + entry) * p2p_depth + (size_t)(seq % (uint64_t)p2p_depth)) *
               p2p_slot_bytes;
  }

  // streams: one per peer and direction so independent links progress independently
  hipStream_t send_stream[xmpi::kMaxRanks] = {nullptr};
  hipStream_t recv_stream[xmpi::kMaxRanks] = {nullptr};
  hipStream_t local_stream = nullptr;
  hipStream_t batch_send_stream = nullptr, batch_recv_stream = nullptr;  // multi-destination copy launches
  // zero-copy collectives (zcopy.cpp): registered user buffers are read / written in place by peers
  long zero_copy = 1;            // AUTO may choose the zero-copy path (all buffers registered HBM)
  long zc_bcast_push_bytes = 256 << 10;  // bcast up to this size: root pushes to everyone; above: scatter + allgather
  long zc_group_launch = 1;      // co-located ranks (threads on one GPU): the lowest folds all their chunks in one launch
  uint64_t zc_fallbacks_unregistered = 0, zc_fallbacks_unmappable = 0;  // zero-copy attempts that went staged, by reason
  uint64_t zc_seq = 0;          // zero-copy attempts so far (same on every rank)
  std::set<std::pair<uint64_t, uint64_t>> zc_announced;  // {base, gen} already published on this communicator
  std::mutex zc_mu;              // guards zc_retired_seen (collectives and p2p calls process the retire logs)
  long p2p_direct_bytes = 1;  // messages from a registered buffer at least this long: the receiver pulls them directly
  uint64_t p2p_direct_count = 0, p2p_staged_count = 0, p2p_lane_count = 0;  // receives served each way (diagnostic)
  uint64_t zc_retired_seen[xmpi::kMaxRanks] = {0};      // how far each peer's retire log has been processed
  long oneshot_bytes = 1 << 20;  // direct allreduce up to this size: push everything, fold locally
  long batch_copies = 1;  // with the copy kernel: all ready SENDs (RECV_COPYs) go out in one launch
  bool shared_stream = false;  // all of the above alias one per-device stream (co-located ranks)
  bool staged_streams = false;  // the per-peer / batch streams exist (made when a staged schedule first runs)
  bool peer_coloc[xmpi::kMaxRanks] = {false};  // peer is a thread of this process on this GPU
  long prof_every = 1;  // profile every k-th launch (events cost stream bubbles)
  uint64_t prof_seq[xmpi::PROF_KINDS] = {0};

  // device-synchronised collectives (dsync.cpp): ranks meet through flag words in HBM, the host only enqueues
  long dsync = 1;                // AUTO / ZCOPY may take the device-synchronised path (XMPI_DSYNC)
  bool dsync_ok = false;         // ... and this job can: every rank has a flag page, no two ranks share a (process, GPU)
  xmpi::DsyncPage* dpage = nullptr;                       // this rank's flag page (uncached HBM)
  xmpi::DsyncPage* peer_page[xmpi::kMaxRanks] = {nullptr};  // everybody's, as addressable from here
  bool peer_page_opened[xmpi::kMaxRanks] = {false};
  const int32_t* dsync_abort_dev = nullptr;  // the job's abort flag as the GPU reads it
  char* ctl_dev = nullptr;        // the control block as the GPU addresses it (hipHostRegister; kernels read the abort flag and
  bool ctl_registered = false;    //   write the ack of a point-to-point message straight into its mail entry)
  xmpi::DsyncResolved* dsync_res = nullptr;  // split form: what the meet kernel leaves for the data kernel (device memory)
  hipEvent_t dsync_order_ev = nullptr;       // recorded behind every device-synchronised launch: a launch on ANOTHER stream waits for it
  hipStream_t dsync_last_stream = nullptr;   //   (the kernels of one rank share the page's epoch, ticket and slots: one at a time)
  long dsync_split_bytes = 4 << 20;  // collectives moving at least this much per rank run as meet / body / done (0 = never)
  long sched_channels = 0;           // ring channels of the stepped kernels; 0 = every link direction (1 when ranks share a GPU)
  long sched_grid = 0;               // workers (blocks) of a stepped kernel; 0 = by size, bounded like dsync_grid_cap
  long tree_piece_bytes = 256 << 10; // binary-tree broadcast: pieces of this size travel down the tree pipelined
  // stream-ordered Send / Receive (dsync.cpp p2p_*_on_stream): message numbers per ordered pair, completion words
  uint64_t p2p_out_seq[xmpi::kMaxRanks] = {0};
  uint64_t p2p_op_id = 0;
  static constexpr int kP2PDoneSlots = 64;  // ... of the stream-ordered forms; as many again for the blocking Receive's pull kernels
  uint64_t* p2p_done = nullptr;      // pinned host: 4 words per slot {done id, status, bytes, -}
  uint64_t* p2p_done_dev = nullptr;
  uint64_t p2p_done_next = 0;
  std::atomic<uint64_t> p2p_pull_next{0};
  long p2p_grid_cap = 0;             // blocks of the pull kernel; 0 = from where the payload lies (engine.cpp p2p_pull_cap)
  double link_gbps[xmpi::kMaxRanks] = {0};  // what xmpi_link_probe measured towards each peer (best of its calls)
  long p2p_kernel_ack = 1;           // blocking Receive: one kernel copies AND acks (0: hipMemcpyAsync / copy kernel + event + host ack)
  long p2p_agent_us = 40;            // ... by a kernel that stays this long after a message (the receive agent, sched.hip): the next
                                     // Receive hands it a command instead of launching again (0: one launch per message)
  uint64_t* p2p_cmd = nullptr;       // pinned host: the agent's command record (8 words)
  uint64_t* p2p_cmd_dev = nullptr;
  uint64_t* p2p_rec = nullptr;       // device: 8 words (block 0 -> other blocks)
  std::mutex agent_mu;               // one command at a time
  uint64_t agent_seq = 0;            // number of the last command written
  bool agent_running = false;        // launched and not yet known to have gone
  hipStream_t agent_stream = nullptr;
  uint64_t p2p_agent_served = 0, p2p_agent_launches = 0;  // messages the agent copied; times it had to be launched
  uint64_t p2p_agent_launch_no = 0;  // ... numbering those launches (P2PAgentArgs::launch): functional, never reset
  struct P2PPending {
    int slot;
    uint64_t id;
    std::vector<void*> bufs;  // stand-ins to give back
  };
  std::vector<P2PPending> p2p_pending;  // stream-ordered operations whose completion nobody has looked at yet
  // host slices in a blocking collective (dsync.cpp): pinned memory the GPU reads / writes directly -- [0, kHostBounce) on the
  // way in, [kHostBounce, 2 kHostBounce) on the way out; allocated when the first such call comes
  static constexpr size_t kHostBounce = 256u << 10;
  char* host_bounce = nullptr;
  char* host_bounce_dev = nullptr;
  uint64_t host_bounce_calls = 0;
  bool lanes_dev_ok = false;  // the host lanes are pinned and mapped: kernels / DMA engines read them (ctl_dev + offset)
  std::mutex p2p_bounce_mu;   // a message out of a peer's HBM into a host slice: through this pinned block (engine.cpp)
  char* p2p_bounce = nullptr;
  char* p2p_bounce_dev = nullptr;
  uint32_t* p2p_tickets = nullptr;      // device: block counters of the pull kernels (blocking Receive), one per done slot
  // the library's own schedule table (xmpi_tune): algorithm per collective and size class, agreed by all ranks
  static constexpr int kTuneClasses = 24;  // class k: messages of [2^(k+8), 2^(k+9)) bytes per rank
  int8_t tune_algo[4][kTuneClasses];       // [collective][class]: an xmpi_algo, or -1 = not tuned (built-in rule)
  int8_t tune_split[4][kTuneClasses];      // 1 = split form, 0 = one kernel, -1 = by dsync_split_bytes
  int8_t tune_unroll[4][kTuneClasses];
  bool tuned = false;
  long tune_mask = -1;  // candidates xmpi_tune may time (bit per candidate; see xmpi_set_param "tune_mask")
  // The candidates by number: what tune_mask and tune_rejected_<collective> name bit by bit.
  enum { CAND_FOLD = 0, CAND_FOLD_U2 = 1, CAND_SPLIT = 2, CAND_ZPUSH = 3, CAND_RING = 4, CAND_RHD = 5, CAND_LL = 6, CAND_RING_PUSH = 7,
         CAND_RHD_PUSH = 8, CAND_TREE = 9, CAND_TREE_PUSH = 10, CAND_COUNT = 11 };
  // Schedules whose ANSWERS were wrong on this machine (xmpi_tune and xmpi_init's self-check run every candidate once on patterned
  // inputs whose sum is exact in any association, compare with the locally computed result and vote: wrong on ANY rank = rejected on
  // EVERY rank).  A rejected schedule is never AUTO's choice again and a caller who names it is refused -- the same on every rank.
  uint32_t tune_rejected[4] = {0, 0, 0, 0};
  uint32_t p2p_rejected = 0;     // ... and ways of moving a message (the self-check): bit 0 the receiver's direct pull, bit 1 every way
  bool tune_running = false;     // xmpi_tune / the self-check are running candidates themselves (no refusal)
  std::string rejected_why;      // which, where first, how wrong (also appended to degraded_why)
  long selfcheck = -1;           // xmpi_init checks what untuned AUTO can reach; XMPI_SELFCHECK (-1: when the ranks sit on different GPUs)
  double selfcheck_ms = -1;      // what it cost (-1: did not run)
  double selfcheck_setup_ms = 0; // ... of which: its buffers (the first arena: allocated, exported, mapped by every peer)
  double tune_ms = 0, tune_check_ms = 0;  // the last xmpi_tune: all of it / the part spent checking answers
  uint32_t* dsync_status = nullptr;          // pinned host word a kernel writes its first failure to ...
  uint32_t* dsync_status_dev = nullptr;      // ... and its device address
  int xcds = 0;                  // XCDs of this rank's GPU as a 1024-block probe grid found them (0: not probed)
  uint32_t xcd_probe_mask = 0;   // XCDs a grid of kXcdBlocks one-wave blocks reached at start-up (what the meet / done kernels are)
  long xcd_check = 1;            // the split form's done kernel fails the collective when meet or done missed an XCD; XMPI_XCD_CHECK
  long body_sys = 0;             // split form: data kernel with system-scope loads / stores (no L2 assumption); XMPI_BODY_SYS
  uint64_t xcd_short = 0;        // split collectives whose meet or done kernel missed an XCD (reported whether checked or not)
  int dsync_sharers = 1;         // ranks of this job on this rank's GPU (bounds the grid: their kernels spin together)
  int dsync_sharers_job = 1;     // ... on the job's most crowded GPU (the same figure on every rank: what shapes a protocol)
  long dsync_grid_cap = 0;       // blocks per kernel; 0 = 1024 / sharers
  long dsync_unroll = 1;         // 16-byte packets per lane per source in flight (2 = deeper, for links)
  long dsync_tiles = 1;          // tiles (256 lanes x unroll packets) a block walks before the grid grows
  uint64_t dsync_epoch = 0;      // epoch of the last kernel launched: the same number on every rank
  uint64_t dsync_done_seq = 0;   // numbers the calls for the pinned "done" word a blocking call polls
  uint64_t dsync_base = 0;       // where this communicator's epochs start (epoch_floor of its kernels)
  uint64_t dsync_tag = 1;        // tags this communicator's entries in the (pooled, uncleared) page's translation cache
  uint64_t dsync_launches = 0, dsync_bounced = 0;  // diagnostics: kernels; buffers stood in for by arena blocks
  long ll_bytes = 0;             // untuned AUTO: collectives up to this many bytes per rank go as LL lines (ll.hip); XMPI_LL_BYTES
  uint64_t dsync_ll_launches = 0;  // ... collectives that did
  long agent_ll = 1;             // a BLOCKING LL collective of up to agent_ll_bytes per rank is handed to the lingering LL agent
  long agent_ll_bytes = 8192;    // (ll.hip ll_agent_kernel) instead of being launched; XMPI_AGENT_LL, XMPI_AGENT_LL_BYTES
  long ll_agent_us = 40;         // how long that kernel lingers after a collective; XMPI_LL_AGENT_US (default: XMPI_P2P_AGENT_US's value)
  uint64_t dsync_ll_agent = 0;   // ... collectives the agent ran
  uint64_t* ll_cmd = nullptr;    // the LL agent's command record (pinned host, 8 words: the second half of p2p_cmd's allocation)
  uint64_t* ll_cmd_dev = nullptr;
  uint64_t ll_agent_seq = 0;     // number of the last command written (callers hold coll_mu)
  bool ll_agent_running = false; // launched and not yet known to have gone
  hipStream_t ll_agent_stream = nullptr;
  uint64_t ll_agent_launches = 0;
  uint64_t agent_ll_wait_ns = 0; // diagnostics: time between writing a command and seeing its answer, summed
  std::atomic<uint64_t> api_calls{0};  // public entry points taken on this communicator (XMPI_ENTER) ...
  uint64_t agent_quiet_at = ~0ull;     // ... and its value when a BLOCKING device-synchronised collective last returned (whoever ran it):
                                       // the NEXT call knows that nothing was enqueued through the library in between without asking
                                       // the streams (hipStreamQuery of a stream whose last kernel is still retiring costs ~10 us)
  uint64_t agent_epoch_at = ~0ull;     // ... and when the AGENT last ran one: the next call's epoch is that one's plus one
  uint64_t dsync_split_launches = 0, dsync_sched_launches = 0;  // ... collectives run as meet / body / done; as a stepped kernel
  uint64_t dsync_land_bytes = 0;   // the landing block this rank lent to its last push-form stepped collective (0: none)
  void* land_block = nullptr;      // ... kept for the next one (a registered arena block; grown when a collective needs more)
  size_t land_block_bytes = 0;
  xmpi::DsyncEntry* dsync_table = nullptr;            // [kMaxRanks][kDsyncArenas], pinned host memory the kernels read
  const xmpi::DsyncEntry* dsync_table_dev = nullptr;  // ... as the GPU addresses it
  uint64_t dsync_seen[xmpi::kMaxRanks] = {0};         // published entries of each peer processed so far
  uint64_t dsync_slot_gen[xmpi::kDsyncArenas] = {0};  // my registrations the peers hold, by table slot
  uint64_t dsync_slot_pub[xmpi::kDsyncArenas] = {0};  // ... published as entry number (1-based)
  uint64_t dsync_slot_used[xmpi::kDsyncArenas] = {0};  // ... last used by epoch
  std::mutex dsync_mu;           // dsync_service may be entered from any thread of the rank
  std::thread dsync_helper;      // ... and from this one, once a millisecond, whatever the rank's own threads do
  std::atomic<bool> dsync_helper_stop{false};
  struct DsyncDeferred {
    hipEvent_t done;
    std::vector<void*> bufs;
  };
  std::vector<DsyncDeferred> dsync_deferred;  // arena blocks lent to collectives still on a stream
  std::vector<void*> dsync_leaked;            // ... whose event could not even be recorded: given back at finalize
  uint64_t dsync_selftest_token = 0;          // what xmpi_init's flag self-test left in the peers' pages (0: it did not run)
  struct DsyncProf {
    hipEvent_t start, stop;
    size_t bytes;
  };
  std::vector<DsyncProf> dsync_prof_pending;  // sampled launches that were only enqueued: read by the next blocking call

  // per collective pipe: slots issued / consumed so far (monotonic across operations)
  uint64_t sent[xmpi::kMaxRanks][xmpi::kMaxLanes] = {{0}};
  uint64_t recvd[xmpi::kMaxRanks][xmpi::kMaxLanes] = {{0}};
  // ... and how many of those have been published to the peer (head of my sends, tail of my receives)
  uint64_t sent_done[xmpi::kMaxRanks][xmpi::kMaxLanes] = {{0}};
  uint64_t recvd_released[xmpi::kMaxRanks][xmpi::kMaxLanes] = {{0}};

  // tunables (xmpi_set_param)
  long channels = 0;  // ring channels; 0 = all edge-disjoint directed rings of the mesh
  long piece_bytes = 0;  // 0 = choose per operation
  long copy_engine = 0;  // 0 = hipMemcpyAsync (SDMA / runtime blit), 1 = xmpi copy kernel
  long timeout_s = 0;  // no-progress limit of steady-state waits in seconds; 0 = for ever (the reference blocks indefinitely)
  long watchdog_ms = 50;  // how often the helper thread asks whether the peers' processes still exist (0 = never); XMPI_WATCHDOG_MS
  double last_run_us = 0, last_sync_us = 0;  // timing of the most recent collective (diagnostic)

  // scratch
  void* temp = nullptr;
  size_t temp_bytes = 0;
  void* host_stage = nullptr;  // HBM bounce buffer for host-resident p2p payloads
  uint64_t* dev_words = nullptr;  // 4 x u64 result words for the verification kernels

  // event pools
  std::mutex ev_mu;
  std::vector<hipEvent_t> ev_free, ev_timed_free;

  // profiling
  bool prof_on = false;
  xmpi::ProfCounter prof[xmpi::PROF_KINDS];

  std::mutex coll_mu;  // collectives are serialised per communicator
  std::mutex p2p_mu;   // guards the tag registries and the p2p stream pool
  std::set<std::pair<int, int>> send_tags, recv_tags;  // active {peer, tag} (network.go:448-497)
  std::map<std::pair<int, int>, xmpi::MailEntry*> pending_sends;  // xmpi_send_nowait awaiting xmpi_wait
  std::vector<hipStream_t> p2p_streams;

  // non-blocking collectives: one worker per communicator runs them in the order they were issued
  // (every rank issues them in the same order, as with the blocking ones)
  std::thread worker;
  std::mutex wq_mu;
  std::condition_variable wq_cv;
  std::deque<std::pair<std::function<int()>, xmpi_request*>> wq;
  size_t wq_busy = 0;  // queued + running
  bool wq_stop = false, worker_started = false;
  bool finalized = false;
};

namespace xmpi {
int ensure_streams(xmpi_comm* c);
int run_plan(xmpi_comm* c, const Plan& plan, const void* sendbuf, void* recvbuf, int dtype, int op);
int p2p_send(xmpi_comm* c, const void* buf, size_t bytes, int dtype, int dest, int tag, bool wait_ack = true);
int p2p_wait(xmpi_comm* c, int dest, int tag);
int p2p_recv(xmpi_comm* c, void* buf, size_t cap_bytes, int dtype, int src, int tag, size_t* got_bytes);
int p2p_probe(xmpi_comm* c, int src, int tag, size_t* bytes, int* dtype);
void p2p_agent_stop(xmpi_comm* c);
// consecutive: the previous call into the library on this communicator was a collective the agent ran (its epoch + 1 is this one's)
// 1 = done, 0 = not taken (launch instead), -1 = failed after it was taken (engine.cpp)
int agent_submit_ll(xmpi_comm* c, const void* send, void* recv, size_t bytes, int ll_coll, int root, int dtype, int op, bool consecutive);
// the number of the public call this thread is in (XMPI_ENTER: the value its fetch_add of api_calls gave back, plus one)
extern thread_local uint64_t t_api_call;
void ll_agent_stop(xmpi_comm* c);
// zcopy.cpp.  *done = false: some rank's buffers are not registered HBM -- every rank saw that and
// the caller runs the staged schedule instead (no rank is left behind: the decision is collective).
int zero_copy_collective(xmpi_comm* c, int coll, int root, const void* sendbuf, void* recvbuf, size_t count,
                         int dtype, int op, bool push, bool* done, int iters = 1);
int registry_add(void* base, size_t bytes, int device);
void registry_remove(xmpi_comm* c, void* base);
void zc_close_peers(const xmpi_comm* c);
bool zc_export(xmpi_comm* c, const void* p, size_t need, BufRef* ref);
bool zc_import(xmpi_comm* c, int peer, const BufRef& ref, void** out);
bool registry_alive(uint64_t gen);
// api.cpp: blocks that peers map (windows, flag pages) and the mappings of the peers' blocks are kept per
// process across communicators (exported memory is not given back by the runtime while the processes live)
hipError_t ipc_open_shared(int owner_pid, uint64_t owner_addr, const void* handle_bytes, void** out);
void ipc_close_shared(void* ptr);
void* pool_acquire(int device, size_t bytes, int kind, bool* fresh, uint64_t* mark);
void pool_release(void* ptr, uint64_t mark = 0);
hipError_t pool_handle(void* ptr, void* handle_out);
hipStream_t stream_acquire(int device);
void stream_release(int device, hipStream_t s);
// dsync.cpp
int dsync_prepare(xmpi_comm* c);
int dsync_connect(xmpi_comm* c, double timeout_s);
void dsync_finalize(xmpi_comm* c);
void dsync_start_helper(xmpi_comm* c);
void dsync_stop_helper(xmpi_comm* c);
void dsync_service(xmpi_comm* c);
bool dsync_usable(const xmpi_comm* c);
// algo: AUTO / ZCOPY (one fold per rank: one kernel, or meet / body / done), ZPUSH, RING, RHD, TREE (the stepped kernels)
int dsync_collective(xmpi_comm* c, int coll, int root, const void* sendbuf, void* recvbuf, size_t count, int dtype,
                     int op, hipStream_t stream, bool blocking, int algo = 0);
bool dsync_takes(const xmpi_comm* c, int coll, int algo);
void dsync_graph_launched(xmpi_comm* c, hipStream_t stream, bool before);
int dsync_send(xmpi_comm* c, const void* buf, size_t bytes, int dtype, int dest, int tag, hipStream_t stream);
int dsync_recv(xmpi_comm* c, void* buf, size_t cap_bytes, int dtype, int src, int tag, hipStream_t stream);
int dsync_p2p_reap(xmpi_comm* c);
int dsync_check(xmpi_comm* c);
void dsync_prof_harvest(xmpi_comm* c);
// a rank that waits keeps serving its peers
inline void arm(Backoff& bo, xmpi_comm* c) {
  if (c->dsync_ok) {
    bo.idle = [](void* p) { dsync_service((xmpi_comm*)p); };
    bo.idle_arg = c;
  }
}
// heap.cpp: xmpi_malloc carves buffers out of long-lived registered arenas
void* heap_alloc(int device, size_t bytes);
bool heap_free(void* p);
bool heap_owns(const void* p);
void heap_comm_created();
void heap_colour_seed(int rank, bool ranks_share_this_process);  // before heap_comm_created: which rank this process hosts (colours of large blocks)
void heap_comm_destroyed(xmpi_comm* c);
void heap_stats(int device, size_t* arenas, size_t* reserved, size_t* in_use);
int heap_selftest(uint64_t seed, int rounds);
hipEvent_t ev_get(xmpi_comm* c, bool timed);
void ev_put(xmpi_comm* c, hipEvent_t e, bool timed);
bool is_device_pointer(const void* p);
}  // namespace xmpi
