// kernels.h -- internal launch API of the gfx950 kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

namespace xmpi {

constexpr int kMaxReduceSrcs = 16;

// dst[i] = a[i] op b[i]; dst may alias a and/or b exactly (same address), never partially.
// ev_start / ev_stop (optional, timing-enabled events): attached to the dispatch itself
// (hipExtLaunchKernelGGL), i.e. they carry the kernel's own begin / end timestamps.
hipError_t launch_reduce2(void* dst, const void* a, const void* b, size_t count, int dtype, int op,
                          hipStream_t stream, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
// dst[i] = ((s0[i] op s1[i]) op s2[i]) ... left to right.
hipError_t launch_reduce_n(void* dst, const void* const* srcs, int nsrc, size_t count, int dtype,
                           int op, hipStream_t stream, hipEvent_t ev_start = nullptr,
                           hipEvent_t ev_stop = nullptr);
// streaming copy (local HBM -> local HBM, or local HBM -> peer HBM over xGMI)
hipError_t launch_copy(void* dst, const void* src, size_t bytes, hipStream_t stream,
                       hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);

// n <= kMaxBatch independent copies in one launch (a piece pushed to every peer / every peer's slot
// drained at once); falls back to one launch per copy for odd alignments
constexpr int kMaxBatch = 16;
// dst2 (optional array; entries may be null): a second destination of each copy
hipError_t launch_copy_batch(void* const* dst, void* const* dst2, const void* const* src, const size_t* bytes,
                             int n, hipStream_t stream, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);

// n <= kMaxBatch independent dst = a op b reductions in one launch (the ring channels of one step)
// dst[i] may be null (result only forwarded) and dst2 (optional array) is a second destination
hipError_t launch_reduce2_batch(void* const* dst, void* const* dst2, const void* const* a, const void* const* b,
                                const size_t* counts, int n, int dtype, int op, hipStream_t stream,
                                hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);

// The zero-copy collective kernels.  Every dsts[k][i] = left-to-right fold of srcs[.][i]: one pass,
// nsrc reads + ndst writes per element; sources and destinations may be local HBM or peers' buffers
// (reads / writes over xGMI).  dsts[k] may alias srcs[j] exactly.  nsrc, ndst <= kMaxReduceSrcs.
hipError_t launch_reduce_n_multi(void* const* dsts, int ndst, const void* const* srcs, int nsrc, size_t count,
                                 int dtype, int op, hipStream_t stream, hipEvent_t ev_start = nullptr,
                                 hipEvent_t ev_stop = nullptr);
// every dsts[k] = src (one read, ndst writes); entries equal to src are skipped
hipError_t launch_copy_multi(void* const* dsts, int ndst, const void* src, size_t bytes, hipStream_t stream,
                             hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);

// dsts[k] = srcs[k], k < n, in ONE launch with the access pattern of launch_reduce_n_multi on the same pointers (same grid, same
// cache policy, n loads and n stores per 16-byte packet index): the fold without its arithmetic.  16-byte aligned pointers only.
hipError_t launch_copy_pairs(void* const* dsts, const void* const* srcs, int n, size_t bytes, hipStream_t stream, hipEvent_t ev_start = nullptr,
                             hipEvent_t ev_stop = nullptr);

// *d_out += number of differing bytes (d_out: 8-byte device word, caller zeroes it)
hipError_t launch_count_mismatch(const void* a, const void* b, size_t bytes, uint64_t* d_out,
                                 hipStream_t stream);
// *d_out += sum of LE u32 words + trailing bytes
hipError_t launch_checksum(const void* buf, size_t bytes, uint64_t* d_out, hipStream_t stream);
// d_out[0] = bits of max|a-b| (as u64 of a non-negative double), d_out[1] = sum|b| (double),
// d_out[2] = NaN-mismatch count (u64), d_out[3] = bits of max_i |a_i-b_i| / |b_i|; caller zeroes the 32 bytes
hipError_t launch_diff_stats(const void* a, const void* b, size_t count, int dtype, void* d_out,
                             hipStream_t stream);
hipError_t launch_fill(void* buf, size_t count, int dtype, int pattern, uint64_t seed,
                       hipStream_t stream);
// cache policy of the streaming kernels: -1 = chosen per launch by size, 0 plain, 1 nt loads+stores,
// 2 nt loads only; grid cap in blocks (0 = one tile per block)
void set_kernel_mode(int mode);
int get_kernel_mode();
void set_grid_cap(int cap);

// ---- device-synchronised collectives (dsync.cpp) ------------------------------------------------
// One kernel per rank IS the collective: it tells the peers (flag words in their HBM, written over xGMI)
// that this rank's buffers are ready and where they are, waits until every peer said the same, moves the
// data straight between the user buffers, and leaves only after every peer has finished reading this
// rank's input and writing its output.  The host enqueues it on a stream and never polls.
constexpr int kDsyncRanks = 8;    // the machine is one node of eight GPUs, one rank per GPU; larger jobs (<= kMaxRanks of ctl.h) meet on the host
constexpr int kDsyncArenas = 32;  // live allocations of one peer this rank can translate

// written by ONE peer (slot p of rank q's page by rank p), read by the owner's kernels
struct DsyncSlot {
  uint64_t epoch;  // written last, system-scope release: the fields below belong to this collective
  uint64_t send_gen, send_off, recv_gen, recv_off;  // the peer's buffers: registration number + byte offset
  uint64_t slots;               // ... and the table slots the peer published them under: send | recv << 8 | landing << 16; bit 24:
                                // it lends a landing block; bits 32 ... 63: what call the peer is in (DsyncArgs::sig)
  uint64_t land_gen, land_off;  // the peer's LANDING block (push forms of the stepped kernels: what the peers store into before
                                // the owner has combined it); written -- and meaningful -- only with bit 24 of `slots`
};
static_assert(sizeof(DsyncSlot) == 64, "one announce = one 64-byte record");
struct DsyncEntry {  // a peer's registration number -> where this process mapped that allocation
  uint64_t gen;      // 0 = free
  uint64_t base, bytes;
  uint64_t tag;      // cache entries only: the communicator that wrote it (DsyncArgs::tag)
};
// One per rank, in that rank's HBM, allocated uncached (never held in an L2): polled by the owner, written by peers.
struct DsyncPage {
  DsyncSlot ready[kDsyncRanks];
  uint64_t done[kDsyncRanks][8];  // 64 bytes apart
  uint32_t ticket;                // blocks of the running kernel that have finished their stores
  uint32_t failword;              // first failure any block of the running kernel met (the closing block reports and clears it)
  uint64_t epoch_now;             // epoch of the last kernel of this rank that has ended (written by its closing block):
                                  // the next kernel's epoch is derived from it ON THE DEVICE, so a captured hipGraph
                                  // that replays the same launch keeps counting
  uint32_t xcc_meet, xcc_done;    // split form: the XCDs the meet / done kernel's blocks of the running collective ran on (one bit
                                  // each, HW_REG_XCC_ID); the closing block of the done kernel reads and clears both
  uint32_t pad[10];
  // what this rank's kernels have looked up in the host's translation table (DsyncArgs::table) so far: the host
  // never writes device memory for this (a copy would need a hardware queue -- possibly the one a waiting kernel
  // occupies), the kernels fill the cache themselves and re-fetch when the registration number differs
  DsyncEntry cache[kDsyncRanks][kDsyncArenas];
};

// The flag allocation of a rank (kDsyncPageBytes of uncached HBM, mapped by every peer) holds more than the page:
//   [0, 64 KiB)               DsyncPage
//   [kStepOff, +256 KiB)      uint64_t step[kDsyncRanks][kStepSlots]: step[p][w] is written by rank p only -- "worker w of
//                             rank p has finished step k of epoch e" as (e << 8) | k (stepped kernels: ring, halving, tree)
//   [kBoxOff, +8 KiB)         P2PBox box[kDsyncRanks][kP2PBoxes]: box[p][b] is written by rank p only -- a message p sends here
//   [kAckOff, +8 KiB)         P2PAck ack[kDsyncRanks][kP2PBoxes]: ack[q][b] is written by rank q only -- q's verdict on the
//                             message this rank put into box b of q's page
//   [kLLHereOff, +1 KiB)      uint64_t here[kDsyncRanks][8]: here[p][0] is written by rank p only -- "rank p has started epoch e"
//                             (the LL broadcast, whose data alone does not tell every rank that every other has arrived)
//   [kLLOff, +2 MiB)          the LL slots (below): slot[p][parity] is written by rank p only
constexpr size_t kDsyncPageBytes = 4u << 20;
constexpr int kStepSlots = 2048;
constexpr size_t kStepOff = 65536;
constexpr int kP2PBoxes = 8;  // messages in flight per ordered rank pair (stream-ordered Send / Receive)
constexpr size_t kBoxOff = kStepOff + sizeof(uint64_t) * kDsyncRanks * kStepSlots;
struct P2PBox {
  uint64_t seq;    // written last (release): message number of the ordered pair, 1-based; box = (seq - 1) % kP2PBoxes
  uint64_t tag;    // low 32 bits: the tag (as int32), bits 32..39: dtype
  uint64_t bytes;
  uint64_t gen, slot, off;  // where the payload lives: the sender's registration number, table slot, byte offset
  uint64_t pad[2];
};
struct P2PAck {
  uint64_t seq;     // the message this answers (written last, release)
  uint64_t status;  // 0 = consumed, otherwise the receiver's verdict as a negated xmpi code (truncate, dtype mismatch)
  uint64_t pad[6];
};
constexpr size_t kAckOff = kBoxOff + sizeof(P2PBox) * kDsyncRanks * kP2PBoxes;
// local state of the receive kernels (never written by a peer):
//   [kTakenOff, +1 KiB)   uint64_t taken[kDsyncRanks][kP2PBoxes]: number of the last message consumed from box b of rank p
//   [kGoOff, +2 KiB)      P2PGo go[kP2PGoSlots]: what block 0 of a receive kernel found, for its other blocks
constexpr size_t kTakenOff = kAckOff + sizeof(P2PAck) * kDsyncRanks * kP2PBoxes;
constexpr int kP2PGoSlots = 64;  // >= the operations that may be outstanding at once (xmpi_comm::kP2PDoneSlots): a record is never reused under a waiter
struct P2PGo {
  uint64_t id;      // written last: the receive operation this belongs to
  uint64_t src;     // payload address as mapped here
  uint64_t bytes;
  uint64_t status;
  uint32_t ticket, pad0;
  uint64_t seq, box;  // which message / box of the pair (the block that finishes last answers the sender)
  uint64_t pad[1];
};
constexpr size_t kGoOff = kTakenOff + sizeof(uint64_t) * kDsyncRanks * kP2PBoxes;
constexpr size_t kLLHereOff = 768u << 10;
static_assert(kGoOff + sizeof(P2PGo) * kP2PGoSlots <= kLLHereOff, "flag allocation");

// ---- LL ("low latency") small collectives (ll.hip) -----------------------------------------------------------------------
// The data IS the flag.  A rank pushes its payload straight into a slot of every peer's flag allocation as 16-byte lines
//   { data[0..3], flag, data[4..7], flag }      flag = low 32 bits of the collective's epoch
// written as two 8-byte system-scope stores (8 bytes are atomic everywhere: a half is either the old line or the new one),
// polls its OWN slots until every line it needs carries this epoch's flag, and folds locally in rank order -- one one-way
// hop; no announce, no remote read, no translation, no "done" exchange (kdev.h dsync_begin / dsync_end cost two round trips
// between peers), and the user buffers need not be registered: only their owner's kernel touches them.  What the reference
// does per message is one message + one ack (network.go:562-571); this is the message alone.
// Slots are double-buffered by the parity of the epoch.  Rank X may overwrite slot[X][e & 1] of rank Y at epoch e because Y
// has finished reading epoch e-2: X has finished epoch e-1, EVERY device-synchronised collective completes on a rank only
// after every peer has started it (the LL forms: a line or a `here` word from everybody; the others: dsync_begin), and Y
// started e-1 after it finished e-2 (the kernels of one rank run one at a time).  tests/ll_sim.py checks exactly this.
constexpr size_t kLLOff = 1u << 20;
constexpr size_t kLLSlotBytes = 64u << 10;          // lines of one (source rank, parity)
constexpr size_t kLLMaxPayload = kLLSlotBytes / 2;  // 32 KiB per rank
static_assert(kLLHereOff + 64 * kDsyncRanks <= kLLOff, "flag allocation");
static_assert(kLLOff + kLLSlotBytes * 2 * kDsyncRanks <= kDsyncPageBytes, "flag allocation");

// what a block of the kernel moves: the fold of the source ranks' buffers (rank order) -> the destination ranks'
struct DsyncSeg {
  uint64_t src_off, dst_off;  // byte offsets into the send / receive buffers
  uint64_t count;             // elements
  uint32_t src_mask, dst_mask;  // ranks whose SEND buffer is read / whose RECEIVE buffer is written
  uint32_t src_from_recv;       // 1: the sources are the ranks' RECEIVE buffers (forwarding what a previous step put there);
                                // 2: they are all LOCAL -- rank r's contribution lies at stage_stride * r of this rank's landing
                                // block, this rank's own at src_off of its send buffer (push-only allreduce: the fold of what landed)
  uint32_t dst_to_land;         // 1: the destinations are the ranks' LANDING blocks (push-only allreduce: contributions the owner has not folded yet)
  uint64_t stage_stride;        // (src_from_recv == 2)
};

// DSYNC_XCD: the split form's meet or done kernel did not have a block on every XCD (DsyncArgs::xcc_need) -- the acquire / release
// once per L2 it relies on did not happen everywhere, the result cannot be trusted
// DSYNC_MISMATCH: a peer announced another call than this rank's (DsyncArgs::sig: collective, schedule, bytes, dtype, operation, root).
enum DsyncStatus : uint32_t { DSYNC_OK = 0, DSYNC_TIMEOUT = 1, DSYNC_ABORTED = 2, DSYNC_UNMAPPED = 3, DSYNC_XCD = 4, DSYNC_MISMATCH = 5 };

struct DsyncArgs {
  DsyncPage* page[kDsyncRanks];  // [me]: own page, others: the peers' pages as mapped here
  int32_t me, n;
  uint64_t epoch_floor;       // this kernel's epoch = max(page.epoch_now, epoch_floor) + 1: the floor is where the
                              // communicator's epochs start (above what earlier users left in the pooled, uncleared pages)
  uint64_t* host_epoch;       // pinned host word that follows page.epoch_now (what the host gives back to the pool), may be null
  uint64_t* host_done;        // pinned host word the closing block writes `done_value` into when the collective is over (a blocking
  uint64_t done_value;        //   call polls it: no event, no stream query between the kernel and the caller); may be null
  uint64_t send_gen, send_off, recv_gen, recv_off;  // what this rank tells its peers
  uint64_t send_slot, recv_slot;
  uint64_t land_gen, land_off, land_slot;  // the landing block this rank lends to a push form (land_gen = 0: none)
  const void* my_send;
  void* my_recv;
  void* my_land;
  uint64_t tag;               // of this communicator: cache entries written under another tag are somebody else's
  uint64_t sig;               // low 32 bits: what this call is (collective, schedule, bytes, dtype, operation, root), announced in the
                              // upper half of DsyncSlot::slots -- a peer that announces another one ends both kernels with
                              // DSYNC_MISMATCH before either touches the other's memory; 0 = not checked
  const DsyncEntry* table;    // [kDsyncRanks][kDsyncArenas] in pinned host memory, written by the host (dsync_service)
  const int32_t* abort_word;  // host memory the GPU can read (the job's abort flag), may be null
  uint32_t* status;           // host memory the GPU can write: first failure (DsyncStatus), may be null
  uint64_t spin_limit;        // wall-clock ticks (100 MHz) a wait may last, 0 = for ever
  int32_t nseg;               // gridDim.y; 0 = synchronise only
  int32_t xcc_need;           // split form: XCDs the meet and the done kernel must each have covered (0 = not checked); the masks
                              // they saw go to status[6] / status[7]
  DsyncSeg seg[kDsyncRanks];
};

enum DsyncLLColl : int32_t { LL_ALLREDUCE = 0, LL_REDUCE = 1, LL_BCAST = 2, LL_ALLGATHER = 3 };
struct DsyncLLArgs {
  DsyncPage* page[kDsyncRanks];  // [me]: own flag allocation, others: the peers' as mapped here
  int32_t me, n;
  int32_t coll, root;            // DsyncLLColl
  uint64_t epoch_floor;          // as DsyncArgs: epoch = max(page.epoch_now, epoch_floor) + 1, counted on the device
  uint64_t* host_epoch;
  uint64_t* host_done;
  uint64_t done_value;
  const void* send;              // this rank's buffers: any memory its own GPU can address
  void* recv;
  uint64_t bytes;                // payload per rank (allgather: one rank's block), <= kLLMaxPayload
  const int32_t* abort_word;
  uint32_t* status;
  uint64_t spin_limit;
};
hipError_t launch_dsync_ll(const DsyncLLArgs& a, int dtype, int op, hipStream_t stream, hipEvent_t ev_start = nullptr,
                           hipEvent_t ev_stop = nullptr);

// The LL AGENT (ll.hip ll_agent_kernel): a one-block kernel that lingers behind a BLOCKING small collective and runs the next
// one without a launch.  The host writes a 32-byte command into pinned memory (as engine.cpp agent_submit does for the receive
// agent, sched.hip p2p_agent_kernel):
//   w0 = doorbell (1 = an LL collective, 2 = stop) | bytes per rank << 2 (22 bits) | seq << 24     w1 = send buffer
//   w2 = receive buffer            w3 = collective (DsyncLLColl) | root << 2 | dtype << 6 | operation << 9 | consecutive << 11 | seq << 32
// consecutive: no kernel of this rank has touched the page's epoch since the collective this agent ran last (the host knows: the
// previous call into the library on this communicator was that collective) -- the epoch is the last one plus one, no load.
// cmd[6] = number of the last command served, cmd[7] = "gone" (the number it was waiting for, plus one).
struct LLAgentArgs {
  uint64_t* cmd;            // pinned host memory, 8 words, 64-byte aligned
  uint64_t seq0;            // the number of the first command this launch serves
  uint64_t patience_ticks;  // wall_clock64 ticks (100 MHz) the agent waits for a command
  DsyncLLArgs ll;           // the communicator's fields: page, me, n, epoch_floor, host_epoch, abort_word, status, spin_limit
};
constexpr uint32_t kAgentLLRootShift = 2, kAgentLLDtypeShift = 6, kAgentLLOpShift = 9, kAgentLLConsecutiveShift = 11;  // root: 4 bits (kDsyncRanks <= 16)
static_assert(kDsyncRanks <= 16, "an LL command names its root in four bits");
constexpr int kLLAgentBlock = 512;  // lanes of the agent's one block: 4 KiB per rank is one line per lane
hipError_t launch_ll_agent(const LLAgentArgs& a, hipStream_t stream);

// grid_x blocks per segment (the caller bounds it: every block spins until the peers arrive, so the kernels of
// all ranks sharing a GPU must be resident together); unroll = 16-byte packets per lane per source in flight
hipError_t launch_dsync_fold(const DsyncArgs& a, int nsrc, int dtype, int op, int grid_x, int unroll, hipStream_t stream,
                             hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);

// one lane: *dst (pinned host memory, as the device addresses it) = *src (device memory), system-scope release -- in stream order
hipError_t launch_word_to_host(uint64_t* dst, const uint64_t* src, hipStream_t stream);
// one lane: system-scope release store of `value` to *flag (host-registered or device memory)
hipError_t launch_signal(uint64_t* flag, uint64_t value, hipStream_t stream);

// ---- split form of a device-synchronised collective (sched.hip): three launches on one stream -----------------------------
// The one-kernel form keeps every block spinning until the peers arrive, so its grid must stay resident (bounded).
// For large messages the collective is split: a ONE-block kernel meets the peers and writes the translated buffer
// addresses into device memory (DsyncResolved), the data kernel is an ordinary streaming kernel with one tile per block
// and no flag anywhere (what reduce_n_multi_kernel is), and a one-block kernel exchanges "done".  Only two blocks ever
// wait for a peer; the chip is free for the caller's other streams meanwhile.
struct DsyncResolvedSeg {
  uint64_t src[kDsyncRanks], dst[kDsyncRanks];  // sources in rank order, destinations local first
  uint64_t count;   // elements
  int32_t nsrc, ndst;
  uint32_t vec, pad;  // vec: every pointer is 16-byte aligned
};
struct DsyncResolved {
  uint64_t epoch;
  uint32_t fail, nseg;
  DsyncResolvedSeg seg[kDsyncRanks];
};
hipError_t launch_dsync_meet(const DsyncArgs& a, DsyncResolved* out, hipStream_t stream);
// nsrc_hint: sources per segment when it is the same for all (unrolled loads), 0 = read it from the table
// sys: every load and store of the data kernel at system scope (sc0 sc1: nothing is served from or left in an L2) -- the form
// that does not depend on the meet / done kernels having reached every XCD's L2
hipError_t launch_dsync_body(const DsyncResolved* res, int nseg, size_t max_packets, int nsrc_hint, int dtype, int op,
                             size_t traffic_bytes, bool sys, hipStream_t stream, hipEvent_t ev_start = nullptr,
                             hipEvent_t ev_stop = nullptr);
// *mask_out (device or pinned host word, zeroed by the caller) |= 1 << XCC_ID of every block of a `blocks` x one-wave grid:
// which XCDs a grid of that size reaches on this device (blocks = 1024: the XCDs the device has)
hipError_t launch_xcc_probe(uint32_t* mask_out, int blocks, hipStream_t stream);
// Do flag words WORK between these ranks?  One wave: lane p stores `token` into word 1 of done[me] in rank p's page and waits (at
// most spin_limit ticks) for rank p's token in its own page; *seen_out (device-visible host word) = the ranks whose token arrived
// (bit per rank, own bit set).  A mapping that opens and then does not carry a peer's store to the owner's polling lane -- the way an
// uncached allocation opened on ANOTHER device could fail -- shows here, inside xmpi_init, instead of as a first collective that
// never ends.  Tokens grow from communicator to communicator (the pages are never cleared): a stale one never passes.
hipError_t launch_flag_selftest(DsyncPage* const* page, int me, int n, uint64_t token, uint64_t spin_limit, const int32_t* abort_word,
                                uint32_t* seen_out, hipStream_t stream);
constexpr int kXcdBlocks = 16;  // blocks of the meet / done kernels
hipError_t launch_dsync_done(const DsyncArgs& a, const DsyncResolved* res, hipStream_t stream);

// ---- stepped collectives in ONE kernel per rank (sched.hip): ring, recursive halving / doubling, binary tree ------------
// All steps of the schedule run inside the kernel: worker w (a block) owns the tiles T of the buffer with T % W == w on
// EVERY rank, so step k of worker w depends on step k-1 of worker w of one peer only -- one flag word per (peer, worker)
// (DsyncPage allocation, `step`).  Every schedule has two forms (sched_steps.h):
//   PULL  a step reads the peer's buffer (send or receive buffer, in place included) and writes only local memory; the flag
//         says "my step k is in my buffer" -- every payload byte crosses the link as a LOAD (a round trip per packet);
//   PUSH  a step reads only local memory -- its own input and what the peers have stored here -- combines, and stores the
//         result into the PEER's memory (its receive buffer, or its landing block where the receive buffer still holds an
//         operand nobody has read: in place); the flag says "my step k is in YOUR memory" -- every payload byte crosses the
//         link as a posted STORE, the reference's one-way message (network.go:562-571) without the ack.
// Both forms combine the same operands in the same association: their results are bit-identical.
enum DsyncSched : int32_t {
  SCHED_RING_ALLREDUCE = 1,  // reduce-scatter + allgather round each channel's ring, 2(N-1) steps
  SCHED_RHD_ALLREDUCE = 2,   // recursive halving + doubling, 2 log2 N steps; N no power of two: a fold-in and a fold-out step more
  SCHED_RING_ALLGATHER = 3,  // N steps (the first is the local copy of the own block)
  SCHED_TREE_BCAST = 4,      // binary tree rooted at `root`, the buffer cut into `pieces` pipelined pieces
  SCHED_TREE_REDUCE = 5,     // the same tree upwards: every node folds its children's partial results into its own, piece by piece
};
constexpr int kMaxSchedChannels = 8;
struct DsyncSchedArgs {
  DsyncArgs d;          // pages, this rank's buffers, epoch floor ... (nseg / seg unused)
  int32_t sched, nchan; // nchan = gridDim.y: ring channels (each a different cyclic order of the ranks)
  int32_t root, pieces;
  uint64_t count;       // elements (allgather: per rank)
  uint32_t elem_size;
  uint32_t push;        // 1 = the push form
  uint8_t order[kMaxSchedChannels][kDsyncRanks];  // order[c][i]: the rank at position i of channel c's ring
};
hipError_t launch_dsync_sched(const DsyncSchedArgs& a, int dtype, int op, int grid_x, hipStream_t stream,
                              hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
// tile bytes of the stepped kernels (what a worker moves per iteration): grid_x = min(cap, ceil(step bytes / this))
constexpr size_t kSchedTileBytes = 16384;
// dst = src exactly as a step of the stepped kernels moves data: system-scope loads and written-through stores (sc0 sc1), 16 KiB
// tiles, `grid_x` workers -- the link probe's third engine: what a link gives the accesses the ring / halving / tree kernels make
hipError_t launch_sys_copy(void* dst, const void* src, size_t bytes, int grid_x, hipStream_t stream);

// ---- stream-ordered Send / Receive (sched.hip) ---------------------------------------------------------------------------
// The sender's kernel puts {seq, tag, dtype, bytes, where the payload is} into box (seq-1) % kP2PBoxes of the receiver's
// flag allocation and waits for the receiver's ack; the receiver's kernel waits for a box with its tag, pulls the payload
// straight out of the sender's buffer and acks with its verdict.  What the reference does with a gob message and an ack
// message over a net.Conn (network.go:562-571, 616-624).
struct P2PArgs {
  DsyncPage* my_page;    // this rank's flag allocation
  DsyncPage* peer_page;  // the other rank's, as mapped here
  int32_t me, peer;
  int32_t tag, dtype;
  uint64_t seq;          // send: number of this message of the pair (me -> peer); its box must be free (acked) -- the host checks
  uint64_t bytes;        // send: payload size; receive: capacity of `buf`
  uint64_t gen, slot, off;  // send: where the payload lives
  void* buf;             // receive: destination (local memory)
  uint64_t comm_tag;     // translation cache tag (DsyncArgs::tag)
  const DsyncEntry* table;
  const int32_t* abort_word;
  uint64_t spin_limit;
  uint64_t* host_done;   // pinned host: [0] = done flag (written last), [1] = status, [2] = bytes received; may be null
  uint64_t done_value;   // what to write into host_done[0]
  uint64_t op_id;        // receive: (communicator number << 32) | number of this operation -- unique per PAGE, which outlives
                         // communicators and is never cleared; go slot = op_id % kP2PGoSlots
};
hipError_t launch_p2p_send(const P2PArgs& a, hipStream_t stream);
// the copy of a blocking Receive the host has already matched: dst = src, then the acks (see sched.hip)
struct P2PPullArgs {
  void* dst;
  const void* src;
  uint64_t bytes;
  uint32_t* ticket;       // device word, zero between launches (grids of more than one block)
  uint64_t* host_done;    // pinned host word this rank's thread polls
  uint64_t done_value;
  uint32_t* mail_state;   // the message's mail entry in the shared control block, as the GPU addresses it (may be null)
  int32_t* mail_status;
  uint32_t mail_done_value;
  uint32_t pad;
};
hipError_t launch_p2p_pull(const P2PPullArgs& a, int grid_x, hipStream_t stream);
// ... by a kernel that stays for a while after a message and takes the next one from a command record in pinned host memory
// instead of being launched again (see sched.hip p2p_agent_kernel)
struct P2PAgentArgs {
  uint64_t* cmd;            // pinned host memory, 8 words, 64-byte aligned: [0..3] the command, [6] last number served, [7] "gone"
  uint64_t* rec;            // device memory, 8 words: what block 0 tells the other blocks; [5] ticket, [6] number all blocks finished
  uint64_t ctl_dev;         // the control block as the GPU addresses it (a command names its mail entry by offset)
  uint64_t seq0;            // the number of the first command this launch serves
  uint64_t launch;          // number of this launch (never 0; < 2^23)
  uint64_t alone_bytes;     // messages up to this long are copied by block 0 alone (waking the other blocks costs ~2.5 us)
  uint64_t patience_ticks;  // wall_clock64 ticks (100 MHz) block 0 waits for a command
  uint32_t mail_done_value;
  uint32_t pad;
};
hipError_t launch_p2p_agent(const P2PAgentArgs& a, int grid_x, hipStream_t stream);
hipError_t launch_p2p_recv(const P2PArgs& a, int grid_x, hipStream_t stream);

}  // namespace xmpi
