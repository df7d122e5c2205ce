// ctl.h -- the shared-memory control block all ranks of one job map (host logic only, no HIP).
//
// It replaces the control traffic of the reference's TCP mesh: the password/id handshake of
// (*Network).listenHandshake / dialHandshake (network.go:211-339) becomes "join the block and
// publish a RankInfo", the per-connection tagManager (network.go:448-497) becomes the mail
// entries of an ordered rank pair, and the ack message of receiveReader (network.go:616-624)
// becomes the entry's DONE state.  Payload bytes never pass through here: they move
// GPU-to-GPU into HBM windows; the block only carries counters and small headers.
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <string>

namespace xmpi {

// A job has at most 16 ranks (the reference takes len(addrs): network.go:94-109).  The DEVICE side -- flag pages, the kernels'
// tables -- is sized for the machine this is written for, one node of eight MI355X with one rank per GPU (kernels.h kDsyncRanks = 8:
// registers and LDS of the hot kernels); a job of 9 .. 16 ranks (two per GPU, say) runs on the generic path: its ranks meet on the
// host (zcopy.cpp's rendezvous, the step tables through the windows), whose tables are sized by this constant.
constexpr int kMaxRanks = 16;
constexpr int kMaxLanes = 4;
constexpr int kMailEntries = 4;  // concurrent messages per ordered rank pair
constexpr int kHostLaneSlots = 4;  // pieces of a host-resident payload in flight per mail entry (the host lanes, below)
constexpr uint64_t kCtlMagic = 0x584D504943544C31ull;  // "XMPICTL1"
constexpr uint32_t kCtlVersion = 11;  // layout of the block: bump with every change of the structs below

struct alignas(64) Counter {
  std::atomic<uint64_t> v;
  char pad[56];
};

// One-directional FIFO of slots in the receiver's HBM window.  head = slots filled (written by
// the sender after its copy completed), tail = slots drained (written by the receiver).
struct PipeCtl {
  Counter head;
  Counter tail;
};

enum MailState : uint32_t { MAIL_FREE = 0, MAIL_CLAIMED = 1, MAIL_POSTED = 2, MAIL_MATCHED = 3, MAIL_DONE = 4 };

// Where a registered buffer lives: enough for a peer to map it (hipIpc) and address the same bytes.
struct BufRef {
  uint64_t base;    // allocation base VA in the owner's process (0 = no buffer)
  uint64_t gen;     // owner's registration number of that allocation (a re-used VA gets a new one)
  uint64_t offset;  // of the user pointer inside the allocation
  uint64_t bytes;   // of the allocation
  uint8_t handle[64];
};

// how the payload of a message travels
enum MailDirect : int32_t {
  DIRECT_NONE = 0,      // through the mail slots of the receiver's window (two copies, pipelined)
  DIRECT_OFFERED = 1,   // the sender's buffer is registered: `src` says where it is
  DIRECT_ACCEPTED = 2,  // the receiver copies straight out of it (one copy, one xGMI crossing)
  DIRECT_DECLINED = 3,  // the receiver cannot (mapping failed): sender uses the slots
  DIRECT_HOST = 4,      // the payload is in HOST memory (a Go slice, network.go:518): it travels through the entry's host lane
};

struct alignas(64) MailEntry {
  std::atomic<uint32_t> state;
  int32_t tag;
  int32_t dtype;
  std::atomic<int32_t> status;  // receiver's verdict (XMPI_OK / XMPI_ERR_TRUNCATE / ...)
  uint64_t bytes;
  std::atomic<int32_t> direct;  // MailDirect
  char pad[36];
  PipeCtl pipe;
  BufRef src;
};

struct alignas(64) RankInfo {
  std::atomic<int32_t> state;  // 0 absent, 1 joined, 2 window published, 3 left
  int32_t pid;
  int32_t device;
  int32_t maps;  // what this rank could map of its peers' memory (xmpi_init, before the vote): kMapsWindows | kMapsFlags
  uint64_t window_addr;   // device VA in the owner's process (used when peer pid == own pid)
  uint64_t window_bytes;
  uint8_t ipc_handle[64];  // hipIpcMemHandle_t of the window
  char busid[32];
  uint64_t flag_addr;       // this rank's flag page (uncached HBM; device-synchronised collectives), 0 = none
  uint64_t flag_epoch;      // the highest epoch an earlier communicator left in that (pooled, never cleared) page
  uint8_t flag_handle[64];  // its hipIpcMemHandle_t
  char maps_why[96];        // the first mapping this rank could not make, in words ("" = none)
  int64_t ll_choice;        // the LL limit this rank would choose from what IT sees (ranks on its GPU, its environment): the job
                            // takes the smallest -- LL or fold is a protocol choice every rank must make alike
  uint64_t start_time;      // /proc/<pid>/stat starttime of the process that joined as this rank (with pid: is it still THAT process?)
};
constexpr int32_t kMapsWindows = 1, kMapsFlags = 2;

// What a rank tells its peers about the user buffers of one zero-copy collective.  Two descriptors
// per rank, used alternately (a rank may publish the next collective's while a slower peer still
// reads the current one).
struct alignas(64) BufDesc {
  std::atomic<uint64_t> seq;  // which zero-copy collective of this communicator the content belongs to
  int32_t ok;     // 1 = both buffers are registered HBM of this rank's device
  int32_t fresh;  // 1 = an allocation named here was not announced on this communicator before
  std::atomic<int32_t> verdict;  // after mapping the peers' buffers: 1 = all mapped, -1 = failed
  int32_t in_place;  // 1 = send and receive buffer are the same memory on this rank
  BufRef send, recv;
  uint32_t sig;      // what call this is (collective, bytes, dtype, operation, root, push-only or not): ranks that publish different
                     // ones are not in the same call -- an error on every rank, before anything is mapped or moved
};
static_assert(sizeof(BufDesc) == 256, "a descriptor is four cache lines");

// Registered allocations a rank has freed since the job began (their `gen`s, in order).  A peer that
// mapped one must unmap it before it maps anything new of that rank: the runtime may hand the same
// hipIpc handle to a later allocation, and a mapping of the dead one would shadow it.
constexpr int kRetireRing = 512;
struct alignas(64) RetireLog {
  std::atomic<uint64_t> count;  // entries ever appended; entry k lives in gen[k % kRetireRing]
  uint64_t gen[kRetireRing];
};

// Device-synchronised collectives (dsync.cpp): what a rank has registered for its peers to map, in
// registration order.  A peer maps entry k (hipIpc), adds {gen -> its mapping} to the translation table its
// kernels read, and then says so in `acked`: a rank uses a buffer in a device-synchronised collective only
// once every peer has acknowledged the allocation it lives in, so no kernel ever meets an address it cannot
// translate.  Entries are only appended; count - min(acked) <= kPubRing.
constexpr int kPubRing = 64;
struct PubEntry {
  uint64_t gen;    // owner's registration number (same numbering as BufRef.gen / the retire log)
  uint64_t base;   // allocation base VA in the owner's process
  uint64_t bytes;
  uint64_t reserved;
  uint8_t handle[64];  // hipIpcMemHandle_t
};
struct alignas(64) PubTable {
  std::atomic<uint64_t> count;  // entries ever published; entry k lives in e[k % kPubRing]
  char pad[56];
  PubEntry e[kPubRing];
};

// xmpi_tune / xmpi_init's self-check: every rank publishes a row -- mean time and number of WRONG bytes per candidate schedule --,
// all meet at a barrier and read the same maxima.  Through this block and not through a collective: the vote judges the transports,
// it must not ride on one of them.
constexpr int kTuneCands = 16;
struct alignas(64) TuneVote {
  double us[kTuneCands];
  uint64_t bad[kTuneCands];
};

struct CtlConfig {
  int32_t lanes;        // FIFO lanes per ordered pair for collectives
  int32_t fifo_depth;   // slots per collective pipe
  uint64_t slot_bytes;  // bytes per collective slot
  int32_t p2p_depth;    // slots per mail entry
  uint64_t p2p_slot_bytes;
  uint64_t host_lane_bytes;  // per mail entry, kHostLaneSlots pieces; 0 = no host lanes (the creator could not reserve them)
};

struct alignas(64) CtlHeader {
  std::atomic<uint64_t> magic;
  uint32_t version;
  int32_t size;
  uint64_t total_bytes;
  int32_t creator_pid;
  uint64_t creator_start;  // /proc/<pid>/stat starttime of the creator (stale-segment check)
  CtlConfig cfg;
  std::atomic<int32_t> abort_code;  // != 0: some rank failed; everybody stops waiting
  std::atomic<int32_t> dead_rank;   // rank + 1 of the first rank whose PROCESS was found gone (peer_gone below); 0 = none
  alignas(64) std::atomic<uint32_t> bar_count;
  alignas(64) std::atomic<uint32_t> bar_gen;
};

// polite spin: pause a while, then yield the core (ranks may outnumber cores)
struct Backoff {
  unsigned n = 0;
  // called now and then while waiting: a rank that blocks still serves its peers (maps the buffers they
  // registered and acknowledges them -- dsync.cpp), so nobody waits on somebody who is waiting
  void (*idle)(void*) = nullptr;
  void* idle_arg = nullptr;
  void pause();
};
class Ctl {
 public:
  // Joins (rank 0: creates) the block named after `key`.  Returns 0 or a negative xmpi code.
  static int join(const std::string& key, int rank, int size, const CtlConfig& cfg_if_creator,
                  double timeout_s, Ctl** out, std::string* err);
  ~Ctl();

  int rank() const { return rank_; }
  int size() const { return size_; }
  const CtlConfig& cfg() const { return hdr_->cfg; }
  CtlHeader* header() { return hdr_; }
  RankInfo* info(int r) { return &ranks_[r]; }
  PipeCtl* pipe(int src, int dst, int lane) { return &pipes_[((size_t)src * size_ + dst) * kMaxLanes + lane]; }
  MailEntry* mail(int src, int dst, int e) { return &mail_[((size_t)src * size_ + dst) * kMailEntries + e]; }
  BufDesc* desc(int r, uint64_t seq) { return &desc_[(size_t)r * 2 + (size_t)(seq & 1)]; }
  RetireLog* retired(int r) { return &retire_[r]; }
  PubTable* published(int r) { return &pub_[r]; }
  TuneVote* vote(int r) { return &vote_[r]; }
  // how many of rank `owner`'s published entries rank `reader` has mapped (written by `reader` only)
  std::atomic<uint64_t>* acked(int reader, int owner) { return &acked_[((size_t)reader * size_ + owner) * 8]; }
  // Host lanes: host-resident payloads (what a Go program hands to Send: slices) travel between the processes of a node
  // through shared memory -- a ring of kHostLaneSlots pieces per mail entry, after the control structures in the same
  // segment (reserved with posix_fallocate by the creator: a full /dev/shm disables them, it never faults a writer).
  size_t host_lane_bytes() const { return hdr_->cfg.host_lane_bytes; }
  char* host_lane(int src, int dst, int e) {
    return lanes_ + (((size_t)src * size_ + dst) * kMailEntries + e) * hdr_->cfg.host_lane_bytes;
  }
  static size_t lane_bytes_for(int size);  // what a job of `size` ranks asks for per entry
  void* base() const { return base_; }
  size_t bytes() const { return bytes_; }  // control structures + host lanes

  // all ranks reach `state` (RankInfo.state >= state) or timeout / abort
  int wait_all_state(int state, double timeout_s);
  int barrier(double timeout_s, Backoff* bo = nullptr);
  void set_abort(int code) {
    int32_t z = 0;
    hdr_->abort_code.compare_exchange_strong(z, code);
  }
  int aborted() const { return hdr_->abort_code.load(std::memory_order_acquire); }
  // The process that joined as rank r no longer exists (it exited, crashed or was killed) and never left the job properly
  // (RankInfo.state 3, xmpi_finalize).  What the reference's peers learn from their sockets at once (network.go:555,611,623: a lost
  // connection is an I/O error in Send / Receive) has to be looked for here: nobody closes a shared-memory block on a crash.
  bool peer_gone(int r);
  // the first rank found gone, or -1; check_peers looks at every other rank and raises the job's abort flag for the first one gone
  int dead_rank() const { return hdr_->dead_rank.load(std::memory_order_acquire) - 1; }
  int check_peers();
  void set_watch(bool on) { watch_ = on; }  // barrier / wait_all_state look for dead peers while they wait (xmpi_init: XMPI_WATCHDOG_MS > 0)
  std::string abort_reason();  // "rank 3's process (pid 1234) is gone" / "a peer rank aborted the job"
  // rank 0 removes the name (the mapping stays valid until every rank unmaps)
  void unlink_name();

  static size_t layout_bytes(int size);

 private:
  Ctl() = default;
  std::string name_;
  int rank_ = 0, size_ = 0;
  void* base_ = nullptr;
  size_t bytes_ = 0;
  CtlHeader* hdr_ = nullptr;
  RankInfo* ranks_ = nullptr;
  PipeCtl* pipes_ = nullptr;
  MailEntry* mail_ = nullptr;
  BufDesc* desc_ = nullptr;
  RetireLog* retire_ = nullptr;
  PubTable* pub_ = nullptr;
  TuneVote* vote_ = nullptr;
  std::atomic<uint64_t>* acked_ = nullptr;  // [reader][owner], one cache line each
  char* lanes_ = nullptr;
  bool creator_ = false;
  bool watch_ = false;
};

double now_seconds();

}  // namespace xmpi
