// ctl.cpp -- shared-memory control block (see ctl.h).  Host logic only, no HIP.
#include "ctl.h"

#include <dirent.h>
#include <fcntl.h>
#include <sched.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/xmpi.h"
#include "../../include/xmpi_test.h"

namespace xmpi {

double now_seconds() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void Backoff::pause() {
  n++;
  if (idle && (n & 255u) == 0) idle(idle_arg);
  if (n < 4096) {
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  } else if (n < 65536) {
    sched_yield();
  } else {
    timespec ts{0, 50000};
    nanosleep(&ts, nullptr);
  }
}

static uint64_t proc_start_time(int pid) {
  char path[64], buf[1024];
  snprintf(path, sizeof path, "/proc/%d/stat", pid);
  FILE* f = fopen(path, "r");
  if (!f) return 0;
  size_t n = fread(buf, 1, sizeof buf - 1, f);
  fclose(f);
  buf[n] = 0;
  const char* p = strrchr(buf, ')');  // comm may contain spaces
  if (!p) return 0;
  p++;
  unsigned long long start = 0;
  int field = 2;  // the field after ')' is #3 (state)
  while (*p) {
    while (*p == ' ') p++;
    field++;
    if (field == 22) {
      sscanf(p, "%llu", &start);
      break;
    }
    while (*p && *p != ' ') p++;
  }
  return start;
}

// is process `pid` still the one that started at `start` (and not a zombie waiting to be reaped)?
static bool process_alive(int pid, uint64_t start) {
  char path[64], buf[1024];
  snprintf(path, sizeof path, "/proc/%d/stat", pid);
  FILE* f = fopen(path, "r");
  // no such entry: the process is gone (ranks that cannot see each other's /proc entries at all -- other pid namespaces -- never get
  // here: xmpi_init tries every LIVE peer once and switches the watchdog off); any other failure to look: assume it lives
  if (!f) return errno != ENOENT && errno != ESRCH;
  const size_t n = fread(buf, 1, sizeof buf - 1, f);
  fclose(f);
  buf[n] = 0;
  const char* p = strrchr(buf, ')');
  if (!p || !p[1] || !p[2]) return true;
  const char state = p[2];
  if (state == 'Z' || state == 'X' || state == 'x') return false;  // exited, not yet reaped by its parent
  return start == 0 || proc_start_time(pid) == start;  // (a recycled pid is another process)
}

bool Ctl::peer_gone(int r) {
  if (r < 0 || r >= size_ || r == rank_) return false;
  RankInfo* ri = &ranks_[r];
  const int st = ri->state.load(std::memory_order_acquire);
  if (st < 1 || st >= 3) return false;  // never joined (the bootstrap's clock covers that) / left properly
  if (ri->pid == (int32_t)getpid()) return false;  // a thread of this process
  return !process_alive(ri->pid, ri->start_time);
}

int Ctl::check_peers() {
  for (int r = 0; r < size_; r++)
    if (peer_gone(r)) {
      int32_t none = 0;
      if (hdr_->dead_rank.compare_exchange_strong(none, r + 1, std::memory_order_acq_rel))
        fprintf(stderr, "xmpi: rank %d: the process of rank %d (pid %d) is gone: the job is aborted\n", rank_, r, (int)ranks_[r].pid);
      set_abort(XMPI_ERR_PEER);
      return r;
    }
  return -1;
}

std::string Ctl::abort_reason() {
  const int d = dead_rank();
  if (d >= 0 && d < size_) return "the process of rank " + std::to_string(d) + " (pid " + std::to_string(ranks_[d].pid) + ") is gone";
  return "a peer rank aborted the job";
}

size_t Ctl::layout_bytes(int size) {
  size_t b = sizeof(CtlHeader);
  b += sizeof(RankInfo) * (size_t)size;
  b += sizeof(PipeCtl) * (size_t)size * size * kMaxLanes;
  b += sizeof(MailEntry) * (size_t)size * size * kMailEntries;
  b += sizeof(BufDesc) * (size_t)size * 2;
  b += sizeof(RetireLog) * (size_t)size;
  b += sizeof(PubTable) * (size_t)size;
  b += sizeof(TuneVote) * (size_t)size;
  b += 64 * (size_t)size * size;  // acked[reader][owner]
  return (b + 4095) / 4096 * 4096;
}

// 32 MiB of lanes per job at most: 256 KiB per entry for 2 ... 4 ranks, 128 KiB for 8, 32 KiB for 16
size_t Ctl::lane_bytes_for(int size) {
  size_t per = (size_t)256 << 10;
  while (per > ((size_t)16 << 10) && per * (size_t)size * size * kMailEntries > ((size_t)32 << 20)) per /= 2;
  return per;
}

static std::string shm_name(const std::string& key) {
  std::string n = "/xmpi-" + std::to_string((unsigned)getuid()) + "-";
  for (char c : key) n += (isalnum((unsigned char)c) || c == '-' || c == '_' || c == '.') ? c : '_';
  if (n.size() > 200) n.resize(200);
  return n;
}

// XMPI_CTL_SHARE_MAPPING=1: ranks hosted by threads of ONE process address the block through one mapping (the creator's)
// instead of one mmap each.  What a thread sanitizer needs -- it tells accesses apart by virtual address, so the atomics of
// two ranks meet in its eyes only when both use the same one (tests/tsan_host_driver.cpp); off otherwise (every communicator
// registers its own mapping with the GPU).
struct SharedMapping {
  void* base = nullptr;
  size_t bytes = 0;
  int refs = 0;
};
static std::mutex g_shared_mu;
static std::map<std::string, SharedMapping> g_shared;
static bool share_mappings() {
  static const bool on = getenv("XMPI_CTL_SHARE_MAPPING") && atoi(getenv("XMPI_CTL_SHARE_MAPPING")) != 0;
  return on;
}

// Control blocks of jobs that died without finalising (kill -9, a crash) would otherwise pile up in
// /dev/shm: rank 0 of a new job removes every block of this user whose creator process is gone.
static void reap_dead_blocks() {
  const std::string prefix = "xmpi-" + std::to_string((unsigned)getuid()) + "-";
  DIR* d = opendir("/dev/shm");
  if (!d) return;
  std::vector<std::string> dead;
  while (dirent* ent = readdir(d)) {
    const std::string fn = ent->d_name;
    if (fn.compare(0, prefix.size(), prefix) != 0) continue;
    const std::string path = "/dev/shm/" + fn;
    int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) continue;
    CtlHeader h;
    const ssize_t got = pread(fd, &h, sizeof h, 0);
    close(fd);
    if (got != (ssize_t)sizeof h || h.magic.load(std::memory_order_relaxed) != kCtlMagic) continue;
    const int cp = h.creator_pid;
    const bool alive = (kill(cp, 0) == 0 || errno == EPERM) && proc_start_time(cp) == h.creator_start;
    if (!alive) dead.push_back("/" + fn);
  }
  closedir(d);
  for (const std::string& n : dead) shm_unlink(n.c_str());
}

int Ctl::join(const std::string& key, int rank, int size, const CtlConfig& cfg, double timeout_s, Ctl** out,
              std::string* err) {
  if (size < 1 || size > kMaxRanks || rank < 0 || rank >= size) {
    *err = "rank/size out of range (size <= " + std::to_string(kMaxRanks) + ")";
    return XMPI_ERR_ARG;
  }
  const std::string name = shm_name(key);
  const size_t ctl_bytes = layout_bytes(size);
  const size_t lanes_total = lane_bytes_for(size) * (size_t)size * size * kMailEntries;
  const size_t bytes = ctl_bytes + lanes_total;
  const double t0 = now_seconds();
  void* base = nullptr;
  bool creator = false;

  if (rank == 0) {
    reap_dead_blocks();
    // A block of this name whose creator is still running belongs to a LIVE job using the same key (two jobs
    // started with the default key, say): taking the name away from it would split that job.  Refuse instead.
    {
      int fd = shm_open(name.c_str(), O_RDONLY, 0600);
      if (fd >= 0) {
        CtlHeader h;
        const ssize_t got = pread(fd, &h, sizeof h, 0);
        close(fd);
        if (got == (ssize_t)sizeof h && h.magic.load(std::memory_order_relaxed) == kCtlMagic) {
          const int cp = h.creator_pid;
          const bool alive = (kill(cp, 0) == 0 || errno == EPERM) && proc_start_time(cp) == h.creator_start;
          if (alive && h.abort_code.load(std::memory_order_relaxed) == 0) {
            *err = "job key already in use by a running job (pid " + std::to_string(cp) + "): " + name;
            return XMPI_ERR_BOOTSTRAP;
          }
        }
      }
    }
    shm_unlink(name.c_str());  // a stale block of a crashed (or aborted) job with the same key
    int fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) {
      *err = "shm_open(create " + name + "): " + strerror(errno);
      return XMPI_ERR_BOOTSTRAP;
    }
    if (ftruncate(fd, (off_t)bytes) != 0) {
      *err = std::string("ftruncate: ") + strerror(errno);
      close(fd);
      shm_unlink(name.c_str());
      return XMPI_ERR_BOOTSTRAP;
    }
    base = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    const int fd_keep = fd;
    if (base == MAP_FAILED) {
      *err = std::string("mmap: ") + strerror(errno);
      close(fd);
      shm_unlink(name.c_str());
      return XMPI_ERR_BOOTSTRAP;
    }
    memset(base, 0, ctl_bytes);  // every counter, state and flag starts at zero (the lanes need no initial value)
    // the lanes' pages exist before anybody writes to them, or the lanes are not used
    const bool lanes_ok = cfg.host_lane_bytes != 0 && posix_fallocate(fd_keep, (off_t)ctl_bytes, (off_t)lanes_total) == 0;
    close(fd_keep);
    CtlHeader* h = new (base) CtlHeader;
    h->version = kCtlVersion;
    h->size = size;
    h->total_bytes = bytes;
    h->creator_pid = (int32_t)getpid();
    h->creator_start = proc_start_time(getpid());
    h->cfg = cfg;
    h->cfg.host_lane_bytes = lanes_ok ? lane_bytes_for(size) : 0;
    h->abort_code.store(0);
    h->dead_rank.store(0);
    h->bar_count.store(0);
    h->bar_gen.store(0);
    h->magic.store(kCtlMagic, std::memory_order_release);
    creator = true;
    if (share_mappings()) {
      std::lock_guard<std::mutex> g(g_shared_mu);
      g_shared[name] = SharedMapping{base, bytes, 1};
    }
  } else {
    Backoff bo;
    for (;;) {
      if (now_seconds() - t0 > timeout_s) {
        *err = "timed out waiting for rank 0 to create " + name;
        return XMPI_ERR_TIMEOUT;
      }
      if (share_mappings()) {  // rank 0 is a thread of this process: its mapping
        {
          std::lock_guard<std::mutex> g(g_shared_mu);
          auto it = g_shared.find(name);
          if (it != g_shared.end() && it->second.bytes == bytes) {
            it->second.refs++;
            base = it->second.base;
          }
        }
        if (base) break;
        timespec ts{0, 200000};
        nanosleep(&ts, nullptr);
        continue;
      }
      int fd = shm_open(name.c_str(), O_RDWR, 0600);
      if (fd < 0) {
        timespec ts{0, 1000000};
        nanosleep(&ts, nullptr);
        continue;
      }
      struct stat st;
      if (fstat(fd, &st) != 0 || (size_t)st.st_size < bytes) {  // still being sized, or another job's
        close(fd);
        timespec ts{0, 1000000};
        nanosleep(&ts, nullptr);
        continue;
      }
      void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      close(fd);
      if (m == MAP_FAILED) {
        *err = std::string("mmap: ") + strerror(errno);
        return XMPI_ERR_BOOTSTRAP;
      }
      CtlHeader* h = reinterpret_cast<CtlHeader*>(m);
      bool ok = false;
      const double tw = now_seconds();
      while (now_seconds() - tw < 0.05) {
        if (h->magic.load(std::memory_order_acquire) == kCtlMagic) {
          ok = true;
          break;
        }
        bo.pause();
      }
      if (ok) {
        const int cp = h->creator_pid;
        const bool alive = (kill(cp, 0) == 0 || errno == EPERM) && proc_start_time(cp) == h->creator_start;
        ok = alive && h->size == size && h->total_bytes == bytes && h->abort_code.load() == 0 &&
             h->version == kCtlVersion;
      }
      if (ok) {
        base = m;
        break;
      }
      munmap(m, bytes);  // stale (dead creator) or foreign: wait for rank 0 to replace it
      timespec ts{0, 2000000};
      nanosleep(&ts, nullptr);
    }
  }

  Ctl* c = new Ctl;
  c->name_ = name;
  c->rank_ = rank;
  c->size_ = size;
  c->base_ = base;
  c->bytes_ = bytes;
  c->creator_ = creator;
  char* p = reinterpret_cast<char*>(base);
  c->hdr_ = reinterpret_cast<CtlHeader*>(p);
  p += sizeof(CtlHeader);
  c->ranks_ = reinterpret_cast<RankInfo*>(p);
  p += sizeof(RankInfo) * (size_t)size;
  c->pipes_ = reinterpret_cast<PipeCtl*>(p);
  p += sizeof(PipeCtl) * (size_t)size * size * kMaxLanes;
  c->mail_ = reinterpret_cast<MailEntry*>(p);
  p += sizeof(MailEntry) * (size_t)size * size * kMailEntries;
  c->desc_ = reinterpret_cast<BufDesc*>(p);
  p += sizeof(BufDesc) * (size_t)size * 2;
  c->retire_ = reinterpret_cast<RetireLog*>(p);
  p += sizeof(RetireLog) * (size_t)size;
  c->pub_ = reinterpret_cast<PubTable*>(p);
  p += sizeof(PubTable) * (size_t)size;
  c->vote_ = reinterpret_cast<TuneVote*>(p);
  p += sizeof(TuneVote) * (size_t)size;
  c->acked_ = reinterpret_cast<std::atomic<uint64_t>*>(p);
  c->lanes_ = reinterpret_cast<char*>(base) + ctl_bytes;

  RankInfo* me = c->info(rank);
  int32_t unclaimed = 0;
  if (!me->state.compare_exchange_strong(unclaimed, -1, std::memory_order_acq_rel)) {
    // Two processes claim the same rank: this one never joined, so it must not touch the block (the
    // other claimant's job may be perfectly healthy) -- it only reports the clash.
    *err = "rank " + std::to_string(rank) + " already joined " + name;
    c->creator_ = false;
    delete c;
    return XMPI_ERR_BOOTSTRAP;
  }
  me->pid = (int32_t)getpid();
  me->start_time = proc_start_time(getpid());
  me->device = -1;
  me->state.store(1, std::memory_order_release);
  const double left = timeout_s - (now_seconds() - t0);
  int rc = c->wait_all_state(1, left > 1.0 ? left : 1.0);
  if (rc != XMPI_OK) {
    *err = "not every rank joined " + name;
    delete c;
    return rc;
  }
  *out = c;
  return XMPI_OK;
}

Ctl::~Ctl() {
  if (base_) {
    if (creator_) shm_unlink(name_.c_str());
    if (share_mappings()) {
      std::lock_guard<std::mutex> g(g_shared_mu);
      auto it = g_shared.find(name_);
      if (it != g_shared.end() && it->second.base == base_) {
        if (--it->second.refs > 0) return;  // another rank of this process still uses the mapping
        g_shared.erase(it);
      }
    }
    munmap(base_, bytes_);
  }
}

void Ctl::unlink_name() {
  if (creator_) {
    shm_unlink(name_.c_str());
    creator_ = false;
  }
}

int Ctl::wait_all_state(int state, double timeout_s) {
  const double t0 = now_seconds();
  double looked = t0;
  Backoff bo;
  for (;;) {
    bool all = true;
    for (int r = 0; r < size_; r++)
      if (ranks_[r].state.load(std::memory_order_acquire) < state) {
        all = false;
        break;
      }
    if (all) return XMPI_OK;
    if (aborted()) return XMPI_ERR_PEER;
    const double t = now_seconds();
    if (t - t0 > timeout_s) return XMPI_ERR_TIMEOUT;
    if (watch_ && t - looked > 0.05) {  // (a rank that waits for a peer that is gone: an error now, not when a clock runs out)
      looked = t;
      (void)check_peers();
    }
    bo.pause();
  }
}

int Ctl::barrier(double timeout_s, Backoff* ext) {
  if (size_ == 1) return XMPI_OK;
  const uint32_t gen = hdr_->bar_gen.load(std::memory_order_acquire);
  const uint32_t arrived = hdr_->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1;
  if (arrived == (uint32_t)size_) {
    hdr_->bar_count.store(0, std::memory_order_relaxed);
    hdr_->bar_gen.store(gen + 1, std::memory_order_release);
    return XMPI_OK;
  }
  const double t0 = now_seconds();
  double looked = t0;
  Backoff own;
  Backoff& bo = ext ? *ext : own;
  while (hdr_->bar_gen.load(std::memory_order_acquire) == gen) {
    if (aborted()) return XMPI_ERR_PEER;
    const double t = now_seconds();
    if (t - t0 > timeout_s) return XMPI_ERR_TIMEOUT;
    if (watch_ && t - looked > 0.05) {
      looked = t;
      (void)check_peers();
    }
    bo.pause();
  }
  return XMPI_OK;
}

}  // namespace xmpi
