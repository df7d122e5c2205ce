// tests/devsim/runtime.cpp -- the HIP runtime of tests/devsim (TEST INFRASTRUCTURE; see include/hip/hip_runtime_api.h).
//
//   devices     DEVSIM_DEVICES virtual devices (default 1); the current device is per thread, as in HIP
//   memory      device memory = POSIX shared memory objects (/dev/shm/devsim.<pid>.<n>), so that hipIpcOpenMemHandle in
//               ANOTHER process maps the same bytes at another address -- exactly what a peer's mapping is; pinned and
//               registered host memory is ordinary memory that is remembered
//   streams     one worker thread each, operations in order; events, cross-stream waits, capture into replayable graphs
//   kernels     one OS thread per resident block (DEVSIM_RESIDENT at most per launch, the rest queue behind them like
//               blocks beyond a GPU's wave slots), the block's lanes are fibers of that thread (hip_runtime.h)
//   schedule    DEVSIM_FUZZ=<seed>: lanes and threads give way at random before system-scope accesses
//   XCDs        DEVSIM_XCD_MAP: how blocks are dealt round the 8 XCDs (what HW_REG_XCC_ID reads), including dispatchers
//               that do NOT deal small grids round all of them (the split form's guard and its fallback)
#include <errno.h>
#include <fcntl.h>
#include <sched.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <dirent.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

#if defined(__has_feature)
#if __has_feature(thread_sanitizer)
#define DEVSIM_TSAN 1
#endif
#endif
// Under the sanitizer a block is ONE sanitizer thread: its lanes change stacks behind the sanitizer's back, which is sound only
// because build.py compiles the kernel sources and this file with -tsan-instrument-func-entry-exit=0 (no shadow call stack to
// corrupt; a report names the racing access by file:line, without its callers).  One sanitizer "fiber" per lane
// (-DDEVSIM_TSAN_FIBERS) keeps the call stacks, but a fold of 8 ranks keeps thousands of lanes alive and the sanitizer has 256
// thread slots: slots are recycled, their history is merged, and a seeded race goes unreported (driver.cpp --seed-race).
#ifdef DEVSIM_TSAN
extern "C" {
void* __tsan_get_current_fiber(void);
void* __tsan_create_fiber(unsigned flags);
void __tsan_destroy_fiber(void* fiber);
void __tsan_switch_to_fiber(void* fiber, unsigned flags);
void __tsan_acquire(void* addr);
void __tsan_release(void* addr);
}
#endif

// ---- lane switch: callee-saved registers + stack pointer (x86-64 System V) ---------------------------------------------------
extern "C" void devsim_ctx_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl devsim_ctx_switch
.type devsim_ctx_switch,@function
devsim_ctx_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size devsim_ctx_switch,.-devsim_ctx_switch
)");

namespace devsim {

thread_local LaneRegs tl_lane;

namespace {

double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

long env_long(const char* name, long dflt) {
  const char* v = getenv(name);
  return (v && *v) ? strtol(v, nullptr, 0) : dflt;
}

struct Config {
  int devices;
  int resident;
  long fuzz;  // 0 = off
  int xcd_map;  // 0 round robin, 1 round robin continuing where the last grid stopped, 2 small grids never reach XCD 7,
                // 3 small grids miss XCD 7 now and then (seeded), 4 pairs of consecutive blocks share an XCD
  size_t stack_bytes;
  Config() {
    devices = (int)std::max(1l, std::min(16l, env_long("DEVSIM_DEVICES", 1)));
    resident = (int)std::max(1l, std::min(1024l, env_long("DEVSIM_RESIDENT", 24)));
    fuzz = env_long("DEVSIM_FUZZ", 0);
    const char* m = getenv("DEVSIM_XCD_MAP");
    std::string s = m ? m : "rr";
    xcd_map = s == "rr" ? 0 : s == "continue" ? 1 : s == "small_miss" ? 2 : s == "flaky" ? 3 : s == "pairs" ? 4 : 0;
#ifdef DEVSIM_TSAN
    stack_bytes = 512u << 10;
#else
    stack_bytes = 128u << 10;
#endif
    stack_bytes = (size_t)env_long("DEVSIM_STACK_KB", (long)(stack_bytes >> 10)) << 10;
  }
};
Config& cfg() {
  static Config* c = new Config;
  return *c;
}

thread_local int tl_device = 0;
thread_local hipError_t tl_error = hipSuccess;
hipError_t fail(hipError_t e) {
  tl_error = e;
  return e;
}

// ---- schedule perturbation ------------------------------------------------------------------------------------------------
std::atomic<uint64_t> g_thread_counter{1};
thread_local uint64_t tl_rng = 0;
inline uint64_t rng_next() {
  if (tl_rng == 0) tl_rng = (uint64_t)cfg().fuzz * 0x9E3779B97F4A7C15ull + g_thread_counter.fetch_add(1) * 0xBF58476D1CE4E5B9ull + 1;
  uint64_t x = tl_rng;
  x ^= x << 13;
  x ^= x >> 7;
  x ^= x << 17;
  tl_rng = x;
  return x;
}

// ---- memory -------------------------------------------------------------------------------------------------------------------
enum { KIND_DEVICE = 0, KIND_PINNED = 1, KIND_REGISTERED = 2, KIND_IPC = 3 };
struct Alloc {
  char* base;
  size_t bytes;
  int device, kind;
  unsigned flags;
  uint64_t id;        // device allocations: the number in the shared-memory object's name
  int owner_pid;      // KIND_IPC: whose allocation this maps
  int refs;           // KIND_IPC: opens of the same handle
};
std::mutex g_mem_mu;
std::atomic<uint64_t> g_alloc_version{1};  // bumped whenever the table changes (the traffic counters' per-thread lookup cache)
std::map<uintptr_t, Alloc>& allocs() {
  static auto* m = new std::map<uintptr_t, Alloc>;
  return *m;
}
std::atomic<uint64_t> g_alloc_id{1};
std::atomic<size_t> g_device_bytes{0};

std::string shm_name(int pid, uint64_t id) { return "/devsim." + std::to_string(pid) + "." + std::to_string(id); }

Alloc* find_alloc_locked(const void* p) {
  auto& m = allocs();
  auto it = m.upper_bound((uintptr_t)p);
  if (it == m.begin()) return nullptr;
  --it;
  Alloc& a = it->second;
  return ((uintptr_t)p < (uintptr_t)a.base + a.bytes) ? &a : nullptr;
}

// ---- traffic accounting (DEVSIM_TRAFFIC=1) ---------------------------------------------------------------------------------------
// Which device's memory the kernels of which device read and wrote, in bytes: build.py --traffic compiles the kernel sources
// with -fsanitize-coverage=trace-loads,trace-stores, every load / store of theirs calls a hook below, the hook looks the address
// up in the allocation table (device memory and IPC mappings: the owning device; pinned / registered host memory: "host"; the
// rest -- stacks, shared variables -- is not memory traffic).  Copies by the runtime (hipMemcpyAsync: the copy engines) count as
// a read of the source and a write of the destination by the stream's device.  What a link would carry is row != column.
constexpr int kOwnerHost = 16;
std::atomic<uint64_t> g_traffic[3][16][kOwnerHost + 1][2];  // [0 payload, 1 flag pages, 2 small blocks][executing device][owner][0 load, 1 store]
bool traffic_on() {
  static const bool on = env_long("DEVSIM_TRAFFIC", 0) != 0;
  return on;
}
// DEVSIM_TRAFFIC=2: also by kernel name, printed when the process ends (local / remote / host, data and flag words)
std::mutex g_by_kernel_mu;
std::map<std::string, std::array<uint64_t, 12>>& by_kernel() {
  static auto* m = new std::map<std::string, std::array<uint64_t, 12>>;
  return *m;
}
void print_by_kernel() {
  std::lock_guard<std::mutex> g(g_by_kernel_mu);
  for (auto& kv : by_kernel()) {
    const auto& v = kv.second;
    fprintf(stderr, "devsim[%d] traffic %-28s payload: local %llu/%llu remote %llu/%llu host+tables %llu/%llu  flag pages: local %llu/%llu remote %llu/%llu - %llu/%llu (loads/stores)\n",
            (int)getpid(), kv.first.c_str(), (unsigned long long)v[0], (unsigned long long)v[1], (unsigned long long)v[2], (unsigned long long)v[3],
            (unsigned long long)v[4], (unsigned long long)v[5], (unsigned long long)v[6], (unsigned long long)v[7], (unsigned long long)v[8],
            (unsigned long long)v[9], (unsigned long long)v[10], (unsigned long long)v[11]);
  }
}
struct TrafficLocal {
  uint64_t n[3][kOwnerHost + 1][2] = {};
  bool dirty = false;
  uintptr_t lo = 1, hi = 0;  // the range the last lookup fell into (an allocation, or the gap between two)
  int owner = -1, cat = 0;
  uint64_t version = 0;
};
thread_local TrafficLocal tl_traffic;

// owner of the address (-1: not memory a link or an HBM channel would see) and, in *cat, what it is: 0 payload (the windows, the
// heaps of xmpi_malloc, whatever a caller allocated: blocks of 256 KiB and more), 1 a flag page (the allocations the library asks
// for as uncached / fine-grained: flag words, boxes, LL lines), 2 a small block of device memory (pointer tables and status
// words every lane of a kernel reads: a scalar load from a cache on the GPU, counted per lane here)
int traffic_owner(uintptr_t a, int* cat) {
  TrafficLocal& t = tl_traffic;
  const uint64_t v = g_alloc_version.load(std::memory_order_acquire);
  if (t.version == v && a >= t.lo && a < t.hi) {
    *cat = t.cat;
    return t.owner;
  }
  std::lock_guard<std::mutex> g(g_mem_mu);
  auto& m = allocs();
  auto it = m.upper_bound(a);
  uintptr_t lo = 0, hi = ~(uintptr_t)0;
  int owner = -1, c = 0;
  if (it != m.end()) hi = it->first;
  if (it != m.begin()) {
    --it;
    const Alloc& al = it->second;
    const uintptr_t end = (uintptr_t)al.base + al.bytes;
    if (a < end) {
      lo = (uintptr_t)al.base;
      hi = end;
      owner = (al.kind == KIND_DEVICE || al.kind == KIND_IPC) ? al.device : kOwnerHost;
      c = owner == kOwnerHost ? 0 : (al.flags & (hipDeviceMallocUncached | hipDeviceMallocFinegrained)) ? 1 : al.bytes < ((size_t)256 << 10) ? 2 : 0;
    } else {
      lo = end;
    }
  }
  t.lo = lo;
  t.hi = hi;
  t.owner = owner;
  t.cat = c;
  t.version = g_alloc_version.load(std::memory_order_acquire);
  *cat = c;
  return owner;
}
inline void traffic_count(const void* p, size_t bytes, int store) {
  int cat = 0;
  const int owner = traffic_owner((uintptr_t)p, &cat);
  if (owner < 0) return;
  tl_traffic.n[cat][owner][store] += bytes;
  tl_traffic.dirty = true;
}
void traffic_flush(int device, const char* kernel = nullptr) {
  TrafficLocal& t = tl_traffic;
  if (!t.dirty) return;
  static const bool named = env_long("DEVSIM_TRAFFIC", 0) >= 2;
  if (named) {
    std::string name = kernel ? kernel : "(copy)";
    const size_t lt = name.find('<');
    if (lt != std::string::npos) name.resize(lt);
    while (!name.empty() && name[0] == '(' && kernel) name.erase(0, 1);
    std::lock_guard<std::mutex> g(g_by_kernel_mu);
    if (by_kernel().empty()) atexit(print_by_kernel);
    auto& v = by_kernel()[name];
    for (int f = 0; f < 3; f++)
      for (int o = 0; o <= kOwnerHost; o++)
        for (int k = 0; k < 2; k++) v[(size_t)((f == 1 ? 6 : 0) + (f == 2 ? 4 : o == kOwnerHost ? 4 : o == device ? 0 : 2) + k)] += t.n[f][o][k];
  }
  for (int f = 0; f < 3; f++)
    for (int o = 0; o <= kOwnerHost; o++)
      for (int k = 0; k < 2; k++)
        if (t.n[f][o][k]) {
          g_traffic[f][device][o][k].fetch_add(t.n[f][o][k], std::memory_order_relaxed);
          t.n[f][o][k] = 0;
        }
  t.dirty = false;
}

void unlink_own_objects() {
  std::lock_guard<std::mutex> g(g_mem_mu);
  for (auto& kv : allocs())
    if (kv.second.kind == KIND_DEVICE) shm_unlink(shm_name((int)getpid(), kv.second.id).c_str());
}

// objects of processes that are gone (a test that was killed): give the memory back
void sweep_dead_objects() {
  DIR* d = opendir("/dev/shm");
  if (!d) return;
  while (dirent* e = readdir(d)) {
    int pid = 0;
    unsigned long long id = 0;
    if (sscanf(e->d_name, "devsim.%d.%llu", &pid, &id) == 2 && pid > 0 && kill(pid, 0) != 0 && errno == ESRCH)
      shm_unlink((std::string("/") + e->d_name).c_str());
  }
  closedir(d);
}

struct Startup {
  Startup() {
    sweep_dead_objects();
    atexit(unlink_own_objects);
  }
};
void startup() { static Startup* s = new Startup; (void)s; }

constexpr uint64_t kIpcMagic = 0x64657673696d3031ull;  // "devsim01"
struct IpcHandle {
  uint64_t magic;
  int32_t pid, device;
  uint64_t id, bytes, offset;
  uint32_t flags;  // of the allocation (uncached / fine-grained: the library's flag pages)
};
static_assert(sizeof(IpcHandle) <= sizeof(hipIpcMemHandle_t), "handle");

// ---- events, streams, graphs ----------------------------------------------------------------------------------------------------
}  // namespace
}  // namespace devsim

struct ihipEvent_t {
  std::mutex mu;
  std::condition_variable cv;
  uint64_t recorded = 0, done = 0;  // number of the last record enqueued / completed
  double t_done = 0;
  unsigned flags = 0;
};

namespace devsim {
namespace {
struct Op {
  enum Kind { FUNC, KERNEL, RECORD, WAIT } kind = FUNC;
  std::function<void()> fn;
  std::shared_ptr<KernelLaunch> kernel;
  hipEvent_t ev = nullptr;
  uint64_t gen = 0;  // RECORD: number of this record (0: take the next one when it executes -- graph replays); WAIT: the record waited for
};
}  // namespace
}  // namespace devsim

struct ihipGraph {
  std::vector<devsim::Op> ops;
};
struct hipGraphExec {
  std::vector<devsim::Op> ops;
};

struct ihipStream_t {
  int device = 0;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<devsim::Op> q;
  uint64_t enqueued = 0, completed = 0;
  bool capturing = false;
  ihipGraph* capture = nullptr;
  std::thread worker;
};

namespace devsim {
namespace {

void event_complete(hipEvent_t e, uint64_t gen) {
  std::lock_guard<std::mutex> g(e->mu);
  if (gen == 0) gen = ++e->recorded;
  if (gen > e->done) {
    e->done = gen;
    e->t_done = now_s();
  }
  e->cv.notify_all();
}

void run_kernel(const KernelLaunch& k, int device);

void execute(Op& op, int device) {
  switch (op.kind) {
    case Op::FUNC: op.fn(); break;
    case Op::KERNEL: run_kernel(*op.kernel, device); break;
    case Op::RECORD: event_complete(op.ev, op.gen); break;
    case Op::WAIT: {
      std::unique_lock<std::mutex> g(op.ev->mu);
      op.ev->cv.wait(g, [&] { return op.ev->done >= op.gen; });
      break;
    }
  }
}

void stream_main(ihipStream_t* s) {
  tl_device = s->device;
  for (;;) {
    Op op;
    {
      std::unique_lock<std::mutex> g(s->mu);
      s->cv.wait(g, [&] { return !s->q.empty(); });
      op = std::move(s->q.front());
      s->q.pop_front();
    }
    execute(op, s->device);
    {
      std::lock_guard<std::mutex> g(s->mu);
      s->completed++;
    }
    s->cv.notify_all();
  }
}

std::mutex g_streams_mu;
std::vector<ihipStream_t*>& all_streams() {
  static auto* v = new std::vector<ihipStream_t*>;
  return *v;
}
ihipStream_t* new_stream(int device) {
  startup();
  ihipStream_t* s = new ihipStream_t;
  s->device = device;
  s->worker = std::thread(stream_main, s);
  s->worker.detach();  // streams live as long as the process (the library pools them)
  std::lock_guard<std::mutex> g(g_streams_mu);
  all_streams().push_back(s);
  return s;
}
ihipStream_t* null_stream(int device) {
  static std::mutex mu;
  static ihipStream_t* per_device[16] = {};
  std::lock_guard<std::mutex> g(mu);
  if (!per_device[device]) per_device[device] = new_stream(device);
  return per_device[device];
}
ihipStream_t* resolve(hipStream_t s) { return s ? s : null_stream(tl_device); }

void enqueue(ihipStream_t* s, Op&& op) {
  std::lock_guard<std::mutex> g(s->mu);
  if (s->capturing) {
    s->capture->ops.push_back(std::move(op));
    return;
  }
  s->q.push_back(std::move(op));
  s->enqueued++;
  s->cv.notify_all();
}

void stream_sync(ihipStream_t* s) {
  std::unique_lock<std::mutex> g(s->mu);
  const uint64_t want = s->enqueued;
  s->cv.wait(g, [&] { return s->completed >= want; });
}

void device_sync(int device) {
  std::vector<ihipStream_t*> v;
  {
    std::lock_guard<std::mutex> g(g_streams_mu);
    for (ihipStream_t* s : all_streams())
      if (s->device == device) v.push_back(s);
  }
  for (ihipStream_t* s : v) stream_sync(s);
}

// ---- kernel execution: blocks as threads, lanes as fibers -----------------------------------------------------------------------
enum { L_READY = 0, L_BARRIER = 1, L_DONE = 2 };
struct Lane {
  void* sp = nullptr;
  int state = L_DONE;
  bool spun = false;
#ifdef DEVSIM_TSAN_FIBERS
  void* fiber = nullptr;
#endif
};

struct BlockRunner {
  const KernelLaunch* k = nullptr;
  std::vector<Lane> lanes;
  char* stacks = nullptr;
  size_t nstacks = 0;
  void* sched_sp = nullptr;
  int cur = -1, live = 0, waiting = 0;
  unsigned linear_block = 0, xcc = 0;
  uint64_t barrier_tag = 0;  // (address the sanitizer hangs the barrier's happens-before on)
  std::vector<uint64_t> xchg;
#ifdef DEVSIM_TSAN_FIBERS
  void* sched_fiber = nullptr;
#endif
  // Every resident block of every launch is a thread with a runner of its own: mapping, faulting in (a page or two per lane) and
  // unmapping 256 .. 1024 lane stacks per block-thread per launch was most of what a collective cost here (system time twice the user
  // time: eight ranks x 24 threads a launch).  Stack regions are kept and handed from launch to launch -- the pages stay.
  struct StackPool {
    std::mutex mu;
    std::vector<std::pair<char*, size_t>> idle;
  };
  static StackPool& pool() {
    static StackPool* p = new StackPool;  // (never destroyed: kernels may still run while the process exits)
    return *p;
  }
  void give_back() {
    if (!stacks) return;
    {
      std::lock_guard<std::mutex> g(pool().mu);  // (also the happens-before edge the sanitizer needs between two threads' uses of a region)
      if (pool().idle.size() < 48) {  // (two launches in flight of 24 resident blocks each; a region keeps a page or two per lane)
        pool().idle.emplace_back(stacks, nstacks);
        stacks = nullptr;
        nstacks = 0;
        return;
      }
    }
    munmap(stacks, nstacks * cfg().stack_bytes);
    stacks = nullptr;
    nstacks = 0;
  }
  ~BlockRunner() {
#ifdef DEVSIM_TSAN_FIBERS
    for (Lane& l : lanes)
      if (l.fiber) __tsan_destroy_fiber(l.fiber);
#endif
    give_back();
  }
  void ensure(size_t n) {
    if (n <= nstacks) return;
    give_back();
    {
      std::lock_guard<std::mutex> g(pool().mu);
      auto& idle = pool().idle;
      for (size_t i = idle.size(); i-- > 0;)
        if (idle[i].second >= n) {
          stacks = idle[i].first;
          nstacks = idle[i].second;
          idle.erase(idle.begin() + (long)i);
          break;
        }
    }
    if (!stacks) {
      stacks = (char*)mmap(nullptr, n * cfg().stack_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
      if (stacks == MAP_FAILED) {
        perror("devsim: lane stacks");
        abort();
      }
      nstacks = n;
    }
    lanes.resize(nstacks);
    xchg.resize(nstacks);
  }
  void to_lane(int i);
  void to_sched();
  void run_block(const KernelLaunch& kl, unsigned linear, unsigned xcc_id);
};
thread_local BlockRunner* tl_runner = nullptr;

__attribute__((no_sanitize("thread"))) void lane_entry() {
  BlockRunner* r = tl_runner;
  r->k->lane();
  r->lanes[(size_t)r->cur].state = L_DONE;
  r->live--;
  r->to_sched();
  abort();  // a finished lane is never resumed
}

void BlockRunner::to_lane(int i) {
  cur = i;
  const unsigned bx = k->block.x, by = k->block.y;
  tl_lane.tid = {(unsigned)i % bx, ((unsigned)i / bx) % by, (unsigned)i / (bx * by)};
#ifdef DEVSIM_TSAN_FIBERS
  __tsan_switch_to_fiber(lanes[(size_t)i].fiber, 0);
#endif
  devsim_ctx_switch(&sched_sp, lanes[(size_t)i].sp);
}
void BlockRunner::to_sched() {
#ifdef DEVSIM_TSAN_FIBERS
  __tsan_switch_to_fiber(sched_fiber, 0);
#endif
  devsim_ctx_switch(&lanes[(size_t)cur].sp, sched_sp);
}

void BlockRunner::run_block(const KernelLaunch& kl, unsigned linear, unsigned xcc_id) {
  k = &kl;
  linear_block = linear;
  xcc = xcc_id;
  const size_t n = (size_t)kl.block.x * kl.block.y * kl.block.z;
  ensure(n);
  tl_lane.bdim = {kl.block.x, kl.block.y, kl.block.z};
  tl_lane.gdim = {kl.grid.x, kl.grid.y, kl.grid.z};
  tl_lane.bid = {linear % kl.grid.x, (linear / kl.grid.x) % kl.grid.y, linear / (kl.grid.x * kl.grid.y)};
#ifdef DEVSIM_TSAN_FIBERS
  sched_fiber = __tsan_get_current_fiber();
#endif
  for (size_t i = 0; i < n; i++) {
    Lane& l = lanes[i];
    uintptr_t top = ((uintptr_t)stacks + (i + 1) * cfg().stack_bytes) & ~(uintptr_t)15;
    uint64_t* sp = (uint64_t*)top;
    *--sp = 0;                          // (where lane_entry would return to: it never does)
    *--sp = (uint64_t)(uintptr_t)&lane_entry;
    for (int r = 0; r < 6; r++) *--sp = 0;  // rbp rbx r12 r13 r14 r15
    l.sp = sp;
    l.state = L_READY;
    l.spun = false;
#ifdef DEVSIM_TSAN_FIBERS
    if (!l.fiber) l.fiber = __tsan_create_fiber(0);
#endif
  }
  live = (int)n;
  waiting = 0;
  unsigned idle_rounds = 0;
  while (live > 0) {
    bool progress = false;
    for (size_t i = 0; i < n && live > 0; i++) {
      Lane& l = lanes[i];
      if (l.state != L_READY) continue;
      l.spun = false;
      to_lane((int)i);
      if (!l.spun) progress = true;
    }
    if (live > 0 && waiting == live) {  // the barrier opens
      for (size_t i = 0; i < n; i++)
        if (lanes[i].state == L_BARRIER) lanes[i].state = L_READY;
      waiting = 0;
      progress = true;
    }
    if (!progress) {  // every lane that can run is waiting for somebody else's store: let them
      if (++idle_rounds < 64) sched_yield();
      else {
        timespec ts{0, 20000};
        nanosleep(&ts, nullptr);
      }
    } else {
      idle_rounds = 0;
    }
  }
  k = nullptr;
}

std::atomic<unsigned> g_dispatch_pos[16];  // where each device's dispatcher stands (xcd_map "continue")
std::atomic<uint64_t> g_launch_counter{0};

unsigned xcc_for(unsigned linear, unsigned nblocks, unsigned start, uint64_t launch_no) {
  switch (cfg().xcd_map) {
    case 1: return (start + linear) % 8;
    case 2: return (nblocks <= 64 && linear % 8 == 7) ? 0u : linear % 8;
    case 3: {
      // one small grid in four misses XCD 7
      const bool miss = nblocks <= 64 && ((launch_no * 0x9E3779B97F4A7C15ull + (uint64_t)cfg().fuzz) >> 61) == 0;
      return (miss && linear % 8 == 7) ? 0u : linear % 8;
    }
    case 4: return (linear / 2) % 8;
    default: return linear % 8;
  }
}

// DEVSIM_STATS=1: what ran, by kernel name, when the process ends
std::mutex g_stats_mu;
std::map<std::string, std::pair<uint64_t, uint64_t>>& stats() {
  static auto* m = new std::map<std::string, std::pair<uint64_t, uint64_t>>;
  return *m;
}
void print_stats() {
  std::lock_guard<std::mutex> g(g_stats_mu);
  for (auto& kv : stats())
    fprintf(stderr, "devsim[%d]: %8llu launches %10llu blocks  %s\n", (int)getpid(), (unsigned long long)kv.second.first,
            (unsigned long long)kv.second.second, kv.first.c_str());
}
void count_launch(const KernelLaunch& k, size_t nblocks) {
  static const bool on = env_long("DEVSIM_STATS", 0) != 0;
  if (!on) return;
  std::lock_guard<std::mutex> g(g_stats_mu);
  if (stats().empty()) atexit(print_stats);
  std::string name = k.name ? k.name : "?";
  const size_t lt = name.find('<');  // (template arguments: one line per kernel)
  if (lt != std::string::npos) name.resize(lt);
  while (!name.empty() && name[0] == '(') name.erase(0, 1);
  auto& e = stats()[name];
  e.first++;
  e.second += nblocks;
}

void run_kernel(const KernelLaunch& k, int device) {
  count_launch(k, (size_t)k.grid.x * k.grid.y * k.grid.z);
  if (k.ev_start) event_complete(k.ev_start, k.gen_start);
  const size_t nblocks = (size_t)k.grid.x * k.grid.y * k.grid.z;
  const unsigned start = g_dispatch_pos[device].fetch_add((unsigned)nblocks);
  const uint64_t launch_no = g_launch_counter.fetch_add(1);
  std::atomic<size_t> next{0};
  auto body = [&]() {
    tl_device = device;
    BlockRunner runner;
    tl_runner = &runner;
    for (;;) {
      const size_t b = next.fetch_add(1);
      if (b >= nblocks) break;
      runner.run_block(k, (unsigned)b, xcc_for((unsigned)b, (unsigned)nblocks, start, launch_no));
    }
    traffic_flush(device, k.name);
    tl_runner = nullptr;
  };
  const size_t threads = std::min(nblocks, (size_t)cfg().resident);
  if (threads <= 1) {
    body();
  } else {
    std::vector<std::thread> ts;
    ts.reserve(threads);
    for (size_t i = 0; i < threads; i++) ts.emplace_back(body);
    for (auto& t : ts) t.join();
  }
  if (k.ev_stop) event_complete(k.ev_stop, k.gen_stop);
}

}  // namespace

// ---- DEVSIM_CORRUPT_FORM=<form>[@<device>] ------------------------------------------------------------------------------------------
// A node on which ONE schedule gives wrong answers -- the things only a node can say ("Open": plain stores into
// peer memory, system-scope loads / stores over a link, 8-byte lines into an uncached allocation of another device).  On <device>
// (default 1) the named kernel's data accesses of the named kind THAT TOUCH ANOTHER DEVICE'S MEMORY get bit 0 of their first byte
// flipped:
//   fold       dsync_fold_kernel: its (non-temporal) stores into the peers' receive buffers -- the one-kernel fold, push-only, bcast
//   split      dsync_body_kernel: the same stores of the split form's data kernel (run with XMPI_KERNEL_MODE=1: non-temporal stores at
//              every size) -- its system-scope form (body_sys) is NOT affected: the ladder's first rung
//   split_sys  dsync_body_kernel in both forms
//   ring | rhd | tree             dsync_sched_kernel, pull form of that schedule: what it LOADS from a peer
//   ring_push | rhd_push | tree_push    ... push form: what it STORES into a peer
//   ll         the LL kernels (launched and agent): the lines they store into the peers' flag allocations (data half of a line)
//   split_stale | fold_stale   the (non-temporal) LOADS of the split form's data kernel / the one-kernel fold from a peer's memory return
//              what the FIRST load of that address returned -- an L2 that keeps lines of a peer's buffer across collectives because
//              the schedule's acquire never reached it.  Invisible to a check that reads every buffer once; the check's second pass
//              (the inputs changed in place) is there for exactly this
//   p2p        the blocking Receive's copy kernels (the lingering agent, the pull kernel): what they LOAD out of the sender's memory
// Found by xmpi_tune / xmpi_init's self-check, or not at all: that is the test (tests/test_devsim.py).
namespace {
struct CorruptSpec {
  bool on = false;
  int device = 1, kinds = 0;
  const char* names[3] = {nullptr, nullptr, nullptr};
  unsigned scheds = 0;  // bit per DsyncSched value; 0 = the kernel carries no schedule
  int push = -1;
  bool ll_lines = false;
  bool stale = false;  // not a flipped bit: a load returns what the FIRST load of that address returned (a cache nobody invalidated)
  CorruptSpec() {
    const char* e = getenv("DEVSIM_CORRUPT_FORM");
    if (!e || !*e) return;
    std::string f = e;
    const size_t at = f.find('@');
    if (at != std::string::npos) {
      device = atoi(f.c_str() + at + 1);
      f.resize(at);
    }
    auto sched = [&](unsigned mask, int p) {
      names[0] = "dsync_sched_kernel";
      scheds = mask;
      push = p;
      kinds = p ? ACC_SYS_STORE : ACC_SYS_LOAD;
    };
    if (f == "fold") names[0] = "dsync_fold_kernel", kinds = ACC_NT_STORE;
    else if (f == "split") names[0] = "dsync_body_kernel", kinds = ACC_NT_STORE;
    else if (f == "split_sys") names[0] = "dsync_body_kernel", kinds = ACC_NT_STORE | ACC_SYS_STORE;
    else if (f == "ring" || f == "ring_push") sched(1u << 1 | 1u << 3, f == "ring_push");
    else if (f == "rhd" || f == "rhd_push") sched(1u << 2, f == "rhd_push");
    else if (f == "tree" || f == "tree_push") sched(1u << 4 | 1u << 5, f == "tree_push");
    else if (f == "split_stale") names[0] = "dsync_body_kernel", kinds = ACC_NT_LOAD, stale = true;
    else if (f == "fold_stale") names[0] = "dsync_fold_kernel", kinds = ACC_NT_LOAD, stale = true;
    else if (f == "p2p") names[0] = "p2p_agent_kernel", names[1] = "p2p_pull_kernel", kinds = ACC_NT_LOAD;
    else if (f == "ll") names[0] = "ll_reduce_kernel", names[1] = "ll_copy_kernel", names[2] = "ll_agent_kernel", kinds = ACC_FLAG_STORE, ll_lines = true;
    else {
      fprintf(stderr, "devsim: DEVSIM_CORRUPT_FORM=%s: no such form\n", e);
      abort();
    }
    on = true;
  }
};
CorruptSpec& corrupt_spec() {
  static CorruptSpec* s = new CorruptSpec;
  return *s;
}
std::atomic<uint64_t> g_corrupted{0};
}  // namespace
bool g_corrupt_armed = corrupt_spec().on;

void corrupt_bits(const void* addr, void* value, unsigned bytes, int kind) {
  const CorruptSpec& cs = corrupt_spec();
  const BlockRunner* r = tl_runner;
  if (!(cs.kinds & kind) || !r || !r->k || !r->k->name || tl_device != cs.device) return;
  bool named = false;
  for (const char* n : cs.names) named = named || (n && strstr(r->k->name, n));
  if (!named) return;
  if (cs.scheds) {
    const unsigned tag = r->k->tag;
    if (!(tag & 0x100u) || !((cs.scheds >> ((tag & 0xffu) >> 1)) & 1u) || (int)(tag & 1u) != cs.push) return;
  }
  {
    std::lock_guard<std::mutex> g(g_mem_mu);
    const Alloc* a = find_alloc_locked(addr);
    if (!a || !(a->kind == KIND_DEVICE || a->kind == KIND_IPC) || a->device == tl_device) return;  // (another device's memory only)
    if (cs.ll_lines) {  // the LL slots of a flag allocation (kernels.h kLLOff = 1 MiB), not its flag words
      if (!(a->flags & (hipDeviceMallocUncached | hipDeviceMallocFinegrained)) || (uintptr_t)addr - (uintptr_t)a->base < ((uintptr_t)1 << 20)) return;
    }
  }
  if (cs.stale) {
    static std::mutex mu;
    static std::map<uintptr_t, std::array<unsigned char, 16>>* first = new std::map<uintptr_t, std::array<unsigned char, 16>>;
    std::lock_guard<std::mutex> g(mu);
    auto it = first->find((uintptr_t)addr);
    if (it == first->end()) {
      std::array<unsigned char, 16> v{};
      memcpy(v.data(), value, std::min<unsigned>(bytes, 16));
      (*first)[(uintptr_t)addr] = v;
    } else {
      if (memcmp(value, it->second.data(), std::min<unsigned>(bytes, 16)) != 0) g_corrupted.fetch_add(1);
      memcpy(value, it->second.data(), std::min<unsigned>(bytes, 16));
    }
    return;
  }
  (void)bytes;
  *reinterpret_cast<unsigned char*>(value) ^= 1u;
  if (g_corrupted.fetch_add(1) == 0 && env_long("DEVSIM_CORRUPT_VERBOSE", 0))
    fprintf(stderr, "devsim[%d]: DEVSIM_CORRUPT_FORM: first flipped bit in %s on device %d\n", (int)getpid(), r->k->name, tl_device);
}

// ---- what the kernels call ---------------------------------------------------------------------------------------------------------
void syncthreads() {
  BlockRunner* r = tl_runner;
  if (!r) return;
  Lane& l = r->lanes[(size_t)r->cur];
#ifdef DEVSIM_TSAN_FIBERS
  __tsan_release(&r->barrier_tag);
#endif
  l.state = L_BARRIER;
  r->waiting++;
  r->to_sched();
#ifdef DEVSIM_TSAN_FIBERS
  __tsan_acquire(&r->barrier_tag);
#endif
}

void yield_lane() {
  BlockRunner* r = tl_runner;
  if (!r) {
    sched_yield();
    return;
  }
  r->lanes[(size_t)r->cur].spun = true;
  r->to_sched();
}

void sync_point() {
  if (cfg().fuzz == 0) return;
  const uint64_t x = rng_next();
  if ((x & 7u) == 0) {
    BlockRunner* r = tl_runner;
    if (r) r->to_sched();  // (not "spun": the lane is not waiting for anything, it is merely late)
    else sched_yield();
  } else if ((x & 0xffu) == 1) {
    sched_yield();
  } else if ((x & 0xfffu) == 2) {
    timespec ts{0, (long)(1000 + (x >> 12) % 50000)};
    nanosleep(&ts, nullptr);
  }
}

uint64_t wall_clock_ticks() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (uint64_t)ts.tv_sec * 100000000ull + (uint64_t)ts.tv_nsec / 10u;
}

unsigned xcc_of_block() {
  BlockRunner* r = tl_runner;
  return r ? r->xcc : 0u;
}

uint64_t shfl_down_bits(uint64_t bits, unsigned delta, unsigned width) {
  BlockRunner* r = tl_runner;
  if (!r) return bits;
  const unsigned t = (unsigned)r->cur;
  r->xchg[t] = bits;
  syncthreads();
  const unsigned src = (t % width) + delta < width ? t + delta : t;
  const uint64_t out = src < r->xchg.size() && r->lanes[src].state != L_DONE ? r->xchg[src] : bits;
  syncthreads();
  return out;
}

void enqueue_kernel(hipStream_t stream, KernelLaunch&& k) {
  startup();
  ihipStream_t* s = resolve(stream);
  Op op;
  op.kind = Op::KERNEL;
  op.kernel = std::make_shared<KernelLaunch>(std::move(k));
  // the events of an extended launch count as recorded from now on (in a capture: from every replay on)
  bool capturing;
  {
    std::lock_guard<std::mutex> g(s->mu);
    capturing = s->capturing;
  }
  if (!capturing) {
    if (hipEvent_t e = op.kernel->ev_start) {
      std::lock_guard<std::mutex> g(e->mu);
      op.kernel->gen_start = ++e->recorded;
    }
    if (hipEvent_t e = op.kernel->ev_stop) {
      std::lock_guard<std::mutex> g(e->mu);
      op.kernel->gen_stop = ++e->recorded;
    }
  }
  enqueue(s, std::move(op));
}

}  // namespace devsim

// =========================================================================================================================================
// the API
// =========================================================================================================================================
using namespace devsim;

extern "C" {

hipError_t hipGetDeviceCount(int* count) {
  if (!count) return fail(hipErrorInvalidValue);
  *count = cfg().devices;
  return hipSuccess;
}
hipError_t hipSetDevice(int device) {
  if (device < 0 || device >= cfg().devices) return fail(hipErrorInvalidDevice);
  tl_device = device;
  return hipSuccess;
}
hipError_t hipGetDevice(int* device) {
  if (!device) return fail(hipErrorInvalidValue);
  *device = tl_device;
  return hipSuccess;
}
hipError_t hipDeviceSynchronize(void) {
  device_sync(tl_device);
  return hipSuccess;
}
hipError_t hipDeviceGetPCIBusId(char* busid, int len, int device) {
  if (!busid || len < 13 || device < 0 || device >= cfg().devices) return fail(hipErrorInvalidValue);
  snprintf(busid, (size_t)len, "0000:%02x:00.0", 0x10 + device);
  return hipSuccess;
}

namespace {
std::mutex g_peer_mu;
bool g_peer[16][16];
}  // namespace
hipError_t hipDeviceEnablePeerAccess(int peer, unsigned) {
  if (peer < 0 || peer >= cfg().devices || peer == tl_device) return fail(hipErrorInvalidDevice);
  std::lock_guard<std::mutex> g(g_peer_mu);
  if (g_peer[tl_device][peer]) return fail(hipErrorPeerAccessAlreadyEnabled);
  g_peer[tl_device][peer] = true;
  return hipSuccess;
}
hipError_t hipDeviceCanAccessPeer(int* can, int device, int peer) {
  if (!can || device < 0 || peer < 0 || device >= cfg().devices || peer >= cfg().devices) return fail(hipErrorInvalidValue);
  *can = device != peer;
  return hipSuccess;
}
hipError_t hipMemGetInfo(size_t* free_bytes, size_t* total_bytes) {
  const size_t total = (size_t)288 << 30;
  if (total_bytes) *total_bytes = total;
  if (free_bytes) *free_bytes = total - std::min(total, g_device_bytes.load());
  return hipSuccess;
}
hipError_t hipGetLastError(void) {
  const hipError_t e = tl_error;
  tl_error = hipSuccess;
  return e;
}
hipError_t hipPeekAtLastError(void) { return tl_error; }
const char* hipGetErrorString(hipError_t e) {
  switch (e) {
    case hipSuccess: return "no error";
    case hipErrorInvalidValue: return "invalid argument";
    case hipErrorOutOfMemory: return "out of memory";
    case hipErrorInvalidDevicePointer: return "invalid device pointer";
    case hipErrorInvalidDevice: return "invalid device ordinal";
    case hipErrorInvalidContext: return "invalid device context";
    case hipErrorInvalidHandle: return "invalid resource handle";
    case hipErrorNotReady: return "device not ready";
    case hipErrorPeerAccessAlreadyEnabled: return "peer access is already enabled";
    case hipErrorHostMemoryAlreadyRegistered: return "part or all of the requested memory range is already mapped";
    case hipErrorHostMemoryNotRegistered: return "pointer does not correspond to a registered memory region";
    default: return "devsim: error";
  }
}

// ---- memory -----------------------------------------------------------------------------------------------------------------------
hipError_t hipExtMallocWithFlags(void** ptr, size_t bytes, unsigned flags) {
  startup();
  if (!ptr) return fail(hipErrorInvalidValue);
  *ptr = nullptr;
  if (bytes == 0) return hipSuccess;
  // DEVSIM_FAIL_UNCACHED_ALLOC=1[@<device>]: the runtime has no uncached / fine-grained device memory (on that device)
  if (flags & (hipDeviceMallocUncached | hipDeviceMallocFinegrained)) {
    static const char* spec = getenv("DEVSIM_FAIL_UNCACHED_ALLOC");
    if (spec) {
      const char* at = strchr(spec, '@');
      if (!at || atoi(at + 1) == tl_device) return fail(hipErrorOutOfMemory);
    }
  }
  const size_t page = 4096, len = (bytes + page - 1) / page * page;
  const uint64_t id = g_alloc_id.fetch_add(1);
  const std::string name = shm_name((int)getpid(), id);
  const int fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0) return fail(hipErrorOutOfMemory);
  if (ftruncate(fd, (off_t)len) != 0) {
    close(fd);
    shm_unlink(name.c_str());
    return fail(hipErrorOutOfMemory);
  }
  void* p = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_NORESERVE, fd, 0);
  close(fd);
  if (p == MAP_FAILED) {
    shm_unlink(name.c_str());
    return fail(hipErrorOutOfMemory);
  }
  Alloc a{(char*)p, len, tl_device, KIND_DEVICE, flags, id, (int)getpid(), 1};
  {
    std::lock_guard<std::mutex> g(g_mem_mu);
    allocs()[(uintptr_t)p] = a;
    g_alloc_version.fetch_add(1);
  }
  g_device_bytes.fetch_add(len);
  *ptr = p;
  return hipSuccess;
}
hipError_t hipMalloc(void** ptr, size_t bytes) { return hipExtMallocWithFlags(ptr, bytes, hipDeviceMallocDefault); }

hipError_t hipFree(void* ptr) {
  if (!ptr) return hipSuccess;
  Alloc a;
  {
    std::lock_guard<std::mutex> g(g_mem_mu);
    auto it = allocs().find((uintptr_t)ptr);
    if (it == allocs().end() || it->second.kind != KIND_DEVICE) return fail(hipErrorInvalidValue);
    a = it->second;
  }
  device_sync(a.device);  // hipFree synchronises the device
  {
    std::lock_guard<std::mutex> g(g_mem_mu);
    allocs().erase((uintptr_t)ptr);
    g_alloc_version.fetch_add(1);
  }
  shm_unlink(shm_name((int)getpid(), a.id).c_str());
  munmap(a.base, a.bytes);
  g_device_bytes.fetch_sub(a.bytes);
  return hipSuccess;
}

// ---- guarded allocations (tests/scenarios.py sc_guard) --------------------------------------------------------------------------
// Device memory whose LAST byte is the last byte of a page, the page behind it inaccessible: a kernel that reads or writes one byte
// past a buffer's end -- a tail handled a packet too wide, a tile rounded up -- faults here, where a GPU serves the access out of the
// 2 MiB granule the allocation sits in and nobody ever knows.  The pointer returned is `bytes` before the guard page (so only as
// aligned as `bytes` is); release with devsim_guarded_free(ptr, bytes): -2 when the bytes in FRONT of the buffer (0xEE) were written.
// This process' mapping only: ranks must be threads.
extern "C" void* devsim_guarded_alloc(size_t bytes) {
  const size_t page = 4096, data = (std::max<size_t>(bytes, 1) + page - 1) / page * page;
  void* base = nullptr;
  if (hipMalloc(&base, data + page) != hipSuccess) return nullptr;
  memset(base, 0xEE, data);
  if (mprotect((char*)base + data, page, PROT_NONE) != 0) {
    (void)hipFree(base);
    return nullptr;
  }
  return (char*)base + data - bytes;
}
extern "C" int devsim_guarded_free(void* ptr, size_t bytes) {
  if (!ptr) return 0;
  const size_t page = 4096, data = (std::max<size_t>(bytes, 1) + page - 1) / page * page;
  char* base = (char*)ptr + bytes - data;
  int rc = 0;
  for (char* q = base; q < (char*)ptr; q++)  // what lies in FRONT of the buffer was filled with 0xEE: a store below the buffer's start shows here
    if ((unsigned char)*q != 0xEE) rc = -2;
  (void)mprotect(base + data, page, PROT_READ | PROT_WRITE);
  return hipFree(base) == hipSuccess ? rc : -1;
}

hipError_t hipHostMalloc(void** ptr, size_t bytes, unsigned flags) {
  startup();
  if (!ptr) return fail(hipErrorInvalidValue);
  *ptr = nullptr;
  if (bytes == 0) return hipSuccess;
  const size_t len = (bytes + 4095) / 4096 * 4096;
  void* p = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (p == MAP_FAILED) return fail(hipErrorOutOfMemory);
  Alloc a{(char*)p, len, tl_device, KIND_PINNED, flags, 0, (int)getpid(), 1};
  std::lock_guard<std::mutex> g(g_mem_mu);
  allocs()[(uintptr_t)p] = a;
  g_alloc_version.fetch_add(1);
  *ptr = p;
  return hipSuccess;
}
hipError_t hipHostFree(void* ptr) {
  if (!ptr) return hipSuccess;
  Alloc a;
  {
    std::lock_guard<std::mutex> g(g_mem_mu);
    auto it = allocs().find((uintptr_t)ptr);
    if (it == allocs().end() || it->second.kind != KIND_PINNED) return fail(hipErrorInvalidValue);
    a = it->second;
    allocs().erase(it);
    g_alloc_version.fetch_add(1);
  }
  munmap(a.base, a.bytes);
  return hipSuccess;
}
hipError_t hipHostRegister(void* ptr, size_t bytes, unsigned flags) {
  if (!ptr || bytes == 0) return fail(hipErrorInvalidValue);
  std::lock_guard<std::mutex> g(g_mem_mu);
  if (Alloc* a = find_alloc_locked(ptr)) {
    // Rank threads that address the control block through ONE mapping (XMPI_CTL_SHARE_MAPPING, so that the sanitizer sees one
    // address per word) all register it: counted here, where HIP would refuse the second one.
    if (a->kind == KIND_REGISTERED && a->base == (char*)ptr && bytes <= a->bytes) {
      a->refs++;
      return hipSuccess;
    }
    return fail(hipErrorHostMemoryAlreadyRegistered);
  }
  if (find_alloc_locked((char*)ptr + bytes - 1)) return fail(hipErrorHostMemoryAlreadyRegistered);
  allocs()[(uintptr_t)ptr] = Alloc{(char*)ptr, bytes, tl_device, KIND_REGISTERED, flags, 0, (int)getpid(), 1};
  g_alloc_version.fetch_add(1);
  return hipSuccess;
}
hipError_t hipHostUnregister(void* ptr) {
  std::lock_guard<std::mutex> g(g_mem_mu);
  auto it = allocs().find((uintptr_t)ptr);
  if (it == allocs().end() || it->second.kind != KIND_REGISTERED) return fail(hipErrorHostMemoryNotRegistered);
  if (--it->second.refs > 0) return hipSuccess;
  allocs().erase(it);
  g_alloc_version.fetch_add(1);
  return hipSuccess;
}
hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned) {
  if (!dev || !host) return fail(hipErrorInvalidValue);
  std::lock_guard<std::mutex> g(g_mem_mu);
  Alloc* a = find_alloc_locked(host);
  if (!a || (a->kind != KIND_PINNED && a->kind != KIND_REGISTERED)) return fail(hipErrorInvalidValue);
  *dev = host;
  return hipSuccess;
}
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* attr, const void* ptr) {
  if (!attr || !ptr) return fail(hipErrorInvalidValue);
  std::lock_guard<std::mutex> g(g_mem_mu);
  Alloc* a = find_alloc_locked(ptr);
  if (!a) return fail(hipErrorInvalidValue);  // (unregistered host memory: what ROCm answers too)
  memset(attr, 0, sizeof *attr);
  attr->device = a->device;
  attr->allocationFlags = a->flags;
  if (a->kind == KIND_DEVICE || a->kind == KIND_IPC) {
    attr->type = hipMemoryTypeDevice;
    attr->devicePointer = const_cast<void*>(ptr);
  } else {
    attr->type = hipMemoryTypeHost;
    attr->devicePointer = attr->hostPointer = const_cast<void*>(ptr);
  }
  return hipSuccess;
}
hipError_t hipMemGetAddressRange(hipDeviceptr_t* base, size_t* bytes, hipDeviceptr_t ptr) {
  std::lock_guard<std::mutex> g(g_mem_mu);
  Alloc* a = find_alloc_locked(ptr);
  if (!a || (a->kind != KIND_DEVICE && a->kind != KIND_IPC)) return fail(hipErrorInvalidDevicePointer);
  if (base) *base = a->base;
  if (bytes) *bytes = a->bytes;
  return hipSuccess;
}

namespace {
bool is_tracked(const void* p) {
  std::lock_guard<std::mutex> g(g_mem_mu);
  return find_alloc_locked(p) != nullptr;
}
}  // namespace

hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t stream) {
  if (bytes == 0) return hipSuccess;
  if (!dst || !src) return fail(hipErrorInvalidValue);
  ihipStream_t* s = resolve(stream);
  Op op;
  bool wait = false;
  if (!is_tracked(src)) {  // pageable source: staged before the call returns (the caller may reuse it at once)
    auto stage = std::make_shared<std::vector<char>>((const char*)src, (const char*)src + bytes);
    op.fn = [dst, stage] { memcpy(dst, stage->data(), stage->size()); };
  } else {
    const int dev = s->device;
    op.fn = [dst, src, bytes, dev] {
      memcpy(dst, src, bytes);
      if (traffic_on()) {
        traffic_count(src, bytes, 0);
        traffic_count(dst, bytes, 1);
        traffic_flush(dev);
      }
    };
    wait = !is_tracked(dst);  // pageable destination: the call returns when the bytes are there
  }
  enqueue(s, std::move(op));
  if (wait && !s->capturing) stream_sync(s);
  return hipSuccess;
}
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
  ihipStream_t* s = null_stream(tl_device);
  const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, s);
  if (e != hipSuccess) return e;
  stream_sync(s);
  return hipSuccess;
}
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return hipSuccess;
  if (!dst) return fail(hipErrorInvalidValue);
  Op op;
  op.fn = [dst, value, bytes] { memset(dst, value, bytes); };
  enqueue(resolve(stream), std::move(op));
  return hipSuccess;
}
hipError_t hipMemset(void* dst, int value, size_t bytes) {
  ihipStream_t* s = null_stream(tl_device);
  const hipError_t e = hipMemsetAsync(dst, value, bytes, s);
  if (e != hipSuccess) return e;
  stream_sync(s);
  return hipSuccess;
}

// ---- IPC ------------------------------------------------------------------------------------------------------------------------------
hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t* handle, void* ptr) {
  if (!handle || !ptr) return fail(hipErrorInvalidValue);
  std::lock_guard<std::mutex> g(g_mem_mu);
  Alloc* a = find_alloc_locked(ptr);
  if (!a || a->kind != KIND_DEVICE) return fail(hipErrorInvalidValue);
  IpcHandle h{kIpcMagic, (int32_t)getpid(), a->device, a->id, a->bytes, (uint64_t)((char*)ptr - a->base), a->flags};
  memset(handle, 0, sizeof *handle);
  memcpy(handle, &h, sizeof h);
  return hipSuccess;
}
hipError_t hipIpcOpenMemHandle(void** ptr, hipIpcMemHandle_t handle, unsigned flags) {
  startup();
  if (!ptr) return fail(hipErrorInvalidValue);
  IpcHandle h;
  memcpy(&h, &handle, sizeof h);
  if (h.magic != kIpcMagic) return fail(hipErrorInvalidHandle);
  if (h.pid == (int32_t)getpid()) return fail(hipErrorInvalidContext);  // HIP does not open a handle in the process that made it
  {  // DEVSIM_FAIL_IPC_OPEN=<n>[@<device>]: the n-th open (of the process on that device) fails -- a node whose driver refuses the mapping
    static std::atomic<long> calls{0};
    static const char* spec = getenv("DEVSIM_FAIL_IPC_OPEN");
    if (spec) {
      long n = 0;
      int dev = -1;
      if (sscanf(spec, "%ld@%d", &n, &dev) >= 1 && (dev < 0 || dev == tl_device) && calls.fetch_add(1) + 1 == n) return fail(hipErrorInvalidValue);
    }
    // DEVSIM_FAIL_IPC_KIND=uncached|default[@<device>]: EVERY open of a handle of that kind of memory fails (on that device) -- a
    // runtime that will not map another device's uncached allocation (the flag pages), or no plain one (the windows, the arenas)
    static const char* kind = getenv("DEVSIM_FAIL_IPC_KIND");
    if (kind) {
      int dev = -1;
      const char* at = strchr(kind, '@');
      if (at) dev = atoi(at + 1);
      const bool uncached = (h.flags & (hipDeviceMallocUncached | hipDeviceMallocFinegrained)) != 0;
      const bool want_uncached = strncmp(kind, "uncached", 8) == 0;
      if ((dev < 0 || dev == tl_device) && uncached == want_uncached) return fail(hipErrorInvalidValue);
    }
  }
  if (h.device != tl_device && !(flags & hipIpcMemLazyEnablePeerAccess)) {
    std::lock_guard<std::mutex> g(g_peer_mu);
    if (!g_peer[tl_device][h.device]) return fail(hipErrorPeerAccessNotEnabled);
  }
  {
    std::lock_guard<std::mutex> g(g_mem_mu);
    for (auto& kv : allocs()) {
      Alloc& a = kv.second;
      if (a.kind == KIND_IPC && a.owner_pid == h.pid && a.id == h.id) {
        a.refs++;
        *ptr = a.base + h.offset;
        return hipSuccess;
      }
    }
  }
  const int fd = shm_open(shm_name(h.pid, h.id).c_str(), O_RDWR, 0600);
  if (fd < 0) return fail(hipErrorInvalidHandle);
  // DEVSIM_PRIVATE_UNCACHED=1[@<device>]: an uncached allocation of another process opens -- and what this device stores there never
  // reaches its owner (a private copy): the mapping that "works" and carries nothing, which only trying it can find
  int map_kind = MAP_SHARED;
  if (h.flags & (hipDeviceMallocUncached | hipDeviceMallocFinegrained)) {
    static const char* spec = getenv("DEVSIM_PRIVATE_UNCACHED");
    if (spec) {
      const char* at = strchr(spec, '@');
      if (!at || atoi(at + 1) == tl_device) map_kind = MAP_PRIVATE;
    }
  }
  void* p = mmap(nullptr, h.bytes, PROT_READ | PROT_WRITE, map_kind | MAP_NORESERVE, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return fail(hipErrorOutOfMemory);
  std::lock_guard<std::mutex> g(g_mem_mu);
  allocs()[(uintptr_t)p] = Alloc{(char*)p, (size_t)h.bytes, h.device, KIND_IPC, h.flags, h.id, h.pid, 1};
  g_alloc_version.fetch_add(1);
  *ptr = (char*)p + h.offset;
  return hipSuccess;
}
hipError_t hipIpcCloseMemHandle(void* ptr) {
  std::lock_guard<std::mutex> g(g_mem_mu);
  Alloc* a = find_alloc_locked(ptr);
  if (!a || a->kind != KIND_IPC) return fail(hipErrorInvalidValue);
  if (--a->refs > 0) return hipSuccess;
  munmap(a->base, a->bytes);
  allocs().erase((uintptr_t)a->base);
  g_alloc_version.fetch_add(1);
  return hipSuccess;
}

// ---- streams ------------------------------------------------------------------------------------------------------------------------
hipError_t hipStreamCreateWithFlags(hipStream_t* stream, unsigned) {
  if (!stream) return fail(hipErrorInvalidValue);
  *stream = new_stream(tl_device);
  return hipSuccess;
}
hipError_t hipStreamCreate(hipStream_t* stream) { return hipStreamCreateWithFlags(stream, 0); }
hipError_t hipStreamDestroy(hipStream_t stream) {
  if (!stream) return fail(hipErrorInvalidHandle);
  stream_sync(stream);  // (the worker stays: a destroyed stream is merely never used again)
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t stream) {
  ihipStream_t* s = resolve(stream);
  if (s->capturing) return fail(hipErrorStreamCaptureUnsupported);
  stream_sync(s);
  return hipSuccess;
}
hipError_t hipStreamQuery(hipStream_t stream) {
  ihipStream_t* s = resolve(stream);
  std::lock_guard<std::mutex> g(s->mu);
  return s->completed >= s->enqueued ? hipSuccess : hipErrorNotReady;  // (not an error that sticks)
}
hipError_t hipStreamWaitEvent(hipStream_t stream, hipEvent_t event, unsigned) {
  if (!event) return fail(hipErrorInvalidHandle);
  Op op;
  op.kind = Op::WAIT;
  op.ev = event;
  {
    std::lock_guard<std::mutex> g(event->mu);
    op.gen = event->recorded;
  }
  enqueue(resolve(stream), std::move(op));
  return hipSuccess;
}
hipError_t hipStreamBeginCapture(hipStream_t stream, hipStreamCaptureMode) {
  ihipStream_t* s = resolve(stream);
  std::lock_guard<std::mutex> g(s->mu);
  if (s->capturing) return fail(hipErrorStreamCaptureUnsupported);
  s->capturing = true;
  s->capture = new ihipGraph;
  return hipSuccess;
}
hipError_t hipStreamEndCapture(hipStream_t stream, hipGraph_t* graph) {
  ihipStream_t* s = resolve(stream);
  std::lock_guard<std::mutex> g(s->mu);
  if (!s->capturing) return fail(hipErrorStreamCaptureInvalidated);
  s->capturing = false;
  if (graph) *graph = s->capture;
  else delete s->capture;
  s->capture = nullptr;
  return hipSuccess;
}
hipError_t hipStreamIsCapturing(hipStream_t stream, hipStreamCaptureStatus* status) {
  if (!status) return fail(hipErrorInvalidValue);
  ihipStream_t* s = resolve(stream);
  std::lock_guard<std::mutex> g(s->mu);
  *status = s->capturing ? hipStreamCaptureStatusActive : hipStreamCaptureStatusNone;
  return hipSuccess;
}

// ---- events ---------------------------------------------------------------------------------------------------------------------------
hipError_t hipEventCreateWithFlags(hipEvent_t* event, unsigned flags) {
  if (!event) return fail(hipErrorInvalidValue);
  *event = new ihipEvent_t;
  (*event)->flags = flags;
  return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t* event) { return hipEventCreateWithFlags(event, 0); }
hipError_t hipEventDestroy(hipEvent_t event) {
  if (!event) return fail(hipErrorInvalidHandle);
  // (operations already enqueued may still refer to it: events are small, they are left to the process)
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t event, hipStream_t stream) {
  if (!event) return fail(hipErrorInvalidHandle);
  ihipStream_t* s = resolve(stream);
  Op op;
  op.kind = Op::RECORD;
  op.ev = event;
  bool capturing;
  {
    std::lock_guard<std::mutex> g(s->mu);
    capturing = s->capturing;
  }
  if (!capturing) {
    std::lock_guard<std::mutex> g(event->mu);
    op.gen = ++event->recorded;
  }
  enqueue(s, std::move(op));
  return hipSuccess;
}
hipError_t hipEventQuery(hipEvent_t event) {
  if (!event) return fail(hipErrorInvalidHandle);
  std::lock_guard<std::mutex> g(event->mu);
  return event->done >= event->recorded ? hipSuccess : hipErrorNotReady;
}
hipError_t hipEventSynchronize(hipEvent_t event) {
  if (!event) return fail(hipErrorInvalidHandle);
  std::unique_lock<std::mutex> g(event->mu);
  const uint64_t want = event->recorded;
  event->cv.wait(g, [&] { return event->done >= want; });
  return hipSuccess;
}
hipError_t hipEventElapsedTime(float* ms, hipEvent_t start, hipEvent_t stop) {
  if (!ms || !start || !stop) return fail(hipErrorInvalidHandle);
  double a, b;
  {
    std::lock_guard<std::mutex> g(start->mu);
    if (start->done == 0) return fail(hipErrorInvalidHandle);
    if (start->done < start->recorded) return hipErrorNotReady;
    a = start->t_done;
  }
  {
    std::lock_guard<std::mutex> g(stop->mu);
    if (stop->done == 0) return fail(hipErrorInvalidHandle);
    if (stop->done < stop->recorded) return hipErrorNotReady;
    b = stop->t_done;
  }
  *ms = (float)((b - a) * 1e3);
  return hipSuccess;
}

// ---- graphs ---------------------------------------------------------------------------------------------------------------------------
hipError_t hipGraphInstantiate(hipGraphExec_t* exec, hipGraph_t graph, hipGraphNode_t*, char*, size_t) {
  if (!exec || !graph) return fail(hipErrorInvalidValue);
  *exec = new hipGraphExec;
  (*exec)->ops = graph->ops;
  return hipSuccess;
}
hipError_t hipGraphLaunch(hipGraphExec_t exec, hipStream_t stream) {
  if (!exec) return fail(hipErrorInvalidHandle);
  ihipStream_t* s = resolve(stream);
  for (const Op& op : exec->ops) {
    Op copy = op;
    enqueue(s, std::move(copy));
  }
  return hipSuccess;
}
hipError_t hipGraphDestroy(hipGraph_t graph) {
  delete graph;
  return hipSuccess;
}
hipError_t hipGraphExecDestroy(hipGraphExec_t exec) {
  if (!exec) return fail(hipErrorInvalidHandle);
  // (launches of it may still be queued and hold copies of its operations)
  delete exec;
  return hipSuccess;
}

}  // extern "C"


// ---- traffic accounting: the hooks the traced kernel sources call, and what a test reads ------------------------------------------------
namespace devsim {
namespace {
inline void traced(const void* p, size_t bytes, int store) {
  if (tl_runner && traffic_on()) traffic_count(p, bytes, store);  // (only the lanes of a kernel: the launchers share the file)
}
}  // namespace
void flag_touch(const void* p, unsigned bytes, int store) {
  if (tl_runner && traffic_on()) traffic_count(p, bytes, store);
}
}  // namespace devsim
extern "C" {
void __sanitizer_cov_load1(uint8_t* p) { devsim::traced(p, 1, 0); }
void __sanitizer_cov_load2(uint16_t* p) { devsim::traced(p, 2, 0); }
void __sanitizer_cov_load4(uint32_t* p) { devsim::traced(p, 4, 0); }
void __sanitizer_cov_load8(uint64_t* p) { devsim::traced(p, 8, 0); }
void __sanitizer_cov_load16(__uint128_t* p) { devsim::traced(p, 16, 0); }
void __sanitizer_cov_store1(uint8_t* p) { devsim::traced(p, 1, 1); }
void __sanitizer_cov_store2(uint16_t* p) { devsim::traced(p, 2, 1); }
void __sanitizer_cov_store4(uint32_t* p) { devsim::traced(p, 4, 1); }
void __sanitizer_cov_store8(uint64_t* p) { devsim::traced(p, 8, 1); }
void __sanitizer_cov_store16(__uint128_t* p) { devsim::traced(p, 16, 1); }
void __sanitizer_cov_trace_pc_guard(uint32_t*) {}
void __sanitizer_cov_trace_pc_guard_init(uint32_t*, uint32_t*) {}

// out[3][17][2]: bytes the kernels (and copies) of `device` in THIS process loaded from / stored to each owner (0..15 devices,
// 16 host) -- [0] payload, [1] the library's flag pages (flag words, boxes, LL lines; a polling lane counts every look), [2] small
// blocks (tables)
void devsim_traffic_read(int device, uint64_t* out) {
  for (int f = 0; f < 3; f++)
    for (int o = 0; o <= devsim::kOwnerHost; o++)
      for (int k = 0; k < 2; k++)
        out[(f * (devsim::kOwnerHost + 1) + o) * 2 + k] = devsim::g_traffic[f][device & 15][o][k].load(std::memory_order_relaxed);
}
void devsim_traffic_reset(void) {
  for (auto& f : devsim::g_traffic)
    for (auto& d : f)
      for (auto& o : d)
        for (auto& k : o) k.store(0, std::memory_order_relaxed);
}
int devsim_traffic_enabled(void) { return devsim::traffic_on() ? 1 : 0; }
}
