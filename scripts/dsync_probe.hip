// dsync_probe.hip -- development probe (not product): can kernels of DIFFERENT processes synchronise through flag
// words in HBM?  N processes share the visible GPUs (process r uses device r % ndev), each allocates an uncached flag
// page + send/recv buffers, exports them with hipIpc, and runs a fused one-kernel allreduce whose ranks meet through
// the flag pages only (no host barrier between iterations).  Every spin has a wall-clock limit, so a GPU that
// time-slices the processes (or cannot run them together) shows up as a timeout count, not as a hang.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/dsync_probe.hip -o scripts/dsync_probe_bin
//   scripts/dsync_probe_bin <nproc> [grid_cap]
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "[%d] %s failed: %s\n", g_rank, #x, hipGetErrorString(e_));      \
      _exit(3);                                                                        \
    }                                                                                  \
  } while (0)

static int g_rank = -1;
constexpr int kMaxN = 8;
typedef unsigned int pack_t __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

struct Shared {
  std::atomic<int> bar_count, bar_gen;
  hipIpcMemHandle_t flag_h[kMaxN], send_h[kMaxN], recv_h[kMaxN];
  double us[kMaxN][16];
  int timeouts[kMaxN];
  int bad[kMaxN];
};

static void host_barrier(Shared* s, int n) {
  const int gen = s->bar_gen.load();
  if (s->bar_count.fetch_add(1) + 1 == n) {
    s->bar_count.store(0);
    s->bar_gen.store(gen + 1);
  } else {
    while (s->bar_gen.load() == gen) usleep(50);
  }
}

// flag page of one rank (uncached HBM): slot p is written by rank p only
struct FlagPage {
  unsigned long long ready[kMaxN * 8];  // 64 B apart
  unsigned long long done[kMaxN * 8];
  unsigned int ticket;
  unsigned int status;  // != 0: a spin timed out
};

struct Args {
  int me, n;
  unsigned long long epoch;
  FlagPage* flag[kMaxN];  // [me] local, others mapped
  const float* send[kMaxN];
  float* recv[kMaxN];
  size_t npack_total;  // 16-byte packets in the whole buffer
};

__device__ __forceinline__ bool spin_until(const unsigned long long* p, unsigned long long want, unsigned int* status) {
  const unsigned long long t0 = wall_clock64();
  if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;  // a spin timed out before: do not wait again
  while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
    __builtin_amdgcn_s_sleep(2);
    if (wall_clock64() - t0 > 300000000ull) {  // 3 s at 100 MHz
      atomicExch(status, 1u);
      return false;
    }
  }
  return true;
}

__global__ __launch_bounds__(256) void fused_allreduce(Args a) {
  const int tid = threadIdx.x, me = a.me, n = a.n;
  FlagPage* mine = a.flag[me];
  // 1. tell everybody my buffers are ready for this epoch
  if (blockIdx.x == 0 && tid < n && tid != me)
    __hip_atomic_store(&a.flag[tid]->ready[me * 8], a.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  // 2. wait until everybody else's are
  if (tid < n && tid != me) spin_until(&mine->ready[tid * 8], a.epoch, &mine->status);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  __syncthreads();
  // 3. fold my chunk of every send buffer in rank order, store it into every receive buffer
  const size_t per = a.npack_total / (size_t)n, lo = per * (size_t)me;
  for (size_t i = (size_t)blockIdx.x * 256 + tid; i < per; i += (size_t)gridDim.x * 256) {
    f4 acc = __builtin_nontemporal_load(reinterpret_cast<const f4*>(a.send[0]) + lo + i);
    for (int s = 1; s < n; s++) acc += __builtin_nontemporal_load(reinterpret_cast<const f4*>(a.send[s]) + lo + i);
    for (int d = 0; d < n; d++) __builtin_nontemporal_store(acc, reinterpret_cast<f4*>(a.recv[d]) + lo + i);
  }
  // 4. my stores are out: per-wave drain, block barrier, one system-scope release, ticket
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __shared__ unsigned int last;
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    last = (__hip_atomic_fetch_add(&mine->ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1u : 0u;
  }
  __syncthreads();
  if (!last) return;
  // 5. the last block: tell everybody I am done, wait until everybody is
  if (tid == 0) __hip_atomic_store(&mine->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid < n && tid != me) {
    __hip_atomic_store(&a.flag[tid]->done[me * 8], a.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    spin_until(&mine->done[tid * 8], a.epoch, &mine->status);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

__global__ void fill(float* p, size_t n, float v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = v + (float)(i & 1023) * 0.125f;
}

static int child(Shared* sh, int r, int n, int grid_cap) {
  g_rank = r;
  int ndev = 0;
  CK(hipGetDeviceCount(&ndev));
  CK(hipSetDevice(r % ndev));
  const size_t max_bytes = 64u << 20;
  FlagPage* fp = nullptr;
  hipError_t e = hipExtMallocWithFlags((void**)&fp, 65536, hipDeviceMallocUncached);
  if (e != hipSuccess) {
    fprintf(stderr, "[%d] uncached alloc failed (%s); trying fine-grained\n", r, hipGetErrorString(e));
    CK(hipExtMallocWithFlags((void**)&fp, 65536, hipDeviceMallocFinegrained));
  }
  CK(hipMemset(fp, 0, 65536));
  float *send = nullptr, *recv = nullptr;
  CK(hipMalloc(&send, max_bytes));
  CK(hipMalloc(&recv, max_bytes));
  CK(hipIpcGetMemHandle(&sh->flag_h[r], fp));
  CK(hipIpcGetMemHandle(&sh->send_h[r], send));
  CK(hipIpcGetMemHandle(&sh->recv_h[r], recv));
  hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, send, max_bytes / 4, (float)(r + 1));
  CK(hipDeviceSynchronize());
  host_barrier(sh, n);
  Args a;
  memset(&a, 0, sizeof a);
  a.me = r;
  a.n = n;
  for (int p = 0; p < n; p++) {
    if (p == r) {
      a.flag[p] = fp;
      a.send[p] = send;
      a.recv[p] = recv;
      continue;
    }
    void *f = nullptr, *s = nullptr, *d = nullptr;
    CK(hipIpcOpenMemHandle(&f, sh->flag_h[p], hipIpcMemLazyEnablePeerAccess));
    CK(hipIpcOpenMemHandle(&s, sh->send_h[p], hipIpcMemLazyEnablePeerAccess));
    CK(hipIpcOpenMemHandle(&d, sh->recv_h[p], hipIpcMemLazyEnablePeerAccess));
    a.flag[p] = (FlagPage*)f;
    a.send[p] = (const float*)s;
    a.recv[p] = (float*)d;
  }
  host_barrier(sh, n);
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  unsigned long long epoch = 0;
  const size_t sizes[] = {1024, 65536, 1u << 20, 16u << 20, 64u << 20};
  int col = 0;
  for (size_t bytes : sizes) {
    a.npack_total = bytes / 16;
    const size_t per = a.npack_total / (size_t)n;
    int grid = (int)((per + 255) / 256);
    if (grid < 1) grid = 1;
    if (grid > grid_cap) grid = grid_cap;
    // (a) one collective per host sync; (b) 50 collectives enqueued back to back, one sync
    for (int mode = 0; mode < 2; mode++) {
      const int iters = bytes <= (1u << 20) ? 200 : 20;
      for (int w = 0; w < 5; w++) {
        a.epoch = ++epoch;
        hipLaunchKernelGGL(fused_allreduce, dim3(grid), dim3(256), 0, st, a);
      }
      CK(hipStreamSynchronize(st));
      host_barrier(sh, n);
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < iters; i++) {
        a.epoch = ++epoch;
        hipLaunchKernelGGL(fused_allreduce, dim3(grid), dim3(256), 0, st, a);
        if (mode == 0) CK(hipStreamSynchronize(st));
      }
      CK(hipStreamSynchronize(st));
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
      sh->us[r][col++] = us;
      host_barrier(sh, n);
    }
    // check: element i of the result = sum over ranks of (p + 1 + (i & 1023) / 8)
    std::vector<float> h(std::min<size_t>(bytes / 4, 4096));
    CK(hipMemcpy(h.data(), recv, h.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < h.size(); i++) {
      float want = 0.f;
      for (int p = 0; p < n; p++) want += (float)(p + 1) + (float)(i & 1023) * 0.125f;
      if (h[i] != want) sh->bad[r]++;
    }
    host_barrier(sh, n);
  }
  unsigned int status = 0;
  CK(hipMemcpy(&status, &fp->status, 4, hipMemcpyDeviceToHost));
  sh->timeouts[r] = (int)status;
  host_barrier(sh, n);
  return 0;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 2;
  const int grid_cap = argc > 2 ? atoi(argv[2]) : 128;
  if (n < 2 || n > kMaxN) return 2;
  Shared* sh = (Shared*)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  memset((void*)sh, 0, sizeof *sh);
  std::vector<pid_t> pids;
  for (int r = 0; r < n; r++) {
    pid_t p = fork();
    if (p == 0) _exit(child(sh, r, n, grid_cap));
    pids.push_back(p);
  }
  int fail = 0;
  for (pid_t p : pids) {
    int st = 0;
    waitpid(p, &st, 0);
    if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) fail++;
  }
  const char* names[] = {"1KiB", "64KiB", "1MiB", "16MiB", "64MiB"};
  printf("{\"nproc\": %d, \"grid_cap\": %d, \"failed_procs\": %d", n, grid_cap, fail);
  for (int c = 0; c < 10; c++) {
    double mx = 0;
    for (int r = 0; r < n; r++) mx = sh->us[r][c] > mx ? sh->us[r][c] : mx;
    printf(", \"%s_%s_us\": %.1f", names[c / 2], c % 2 ? "queued" : "synced", mx);
  }
  int to = 0, bad = 0;
  for (int r = 0; r < n; r++) to += sh->timeouts[r], bad += sh->bad[r];
  printf(", \"spin_timeouts\": %d, \"bad_elements\": %d}\n", to, bad);
  return fail ? 1 : 0;
}
