// bar_probe -- where should a lingering kernel's command record live?  The agents (sched.hip p2p_agent_kernel, ll.hip
// ll_agent_kernel) poll 32 bytes of PINNED HOST memory: every poll is a read across PCIe.  With a large BAR the host can store
// straight into DEVICE memory (fine-grained / uncached), which the kernel polls locally.  This measures the round trip
// host store -> kernel sees it -> kernel stores an answer into pinned host memory -> host sees it, for the record in
// (a) pinned host memory, (b) fine-grained device memory, (c) uncached device memory -- and says whether (b) / (c) are
// writable by the host at all.
//   hipcc --offload-arch=gfx950 -O2 -o scripts/bar_probe_bin scripts/bar_probe.hip
#include <hip/hip_runtime.h>
#include <immintrin.h>

#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdint>
#include <cstdio>

static sigjmp_buf g_jmp;
static bool g_flush = false;
static void on_segv(int) { siglongjmp(g_jmp, 1); }
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void echo(const uint64_t* cmd, uint64_t* ans, uint64_t n) {
  for (uint64_t k = 1; k <= n; k++) {
    while (__hip_atomic_load(cmd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != k) __builtin_amdgcn_s_sleep(1);
    __hip_atomic_store(ans, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

static double round_trips(uint64_t* cmd_host_view, const uint64_t* cmd_dev_view, uint64_t* ans, uint64_t* ans_dev, int n) {
  __atomic_store_n(cmd_host_view, 0, __ATOMIC_RELEASE);
  __atomic_store_n(ans, 0, __ATOMIC_RELEASE);
  hipLaunchKernelGGL(echo, dim3(1), dim3(64), 0, 0, cmd_dev_view, ans_dev, (uint64_t)n);
  double t0 = 0;
  for (int k = 1; k <= n; k++) {
    if (k == n / 10 + 1) t0 = now_us();
    __atomic_store_n(cmd_host_view, (uint64_t)k, __ATOMIC_RELEASE);
    if (g_flush) _mm_sfence();  // (a store into the BAR sits in a write-combining buffer until something pushes it out)
    const double dl = now_us() + 2e6;
    while (__atomic_load_n(ans, __ATOMIC_ACQUIRE) != (uint64_t)k)
      if (now_us() > dl) return -1.0;
  }
  const double t = (now_us() - t0) / (n - n / 10);
  (void)hipDeviceSynchronize();
  return t;
}

int main() {
  int large_bar = -1;
  (void)hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, 0);
  uint64_t *host = nullptr, *host_dev = nullptr, *ans = nullptr, *ans_dev = nullptr;
  if (hipHostMalloc((void**)&host, 4096, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer((void**)&host_dev, host, 0) != hipSuccess) return 1;
  ans = host + 64;
  ans_dev = host_dev + 64;
  printf("{\"large_bar\": %d", large_bar);
  const int n = 2000;
  printf(", \"record_in_pinned_host_us\": %.2f", round_trips(host, host_dev, ans, ans_dev, n));
  const unsigned flags[2] = {hipDeviceMallocFinegrained, hipDeviceMallocUncached};
  const char* names[2] = {"finegrained", "uncached"};
  signal(SIGSEGV, on_segv);
  signal(SIGBUS, on_segv);
  for (int f = 0; f < 2; f++) {
    uint64_t* dev = nullptr;
    if (hipExtMallocWithFlags((void**)&dev, 4096, flags[f]) != hipSuccess) {
      (void)hipGetLastError();
      printf(", \"%s\": \"allocation refused\"", names[f]);
      continue;
    }
    (void)hipMemset(dev, 0, 4096);
    (void)hipDeviceSynchronize();
    bool writable = false;
    if (sigsetjmp(g_jmp, 1) == 0) {
      __atomic_store_n(dev + 8, 0x1234ull, __ATOMIC_RELEASE);
      writable = __atomic_load_n(dev + 8, __ATOMIC_ACQUIRE) == 0x1234ull;
    }
    if (!writable) {
      printf(", \"%s\": \"the host cannot store into it\"", names[f]);
      continue;
    }
    g_flush = false;
    printf(", \"record_in_%s_device_memory_us\": %.2f", names[f], round_trips(dev, dev, ans, ans_dev, 200));
    g_flush = true;
    printf(", \"record_in_%s_device_memory_sfence_us\": %.2f", names[f], round_trips(dev, dev, ans, ans_dev, n));
  }
  printf("}\n");
  return 0;
}
