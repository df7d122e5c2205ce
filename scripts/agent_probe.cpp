// agent_probe -- the receive agent's kernel (sched.hip p2p_agent_kernel) driven by hand in ONE process: commands written
// into the pinned record, outcomes read back, with timeouts everywhere (debugging aid).
#include <hip/hip_runtime_api.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../mpi_amd/csrc/kernels.h"

using namespace xmpi;
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int grid = argc > 1 ? atoi(argv[1]) : 8;
  const long patience_us = argc > 2 ? atol(argv[2]) : 40;
  uint64_t *cmd = nullptr, *cmd_dev = nullptr, *rec = nullptr;
  uint32_t *mail = nullptr, *mail_dev = nullptr;
  CK(hipHostMalloc((void**)&cmd, 64, hipHostMallocMapped));
  CK(hipHostMalloc((void**)&mail, 4096, hipHostMallocMapped));
  memset(cmd, 0, 64);
  memset(mail, 0, 4096);
  CK(hipHostGetDevicePointer((void**)&cmd_dev, cmd, 0));
  CK(hipHostGetDevicePointer((void**)&mail_dev, mail, 0));
  CK(hipMalloc((void**)&rec, 64));
  CK(hipMemset(rec, 0, 64));
  const size_t n = 1 << 20;
  char *src = nullptr, *dst = nullptr;
  CK(hipMalloc((void**)&src, n));
  CK(hipMalloc((void**)&dst, n));
  std::vector<char> h(n), back(n);
  for (size_t i = 0; i < n; i++) h[i] = (char)(i * 7 + 3);
  CK(hipMemcpy(src, h.data(), n, hipMemcpyHostToDevice));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  volatile uint64_t* c = cmd;
  uint64_t seq = 0;
  bool running = false;
  int launches = 0;
  const bool slow = argc > 3;  // slow: every round lets the agent's patience run out
  for (int round = 0; round < 40; round++) {
    const size_t bytes = round % 3 == 0 ? 8 : (round % 3 == 1 ? 65536 : n);
    if (slow || round % 10 == 9) {
      CK(hipMemset(dst, 0, n));
      CK(hipDeviceSynchronize());
    }
    if (round == 20) usleep(1000);  // let the agent's patience run out once
    const double t0 = now_us();
    ++seq;
    c[1] = (uint64_t)src;
    c[2] = (uint64_t)dst;
    c[3] = (uint64_t)(64 * (round % 8)) | (seq << 32);
    __atomic_store_n((uint64_t*)&c[0], 1ull | ((uint64_t)bytes << 2) | (seq << 24), __ATOMIC_RELEASE);
    auto launch = [&]() {
      __atomic_store_n((uint64_t*)&c[7], 0, __ATOMIC_RELEASE);
      P2PAgentArgs a;
      memset(&a, 0, sizeof a);
      a.cmd = cmd_dev;
      a.rec = rec;
      a.ctl_dev = (uint64_t)(uintptr_t)mail_dev;
      a.seq0 = seq;
      a.launch = (uint64_t)launches + 1;
      a.alone_bytes = 128 << 10;
      a.patience_ticks = (uint64_t)patience_us * 100;
      a.mail_done_value = 4;
      hipError_t e = launch_p2p_agent(a, grid, st);
      if (e != hipSuccess) printf("launch: %s\n", hipGetErrorString(e));
      running = true;
      launches++;
    };
    if (!running) launch();
    bool ok = false;
    while (now_us() - t0 < 2e6) {
      if (__atomic_load_n((const uint64_t*)&c[6], __ATOMIC_ACQUIRE) == seq) { ok = true; break; }
      if (__atomic_load_n((const uint64_t*)&c[7], __ATOMIC_ACQUIRE) != 0) {
        if (__atomic_load_n((const uint64_t*)&c[6], __ATOMIC_ACQUIRE) == seq) { ok = true; break; }
        running = false;
        launch();
      }
    }
    const double dt = now_us() - t0;
    if (!ok) {
      printf("round %d seq %llu: TIMED OUT: cmd = %llx %llx %llx %llx served %llu gone %llu\n", round, (unsigned long long)seq,
             (unsigned long long)c[0], (unsigned long long)c[1], (unsigned long long)c[2], (unsigned long long)c[3],
             (unsigned long long)c[6], (unsigned long long)c[7]);
      fflush(stdout);
      _exit(2);
    }
    const bool check = slow || round % 10 == 9;  // (the checks take longer than the agent stays: mostly skip them)
    if (check) CK(hipMemcpy(back.data(), dst, n, hipMemcpyDeviceToHost));
    const bool same = !check || (memcmp(back.data(), h.data(), bytes) == 0 && (bytes == n || back[bytes] == 0));
    printf("round %2d bytes %7zu: %6.1f us  mail[%d] = %u  copy %s  launches %d\n", round, bytes, dt, 16 * (round % 8), mail[16 * (round % 8)],
           same ? "ok" : "WRONG", launches);
    mail[16 * (round % 8)] = 0;
  }
  // stop
  ++seq;
  c[3] = seq << 32;
  __atomic_store_n((uint64_t*)&c[0], 2ull | (seq << 24), __ATOMIC_RELEASE);
  const double t0 = now_us();
  while (__atomic_load_n((const uint64_t*)&c[7], __ATOMIC_ACQUIRE) == 0 && now_us() - t0 < 2e6) {}
  printf("stop: gone = %llu after %.1f us\n", (unsigned long long)c[7], now_us() - t0);
  CK(hipDeviceSynchronize());
  printf("done\n");
  return 0;
}
