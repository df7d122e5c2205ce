// zc_latency.cpp -- per-call latency of the collectives through the C ABI, without Python in the loop
// (development tool).  R ranks as threads of this process on device 0.
//   g++ -O2 -std=c++17 -pthread scripts/zc_latency.cpp -Iinclude -Lmpi_amd -lxmpi -Wl,-rpath,$PWD/mpi_amd -o scripts/zc_latency_bin
//   scripts/zc_latency_bin [ranks=8] [iters=200]
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "xmpi.h"

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
  const int R = argc > 1 ? atoi(argv[1]) : 8;
  const int iters = argc > 2 ? atoi(argv[2]) : 200;
  const std::string key = "zclat-" + std::to_string((int)getpid());
  const size_t sizes[] = {1024, 1 << 20, 16 << 20, 256u << 20};
  const int algos[] = {XMPI_ALGO_ZCOPY, XMPI_ALGO_DIRECT, XMPI_ALGO_RING};
  const char* names[] = {"zcopy", "direct", "ring"};
  std::vector<std::thread> th;
  for (int r = 0; r < R; r++)
    th.emplace_back([&, r]() {
      xmpi_comm* c = nullptr;
      if (xmpi_init(r, R, 0, key.c_str(), &c) != XMPI_OK) {
        fprintf(stderr, "rank %d: init failed: %s\n", r, xmpi_last_error());
        exit(1);
      }
      const size_t maxb = 256u << 20;
      void* s = xmpi_malloc(c, maxb);
      void* d = xmpi_malloc(c, maxb);
      xmpi_fill_pattern(c, s, maxb / 4, XMPI_F32, 0, 1000 + r);
      for (size_t bytes : sizes)
        for (int a = 0; a < 3; a++) {
          const int n = bytes >= (16u << 20) ? iters / 10 + 2 : iters;
          for (int w = 0; w < 3; w++) xmpi_allreduce(c, s, d, bytes / 4, XMPI_F32, XMPI_SUM, algos[a]);
          xmpi_barrier(c);
          const double t0 = now();
          for (int i = 0; i < n; i++)
            if (xmpi_allreduce(c, s, d, bytes / 4, XMPI_F32, XMPI_SUM, algos[a]) != XMPI_OK) {
              fprintf(stderr, "rank %d: %s\n", r, xmpi_last_error());
              exit(1);
            }
          xmpi_barrier(c);
          const double t = (now() - t0) / n;
          if (r == 0)
            printf("%2d ranks %-6s %10zu B : %9.1f us  algbw %8.2f GB/s  aggregate %9.1f GB/s\n", R, names[a], bytes,
                   t * 1e6, bytes / t / 1e9, R * (double)bytes / t / 1e9);
        }
      xmpi_free(c, s);
      xmpi_free(c, d);
      xmpi_finalize(c);
    });
  for (auto& t : th) t.join();
  return 0;
}
