// overlap_probe -- what a device-synchronised collective costs the caller's OTHER work.
// Every block of the one-kernel form spins until the peers arrive and holds its wave slots meanwhile; the meet / body /
// done form only ever keeps a few one-wave blocks waiting.  What that costs shows when a peer is LATE: rank 0 enqueues a
// float32 allreduce on stream B and, right behind it on stream A, a train of compute kernels (an FMA chain per lane, a
// grid that fills every wave slot of the chip); rank 1 enqueues its allreduce `delay` milliseconds later.  The train's
// time next to the waiting collective, against the train alone, for the one-kernel form with its grid capped at
// 1024 / 256 / 64 blocks and for the split form.  Run under the launcher with 2 processes:
//   xmpirun 2 overlap_probe_bin [bytes] [reps] [delay_ms]
// Rank 0 prints one JSON line.
#include <hip/hip_runtime.h>

#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../include/xmpi.h"

__global__ void burn(float* out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  for (int i = 0; i < iters; i++) a = a * b + 0.5f;
  if (a == 12345.678f) out[0] = a;  // (never: keeps the loop)
}

#define CK(x)                                                            \
  do {                                                                   \
    hipError_t e_ = (x);                                                 \
    if (e_ != hipSuccess) {                                              \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));            \
      return 1;                                                          \
    }                                                                    \
  } while (0)
#define XK(x)                                                            \
  do {                                                                   \
    int r_ = (x);                                                        \
    if (r_ != XMPI_OK) {                                                 \
      fprintf(stderr, "%s: %s (%s)\n", #x, xmpi_strerror(r_), xmpi_last_error()); \
      return 1;                                                          \
    }                                                                    \
  } while (0)

int main(int argc, char** argv) {
  const int rank = atoi(getenv("XMPI_RANK") ? getenv("XMPI_RANK") : "0"), size = atoi(getenv("XMPI_SIZE") ? getenv("XMPI_SIZE") : "1");
  const char* key = getenv("XMPI_JOB") ? getenv("XMPI_JOB") : "overlap";
  const size_t bytes = argc > 1 ? (size_t)atoll(argv[1]) : (size_t)256 << 20;
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  xmpi_comm* c = nullptr;
  XK(xmpi_init(rank, size, getenv("XMPI_DEVICE") ? atoi(getenv("XMPI_DEVICE")) : -1, key, &c));
  float* send = (float*)xmpi_malloc(c, bytes);
  float* recv = (float*)xmpi_malloc(c, bytes);
  if (!send || !recv) return 1;
  XK(xmpi_memset(c, send, 0, bytes));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  float* sink = nullptr;
  CK(hipMalloc(&sink, 64));
  hipEvent_t a0, a1, b0, b1;
  CK(hipEventCreate(&a0));
  CK(hipEventCreate(&a1));
  CK(hipEventCreate(&b0));
  CK(hipEventCreate(&b1));
  const int delay_us = (argc > 3 ? atoi(argv[3]) : 3) * 1000;
  const int iters = 12000, train = 16;      // ~0.3 ms of FMAs per lane and launch
  const dim3 grid(256 * 8), block(256);     // 8 blocks of 4 waves per CU: every wave slot of the chip
  auto run_train = [&](float* ms) -> int {
    CK(hipEventRecord(a0, sa));
    for (int k = 0; k < train; k++) hipLaunchKernelGGL(burn, grid, block, 0, sa, sink, iters);
    CK(hipEventRecord(a1, sa));
    CK(hipStreamSynchronize(sa));
    CK(hipEventElapsedTime(ms, a0, a1));
    return 0;
  };
  float alone = 0, t = 0;
  for (int i = 0; i < 3; i++)
    if (run_train(&t)) return 1;
  for (int i = 0; i < reps; i++) {
    if (run_train(&t)) return 1;
    alone += t / reps;
  }
  struct Form {
    const char* name;
    long split, grid;
  } forms[] = {{"one kernel, grid 1024", 0, 1024}, {"one kernel, grid 256", 0, 256}, {"one kernel, grid 64", 0, 64}, {"meet / body / done", 1, 0}};
  std::string rows;
  for (const Form& f : forms) {
    xmpi_set_param(c, "dsync_split_bytes", f.split);
    xmpi_set_param(c, "dsync_grid", f.grid);
    XK(xmpi_allreduce_on_stream(c, send, recv, bytes / 4, XMPI_F32, XMPI_SUM, sb));  // warm
    XK(xmpi_stream_sync(c, sb));
    float coll_alone = 0, coll_with = 0, comp_with = 0, base = 0;
    for (int i = 0; i < reps; i++) {
      XK(xmpi_barrier(c));
      CK(hipEventRecord(b0, sb));
      XK(xmpi_allreduce_on_stream(c, send, recv, bytes / 4, XMPI_F32, XMPI_SUM, sb));
      CK(hipEventRecord(b1, sb));
      XK(xmpi_stream_sync(c, sb));
      CK(hipEventElapsedTime(&t, b0, b1));
      coll_alone += t / reps;
      XK(xmpi_barrier(c));
      if (rank == 0) {  // the baseline right before each measurement: clocks drift over a run
        if (run_train(&t)) return 1;
        base += t / reps;
      }
      XK(xmpi_barrier(c));
      if (rank == 0) {
        // the collective first (its blocks take their places and wait for the late peer), the caller's kernels behind it
        CK(hipEventRecord(b0, sb));
        XK(xmpi_allreduce_on_stream(c, send, recv, bytes / 4, XMPI_F32, XMPI_SUM, sb));
        CK(hipEventRecord(b1, sb));
        if (run_train(&t)) return 1;
        comp_with += t / reps;
        XK(xmpi_stream_sync(c, sb));
        CK(hipEventElapsedTime(&t, b0, b1));
        coll_with += t / reps;
      } else {
        usleep((useconds_t)delay_us);
        XK(xmpi_allreduce_on_stream(c, send, recv, bytes / 4, XMPI_F32, XMPI_SUM, sb));
        XK(xmpi_stream_sync(c, sb));
      }
    }
    char row[360];
    snprintf(row, sizeof row, "%s{\"form\": \"%s\", \"allreduce_alone_ms\": %.3f, \"allreduce_with_late_peer_ms\": %.3f, "
             "\"compute_train_alone_ms\": %.3f, \"compute_train_next_to_waiting_allreduce_ms\": %.3f, \"compute_slowdown\": %.3f}",
             rows.empty() ? "" : ", ", f.name, coll_alone, coll_with, base, comp_with, base > 0 ? comp_with / base : 0.f);
    rows += row;
  }
  XK(xmpi_barrier(c));
  if (rank == 0)
    printf("{\"ranks\": %d, \"bytes_per_rank\": %zu, \"peer_delay_ms\": %.1f, \"compute_train_alone_ms\": %.3f, \"compute_grid_blocks\": %d, "
           "\"launches_per_train\": %d, \"rows\": [%s]}\n", size, bytes, delay_us / 1e3, alone, (int)grid.x, train, rows.c_str());
  xmpi_free(c, send);
  xmpi_free(c, recv);
  XK(xmpi_finalize(c));
  return 0;
}
