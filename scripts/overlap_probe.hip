// overlap_probe -- what a device-synchronised collective costs the caller's OTHER work.
// Every block of the one-kernel form spins until the peers arrive and holds its wave slots meanwhile; the meet / body /
// done form only ever keeps two small kernels waiting.  This program measures it: a compute kernel of fixed work (an FMA
// chain per lane, a grid that fills the chip) on stream A, alone and then concurrently with a float32 allreduce on
// stream B (xmpi_allreduce_on_stream), for the one-kernel form with its grid capped at 1024 / 256 / 64 blocks and for the
// split form.  Run under the launcher with 2 processes:  xmpirun 2 overlap_probe_bin [bytes] [reps]
// Rank 0 prints one JSON line: compute time alone, compute time next to each form, the allreduce's time next to compute.
#include <hip/hip_runtime.h>

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
  const int iters = 400000;                 // ~ a millisecond of FMAs per lane
  const dim3 grid(256 * 4), block(256);     // 4 blocks of 4 waves per CU: half of the chip's wave slots
  auto compute_alone = [&](float* ms) -> int {
    CK(hipEventRecord(a0, sa));
    hipLaunchKernelGGL(burn, grid, block, 0, sa, sink, iters);
    CK(hipEventRecord(a1, sa));
    CK(hipStreamSynchronize(sa));
    CK(hipEventElapsedTime(ms, a0, a1));
    return 0;
  };
  float alone = 0, t = 0;
  for (int i = 0; i < 3; i++)
    if (compute_alone(&t)) return 1;
  for (int i = 0; i < reps; i++) {
    if (compute_alone(&t)) return 1;
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
    float coll_alone = 0, coll_with = 0, comp_with = 0;
    for (int i = 0; i < reps; i++) {
      XK(xmpi_barrier(c));
      CK(hipEventRecord(b0, sb));
      XK(xmpi_allreduce_on_stream(c, send, recv, bytes / 4, XMPI_F32, XMPI_SUM, sb));
      CK(hipEventRecord(b1, sb));
      XK(xmpi_stream_sync(c, sb));
      CK(hipEventElapsedTime(&t, b0, b1));
      coll_alone += t / reps;
      XK(xmpi_barrier(c));
      // the collective first (its blocks take their places), the caller's kernel right behind it on the other stream
      CK(hipEventRecord(b0, sb));
      XK(xmpi_allreduce_on_stream(c, send, recv, bytes / 4, XMPI_F32, XMPI_SUM, sb));
      CK(hipEventRecord(b1, sb));
      CK(hipEventRecord(a0, sa));
      hipLaunchKernelGGL(burn, grid, block, 0, sa, sink, iters);
      CK(hipEventRecord(a1, sa));
      CK(hipStreamSynchronize(sa));
      XK(xmpi_stream_sync(c, sb));
      CK(hipEventElapsedTime(&t, a0, a1));
      comp_with += t / reps;
      CK(hipEventElapsedTime(&t, b0, b1));
      coll_with += t / reps;
    }
    char row[320];
    snprintf(row, sizeof row, "%s{\"form\": \"%s\", \"allreduce_alone_ms\": %.3f, \"allreduce_next_to_compute_ms\": %.3f, "
             "\"compute_next_to_allreduce_ms\": %.3f, \"compute_slowdown\": %.3f}", rows.empty() ? "" : ", ", f.name, coll_alone, coll_with,
             comp_with, comp_with / alone);
    rows += row;
  }
  XK(xmpi_barrier(c));
  if (rank == 0)
    printf("{\"ranks\": %d, \"bytes_per_rank\": %zu, \"compute_alone_ms\": %.3f, \"compute_grid_blocks\": %d, \"rows\": [%s]}\n", size, bytes,
           alone, (int)grid.x, rows.c_str());
  xmpi_free(c, send);
  xmpi_free(c, recv);
  XK(xmpi_finalize(c));
  return 0;
}
