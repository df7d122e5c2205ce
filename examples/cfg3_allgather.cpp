// cfg3_allgather -- BASELINE.json config 3: allgather of int64, 16 MiB per rank, 4 ranks, RING -- with ONE OS PROCESS PER
// RANK (xmpirun), where the ring is the stepped kernel of sched.hip (every hop inside one kernel per rank) -- beside the
// library's own choice.  Rank r contributes x[i] = (r << 40) | i (xmpi_fill_pattern, pattern 1): the result is checked
// bit for bit AND by position on every rank, every element (xmpi_count_mismatch against a locally generated expectation).
// The reference user's idiom for the same thing is the all-to-all of helloworld.go:53-81.
//
//   xmpirun 4 cfg3_allgather [elements per rank = 2097152] [iterations = 20]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "mpi.hpp"
#include "xmpi_test.h"  // xmpi_fill_pattern: deterministic inputs a checker can regenerate

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static int fail(const char* what, int rc) {
  fprintf(stderr, "%s: %s (%s)\n", what, xmpi_strerror(rc), xmpi_last_error());
  return 1;
}

int main(int argc, char** argv) {
  mpi::ParseFlags(&argc, argv);
  if (mpi::Error err = mpi::Init()) {
    fprintf(stderr, "init: %s\n", err.What().c_str());
    return 1;
  }
  const int rank = mpi::Rank(), size = mpi::Size();
  const size_t n = argc > 1 ? (size_t)atoll(argv[1]) : (size_t)2097152;
  const int iters = argc > 2 ? atoi(argv[2]) : 20;
  mpi::XGMI* gpu = mpi::DefaultBackend();
  xmpi_comm* c = gpu->Handle();
  int64_t* send = (int64_t*)gpu->Malloc(n * 8);
  int64_t* recv = (int64_t*)gpu->Malloc(n * 8 * (size_t)size);
  int64_t* want = (int64_t*)gpu->Malloc(n * 8 * (size_t)size);
  if (!send || !recv || !want) return fail("malloc", XMPI_ERR_NOMEM);
  int rc = xmpi_fill_pattern(c, send, n, XMPI_I64, 1, (uint64_t)rank);
  for (int r = 0; r < size && rc == XMPI_OK; r++) rc = xmpi_fill_pattern(c, want + (size_t)r * n, n, XMPI_I64, 1, (uint64_t)r);
  if (rc != XMPI_OK) return fail("fill", rc);
  rc = xmpi_tune(c, n * 8);
  if (rc != XMPI_OK) return fail("tune", rc);
  struct Sched {
    const char* name;
    int algo;
  } scheds[] = {{"ring", XMPI_ALGO_RING}, {"ring_push", XMPI_ALGO_RING_PUSH}, {"auto", XMPI_ALGO_AUTO}};
  std::string rows;
  uint64_t wrong = 0;
  for (const Sched& s : scheds) {
    rc = xmpi_memset(c, recv, 0, n * 8 * (size_t)size);
    if (rc == XMPI_OK) rc = xmpi_allgather(c, send, recv, n, XMPI_I64, s.algo);
    uint64_t diff = 0;
    if (rc == XMPI_OK) rc = xmpi_count_mismatch(c, recv, want, n * 8 * (size_t)size, &diff);
    if (rc != XMPI_OK) return fail(s.name, rc);
    wrong += diff;
    void* st = gpu->Stream();
    mpi::Barrier();
    xmpi_sync(c);
    double t0 = now_us();
    for (int i = 0; i < iters && rc == XMPI_OK; i++) rc = xmpi_allgather(c, send, recv, n, XMPI_I64, s.algo);
    const double blocking = (now_us() - t0) / iters;
    mpi::Barrier();
    t0 = now_us();
    if (s.algo == XMPI_ALGO_AUTO) {  // the stream-ordered form follows the library's table
      for (int i = 0; i < iters && rc == XMPI_OK; i++) rc = xmpi_allgather_on_stream(c, send, recv, n, XMPI_I64, st);
      if (rc == XMPI_OK) rc = xmpi_stream_sync(c, st);
    }
    const double queued = s.algo == XMPI_ALGO_AUTO ? (now_us() - t0) / iters : 0.0;
    gpu->StreamDestroy(st);
    if (rc != XMPI_OK) return fail(s.name, rc);
    std::vector<double> mine = {blocking, queued}, worst(2);
    (void)mpi::Allreduce(mpi::Slice(mine), mpi::Into(&worst), XMPI_MAX);
    const double total = (double)n * 8.0 * size;  // nccl-tests: S = the output's bytes
    char row[320];
    snprintf(row, sizeof row, "%s\"%s\": {\"blocking_us\": %.1f, \"queued_us\": %.1f, \"algbw_GBps\": %.2f, \"busbw_GBps\": %.2f, \"bit_exact_and_in_place\": %s}",
             rows.empty() ? "" : ", ", s.name, worst[0], worst[1], total / worst[0] / 1e3, total / worst[0] / 1e3 * (size - 1) / size,
             diff ? "false" : "true");
    rows += row;
  }
  mpi::Barrier();
  if (rank == 0)
    printf("{\"config\": \"BASELINE cfg 3: allgather int64\", \"ranks\": %d, \"bytes_per_rank\": %zu, \"one_process_per_rank\": true, \"meet\": \"%s\", "
           "\"iterations\": %d, \"exact\": %s, %s}\n",
           size, n * 8, xmpi_get_param(c, "dsync") == 1 ? "on the device: ring = the stepped kernel" : "on the host", iters,
           wrong ? "false" : "true", rows.c_str());
  gpu->Free(send);
  gpu->Free(recv);
  gpu->Free(want);
  mpi::Finalize();
  return wrong ? 1 : 0;
}
