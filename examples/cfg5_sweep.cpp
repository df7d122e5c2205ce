// cfg5_sweep -- BASELINE.json config 5: allreduce-sum of fp16, recursive halving + doubling against the ring, message
// sizes 1 MiB ... max (x4), with ONE OS PROCESS PER RANK (xmpirun) so that both are what the north-star layout runs: the
// stepped kernels of sched.hip -- every step of the schedule inside one kernel per rank -- beside the library's own
// choice (XMPI_ALGO_AUTO after xmpi_tune).  Inputs are k/64 with k < 64 (exactly summable in fp16 for up to 16 ranks),
// so every schedule must reproduce the rank-order result BIT FOR BIT; the program checks that on the device
// (xmpi_count_mismatch against the zero-copy fold, whose rank order the GPU test-suite pins to the CPU oracle).
//
//   xmpirun N cfg5_sweep [max bytes per rank = 1 GiB] [iterations = 5]
//
// Rank 0 prints one JSON line: per size, per schedule: microseconds (max over ranks, steps enqueued back to back),
// algbw, busbw, bit identity.  The reference has no counterpart (mpi.go:130); its only timing loop is bounce.go:83-151.
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
  const size_t max_bytes = argc > 1 ? (size_t)atoll(argv[1]) : (size_t)1 << 30;
  const int iters = argc > 2 ? atoi(argv[2]) : 5;
  mpi::XGMI* gpu = mpi::DefaultBackend();
  xmpi_comm* c = gpu->Handle();
  void* send = gpu->Malloc(max_bytes);
  void* recv = gpu->Malloc(max_bytes);
  void* ref = gpu->Malloc(max_bytes);
  if (!send || !recv || !ref) return fail("malloc", XMPI_ERR_NOMEM);
  // pattern 0 of xmpi_fill_pattern for fp16: (hash & 63) / 64 -- BASELINE cfg 5's inputs, seed 2000 + rank
  int rc = xmpi_fill_pattern(c, send, max_bytes / 2, XMPI_F16, 0, 2000 + (uint64_t)rank);
  if (rc != XMPI_OK) return fail("fill", rc);
  rc = xmpi_tune(c, max_bytes > ((size_t)256 << 20) ? (size_t)256 << 20 : max_bytes);
  if (rc != XMPI_OK) return fail("tune", rc);
  struct Sched {
    const char* name;
    int algo;
  };
  std::vector<Sched> scheds = {{"ring", XMPI_ALGO_RING}, {"ring_push", XMPI_ALGO_RING_PUSH}, {"auto", XMPI_ALGO_AUTO}};
  if ((size & (size - 1)) == 0) {
    scheds.insert(scheds.begin() + 2, {"rhd_push", XMPI_ALGO_RHD_PUSH});
    scheds.insert(scheds.begin() + 2, {"rhd", XMPI_ALGO_RHD});
  }
  std::string rows;
  int not_identical = 0;
  for (size_t bytes = (size_t)1 << 20; bytes <= max_bytes; bytes *= 4) {
    const size_t n = bytes / 2;
    xmpi_set_param(c, "tuned", 0);  // the reference result: the zero-copy fold in rank order
    rc = xmpi_allreduce(c, send, ref, n, XMPI_F16, XMPI_SUM, XMPI_ALGO_ZCOPY);
    xmpi_set_param(c, "tuned", 1);
    if (rc != XMPI_OK) return fail("reference allreduce", rc);
    char head[64];
    snprintf(head, sizeof head, "%s{\"bytes\": %zu", rows.empty() ? "" : ", ", bytes);
    rows += head;
    for (const Sched& s : scheds) {
      rc = xmpi_memset(c, recv, 0, bytes);
      if (rc == XMPI_OK) rc = xmpi_allreduce(c, send, recv, n, XMPI_F16, XMPI_SUM, s.algo);
      uint64_t diff = 0;
      if (rc == XMPI_OK) rc = xmpi_count_mismatch(c, recv, ref, bytes, &diff);
      if (rc != XMPI_OK) return fail(s.name, rc);
      if (diff) not_identical++;
      mpi::Barrier();
      xmpi_sync(c);
      const double t0 = now_us();
      rc = xmpi_allreduce_repeat(c, send, recv, n, XMPI_F16, XMPI_SUM, s.algo, iters);
      xmpi_sync(c);
      mpi::Barrier();
      if (rc != XMPI_OK) return fail(s.name, rc);
      std::vector<double> mine = {(now_us() - t0) / iters}, worst(1);
      (void)mpi::Allreduce(mpi::Slice(mine), mpi::Into(&worst), XMPI_MAX);
      char cell[256];
      snprintf(cell, sizeof cell, ", \"%s\": {\"us\": %.1f, \"algbw_GBps\": %.2f, \"busbw_GBps\": %.2f, \"bit_identical_to_rank_order\": %s}", s.name,
               worst[0], (double)bytes / worst[0] / 1e3, (double)bytes / worst[0] / 1e3 * 2.0 * (size - 1) / size, diff ? "false" : "true");
      rows += cell;
    }
    rows += "}";
  }
  mpi::Barrier();
  if (rank == 0)
    printf("{\"config\": \"BASELINE cfg 5: allreduce-sum fp16, recursive halving + doubling vs ring\", \"ranks\": %d, \"one_process_per_rank\": true, "
           "\"meet\": \"%s\", \"iterations\": %d, \"all_bit_identical\": %s, \"rows\": [%s]}\n",
           size, xmpi_get_param(c, "dsync") == 1 ? "on the device: ring / rhd are the stepped kernels" : "on the host: ring / rhd are host-driven step tables",
           iters, not_identical ? "false" : "true", rows.c_str());
  gpu->Free(send);
  gpu->Free(recv);
  gpu->Free(ref);
  mpi::Finalize();
  return not_identical ? 1 : 0;
}
