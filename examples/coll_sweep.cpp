// coll_sweep -- allreduce latency / bandwidth over message sizes, one OS process per rank (the production
// layout), through the C++ mirror of package mpi: the reference's only benchmark is the bounce timing loop
// (examples/bounce/bounce.go:83-151); this is its counterpart for the collectives the reference stubs
// (mpi.go:130).  Two figures per size: a blocking Allreduce per iteration (enqueue + wait), and the same
// collectives enqueued back to back on a stream with ONE wait at the end (what a caller that overlaps pays).
// Every size is checked against its closed form.  Rank 0 prints one JSON line.
//   xmpirun N coll_sweep [max_bytes] [iters]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "mpi.hpp"

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static int fail(const char* what, const mpi::Error& err) {
  fprintf(stderr, "%s: %s\n", what, err.What().c_str());
  return 1;
}

int main(int argc, char** argv) {
  mpi::ParseFlags(&argc, argv);
  if (mpi::Error err = mpi::Init()) return fail("init", err);
  const int rank = mpi::Rank(), size = mpi::Size();
  const size_t max_bytes = argc > 1 ? (size_t)atoll(argv[1]) : (size_t)16 << 20;
  const int iters = argc > 2 ? atoi(argv[2]) : 200;
  const double think_us = getenv("COLL_SWEEP_THINK_US") ? atof(getenv("COLL_SWEEP_THINK_US")) : 0.0;
  const size_t factor = argc > 3 && atoi(argv[3]) >= 2 ? (size_t)atoi(argv[3]) : 4;  // sizes 1 KiB, x factor, ...
  mpi::XGMI* gpu = mpi::DefaultBackend();
  const size_t nmax = max_bytes / 4;
  float* send = (float*)gpu->Malloc(max_bytes);
  float* recv = (float*)gpu->Malloc(max_bytes);
  if (!send || !recv) return fail("malloc", mpi::Error(XMPI_ERR_NOMEM, "out of HBM"));
  std::vector<float> x(nmax), got(4096);
  for (size_t i = 0; i < nmax; i++) x[i] = (float)(rank + 1) + (float)(i % 7);  // partial sums are small integers: exact
  if (mpi::Error err = gpu->Memcpy(send, x.data(), max_bytes)) return fail("upload", err);
  void* st = gpu->Stream();
  const bool on_device = xmpi_get_param(gpu->Handle(), "dsync") == 1;
  std::string rows;
  int bad = 0;
  for (size_t bytes = 1024; bytes <= max_bytes; bytes *= factor) {
    const size_t n = bytes / 4;
    const int k = bytes >= ((size_t)4 << 20) ? iters / 10 + 2 : iters;
    for (int w = 0; w < 5; w++)
      if (mpi::Error err = mpi::Allreduce(mpi::Span(send, n), mpi::Span(recv, n))) return fail("allreduce", err);
    mpi::Barrier();
    const long ag0 = xmpi_get_param(gpu->Handle(), "dsync_ll_agent"), wt0 = xmpi_get_param(gpu->Handle(), "agent_ll_wait_ns");
    double t0 = now_us(), thought = 0.0;
    for (int i = 0; i < k; i++) {
      if (mpi::Error err = mpi::Allreduce(mpi::Span(send, n), mpi::Span(recv, n))) return fail("allreduce", err);
      if (think_us > 0) {  // the caller's own work between two collectives (COLL_SWEEP_THINK_US): not part of the figure
        const double w0 = now_us();
        while (now_us() - w0 < think_us) {
        }
        thought += now_us() - w0;
      }
    }
    const double blocking_us = (now_us() - t0 - thought) / k;
    // of which: between the command to the lingering LL agent and its answer (when it ran them: ll.hip ll_agent_kernel)
    const long ag = xmpi_get_param(gpu->Handle(), "dsync_ll_agent") - ag0;
    const double agent_wait_us = ag > 0 ? (double)(xmpi_get_param(gpu->Handle(), "agent_ll_wait_ns") - wt0) / 1e3 / (double)ag : 0.0;
    {  // the blocking calls' result (whoever ran them: a launched kernel or a lingering agent), then garbage for the enqueued ones to overwrite
      const size_t m = n < got.size() ? n : got.size();
      if (mpi::Error err = gpu->Memcpy(got.data(), recv, m * 4)) return fail("download", err);
      for (size_t i = 0; i < m; i++)
        if (got[i] != (float)(size * (size + 1) / 2) + (float)size * (float)(i % 7)) bad++;
      std::vector<float> junk(m, -1.0f);
      if (mpi::Error err = gpu->Memcpy(recv, junk.data(), m * 4)) return fail("upload", err);
    }
    // ... and the same call on HOST slices -- what a program written against the reference passes (helloworld.go:58, bounce.go:96:
    // Go slices): in through pinned memory the kernel reads itself, out the same way (PCIe-inclusive; never the bench's `value`)
    double host_us = 0.0;
    if (bytes <= ((size_t)64 << 10)) {
      std::vector<float> hx(x.begin(), x.begin() + (long)n), hy(n);
      for (int w = 0; w < 3; w++)
        if (mpi::Error err = mpi::Allreduce(mpi::Slice(hx), mpi::Into(&hy))) return fail("allreduce(host slices)", err);
      mpi::Barrier();
      t0 = now_us();
      for (int i = 0; i < k; i++)
        if (mpi::Error err = mpi::Allreduce(mpi::Slice(hx), mpi::Into(&hy))) return fail("allreduce(host slices)", err);
      host_us = (now_us() - t0) / k;
      for (size_t i = 0; i < n; i++)
        if (hy[i] != (float)(size * (size + 1) / 2) + (float)size * (float)(i % 7)) bad++;
    }
    mpi::Barrier();
    t0 = now_us();
    for (int i = 0; i < k; i++)
      if (mpi::Error err = gpu->AllreduceOnStream(mpi::Span(send, n), mpi::Span(recv, n), XMPI_SUM, st)) return fail("enqueue", err);
    if (mpi::Error err = gpu->StreamSync(st)) return fail("stream sync", err);
    const double queued_us = (now_us() - t0) / k;
    const size_t m = n < got.size() ? n : got.size();
    if (mpi::Error err = gpu->Memcpy(got.data(), recv, m * 4)) return fail("download", err);
    for (size_t i = 0; i < m; i++)
      if (got[i] != (float)(size * (size + 1) / 2) + (float)size * (float)(i % 7)) bad++;
    // max over ranks of both figures
    std::vector<double> mine = {blocking_us, queued_us, host_us}, worst(3);
    if (mpi::Error err = mpi::Allreduce(mpi::Slice(mine), mpi::Into(&worst), XMPI_MAX)) return fail("allreduce(max)", err);
    char row[384];
    snprintf(row, sizeof row, "%s{\"bytes\": %zu, \"blocking_us\": %.2f, \"queued_us\": %.2f, \"queued_busbw_GBps\": %.3f, \"by_agent\": %ld, \"agent_wait_us\": %.2f, \"host_slices_blocking_us\": %.2f}",
             rows.empty() ? "" : ", ", bytes, worst[0], worst[1], (double)bytes / worst[1] / 1e3 * 2.0 * (size - 1) / size, ag, agent_wait_us, worst[2]);
    rows += row;
  }
  mpi::Barrier();
  if (rank == 0)
    printf("{\"ranks\": %d, \"one_process_per_rank\": true, \"meet\": \"%s\", \"exact\": %s, \"ll_collectives\": %ld, \"run_by_the_ll_agent\": %ld, "
           "\"ll_agent_launches\": %ld, \"rows\": [%s]}\n", size,
           on_device ? "on the device (flag words in HBM)" : "on the host (control block)", bad ? "false" : "true",
           xmpi_get_param(gpu->Handle(), "dsync_ll_launches"), xmpi_get_param(gpu->Handle(), "dsync_ll_agent"),
           xmpi_get_param(gpu->Handle(), "ll_agent_launches"), rows.c_str());
  gpu->StreamDestroy(st);
  gpu->Free(send);
  gpu->Free(recv);
  mpi::Finalize();
  return bad ? 1 : 0;
}
