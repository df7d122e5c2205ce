// allreduce -- what a user of the reference would write once `//func AllReduce() {}` (mpi.go:130) is
// filled in: the package-level collectives of the C++ mirror of package mpi, on HBM buffers and on
// plain host slices, plus the non-blocking form.  Every result is checked against
// its closed form, so the exit status is the verdict.
//   xmpirun N allreduce [elements] [--tcp]
// --tcp registers the reference's own backend instead (mpi::Network: TCP + gob, host slices only): the same
// package-level calls, the same closed forms -- two backends, one answer.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mpi.hpp"
#include "network.hpp"

static int fail(const char* what, const mpi::Error& err) {
  fprintf(stderr, "%s: %s\n", what, err.What().c_str());
  return 1;
}

int main(int argc, char** argv) {
  mpi::ParseFlags(&argc, argv);
  static mpi::Network net;
  bool tcp = false;
  for (int i = 1; i < argc; i++)
    if (!strcmp(argv[i], "--tcp")) tcp = true;
  if (tcp) mpi::Register(&net);
  if (mpi::Error err = mpi::Init()) return fail("init", err);
  const int rank = mpi::Rank(), size = mpi::Size();
  const size_t n = (argc > 1 && argv[1][0] != '-') ? (size_t)atoll(argv[1]) : (size_t)1 << 20;
  mpi::XGMI* gpu = mpi::DefaultBackend();

  // x_r[i] = (r + 1) + (i mod 7): every partial sum is a small integer, exact in float32
  std::vector<float> x(n), got(n);
  for (size_t i = 0; i < n; i++) x[i] = (float)(rank + 1) + (float)(i % 7);
  float *send = nullptr, *recv = nullptr;
  if (!tcp) {
    send = (float*)gpu->Malloc(n * sizeof(float));
    recv = (float*)gpu->Malloc(n * sizeof(float));
    if (!send || !recv) return fail("malloc", mpi::Error(XMPI_ERR_NOMEM, "out of HBM"));
    if (mpi::Error err = gpu->Memcpy(send, x.data(), n * sizeof(float))) return fail("upload", err);
  }

  int bad = 0;
  auto check_sum = [&](const std::vector<float>& v, const char* what) {
    for (size_t i = 0; i < n; i++) {
      const float want = (float)(size * (size + 1) / 2) + (float)size * (float)(i % 7);
      if (v[i] != want) {
        if (bad++ < 3) fprintf(stderr, "node %d: %s[%zu] = %g, want %g\n", rank, what, i, v[i], want);
      }
    }
  };

  // 1. HBM buffers (zero-copy: the peers' buffers are folded in place, in rank order)
  double us = 0;
  if (!tcp) {
    const auto t0 = std::chrono::steady_clock::now();
    if (mpi::Error err = mpi::Allreduce(mpi::Span(send, n), mpi::Span(recv, n))) return fail("allreduce", err);
    us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (mpi::Error err = gpu->Memcpy(got.data(), recv, n * sizeof(float))) return fail("download", err);
    check_sum(got, "allreduce(HBM)");
  }

  // 2. plain host slices, as in the reference's examples (staged through HBM by the library)
  std::vector<float> host_out(n);
  const auto th = std::chrono::steady_clock::now();
  if (mpi::Error err = mpi::Allreduce(mpi::Slice(x), mpi::Into(&host_out))) return fail("allreduce(host)", err);
  if (tcp) us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - th).count();
  check_sum(host_out, "allreduce(host)");

  // 3. non-blocking: the caller is free to do its own work until WaitRequest
  if (!tcp) {
    xmpi_request* req = nullptr;
    if (mpi::Error err = gpu->IAllreduce(mpi::Span(send, n), mpi::Span(recv, n), XMPI_MAX, &req)) return fail("iallreduce", err);
    if (mpi::Error err = gpu->WaitRequest(req)) return fail("wait", err);
    if (mpi::Error err = gpu->Memcpy(got.data(), recv, n * sizeof(float))) return fail("download", err);
    for (size_t i = 0; i < n; i++)
      if (got[i] != (float)size + (float)(i % 7) && bad++ < 3) fprintf(stderr, "node %d: max[%zu] = %g\n", rank, i, got[i]);
  } else {  // the same MAX through the TCP backend's fold
    std::vector<float> mx;
    if (mpi::Error err = mpi::Allreduce(mpi::Slice(x), mpi::Into(&mx), XMPI_MAX)) return fail("allreduce(max)", err);
    for (size_t i = 0; i < n; i++)
      if (mx[i] != (float)size + (float)(i % 7) && bad++ < 3) fprintf(stderr, "node %d: max[%zu] = %g\n", rank, i, mx[i]);
  }

  // 4. allgather (every rank's id and element count), bcast from the last rank, reduce to rank 0
  std::vector<int64_t> mine = {rank, (int64_t)n}, all((size_t)2 * size);
  if (mpi::Error err = mpi::Allgather(mpi::Slice(mine), mpi::Into(&all))) return fail("allgather", err);
  for (int r = 0; r < size; r++)
    if (all[2 * r] != r || all[2 * r + 1] != (int64_t)n) bad++;
  std::vector<double> token = {rank == size - 1 ? 42.5 : 0.0};
  if (mpi::Error err = mpi::Bcast(mpi::Into(&token), size - 1)) return fail("bcast", err);
  if (token[0] != 42.5) bad++;
  std::vector<int64_t> one = {(int64_t)1 << rank}, mask(1);
  if (mpi::Error err = mpi::Reduce(mpi::Slice(one), mpi::Into(&mask), XMPI_SUM, 0)) return fail("reduce", err);
  if (rank == 0 && mask[0] != ((int64_t)1 << size) - 1) bad++;

  if (mpi::Error err = mpi::Barrier()) return fail("barrier", err);
  if (rank == 0)
    printf("allreduce of %zu float32 over %d nodes: %.1f us, %s\n", n, size, us, bad ? "WRONG" : "every result exact");
  if (!tcp) {
    gpu->Free(send);
    gpu->Free(recv);
  }
  mpi::Finalize();
  return bad ? 1 : 0;
}
