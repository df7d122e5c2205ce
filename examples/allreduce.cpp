// allreduce -- what a user of the reference would write once `//func AllReduce() {}` (mpi.go:130) is
// filled in: the package-level collectives of the C++ mirror of package mpi, on HBM buffers and on
// plain host slices, plus the non-blocking form.  Every result is checked against
// its closed form, so the exit status is the verdict.
//   xmpirun N allreduce [elements]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mpi.hpp"

static int fail(const char* what, const mpi::Error& err) {
  fprintf(stderr, "%s: %s\n", what, err.What().c_str());
  return 1;
}

int main(int argc, char** argv) {
  mpi::ParseFlags(&argc, argv);
  if (mpi::Error err = mpi::Init()) return fail("init", err);
  const int rank = mpi::Rank(), size = mpi::Size();
  const size_t n = argc > 1 ? (size_t)atoll(argv[1]) : (size_t)1 << 20;
  mpi::XGMI* gpu = mpi::DefaultBackend();

  // x_r[i] = (r + 1) + (i mod 7): every partial sum is a small integer, exact in float32
  std::vector<float> x(n), got(n);
  for (size_t i = 0; i < n; i++) x[i] = (float)(rank + 1) + (float)(i % 7);
  float* send = (float*)gpu->Malloc(n * sizeof(float));
  float* recv = (float*)gpu->Malloc(n * sizeof(float));
  if (!send || !recv) return fail("malloc", mpi::Error(XMPI_ERR_NOMEM, "out of HBM"));
  if (mpi::Error err = gpu->Memcpy(send, x.data(), n * sizeof(float))) return fail("upload", err);

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
  const auto t0 = std::chrono::steady_clock::now();
  if (mpi::Error err = mpi::Allreduce(mpi::Span(send, n), mpi::Span(recv, n))) return fail("allreduce", err);
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  if (mpi::Error err = gpu->Memcpy(got.data(), recv, n * sizeof(float))) return fail("download", err);
  check_sum(got, "allreduce(HBM)");

  // 2. plain host slices, as in the reference's examples (staged through HBM by the library)
  std::vector<float> host_out(n);
  if (mpi::Error err = mpi::Allreduce(mpi::Slice(x), mpi::Into(&host_out))) return fail("allreduce(host)", err);
  check_sum(host_out, "allreduce(host)");

  // 3. non-blocking: the caller is free to do its own work until WaitRequest
  xmpi_request* req = nullptr;
  if (mpi::Error err = gpu->IAllreduce(mpi::Span(send, n), mpi::Span(recv, n), XMPI_MAX, &req)) return fail("iallreduce", err);
  if (mpi::Error err = gpu->WaitRequest(req)) return fail("wait", err);
  if (mpi::Error err = gpu->Memcpy(got.data(), recv, n * sizeof(float))) return fail("download", err);
  for (size_t i = 0; i < n; i++)
    if (got[i] != (float)size + (float)(i % 7) && bad++ < 3) fprintf(stderr, "node %d: max[%zu] = %g\n", rank, i, got[i]);

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
  gpu->Free(send);
  gpu->Free(recv);
  mpi::Finalize();
  return bad ? 1 : 0;
}
