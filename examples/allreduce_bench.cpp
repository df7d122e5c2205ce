// allreduce_bench -- the production layout's figures: ONE OS PROCESS PER RANK (xmpirun), ranks meeting on the device.
// The reference's only benchmark is the bounce timing loop (examples/bounce/bounce.go:83-151); this is its counterpart
// for the collective the reference stubs (mpi.go:130) and for Send / Receive through the C ABI, written against the
// C++ mirror of package mpi the way a Go program would be written against the cgo shim.
//
//   xmpirun N allreduce_bench <bytes per rank> <steps> <warmup> [mode ...]
//
// For every mode (auto | fused | fused2 | split | zpush | ring | rhd): `steps` float32 allreduces enqueued back to back
// (xmpi_allreduce_repeat), bracketed by barriers, max over ranks; the dominant kernel's own duration from HIP events
// attached to sampled dispatches (rank 0); the result checked against its closed form.  "auto" first lets the library
// tune its schedule table (xmpi_tune) and reports what AUTO resolved to.  Then, between ranks 0 and 1, the ping-pong of
// bounce.go for a few lengths.  Rank 0 prints one JSON line.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mpi.hpp"

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
  const size_t bytes = argc > 1 ? (size_t)atoll(argv[1]) : (size_t)256 << 20;
  const int steps = argc > 2 ? atoi(argv[2]) : 20, warmup = argc > 3 ? atoi(argv[3]) : 5;
  std::vector<std::string> modes;
  for (int i = 4; i < argc; i++) modes.push_back(argv[i]);
  if (modes.empty()) modes = {"auto"};
  mpi::XGMI* gpu = mpi::DefaultBackend();
  xmpi_comm* c = gpu->Handle();
  const size_t n = bytes / 4;
  float* send = (float*)gpu->Malloc(bytes);
  float* recv = (float*)gpu->Malloc(bytes);
  if (!send || !recv) return fail("malloc", XMPI_ERR_NOMEM);
  {  // x_r[i] = (r + 1) + i % 7: every partial sum is a small integer, exact in any order
    std::vector<float> x(n);
    for (size_t i = 0; i < n; i++) x[i] = (float)(rank + 1) + (float)(i % 7);
    if (mpi::Error err = gpu->Memcpy(send, x.data(), bytes)) return fail("upload", err.Code());
  }
  const bool on_device = xmpi_get_param(c, "dsync") == 1;
  const long split0 = xmpi_get_param(c, "dsync_split_bytes"), unroll0 = xmpi_get_param(c, "dsync_unroll");
  std::string rows;
  int bad = 0;
  auto max_over_ranks = [&](double v) {
    std::vector<double> mine = {v}, worst(1);
    (void)mpi::Allreduce(mpi::Slice(mine), mpi::Into(&worst), XMPI_MAX);
    return worst[0];
  };
  for (const std::string& mode : modes) {
    int algo = XMPI_ALGO_AUTO;
    xmpi_set_param(c, "tuned", 0);
    xmpi_set_param(c, "dsync_split_bytes", split0);
    xmpi_set_param(c, "dsync_unroll", unroll0);
    double tune_ms = 0;
    if (mode == "auto") {
      const double t0 = now_us();
      int rc = xmpi_tune(c, bytes);
      if (rc != XMPI_OK) return fail("tune", rc);
      tune_ms = (now_us() - t0) / 1e3;
    } else if (mode == "fused" || mode == "fused2") {
      algo = XMPI_ALGO_ZCOPY;
      xmpi_set_param(c, "dsync_split_bytes", 0);
      xmpi_set_param(c, "dsync_unroll", mode == "fused2" ? 2 : 1);
    } else if (mode == "split") {
      algo = XMPI_ALGO_ZCOPY;
      xmpi_set_param(c, "dsync_split_bytes", 1);
    } else if (mode == "zpush") {
      algo = XMPI_ALGO_ZPUSH;
    } else if (mode == "ring") {
      algo = XMPI_ALGO_RING;
    } else if (mode == "rhd") {
      algo = XMPI_ALGO_RHD;
    } else if (mode == "ring_push") {  // the push forms: every payload byte a posted store over its link
      algo = XMPI_ALGO_RING_PUSH;
    } else if (mode == "rhd_push") {
      algo = XMPI_ALGO_RHD_PUSH;
    } else {
      fprintf(stderr, "unknown mode %s\n", mode.c_str());
      return 1;
    }
    int rc = xmpi_memset(c, recv, 0, bytes);
    if (rc == XMPI_OK) rc = xmpi_allreduce_repeat(c, send, recv, n, XMPI_F32, XMPI_SUM, algo, warmup > 0 ? warmup : 1);
    if (rc != XMPI_OK) return fail("warmup", rc);
    // the result, every element of a few windows, against the closed form
    std::vector<float> got(65536);
    for (size_t off : {(size_t)0, n / 3, n > got.size() ? n - got.size() : (size_t)0}) {
      const size_t m = std::min(got.size(), n - off);
      if (mpi::Error err = gpu->Memcpy(got.data(), recv + off, m * 4)) return fail("download", err.Code());
      for (size_t i = 0; i < m; i++)
        if (got[i] != (float)(size * (size + 1) / 2) + (float)size * (float)((off + i) % 7)) bad++;
    }
    xmpi_prof_reset(c);
    xmpi_set_param(c, "prof_every", 4);
    mpi::Barrier();
    xmpi_sync(c);
    xmpi_prof_enable(c, 1);
    const double t0 = now_us();
    rc = xmpi_allreduce_repeat(c, send, recv, n, XMPI_F32, XMPI_SUM, algo, steps);
    xmpi_sync(c);
    mpi::Barrier();
    const double dt_us = (now_us() - t0) / steps;
    xmpi_prof_enable(c, 0);
    if (rc != XMPI_OK) return fail("timed steps", rc);
    const double t = max_over_ranks(dt_us);
    uint64_t launches = 0, kbytes = 0;
    double kms = 0;
    xmpi_prof_get(c, 4, &launches, &kms, &kbytes);
    int cls = 0;
    while (cls + 1 < 24 && (bytes >> (cls + 9)) != 0) cls++;
    char pat[64], row[640];
    snprintf(pat, sizeof pat, "tune_algo_0_%d", cls);
    const long talgo = xmpi_get_param(c, pat);
    snprintf(pat, sizeof pat, "tune_split_0_%d", cls);
    const long tsplit = xmpi_get_param(c, pat);
    snprintf(pat, sizeof pat, "tune_unroll_0_%d", cls);
    const long tunroll = xmpi_get_param(c, pat);
    snprintf(row, sizeof row,
             "%s{\"mode\": \"%s\", \"us_per_step\": %.2f, \"algbw_GBps\": %.3f, \"busbw_GBps\": %.3f, \"kernel_launches_sampled\": %llu, "
             "\"kernel_avg_us\": %.2f, \"kernel_bytes_per_launch\": %.0f, \"kernel_GBps\": %.1f, \"tuned\": {\"algo\": %ld, \"split\": %ld, "
             "\"unroll\": %ld, \"tune_ms\": %.1f}, \"split_launches\": %ld, \"sched_launches\": %ld}",
             rows.empty() ? "" : ", ", mode.c_str(), t, (double)bytes / t / 1e3, (double)bytes / t / 1e3 * 2.0 * (size - 1) / size,
             (unsigned long long)launches, launches ? kms * 1e3 / (double)launches : 0.0, launches ? (double)kbytes / (double)launches : 0.0,
             kms > 0 ? (double)kbytes / (kms * 1e-3) / 1e9 : 0.0, talgo, tsplit, tunroll, tune_ms, xmpi_get_param(c, "dsync_split_launches"),
             xmpi_get_param(c, "dsync_sched_launches"));
    rows += row;
  }
  // Send / Receive through the C ABI: bounce.go's ping-pong between ranks 0 and 1 (half round trip, device buffers)
  std::string bounce;
  if (size >= 2) {
    for (size_t len : {(size_t)0, (size_t)8, (size_t)1024, (size_t)65536, (size_t)1 << 20, (size_t)16 << 20}) {
      if (len > bytes) break;
      const int reps = len <= ((size_t)1 << 20) ? 200 : 30;
      double half = 0;
      mpi::Barrier();
      if (rank < 2) {
        const int peer = 1 - rank;
        double t0 = 0;
        for (int w = 0; w < reps + 20; w++) {
          if (w == 20) t0 = now_us();
          int rc;
          if (rank == 0) {
            rc = xmpi_send(c, send, len, XMPI_U8, peer, 3);
            if (rc == XMPI_OK) rc = xmpi_recv(c, recv, len, XMPI_U8, peer, 3, nullptr);
          } else {
            rc = xmpi_recv(c, recv, len, XMPI_U8, peer, 3, nullptr);
            if (rc == XMPI_OK) rc = xmpi_send(c, recv, len, XMPI_U8, peer, 3);
          }
          if (rc != XMPI_OK) return fail("bounce", rc);
        }
        half = (now_us() - t0) / reps / 2;
      }
      mpi::Barrier();
      char row[160];
      snprintf(row, sizeof row, "%s{\"bytes\": %zu, \"half_round_trip_us\": %.2f, \"GBps\": %.3f}", bounce.empty() ? "" : ", ", len, half,
               half > 0 ? (double)len / half / 1e3 : 0.0);
      if (rank == 0) bounce += row;
    }
  }
  mpi::Barrier();
  if (rank == 0)
    printf("{\"ranks\": %d, \"one_process_per_rank\": true, \"meet\": \"%s\", \"bytes_per_rank\": %zu, \"steps\": %d, \"warmup\": %d, "
           "\"sharers\": %ld, \"xcds\": %ld, \"xcd_probe_mask\": %ld, \"xcd_meet_mask\": %ld, \"xcd_done_mask\": %ld, \"xcd_short\": %ld, "
           "\"body_sys\": %ld, \"degraded\": %ld, \"tune_rejected\": %ld, \"tune_check_ms\": %.1f, \"init_selfcheck_us\": %ld, \"exact\": %s, \"rows\": [%s], "
           "\"bounce\": [%s]}\n",
           size, on_device ? "on the device (flag words in HBM)" : "on the host (control block)", bytes, steps, warmup,
           xmpi_get_param(c, "dsync_sharers"), xmpi_get_param(c, "xcds"), xmpi_get_param(c, "xcd_probe_mask"),
           xmpi_get_param(c, "xcd_meet_mask"), xmpi_get_param(c, "xcd_done_mask"), xmpi_get_param(c, "xcd_short"),
           xmpi_get_param(c, "body_sys"), xmpi_get_param(c, "degraded"), xmpi_get_param(c, "tune_rejected"), (double)xmpi_get_param(c, "tune_check_us") / 1e3,
           xmpi_get_param(c, "init_selfcheck_us"), bad ? "false" : "true", rows.c_str(), bounce.c_str());
  gpu->Free(send);
  gpu->Free(recv);
  mpi::Finalize();
  return bad ? 1 : 0;
}
