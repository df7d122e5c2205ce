// bounce -- the reference's examples/bounce/bounce.go against the C++ mirror of package mpi, with
// the message buffers resident in HBM (BASELINE config 2).  Even ranks send to rank+1, odd ranks
// echo; the even rank checks the echo is bit-identical (bytes.Equal / floats.Equal upstream, the
// LDS+shuffle compare kernel here) and prints the mean round-trip time in microseconds per length.
//   xmpirun 2 bounce [--host | --tcp]   --host keeps the buffers in host memory (staged through HBM);
//                                        --tcp runs the reference's TCP + gob protocol instead (mpi::Network)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "mpi.hpp"
#include "network.hpp"

static const size_t kLengths[] = {0, 1, 10, 100, 1000, 10000, 100000, 1000000, 10000000};  // bounce.go:33
static int kRepeats = 10;                                                                  // bounce.go:35

int main(int argc, char** argv) {
  mpi::ParseFlags(&argc, argv);
  bool host = false, tcp = false;
  size_t max_length = 10000000;  // the reference's longest message; tests may stop earlier
  for (int i = 1; i < argc; i++) {
    host = host || !strcmp(argv[i], "--host");
    tcp = tcp || !strcmp(argv[i], "--tcp");  // the reference's own backend (TCP + gob, wire-compatible); host buffers
    if (!strcmp(argv[i], "--max-length") && i + 1 < argc) max_length = (size_t)atoll(argv[++i]);
    else if (!strcmp(argv[i], "--repeats") && i + 1 < argc) kRepeats = atoi(argv[++i]);
  }
  static mpi::Network net;
  if (tcp) {
    host = true;
    mpi::Register(&net);
  }
  if (mpi::Error err = mpi::Init()) {
    fprintf(stderr, "error initializing: %s\n", err.What().c_str());
    return 1;
  }
  const int rank = mpi::Rank();
  if (rank < 0) {
    fprintf(stderr, "Incorrect initialization\n");
    return 1;
  }
  const bool even = rank % 2 == 0;
  const int size = mpi::Size();
  if (size % 2 != 0) {
    fprintf(stderr, "Must have an even number of nodes for this example\n");
    return 1;
  }
  if (rank == 0) printf("Number of nodes =  %d\n", size);

  mpi::XGMI* gpu = mpi::DefaultBackend();
  const size_t maxsize = kLengths[sizeof kLengths / sizeof kLengths[0] - 1];
  std::mt19937_64 rng(12345 + (uint64_t)rank);
  std::vector<uint8_t> h_msg(maxsize);
  std::vector<double> h_msgf(maxsize / 8);
  for (size_t i = 0; i < maxsize / 8; i++) {
    const uint64_t v = rng();
    memcpy(&h_msg[i * 8], &v, 8);
    h_msgf[i] = (double)(rng() >> 11) * 0x1p-53;
  }
  uint8_t *msg, *rcv;
  double *msgf, *rcvf;
  if (host) {
    msg = h_msg.data();
    rcv = (uint8_t*)calloc(maxsize, 1);
    msgf = h_msgf.data();
    rcvf = (double*)calloc(maxsize / 8, 8);
  } else {
    msg = (uint8_t*)gpu->Malloc(maxsize);
    rcv = (uint8_t*)gpu->Malloc(maxsize);
    msgf = (double*)gpu->Malloc(maxsize);
    rcvf = (double*)gpu->Malloc(maxsize);
    gpu->Memcpy(msg, h_msg.data(), maxsize);
    gpu->Memcpy(msgf, h_msgf.data(), maxsize / 8 * 8);
  }

  size_t nlen = 0;
  while (nlen < sizeof kLengths / sizeof kLengths[0] && kLengths[nlen] <= max_length) nlen++;
  std::vector<long long> times(nlen, 0), timesf(nlen, 0);
  auto same = [&](const void* a, const void* b, size_t bytes) {
    if (host) return memcmp(a, b, bytes) == 0;
    uint64_t bad = 1;
    return xmpi_count_mismatch(gpu->Handle(), a, b, bytes, &bad) == XMPI_OK && bad == 0;
  };
  for (size_t i = 0; i < nlen; i++) {
    const size_t l = kLengths[i];
    for (int j = 0; j < kRepeats; j++) {
      auto t0 = std::chrono::steady_clock::now();
      if (even) {
        mpi::Send(mpi::Span(msg, l), rank + 1, 0);
        mpi::Receive(mpi::Span(rcv, l), rank + 1, 0);
      } else {
        mpi::Receive(mpi::Span(rcv, l), rank - 1, 0);
        mpi::Send(mpi::Span(rcv, l), rank - 1, 0);
      }
      times[i] += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
      if (even && !same(msg, rcv, l)) {
        fprintf(stderr, "message not the same\n");
        return 1;
      }
      // zero the buffer so the next round cannot pass on stale data (bounce.go:109-112)
      if (host) memset(rcv, 0, l);
      else xmpi_memset(gpu->Handle(), rcv, 0, l);

      t0 = std::chrono::steady_clock::now();
      if (even) {
        mpi::Send(mpi::Span(msgf, l / 8), rank + 1, 0);
        mpi::Receive(mpi::Span(rcvf, l / 8), rank + 1, 0);
      } else {
        mpi::Receive(mpi::Span(rcvf, l / 8), rank - 1, 0);
        mpi::Send(mpi::Span(rcvf, l / 8), rank - 1, 0);
      }
      timesf[i] += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
      if (even && !same(msgf, rcvf, l / 8 * 8)) {
        fprintf(stderr, "message not the same\n");
        return 1;
      }
    }
  }
  if (even) {  // bounce.go:140-152
    printf("Average byte trip time in \xC2\xB5s between node %d and %d: [", rank, rank + 1);
    for (size_t i = 0; i < nlen; i++) printf("%s%lld", i ? " " : "", times[i] / 1000 / kRepeats);
    printf("]\nAverage float64 trip time in \xC2\xB5s between node %d and %d: [", rank, rank + 1);
    for (size_t i = 0; i < nlen; i++) printf("%s%lld", i ? " " : "", timesf[i] / 1000 / kRepeats);
    printf("]\n");
  }
  fflush(stdout);
  mpi::Finalize();
  return 0;
}
