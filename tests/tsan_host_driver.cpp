// tsan_host_driver.cpp -- the library's REAL host-side shared-memory protocol code under ThreadSanitizer (test infrastructure).
//
// tests/*_sim.py check Python models of the protocols; this races the C++ that ships: ctl.cpp (join, barrier, pipe counters, mail
// entries, descriptors, retire logs, publication table, host lanes -- through xmpi_ctl_selftest) and engine.cpp's blocking
// Send / Receive of host slices (TagGuard, mail entries, host lanes, acks, truncation, withdrawal -- p2p_send / p2p_recv /
// p2p_probe), with the ranks as THREADS of this process that address the control block through ONE mapping
// (XMPI_CTL_SHARE_MAPPING=1: the sanitizer tells accesses apart by virtual address).  Built by `python -m mpi_amd.build --tsan`
// from the same sources with -fsanitize=thread; no GPU is needed and none is used (a communicator is put together by hand:
// what xmpi_init does before it touches the device).  What the reference would race here: network.go:448-497 (tagManager),
// :518-625 (Send / Receive + ack), mpi.go:121-125 (concurrent calls with distinct {peer, tag}).
//
// usage: tsan_host_bin <ranks> <rounds>        exit 0 = every payload arrived intact (the sanitizer reports on stderr and
//                                               turns the exit code into 66 through TSAN_OPTIONS=exitcode=66)
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../mpi_amd/csrc/comm.h"

using namespace xmpi;

namespace {

std::atomic<int> g_bad{0};

void fail(int rank, const char* what, int rc) {
  fprintf(stderr, "rank %d: %s: rc=%d (%s)\n", rank, what, rc, xmpi_last_error());
  g_bad.fetch_add(1);
}

uint8_t byte_at(int from, int tag, size_t i) { return (uint8_t)(i * 131u + (size_t)from * 17u + (size_t)tag * 3u + (i >> 11)); }

void fill(std::vector<uint8_t>& v, int from, int tag) {
  for (size_t i = 0; i < v.size(); i++) v[i] = byte_at(from, tag, i);
}
bool check(const std::vector<uint8_t>& v, size_t n, int from, int tag) {
  for (size_t i = 0; i < n; i++)
    if (v[i] != byte_at(from, tag, i)) return false;
  return true;
}

// what xmpi_init sets up before it opens the GPU, and nothing of what it sets up after
xmpi_comm* make_comm(const std::string& key, int rank, int size) {
  CtlConfig cfg{2, 8, 8u << 20, 2, 4u << 20, 1};  // (host lanes requested)
  std::string err;
  Ctl* ctl = nullptr;
  const int rc = Ctl::join(key, rank, size, cfg, 30.0, &ctl, &err);
  if (rc != XMPI_OK) {
    fprintf(stderr, "rank %d: join: %s\n", rank, err.c_str());
    return nullptr;
  }
  xmpi_comm* c = new xmpi_comm;
  c->rank = rank;
  c->size = size;
  c->device = -1;
  c->ctl = ctl;
  c->timeout_s = 60;
  c->p2p_agent_us = 0;
  c->p2p_kernel_ack = 0;
  // a blocking Send / Receive leases a stream for the copies of DEVICE payloads; there are none here: hand it tokens
  for (int k = 0; k < 8; k++) c->p2p_streams.push_back(reinterpret_cast<hipStream_t>((uintptr_t)(0x1000 + k)));
  return c;
}

void rank_main(const std::string& key, int rank, int size, int rounds) {
  // 1. the control plane's own self-test (every structure of the block, host lanes streamed by all ranks at once)
  int rc = xmpi_ctl_selftest((key + "-ctl").c_str(), rank, size, rounds);
  if (rc != XMPI_OK) return fail(rank, "ctl selftest", rc);

  xmpi_comm* c = make_comm(key + "-p2p", rank, size);
  if (!c) {
    g_bad.fetch_add(1);
    return;
  }
  const size_t lane = c->ctl->host_lane_bytes();
  if (lane == 0) fprintf(stderr, "rank %d: no host lanes (/dev/shm full?): the lane paths are not exercised\n", rank);
  const int next = (rank + 1) % size, prev = (rank + size - 1) % size;
  // lengths: empty, one byte, a piece, the ring exactly, beyond the ring (streams while the receiver drains)
  const size_t lens[] = {0, 1, 17, lane / 4, lane / 4 + 1, lane, lane + 13, 3 * lane + 5};
  if (size > 1) {
    for (int k = 0; k < rounds && g_bad.load() == 0; k++) {
      for (size_t li = 0; li < sizeof lens / sizeof lens[0]; li++) {
        const size_t n = lens[li];
        const int tag = 100 + (int)li;
        std::vector<uint8_t> out(n), in(n + 8, 0xEE);
        fill(out, rank, tag);
        // Send and Receive at once from two threads of the rank (the reference's goroutines, helloworld.go:53-81): a ring of
        // blocking rendezvous sends would deadlock otherwise
        std::thread tx([&] {
          const int r = p2p_send(c, out.data(), n, XMPI_U8, next, tag, true);
          if (r != XMPI_OK) fail(rank, "send", r);
        });
        size_t got = ~(size_t)0;
        const int r = p2p_recv(c, in.data(), n + 8, XMPI_U8, prev, tag, &got);
        tx.join();
        if (r != XMPI_OK) fail(rank, "recv", r);
        else if (got != n || !check(in, n, prev, tag) || in[n] != 0xEE) fail(rank, "payload", -1);
      }
      if (c->ctl->barrier(30.0) != XMPI_OK) return fail(rank, "barrier", -1);
    }
    // 2. helloworld's all-to-all: every rank sends to every other with tag = its own rank, receives in the OPPOSITE order;
    //    N-1 concurrent sends per rank on distinct {dest, tag}
    {
      std::vector<std::thread> tx;
      std::vector<std::vector<uint8_t>> outs((size_t)size);
      for (int d = 0; d < size; d++) {
        if (d == rank) continue;
        outs[(size_t)d].resize(1000 + 37 * (size_t)d);
        fill(outs[(size_t)d], rank, rank);
        tx.emplace_back([&, d] {
          const int r = p2p_send(c, outs[(size_t)d].data(), outs[(size_t)d].size(), XMPI_U8, d, rank, true);
          if (r != XMPI_OK) fail(rank, "all-to-all send", r);
        });
      }
      for (int s = size - 1; s >= 0; s--) {
        if (s == rank) continue;
        std::vector<uint8_t> in(1000 + 37 * (size_t)rank);
        size_t got = 0;
        // (probe first: the message's length before it is received)
        size_t pb = 0;
        int pd = -1;
        const int pr = p2p_probe(c, s, s, &pb, &pd);
        if (pr != XMPI_OK || pb != in.size() || pd != XMPI_U8) fail(rank, "probe", pr);
        const int r = p2p_recv(c, in.data(), in.size(), XMPI_U8, s, s, &got);
        if (r != XMPI_OK || got != in.size() || !check(in, got, s, s)) fail(rank, "all-to-all recv", r);
      }
      for (auto& t : tx) t.join();
      if (c->ctl->barrier(30.0) != XMPI_OK) return fail(rank, "barrier", -1);
    }
    // 3. the error paths: a message that does not fit (both sides learn it), a duplicate {peer, tag}
    {
      std::vector<uint8_t> out(4096), in(100);
      fill(out, rank, 7);
      int rs = 0;
      std::thread tx([&] { rs = p2p_send(c, out.data(), out.size(), XMPI_U8, next, 7, true); });
      size_t got = 0;
      const int rr = p2p_recv(c, in.data(), in.size(), XMPI_U8, prev, 7, &got);
      tx.join();
      if (rr != XMPI_ERR_TRUNCATE || got != out.size()) fail(rank, "truncate (receiver)", rr);
      if (rs != XMPI_ERR_TRUNCATE) fail(rank, "truncate (sender)", rs);
      if (c->ctl->barrier(30.0) != XMPI_OK) return fail(rank, "barrier", -1);
    }
  }
  // 3b. Send without waiting + Wait (mpi.go:132-152, the author's sketch): the payload is in the lane when the call returns, the
  //     ack is collected later; no second thread needed
  if (size > 1 && lane > 0) {
    std::vector<uint8_t> out(lane / 2), in(lane / 2);
    fill(out, rank, 11);
    int r = p2p_send(c, out.data(), out.size(), XMPI_U8, next, 11, false);
    if (r != XMPI_OK) fail(rank, "send (no wait)", r);
    size_t got = 0;
    r = p2p_recv(c, in.data(), in.size(), XMPI_U8, prev, 11, &got);
    if (r != XMPI_OK || got != in.size() || !check(in, got, prev, 11)) fail(rank, "recv behind a pending send", r);
    r = p2p_wait(c, next, 11);
    if (r != XMPI_OK) fail(rank, "wait", r);
    if (c->ctl->barrier(30.0) != XMPI_OK) return fail(rank, "barrier", -1);
  }
  // 4. self-send (network.go:388-446): sender and receiver are two threads of the rank
  {
    std::vector<uint8_t> out(lane + 99), in(lane + 99);
    fill(out, rank, 9);
    std::thread tx([&] {
      const int r = p2p_send(c, out.data(), out.size(), XMPI_U8, rank, 9, true);
      if (r != XMPI_OK) fail(rank, "self send", r);
    });
    size_t got = 0;
    const int r = p2p_recv(c, in.data(), in.size(), XMPI_U8, rank, 9, &got);
    tx.join();
    if (r != XMPI_OK || got != out.size() || !check(in, got, rank, 9)) fail(rank, "self recv", r);
  }
  (void)c->ctl->barrier(30.0);
  c->ctl->info(rank)->state.store(3, std::memory_order_release);
  delete c->ctl;
  c->ctl = nullptr;
  c->p2p_streams.clear();
  delete c;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc > 1 && std::string(argv[1]) == "--seed-race") {
    // proof that the sanitizer watches the control block: two "ranks" write one of its plain (non-atomic) fields unordered
    setenv("XMPI_CTL_SHARE_MAPPING", "1", 1);
    const std::string key = "tsan-seed-" + std::to_string((int)getpid());
    auto body = [&](int r) {
      CtlConfig cfg{2, 8, 8u << 20, 2, 4u << 20, 0};
      std::string err;
      Ctl* ctl = nullptr;
      if (Ctl::join(key, r, 2, cfg, 30.0, &ctl, &err) != XMPI_OK) return;
      ctl->mail(0, 1, 0)->tag = r;  // both ranks, no ordering between them
      (void)ctl->barrier(30.0);
      delete ctl;
    };
    std::thread a(body, 0), b(body, 1);
    a.join();
    b.join();
    return 0;
  }
  const int size = argc > 1 ? atoi(argv[1]) : 4, rounds = argc > 2 ? atoi(argv[2]) : 3;
  if (size < 1 || size > 8) return 2;
  setenv("XMPI_CTL_SHARE_MAPPING", "1", 1);
  const std::string key = "tsan-" + std::to_string((int)getpid());
  std::vector<std::thread> ranks;
  for (int r = 0; r < size; r++) ranks.emplace_back(rank_main, key, r, size, rounds);
  for (auto& t : ranks) t.join();
  if (g_bad.load()) {
    fprintf(stderr, "tsan_host_driver: %d failure(s)\n", g_bad.load());
    return 1;
  }
  printf("tsan_host_driver ok: %d ranks as threads, %d rounds, host lanes of %zu bytes per entry\n", size, rounds,
         Ctl::lane_bytes_for(size));
  return 0;
}
