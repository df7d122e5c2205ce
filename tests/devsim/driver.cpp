// tests/devsim/driver.cpp -- the library WHOLE (host sources + the gfx950 kernel sources compiled for the CPU, tests/devsim)
// with the ranks as THREADS of this process, every rank on a virtual HIP device of its own, driven through the C ABI of
// include/xmpi.h.  Built with -fsanitize=thread (tests/devsim/build.py --tsan): what races here is the code that ships --
// kdev.h dsync_begin / dsync_end, the one-kernel fold, meet / body / done, the LL lines, the stepped ring / halving / tree
// kernels, the stream-ordered Send / Receive kernels, the copy-and-ack kernel and the receive agent, and the host code that
// enqueues them -- not a Python model of it ("nothing races the actual atomics").  Ranks that are threads on
// DIFFERENT devices meet on the device exactly like processes do (dsync.cpp dsync_connect), and use each other's pointers
// directly: ONE virtual address per word, which is what the sanitizer needs to see a race.
//
// What the reference would race here: network.go:448-497 (tagManager), :518-625 (Send / Receive + ack); the collectives
// have no upstream counterpart (mpi.go:130).
//
// usage: devsim_tsan_bin <ranks> <rounds> [scenario ...]      scenarios: fold split ll sched bcast reduce allgather
//                                                             walk stream graph p2p_stream p2p_block (default: all)
//        devsim_tsan_bin --shared <ranks> <rounds> [...]     every rank on device 0: the ranks meet on the HOST (zcopy.cpp's rendezvous,
//                                                             one launch folds everybody's chunks; the step tables through the windows)
//        devsim_tsan_bin --seed-race <ranks>                  the same allreduce with a rank that reads its result back before
//                                                             the collective has completed: the sanitizer must report it
// exit 0 = every result was right; the sanitizer reports on stderr and turns the exit code into 66 (TSAN_OPTIONS=exitcode=66)
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../../include/xmpi.h"

namespace {

std::atomic<int> g_bad{0};
std::set<std::string> g_only;
bool g_seed_race = false;
bool g_shared = false;  // --shared: every rank on device 0 (ranks that share a process AND a GPU: they meet on the host, zcopy.cpp)

bool wants(const char* name) { return g_only.empty() || g_only.count(name) > 0; }

#define CHECK(call)                                                                                              \
  do {                                                                                                           \
    const int _rc = (call);                                                                                      \
    if (_rc != XMPI_OK) {                                                                                        \
      fprintf(stderr, "rank %d: %s:%d: %s -> %d (%s)\n", rank, __FILE__, __LINE__, #call, _rc, xmpi_last_error()); \
      g_bad.fetch_add(1);                                                                                        \
      return;                                                                                                    \
    }                                                                                                            \
  } while (0)

// exactly summable in any order: small integers
inline int64_t in_i64(int rank, size_t i, int salt) { return (int64_t)((i * 7 + (size_t)rank * 13 + (size_t)salt * 5) % 1000) - 500; }
inline float in_f32(int rank, size_t i, int salt) { return (float)(int)((i * 3 + (size_t)rank * 11 + (size_t)salt) % 257) - 128.0f; }

struct Rank {
  int rank, size;
  xmpi_comm* c = nullptr;
  void *send = nullptr, *recv = nullptr;  // registered device buffers (xmpi_malloc)
  size_t cap = 0;
  std::vector<char> host;

  void upload(const void* src, size_t bytes) { (void)xmpi_memcpy(c, send, src, bytes); }
  void download(size_t bytes) {
    host.resize(bytes);
    (void)xmpi_memcpy(c, host.data(), recv, bytes);
  }
  bool expect_sum_i64(size_t count, int salt, const char* what) {
    download(count * 8);
    const int64_t* got = (const int64_t*)host.data();
    for (size_t i = 0; i < count; i++) {
      int64_t want = 0;
      for (int r = 0; r < size; r++) want += in_i64(r, i, salt);
      if (got[i] != want) {
        fprintf(stderr, "rank %d: %s: element %zu of %zu is %lld, expected %lld\n", rank, what, i, count, (long long)got[i], (long long)want);
        g_bad.fetch_add(1);
        return false;
      }
    }
    return true;
  }
  bool expect_sum_f32(size_t count, int salt, const char* what) {
    download(count * 4);
    const float* got = (const float*)host.data();
    for (size_t i = 0; i < count; i++) {
      float want = 0;
      for (int r = 0; r < size; r++) want += in_f32(r, i, salt);
      if (got[i] != want) {
        fprintf(stderr, "rank %d: %s: element %zu of %zu is %g, expected %g\n", rank, what, i, count, (double)got[i], (double)want);
        g_bad.fetch_add(1);
        return false;
      }
    }
    return true;
  }
  void fill_i64(size_t count, int salt) {
    std::vector<int64_t> v(count);
    for (size_t i = 0; i < count; i++) v[i] = in_i64(rank, i, salt);
    upload(v.data(), count * 8);
  }
  void fill_f32(size_t count, int salt) {
    std::vector<float> v(count);
    for (size_t i = 0; i < count; i++) v[i] = in_f32(rank, i, salt);
    upload(v.data(), count * 4);
  }
};

// idle_first: the call finds its stream idle (what lets the LL agent take a blocking small collective: ll.hip ll_agent_kernel)
void allreduce_case(Rank& R, size_t count, int algo, int salt, const char* what, bool idle_first = false) {
  const int rank = R.rank;
  if (salt & 1) {
    R.fill_i64(count, salt);
    (void)xmpi_memset(R.c, R.recv, 0xEE, count * 8);
    if (idle_first) (void)xmpi_sync(R.c);
    CHECK(xmpi_allreduce(R.c, R.send, R.recv, count, XMPI_I64, XMPI_SUM, algo));
    (void)R.expect_sum_i64(count, salt, what);
  } else {
    R.fill_f32(count, salt);
    (void)xmpi_memset(R.c, R.recv, 0xEE, count * 4);
    if (idle_first) (void)xmpi_sync(R.c);
    CHECK(xmpi_allreduce(R.c, R.send, R.recv, count, XMPI_F32, XMPI_SUM, algo));
    (void)R.expect_sum_f32(count, salt, what);
  }
}

// in place: the receive buffer is the input (a push-form ring then lands its partial results in lent landing blocks)
void allreduce_in_place_case(Rank& R, size_t count, int algo, int salt, const char* what) {
  const int rank = R.rank;
  std::vector<int64_t> v(count);
  for (size_t i = 0; i < count; i++) v[i] = in_i64(rank, i, salt);
  (void)xmpi_memcpy(R.c, R.recv, v.data(), count * 8);
  CHECK(xmpi_allreduce(R.c, R.recv, R.recv, count, XMPI_I64, XMPI_SUM, algo));
  (void)R.expect_sum_i64(count, salt, what);
}

void rank_main(const std::string& key, int rank, int size, int rounds) {
  Rank R;
  R.rank = rank;
  R.size = size;
  CHECK(xmpi_init(rank, size, g_shared ? 0 : rank, key.c_str(), &R.c));
  xmpi_comm* c = R.c;
  const bool dev = !g_shared;  // the ranks meet on the device
  if (dev && xmpi_get_param(c, "dsync") != 1) {
    fprintf(stderr, "rank %d: the ranks do not meet on the device (dsync = %ld): nothing of interest would run\n", rank, xmpi_get_param(c, "dsync"));
    g_bad.fetch_add(1);
    return;
  }
  R.cap = (size_t)1 << 20;
  R.send = xmpi_malloc(c, R.cap);
  R.recv = xmpi_malloc(c, R.cap * (size_t)size);
  if (!R.send || !R.recv) {
    fprintf(stderr, "rank %d: xmpi_malloc: %s\n", rank, xmpi_last_error());
    g_bad.fetch_add(1);
    return;
  }
  CHECK(xmpi_barrier(c));
  const size_t counts[] = {1, 257, 4099, 40001};
  int salt = 0;

  if (g_seed_race) {
    // the SAME collective, but this rank looks at a peer-written buffer without waiting for the collective: enqueue on a
    // stream and read the result at once.  (Proof that the sanitizer watches the device buffers and the kernels' stores.)
    void* s = xmpi_stream_create(c);
    R.fill_i64(4099, 1);
    CHECK(xmpi_allreduce_on_stream(c, R.send, R.recv, 4099, XMPI_I64, XMPI_SUM, s));
    usleep(200000);  // (after the kernels' stores in every schedule: locks of the runtime passed on the way there cannot order an access that is already behind)
    volatile int64_t peek = ((volatile int64_t*)R.recv)[7];  // unordered against the kernels' stores
    (void)peek;
    CHECK(xmpi_stream_sync(c, s));
    CHECK(xmpi_barrier(c));
    (void)xmpi_finalize(c);
    return;
  }

  for (int round = 0; round < rounds && g_bad.load() == 0; round++) {
    // ---- the zero-copy fold: one kernel (rendezvous + fold + close), every size -------------------------------------------------
    if (wants("fold")) {
      CHECK(xmpi_set_param(c, "dsync_split_bytes", 0));  // never split
      CHECK(xmpi_set_param(c, "ll_bytes", 0));
      for (size_t n : counts) allreduce_case(R, n, XMPI_ALGO_ZCOPY, ++salt, "fold");
      allreduce_case(R, 4099, XMPI_ALGO_ZPUSH, ++salt, "push-only");  // (a ragged count: through the communicators' own blocks)
      allreduce_case(R, (size_t)size * 4096, XMPI_ALGO_ZPUSH, ++salt, "push-only, equal chunks");  // (through the receive buffers)
      allreduce_in_place_case(R, 40001, XMPI_ALGO_ZPUSH, ++salt, "push-only, in place");
      // in place
      R.fill_i64(4099, ++salt);
      (void)xmpi_memcpy(c, R.recv, R.send, 4099 * 8);
      CHECK(xmpi_allreduce(c, R.recv, R.recv, 4099, XMPI_I64, XMPI_SUM, XMPI_ALGO_ZCOPY));
      (void)R.expect_sum_i64(4099, salt, "fold in place");
    }
    // ---- meet / body / done (the split form), with the plain and the system-scope data kernel -----------------------------
    if (wants("split") && dev) {
      CHECK(xmpi_set_param(c, "dsync_split_bytes", 1));  // always split
      CHECK(xmpi_set_param(c, "ll_bytes", 0));
      for (int sys = 0; sys < 2; sys++) {
        CHECK(xmpi_set_param(c, "body_sys", sys));
        for (size_t n : counts) allreduce_case(R, n, XMPI_ALGO_ZCOPY, ++salt, sys ? "split (system scope)" : "split");
      }
      CHECK(xmpi_set_param(c, "body_sys", 0));
      if (xmpi_get_param(c, "dsync_split_launches") <= 0) {
        fprintf(stderr, "rank %d: the split form never ran\n", rank);
        g_bad.fetch_add(1);
      }
      CHECK(xmpi_set_param(c, "dsync_split_bytes", 4 << 20));
    }
    // ---- LL lines ---------------------------------------------------------------------------------------------------------------
    if (wants("ll") && dev) {
      for (size_t n : {(size_t)1, (size_t)33, (size_t)257, (size_t)4096}) {
        allreduce_case(R, n, XMPI_ALGO_LL, ++salt, "LL allreduce");
        allreduce_case(R, n, XMPI_ALGO_LL, ++salt, "LL allreduce");  // (both parities)
      }
      for (int root = 0; root < size; root++) {
        R.fill_i64(300, ++salt);
        if (rank == root) (void)xmpi_memcpy(c, R.recv, R.send, 300 * 8);
        else (void)xmpi_memset(c, R.recv, 0xEE, 300 * 8);
        CHECK(xmpi_bcast(c, R.recv, 300, XMPI_I64, root, XMPI_ALGO_LL));
        R.download(300 * 8);
        for (size_t i = 0; i < 300; i++)
          if (((const int64_t*)R.host.data())[i] != in_i64(root, i, salt)) {
            fprintf(stderr, "rank %d: LL bcast from %d: element %zu\n", rank, root, i);
            g_bad.fetch_add(1);
            break;
          }
        R.fill_i64(300, ++salt);
        CHECK(xmpi_reduce(c, R.send, R.recv, 300, XMPI_I64, XMPI_SUM, root, XMPI_ALGO_LL));
        if (rank == root) (void)R.expect_sum_i64(300, salt, "LL reduce");
      }
      R.fill_i64(200, ++salt);
      CHECK(xmpi_allgather(c, R.send, R.recv, 200, XMPI_I64, XMPI_ALGO_LL));
      R.download(200 * 8 * (size_t)size);
      for (int r = 0; r < size; r++)
        for (size_t i = 0; i < 200; i++)
          if (((const int64_t*)R.host.data())[(size_t)r * 200 + i] != in_i64(r, i, salt)) {
            fprintf(stderr, "rank %d: LL allgather: block %d element %zu\n", rank, r, i);
            g_bad.fetch_add(1);
            r = size;
            break;
          }
    }
    // ---- the same lines run by the lingering LL agent: blocking calls that find their stream idle ------------------------------
    if (wants("ll") && dev && xmpi_get_param(c, "agent_ll") >= 1 && xmpi_get_param(c, "ll_agent_us") > 0) {
      CHECK(xmpi_set_param(c, "agent_ll_bytes", 32768));
      CHECK(xmpi_set_param(c, "agent_ll", 2));
      const long ag0 = xmpi_get_param(c, "dsync_ll_agent");
      long expect = 0;
      for (size_t n : {(size_t)1, (size_t)33, (size_t)257, (size_t)4096}) {
        allreduce_case(R, n, XMPI_ALGO_LL, ++salt, "LL agent allreduce", true);
        allreduce_case(R, n, XMPI_ALGO_LL, ++salt, "LL agent allreduce", true);
        allreduce_case(R, n, XMPI_ALGO_LL, ++salt, "LL allreduce, launched or by the agent");  // (whichever: the stream may be busy)
        expect += 2;
      }
      for (int root = 0; root < size; root++) {
        R.fill_i64(300, ++salt);
        if (rank == root) (void)xmpi_memcpy(c, R.recv, R.send, 300 * 8);
        else (void)xmpi_memset(c, R.recv, 0xEE, 300 * 8);
        (void)xmpi_sync(c);
        CHECK(xmpi_bcast(c, R.recv, 300, XMPI_I64, root, XMPI_ALGO_LL));
        R.download(300 * 8);
        for (size_t i = 0; i < 300; i++)
          if (((const int64_t*)R.host.data())[i] != in_i64(root, i, salt)) {
            fprintf(stderr, "rank %d: LL agent bcast from %d: element %zu\n", rank, root, i);
            g_bad.fetch_add(1);
            break;
          }
        R.fill_i64(300, ++salt);
        (void)xmpi_sync(c);
        CHECK(xmpi_reduce(c, R.send, R.recv, 300, XMPI_I64, XMPI_SUM, root, XMPI_ALGO_LL));
        if (rank == root) (void)R.expect_sum_i64(300, salt, "LL agent reduce");
        expect += 2;
        if (rank == root) usleep(300);  // (the others' agents wait for this rank inside the next collective; its own has gone)
      }
      {  // back to back, nothing in between: the agent counts the epochs itself (no load from the page); then a launched kernel moves it
        R.fill_i64(64, ++salt);
        (void)xmpi_sync(c);
        for (int i = 0; i < 6; i++) {
          CHECK(xmpi_allreduce(c, R.send, R.recv, 64, XMPI_I64, XMPI_SUM, XMPI_ALGO_LL));
          CHECK(xmpi_allreduce(c, R.recv, R.send, 64, XMPI_I64, XMPI_SUM, XMPI_ALGO_LL));
        }
        expect += 12;
        CHECK(xmpi_allreduce(c, R.send, R.recv, 64, XMPI_I64, XMPI_SUM, XMPI_ALGO_ZCOPY));  // launched: the 13th
        CHECK(xmpi_allreduce(c, R.recv, R.send, 64, XMPI_I64, XMPI_SUM, XMPI_ALGO_LL));     // the 14th, wherever it ran
        CHECK(xmpi_allreduce(c, R.send, R.recv, 64, XMPI_I64, XMPI_SUM, XMPI_ALGO_LL));     // the 15th
        R.download(64 * 8);
        for (size_t i = 0; i < 64; i++) {
          uint64_t want = 0;
          for (int r = 0; r < size; r++) want += (uint64_t)in_i64(r, i, salt);
          for (int k = 0; k < 14; k++) want *= (uint64_t)size;
          if (((const uint64_t*)R.host.data())[i] != want) {
            fprintf(stderr, "rank %d: LL agent, back to back: element %zu\n", rank, i);
            g_bad.fetch_add(1);
            break;
          }
        }
      }
      if (xmpi_get_param(c, "dsync_ll_agent") - ag0 < expect) {
        fprintf(stderr, "rank %d: the LL agent ran %ld of %ld blocking collectives\n", rank, xmpi_get_param(c, "dsync_ll_agent") - ag0, expect);
        g_bad.fetch_add(1);
      }
      CHECK(xmpi_set_param(c, "agent_ll_bytes", 8192));
      CHECK(xmpi_set_param(c, "agent_ll", 1));
    }
    // ---- the stepped kernels: ring, recursive halving + doubling (any N) --------------------------------------------------------
    if (wants("sched")) {
      for (size_t n : counts) {
        allreduce_case(R, n, XMPI_ALGO_RING, ++salt, "ring kernel");
        allreduce_case(R, n, XMPI_ALGO_RHD, ++salt, "halving kernel");
        // the push forms: a step stores into the peer's receive buffer / landing block and reads only what landed here
        allreduce_case(R, n, XMPI_ALGO_RING_PUSH, ++salt, "ring kernel, push");
        allreduce_case(R, n, XMPI_ALGO_RHD_PUSH, ++salt, "halving kernel, push");
        allreduce_in_place_case(R, n, XMPI_ALGO_RING_PUSH, ++salt, "ring kernel, push, in place");
        allreduce_in_place_case(R, n, XMPI_ALGO_RHD_PUSH, ++salt, "halving kernel, push, in place");
        allreduce_in_place_case(R, n, XMPI_ALGO_RING, ++salt, "ring kernel, in place");
      }
      // back to back on the stream, nobody waiting in between: the push forms' landing block is the communicator's, used again by
      // the next collective while the host has not seen the last one end (the peers store into it only after this rank's next
      // kernel has announced it)
      {
        const size_t n = 4099;
        R.fill_i64(n, ++salt);
        CHECK(xmpi_allreduce_repeat(c, R.send, R.recv, n, XMPI_I64, XMPI_SUM, XMPI_ALGO_RHD_PUSH, 3));
        (void)R.expect_sum_i64(n, salt, "halving kernel, push, three enqueued back to back");
        std::vector<int64_t> v(n);
        const int s2 = ++salt;
        for (size_t i = 0; i < n; i++) v[i] = in_i64(rank, i, s2);
        (void)xmpi_memcpy(c, R.recv, v.data(), n * 8);
        CHECK(xmpi_allreduce_repeat(c, R.recv, R.recv, n, XMPI_I64, XMPI_SUM, XMPI_ALGO_RING_PUSH, 2));  // in place: sum, then N x sum
        R.download(n * 8);
        for (size_t i = 0; i < n; i++) {
          int64_t want = 0;
          for (int r = 0; r < size; r++) want += in_i64(r, i, s2);
          if (((const int64_t*)R.host.data())[i] != want * size) {
            fprintf(stderr, "rank %d: ring kernel, push, in place, two enqueued back to back: element %zu\n", rank, i);
            g_bad.fetch_add(1);
            break;
          }
        }
      }
      if (!dev) allreduce_case(R, 40001, XMPI_ALGO_DIRECT, ++salt, "direct step table");
      if (dev && xmpi_get_param(c, "dsync_sched_launches") <= 0) {
        fprintf(stderr, "rank %d: the stepped kernels never ran\n", rank);
        g_bad.fetch_add(1);
      }
    }
    // ---- broadcast / reduce: binary tree kernels and the fold --------------------------------------------------------------------
    if (wants("bcast") || wants("reduce")) {
      CHECK(xmpi_set_param(c, "tree_piece_bytes", 4096));
      for (int algo : {(int)XMPI_ALGO_TREE, (int)XMPI_ALGO_TREE_PUSH, (int)XMPI_ALGO_AUTO})
        for (int root : {0, size / 2, size - 1}) {
          const size_t n = 4099;
          R.fill_i64(n, ++salt);
          if (rank == root) (void)xmpi_memcpy(c, R.recv, R.send, n * 8);
          else (void)xmpi_memset(c, R.recv, 0xEE, n * 8);
          CHECK(xmpi_bcast(c, R.recv, n, XMPI_I64, root, algo));
          R.download(n * 8);
          for (size_t i = 0; i < n; i++)
            if (((const int64_t*)R.host.data())[i] != in_i64(root, i, salt)) {
              fprintf(stderr, "rank %d: bcast algo %d from %d: element %zu\n", rank, algo, root, i);
              g_bad.fetch_add(1);
              break;
            }
          R.fill_i64(n, ++salt);
          CHECK(xmpi_reduce(c, R.send, R.recv, n, XMPI_I64, XMPI_SUM, root, algo));
          if (rank == root) (void)R.expect_sum_i64(n, salt, algo == XMPI_ALGO_TREE ? "tree reduce" : algo == XMPI_ALGO_TREE_PUSH ? "tree reduce, push" : "reduce");
        }
    }
    // ---- allgather: ring kernel and the fold ------------------------------------------------------------------------------------------
    if (wants("allgather")) {
      for (int algo : {(int)XMPI_ALGO_RING, (int)XMPI_ALGO_RING_PUSH, (int)XMPI_ALGO_AUTO}) {
        const size_t n = 2051;
        R.fill_i64(n, ++salt);
        CHECK(xmpi_allgather(c, R.send, R.recv, n, XMPI_I64, algo));
        R.download(n * 8 * (size_t)size);
        for (int r = 0; r < size; r++)
          for (size_t i = 0; i < n; i++)
            if (((const int64_t*)R.host.data())[(size_t)r * n + i] != in_i64(r, i, salt)) {
              fprintf(stderr, "rank %d: allgather algo %d: block %d element %zu\n", rank, algo, r, i);
              g_bad.fetch_add(1);
              r = size;
              break;
            }
      }
    }
    // ---- a seeded random walk over the forms (DEVSIM_WALK=<steps>, default 40): what one form leaves behind on the never-cleared flag
    // page -- epochs, slot parities, step words, tickets -- for the next, under the sanitizer.  Every rank draws the same sequence.
    if (wants("walk") && dev) {
      uint64_t x = 0x9E3779B97F4A7C15ull * (uint64_t)(round + 1) + (uint64_t)(getenv("DEVSIM_FUZZ") ? atol(getenv("DEVSIM_FUZZ")) : 0);
      auto draw = [&](uint64_t n) {
        x ^= x << 13;
        x ^= x >> 7;
        x ^= x << 17;
        return (size_t)(x % n);
      };
      const int steps = getenv("DEVSIM_WALK") ? atoi(getenv("DEVSIM_WALK")) : 40;
      void* ws = xmpi_stream_create(c);
      const size_t sizes[] = {1, 17, 300, 2051, 4099, 20011};
      for (int k = 0; k < steps && g_bad.load() == 0; k++) {
        const size_t form = draw(15), n = sizes[draw(6)];
        CHECK(xmpi_set_param(c, "dsync_split_bytes", form == 1 || form == 2 ? 1 : 0));
        CHECK(xmpi_set_param(c, "body_sys", form == 2 ? 1 : 0));
        CHECK(xmpi_set_param(c, "ll_bytes", form == 3 ? xmpi_get_param(c, "ll_max_bytes") : 0));
        switch (form) {
          case 0: allreduce_case(R, n, XMPI_ALGO_ZCOPY, ++salt, "walk: fold"); break;
          case 1: allreduce_case(R, n, XMPI_ALGO_ZCOPY, ++salt, "walk: split"); break;
          case 2: allreduce_case(R, n, XMPI_ALGO_ZCOPY, ++salt, "walk: split, system scope"); break;
          case 3: allreduce_case(R, n > 4096 ? 257 : n, XMPI_ALGO_LL, ++salt, "walk: LL"); break;
          case 4: allreduce_case(R, n, XMPI_ALGO_RING, ++salt, "walk: ring kernel"); break;
          case 5: allreduce_case(R, n, XMPI_ALGO_RHD, ++salt, "walk: halving kernel"); break;
          case 6: allreduce_case(R, n, XMPI_ALGO_ZPUSH, ++salt, "walk: push-only"); break;
          case 12: allreduce_case(R, n, XMPI_ALGO_RING_PUSH, ++salt, "walk: ring kernel, push"); break;
          case 13: allreduce_case(R, n, XMPI_ALGO_RHD_PUSH, ++salt, "walk: halving kernel, push"); break;
          case 14: allreduce_in_place_case(R, n, draw(2) ? XMPI_ALGO_RING_PUSH : XMPI_ALGO_RHD_PUSH, ++salt, "walk: push form in place"); break;
          case 7:
          case 8: {
            const int root = (int)draw((uint64_t)size), algo = form == 7 ? (draw(2) ? (int)XMPI_ALGO_TREE : (int)XMPI_ALGO_TREE_PUSH) : (int)XMPI_ALGO_AUTO;
            R.fill_i64(n, ++salt);
            CHECK(xmpi_reduce(c, R.send, R.recv, n, XMPI_I64, XMPI_SUM, root, algo));
            if (rank == root) (void)R.expect_sum_i64(n, salt, "walk: reduce");
            break;
          }
          case 9: {
            const int root = (int)draw((uint64_t)size), pick = (int)draw(3);
            const int algo = pick == 0 ? (int)XMPI_ALGO_TREE : pick == 1 ? (int)XMPI_ALGO_TREE_PUSH : (int)XMPI_ALGO_AUTO;
            R.fill_i64(n, ++salt);
            if (rank == root) (void)xmpi_memcpy(c, R.recv, R.send, n * 8);
            else (void)xmpi_memset(c, R.recv, 0xEE, n * 8);
            CHECK(xmpi_bcast(c, R.recv, n, XMPI_I64, root, algo));
            R.download(n * 8);
            for (size_t i = 0; i < n; i++)
              if (((const int64_t*)R.host.data())[i] != in_i64(root, i, salt)) {
                fprintf(stderr, "rank %d: walk: bcast algo %d from %d: element %zu\n", rank, algo, root, i);
                g_bad.fetch_add(1);
                break;
              }
            break;
          }
          case 10: {
            const int pick = (int)draw(3);
            const int algo = pick == 0 ? (int)XMPI_ALGO_RING : pick == 1 ? (int)XMPI_ALGO_RING_PUSH : (int)XMPI_ALGO_AUTO;
            const size_t m = n > 4099 ? 4099 : n;
            R.fill_i64(m, ++salt);
            CHECK(xmpi_allgather(c, R.send, R.recv, m, XMPI_I64, algo));
            R.download(m * 8 * (size_t)size);
            for (int r = 0; r < size; r++)
              for (size_t i = 0; i < m; i++)
                if (((const int64_t*)R.host.data())[(size_t)r * m + i] != in_i64(r, i, salt)) {
                  fprintf(stderr, "rank %d: walk: allgather algo %d: block %d element %zu\n", rank, algo, r, i);
                  g_bad.fetch_add(1);
                  r = size;
                  break;
                }
            break;
          }
          default: {  // two collectives back to back on a stream, the second reading what the first wrote
            R.fill_i64(n, ++salt);
            CHECK(xmpi_allreduce_on_stream(c, R.send, R.recv, n, XMPI_I64, XMPI_SUM, ws));
            CHECK(xmpi_allreduce_on_stream(c, R.recv, R.recv, n, XMPI_I64, XMPI_MAX, ws));  // (everybody holds the same: MAX leaves it)
            CHECK(xmpi_stream_sync(c, ws));
            (void)R.expect_sum_i64(n, salt, "walk: two on a stream");
            break;
          }
        }
      }
      CHECK(xmpi_set_param(c, "dsync_split_bytes", 4 << 20));
      CHECK(xmpi_set_param(c, "body_sys", 0));
      CHECK(xmpi_stream_destroy(c, ws));
    }
    // ---- stream-ordered collectives on a stream of the caller's, and a captured graph replayed ---------------------------------
    if ((wants("stream") || wants("graph")) && dev) {
      void* s = xmpi_stream_create(c);
      if (!s) {
        fprintf(stderr, "rank %d: xmpi_stream_create: %s\n", rank, xmpi_last_error());
        g_bad.fetch_add(1);
        return;
      }
      for (size_t n : {(size_t)33, (size_t)4099}) {
        R.fill_i64(n, ++salt);
        CHECK(xmpi_allreduce_on_stream(c, R.send, R.recv, n, XMPI_I64, XMPI_SUM, s));
        CHECK(xmpi_stream_sync(c, s));
        (void)R.expect_sum_i64(n, salt, "allreduce on a stream");
      }
      if (wants("graph")) {
        const size_t n = 4099;
        R.fill_i64(n, ++salt);
        void* g = nullptr;
        CHECK(xmpi_graph_begin(c, s));
        CHECK(xmpi_allreduce_on_stream(c, R.send, R.recv, n, XMPI_I64, XMPI_SUM, s));
        CHECK(xmpi_graph_end(c, s, &g));
        for (int k = 0; k < 3; k++) {
          (void)xmpi_memset(c, R.recv, 0xEE, n * 8);
          CHECK(xmpi_graph_launch(c, g, s));
          CHECK(xmpi_stream_sync(c, s));
          (void)R.expect_sum_i64(n, salt, "graph replay");
        }
        CHECK(xmpi_graph_destroy(c, g));
      }
      CHECK(xmpi_stream_destroy(c, s));
    }
    // ---- stream-ordered Send / Receive: a ring on ONE stream per rank (even ranks send first) ---------------------------------------
    if (wants("p2p_stream") && size > 1 && dev) {
      void* s = xmpi_stream_create(c);
      const int next = (rank + 1) % size, prev = (rank + size - 1) % size;
      for (size_t n : {(size_t)1, (size_t)300, (size_t)9001}) {  // (9001 x 8 bytes: a receive kernel of several blocks)
        R.fill_i64(n, ++salt);
        (void)xmpi_memset(c, R.recv, 0xEE, n * 8);
        if (rank % 2 == 0) {
          CHECK(xmpi_send_on_stream(c, R.send, n, XMPI_I64, next, 40 + (int)(n % 7), s));
          CHECK(xmpi_recv_on_stream(c, R.recv, n, XMPI_I64, prev, 40 + (int)(n % 7), s));
        } else {
          CHECK(xmpi_recv_on_stream(c, R.recv, n, XMPI_I64, prev, 40 + (int)(n % 7), s));
          CHECK(xmpi_send_on_stream(c, R.send, n, XMPI_I64, next, 40 + (int)(n % 7), s));
        }
        // (an odd ring has two neighbours that both send first; the cycle is still broken by every odd rank)
        CHECK(xmpi_stream_sync(c, s));
        R.download(n * 8);
        for (size_t i = 0; i < n; i++)
          if (((const int64_t*)R.host.data())[i] != in_i64(prev, i, salt)) {
            fprintf(stderr, "rank %d: stream-ordered receive of %zu: element %zu\n", rank, n, i);
            g_bad.fetch_add(1);
            break;
          }
      }
      CHECK(xmpi_stream_destroy(c, s));
    }
    // ---- blocking Send / Receive of device payloads: the copy-and-ack kernel and the receive agent ----------------------------------
    if (wants("p2p_block") && size > 1) {
      const int next = (rank + 1) % size, prev = (rank + size - 1) % size;
      for (size_t n : {(size_t)1, (size_t)1000, (size_t)20000, (size_t)100000}) {  // 8 B ... 800 KB: agent alone / wide / pull kernel
        R.fill_i64(n, ++salt);
        (void)xmpi_memset(c, R.recv, 0xEE, n * 8);
        int rs = 0;
        std::thread tx([&] { rs = xmpi_send(c, R.send, n, XMPI_I64, next, 7); });
        size_t got = 0;
        const int rr = xmpi_recv(c, R.recv, n, XMPI_I64, prev, 7, &got);
        tx.join();
        if (rs != XMPI_OK || rr != XMPI_OK || got != n) {
          fprintf(stderr, "rank %d: blocking send / receive of %zu: %d / %d, %zu elements (%s)\n", rank, n, rs, rr, got, xmpi_last_error());
          g_bad.fetch_add(1);
          break;
        }
        R.download(n * 8);
        for (size_t i = 0; i < n; i++)
          if (((const int64_t*)R.host.data())[i] != in_i64(prev, i, salt)) {
            fprintf(stderr, "rank %d: blocking receive of %zu: element %zu\n", rank, n, i);
            g_bad.fetch_add(1);
            break;
          }
      }
    }
    CHECK(xmpi_barrier(c));
  }
  if (rank == 0 && g_bad.load() == 0)
    printf("devsim driver: epochs %ld, fold / split / stepped / LL launches %ld / %ld / %ld / %ld, XCD masks %#lx %#lx, agent served %ld\n",
           xmpi_get_param(c, "dsync_epoch"), xmpi_get_param(c, "dsync_launches"), xmpi_get_param(c, "dsync_split_launches"),
           xmpi_get_param(c, "dsync_sched_launches"), xmpi_get_param(c, "dsync_ll_launches"), xmpi_get_param(c, "xcd_meet_mask"),
           xmpi_get_param(c, "xcd_done_mask"), xmpi_get_param(c, "p2p_agent_served"));
  (void)xmpi_free(c, R.send);
  (void)xmpi_free(c, R.recv);
  CHECK(xmpi_finalize(c));
}

}  // namespace

int main(int argc, char** argv) {
  int a = 1;
  if (argc > 1 && std::string(argv[1]) == "--seed-race") {
    g_seed_race = true;
    a = 2;
  } else if (argc > 1 && std::string(argv[1]) == "--shared") {
    g_shared = true;
    a = 2;
  }
  const int size = argc > a ? atoi(argv[a]) : 2, rounds = argc > a + 1 ? atoi(argv[a + 1]) : 1;
  for (int k = a + 2; k < argc; k++) g_only.insert(argv[k]);
  if (size < 2 || size > 16) return 2;  // (kMaxRanks)
  setenv("XMPI_CTL_SHARE_MAPPING", "1", 1);
  setenv("DEVSIM_DEVICES", g_shared ? "1" : std::to_string(size).c_str(), 1);
  setenv("XMPI_TIMEOUT_S", "120", 0);
  setenv("XMPI_HOST_LANES", "0", 0);
  const std::string key = "devsim-" + std::to_string((int)getpid());
  std::vector<std::thread> ranks;
  for (int r = 0; r < size; r++) ranks.emplace_back(rank_main, key, r, size, rounds);
  for (auto& t : ranks) t.join();
  if (g_bad.load()) {
    fprintf(stderr, "devsim driver: %d failure(s)\n", g_bad.load());
    return 1;
  }
  printf("devsim driver ok: %d ranks as threads on %d virtual device%s, %d round(s)\n", size, g_shared ? 1 : size, g_shared ? "" : "s", rounds);
  return 0;
}
