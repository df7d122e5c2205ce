// ll.hip -- the LL ("low latency") small collectives: the payload travels as {data, flag} lines pushed into the peers'
// flag allocations, every rank folds locally in rank order.  One one-way hop per collective; protocol and slot
// layout: kernels.h (kLLOff ...).
//
// What it bypasses: the rendezvous of the zero-copy forms (kdev.h dsync_begin / dsync_end) -- announce -> wait ->
// translate -> remote READ -> done exchange: two round trips between peers per collective whatever its size.  The
// reference's per-message cost is one message and one ack (network.go:562-571); a reference user's allreduce is N-1
// of those per rank (helloworld.go:53-81) plus a host-side sum in rank order -- that sum, bit for bit, is what the fold
// below computes (oracle/xmpi_oracle.c reduce_ranks).
//
// Ordering needs nothing but 8-byte atomicity of stores: a half-line {4 data bytes, flag} is written by ONE 8-byte
// system-scope store and read by a system-scope load past the caches, so a reader sees either the old half (old flag)
// or the new one, never a mix.  No fence, no cache maintenance, no second hop.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "kdev.h"
#include "kernels.h"

namespace xmpi {
namespace {

#define XMPI_LAUNCH(kern, grid, block, stream, es, ee, ...)                                  \
  do {                                                                                       \
    if ((es) || (ee)) hipExtLaunchKernelGGL(kern, grid, block, 0, stream, es, ee, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kern, grid, block, 0, stream, __VA_ARGS__);                      \
  } while (0)

__device__ __forceinline__ char* ll_slot(DsyncPage* page, int src, uint32_t parity) {
  return reinterpret_cast<char*>(page) + kLLOff + ((size_t)src * 2 + parity) * kLLSlotBytes;
}
__device__ __forceinline__ uint64_t* ll_here(DsyncPage* page, int src) {
  return reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(page) + kLLHereOff + (size_t)src * 64);
}

// one line = 8 payload bytes: two 8-byte stores, each carrying its own copy of the flag
__device__ __forceinline__ void ll_store(char* line, uint64_t data, uint32_t flag) {
  st_sys64(reinterpret_cast<uint64_t*>(line), (uint64_t)(uint32_t)data | ((uint64_t)flag << 32));
  st_sys64(reinterpret_cast<uint64_t*>(line) + 1, (data >> 32) | ((uint64_t)flag << 32));
}

// `valid` (1..8) bytes of this rank's own buffer, the rest zero.  SYS: past the caches, both ways (system-scope loads, stores
// written through) -- for the agent, which lingers across calls: its CU's L1 and its XCD's L2 may hold the send buffer as it was a
// call ago, and what it stores must be in memory before it answers.  A launched kernel has its boundaries for that; the agent
// would need an L2 invalidate before and a write-back after every call (fences at system scope -- its first form had them: the
// two ways measured within noise of each other at 1-4 KiB; this one asks nothing of what else the L2 holds).
// (tests/devsim: plain accesses either way -- under its sanitizer the buffers' bytes must stay DATA, so that a reader no chain of
// flag words has ordered behind the agent's stores is a reported race; what a cache holds is not modelled there.)
#ifdef XMPI_DEVSIM
#define XMPI_LL_SYS(SYS) false
#else
#define XMPI_LL_SYS(SYS) SYS
#endif
template <bool SYS>
__device__ __forceinline__ uint64_t load8(const char* p, uint32_t valid) {
  if (valid == 8 && (reinterpret_cast<uintptr_t>(p) & 7u) == 0) {
    if constexpr (XMPI_LL_SYS(SYS)) return ld_sys64(reinterpret_cast<const uint64_t*>(p));
    else return *reinterpret_cast<const uint64_t*>(p);
  }
  uint64_t v = 0;
  for (uint32_t i = 0; i < valid; i++) {
    uint8_t b;
    if constexpr (XMPI_LL_SYS(SYS)) b = __hip_atomic_load(reinterpret_cast<const uint8_t*>(p) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else b = (uint8_t)p[i];
    v |= (uint64_t)b << (8 * i);
  }
  return v;
}
template <bool SYS>
__device__ __forceinline__ void store8(char* p, uint64_t v, uint32_t valid) {
  if (valid == 8 && (reinterpret_cast<uintptr_t>(p) & 7u) == 0) {
    if constexpr (XMPI_LL_SYS(SYS)) st_sys64(reinterpret_cast<uint64_t*>(p), v);
    else *reinterpret_cast<uint64_t*>(p) = v;
    return;
  }
  for (uint32_t i = 0; i < valid; i++) {
    if constexpr (XMPI_LL_SYS(SYS)) __hip_atomic_store(reinterpret_cast<uint8_t*>(p) + i, (uint8_t)(v >> (8 * i)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else p[i] = (char)(v >> (8 * i));
  }
}

template <typename T, int OP>
__device__ __forceinline__ uint64_t combine8(uint64_t a, uint64_t b) {
  constexpr int N = 8 / sizeof(T);
  union {
    uint64_t u;
    T e[N];
  } x, y, r;
  x.u = a;
  y.u = b;
#pragma unroll
  for (int i = 0; i < N; i++) r.e[i] = combine_any<T, OP>(x.e[i], y.e[i]);
  return r.u;
}

// the ranks' lines in rank order: ((x0 op x1) op x2) ... -- the host-side sum of a reference user (helloworld.go:53-81), bit for bit.
// x(p) = rank p's line (never asked for p == me)
template <typename T, int OP, typename X>
__device__ __forceinline__ uint64_t fold_ranks(X x, uint64_t mine8, int me, int n) {
  uint64_t acc = me == 0 ? mine8 : x(0);
#pragma unroll
  for (int p = 1; p < kDsyncRanks; p++)
    if (p < n) acc = combine8<T, OP>(acc, p == me ? mine8 : x(p));
  return acc;
}
// dtype and operation fixed at compile time (the launched kernels: one instantiation each) ...
template <typename T, int OP>
struct LLFoldStatic {
  template <typename X>
  __device__ __forceinline__ uint64_t operator()(X x, uint64_t mine8, int me, int n) const {
    return fold_ranks<T, OP>(x, mine8, me, n);
  }
};
// ... or read from the command (the agent: one kernel for everything; one uniform branch per line)
struct LLFoldRuntime {
  int dtype, op;
  template <typename T, typename X>
  __device__ __forceinline__ uint64_t with(X x, uint64_t mine8, int me, int n) const {
    switch (op) {
      case OP_SUM: return fold_ranks<T, OP_SUM>(x, mine8, me, n);
      case OP_PROD: return fold_ranks<T, OP_PROD>(x, mine8, me, n);
      case OP_MIN: return fold_ranks<T, OP_MIN>(x, mine8, me, n);
      default: return fold_ranks<T, OP_MAX>(x, mine8, me, n);
    }
  }
  template <typename X>
  __device__ __forceinline__ uint64_t operator()(X x, uint64_t mine8, int me, int n) const {
    switch (dtype) {
      case DT_U8: return with<uint8_t>(x, mine8, me, n);
      case DT_I32: return with<int32_t>(x, mine8, me, n);
      case DT_I64: return with<int64_t>(x, mine8, me, n);
      case DT_F16: return with<_Float16>(x, mine8, me, n);
      case DT_F32: return with<float>(x, mine8, me, n);
      case DT_F64: return with<double>(x, mine8, me, n);
      default: return with<bf16_t>(x, mine8, me, n);
    }
  }
};

struct LLShared {
  uint64_t epoch;
  uint32_t fail, last;
};

// what one call asks for (the launched kernels: fields of their argument; the agent: its command)
struct LLCall {
  const void* send;
  void* recv;
  uint64_t bytes;
  int32_t coll, root;
};

// `a` below: the fields of DsyncLLArgs that belong to the COMMUNICATOR (page, me, n, epoch_floor, abort_word, spin_limit)

__device__ __forceinline__ void ll_begin(const DsyncLLArgs& a, LLShared& sh) {
  if (threadIdx.x == 0) {
    // the epoch is counted on the device, as in kdev.h dsync_begin: a captured launch that is replayed keeps counting
    const uint64_t seen = ld_sys64(&a.page[a.me]->epoch_now);
    sh.epoch = (seen > a.epoch_floor ? seen : a.epoch_floor) + 1;
    sh.fail = DSYNC_OK;
  }
  __syncthreads();
}

// "I have started epoch e", to every peer (collectives whose data does not reach everybody: broadcast, reduce)
__device__ __forceinline__ void ll_say_here(const DsyncLLArgs& a, uint64_t epoch) {
  const int t = threadIdx.x;
  if (blockIdx.x == 0 && t < a.n && t != a.me) st_sys64(ll_here(a.page[t], a.me), epoch);
}
__device__ __forceinline__ void ll_wait_here(const DsyncLLArgs& a, LLShared& sh) {
  const int t = threadIdx.x;
  if (blockIdx.x == 0 && t < a.n && t != a.me) {
    const uint32_t why = spin_until(ll_here(a.page[a.me], t), sh.epoch, a.abort_word, a.spin_limit);
    if (why != DSYNC_OK) atomicMax(&sh.fail, why);
  }
}

// Lines `idx` of the slots of the ranks in `want` (bit per rank), polled until they carry this epoch's flag; out[p] = the
// 8 payload bytes of rank p's line.  All loads of a round are in flight together.  false: gave up (sh.fail says why).
__device__ __forceinline__ bool ll_gather(const DsyncLLArgs& a, LLShared& sh, uint32_t want, uint32_t parity, uint32_t flag,
                                          size_t idx, uint64_t (&out)[kDsyncRanks]) {
  DsyncPage* mine = a.page[a.me];
  pack_t v[kDsyncRanks];
#pragma unroll
  for (int p = 0; p < kDsyncRanks; p++) v[p] = pack_t{0u, 0u, 0u, 0u};
  uint32_t pending = want;
  uint64_t t0 = 0;
  for (uint32_t k = 0; pending; k++) {
#pragma unroll
    for (int p = 0; p < kDsyncRanks; p++)
      if (pending >> p & 1u) ld_sys128_issue(v[p], reinterpret_cast<const pack_t*>(ll_slot(mine, p, parity) + idx * 16));
    XMPI_DRAIN();
#pragma unroll
    for (int p = 0; p < kDsyncRanks; p += 4)  // (the loads' results may be used from here on)
      XMPI_REGS_DEFINED4(v[p], v[p + 1], v[p + 2], v[p + 3]);
#pragma unroll
    for (int p = 0; p < kDsyncRanks; p++)
      if ((pending >> p & 1u) && v[p].y == flag && v[p].w == flag) {
        out[p] = ((uint64_t)v[p].z << 32) | v[p].x;
        pending &= ~(1u << p);
      }
    if (!pending) break;
    if (k == 0) t0 = wall_clock64();
    __builtin_amdgcn_s_sleep(1);
    if ((k & 127u) == 127u) {
      uint32_t why = DSYNC_OK;
      if (a.abort_word && __hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) why = DSYNC_ABORTED;
      else if (a.spin_limit && wall_clock64() - t0 > a.spin_limit) why = DSYNC_TIMEOUT;
      if (why != DSYNC_OK) {
        atomicMax(&sh.fail, why);
        return false;
      }
    }
  }
  return true;
}

// The launched kernels' line (payload bytes [8 idx, 8 idx + 8)), one per lane, in two halves: PUSH this rank's bytes to whoever
// needs them, COLLECT the peers'.  Nothing is kept between the halves: a line's own bytes are read again (in place works: a line
// is overwritten by its own collect only).  (The agent's lanes own several lines each: ll_agent_collective below.)

// allreduce / reduce
__device__ __forceinline__ void ll_reduce_push(const DsyncLLArgs& a, const LLCall& q, uint32_t parity, uint32_t flag, size_t idx) {
  const int me = a.me, n = a.n;
  if (q.coll != LL_ALLREDUCE && me == q.root) return;
  const uint32_t valid = q.bytes - idx * 8 >= 8 ? 8u : (uint32_t)(q.bytes - idx * 8);
  const uint64_t mine8 = load8<false>(reinterpret_cast<const char*>(q.send) + idx * 8, valid);
  if (q.coll == LL_ALLREDUCE) {
#pragma unroll
    for (int d = 1; d < kDsyncRanks; d++) {  // (start with the next rank: the ranks do not all hit rank 0's page first)
      const int p = (me + d) % n;
      if (d < n) ll_store(ll_slot(a.page[p], me, parity) + idx * 16, mine8, flag);
    }
  } else {
    ll_store(ll_slot(a.page[q.root], me, parity) + idx * 16, mine8, flag);
  }
}
// ... gather, fold in rank order, store
template <typename F>
__device__ __forceinline__ void ll_reduce_collect(const DsyncLLArgs& a, const LLCall& q, LLShared& sh, uint32_t parity, uint32_t flag,
                                                  size_t idx, F fold) {
  const int me = a.me, n = a.n;
  if (q.coll != LL_ALLREDUCE && me != q.root) return;
  const uint32_t valid = q.bytes - idx * 8 >= 8 ? 8u : (uint32_t)(q.bytes - idx * 8);
  const uint64_t mine8 = load8<false>(reinterpret_cast<const char*>(q.send) + idx * 8, valid);
  uint64_t x[kDsyncRanks];
  const uint32_t everyone = n >= 32 ? 0xffffffffu : ((1u << n) - 1u);
  if (ll_gather(a, sh, everyone & ~(1u << me), parity, flag, idx, x))
    store8<false>(reinterpret_cast<char*>(q.recv) + idx * 8, fold([&x](int p) { return x[p]; }, mine8, me, n), valid);
}

// broadcast / allgather: bytes only
__device__ __forceinline__ void ll_copy_push(const DsyncLLArgs& a, const LLCall& q, uint32_t parity, uint32_t flag, size_t idx) {
  const int me = a.me, n = a.n;
  const bool gather = q.coll == LL_ALLGATHER;
  if (!gather && me != q.root) return;
  const uint32_t valid = q.bytes - idx * 8 >= 8 ? 8u : (uint32_t)(q.bytes - idx * 8);
  const uint64_t mine8 = load8<false>(reinterpret_cast<const char*>(q.send) + idx * 8, valid);
#pragma unroll
  for (int d = 1; d < kDsyncRanks; d++) {
    const int p = (me + d) % n;
    if (d < n) ll_store(ll_slot(a.page[p], me, parity) + idx * 16, mine8, flag);
  }
  if (gather) store8<false>(reinterpret_cast<char*>(q.recv) + (size_t)me * q.bytes + idx * 8, mine8, valid);
}
__device__ __forceinline__ void ll_copy_collect(const DsyncLLArgs& a, const LLCall& q, LLShared& sh, uint32_t parity, uint32_t flag,
                                                size_t idx) {
  const int me = a.me, n = a.n;
  const uint32_t valid = q.bytes - idx * 8 >= 8 ? 8u : (uint32_t)(q.bytes - idx * 8);
  uint64_t x[kDsyncRanks];
  if (q.coll == LL_ALLGATHER) {
    const uint32_t everyone = n >= 32 ? 0xffffffffu : ((1u << n) - 1u);
    if (ll_gather(a, sh, everyone & ~(1u << me), parity, flag, idx, x)) {
#pragma unroll
      for (int p = 0; p < kDsyncRanks; p++)
        if (p < n && p != me) store8<false>(reinterpret_cast<char*>(q.recv) + (size_t)p * q.bytes + idx * 8, x[p], valid);
    }
  } else if (me != q.root) {
    if (ll_gather(a, sh, 1u << q.root, parity, flag, idx, x)) {
      uint64_t got = 0;
#pragma unroll
      for (int p = 0; p < kDsyncRanks; p++)
        if (p == q.root) got = x[p];
      store8<false>(reinterpret_cast<char*>(q.recv) + idx * 8, got, valid);
    }
  }
}

// Every wave's stores have left; the block that finishes last advances the epoch and tells the host.  No exchange with the
// peers: nobody reads this rank's buffers, and the slots are safe by the parity argument (kernels.h).
__device__ __forceinline__ void ll_end(const DsyncLLArgs& a, LLShared& sh) {
  DsyncPage* mine = a.page[a.me];
  XMPI_DRAIN();
  __syncthreads();
  if (threadIdx.x != 0) return;
  if (gridDim.x > 1) {
    if (sh.fail != DSYNC_OK) __hip_atomic_fetch_max(&mine->failword, sh.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // (system scope: the result may lie in pinned host memory)
    if (__hip_atomic_fetch_add(&mine->ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) != gridDim.x - 1) return;
    __hip_atomic_store(&mine->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t f = __hip_atomic_exchange(&mine->failword, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (f > sh.fail) sh.fail = f;
  }
  st_sys64(&mine->epoch_now, sh.epoch);
  if (a.host_epoch) st_sys64(a.host_epoch, sh.epoch);
  if (sh.fail != DSYNC_OK && a.status) __hip_atomic_store(a.status, sh.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  // the blocking caller's word, last (release: the result is where the caller -- host or a later kernel -- will read it)
  if (a.host_done) __hip_atomic_store(a.host_done, a.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// allreduce / reduce: thread i owns payload bytes [8 i, 8 i + 8)
template <typename T, int OP>
__global__ __launch_bounds__(kBlock) void ll_reduce_kernel(DsyncLLArgs a) {
  XMPI_SHARED(LLShared, sh);
  ll_begin(a, sh);
  const uint64_t epoch = sh.epoch;
  const uint32_t parity = (uint32_t)(epoch & 1u), flag = (uint32_t)epoch ? (uint32_t)epoch : 1u;
  const LLCall q{a.send, a.recv, a.bytes, a.coll, a.root};
  const bool to_all = a.coll == LL_ALLREDUCE;
  if (!to_all) ll_say_here(a, epoch);
  const size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (idx * 8 < a.bytes) {
    ll_reduce_push(a, q, parity, flag, idx);
    ll_reduce_collect(a, q, sh, parity, flag, idx, LLFoldStatic<T, OP>{});
  }
  if (!to_all) ll_wait_here(a, sh);
  ll_end(a, sh);
}

// broadcast / allgather: bytes only
__global__ __launch_bounds__(kBlock) void ll_copy_kernel(DsyncLLArgs a) {
  XMPI_SHARED(LLShared, sh);
  ll_begin(a, sh);
  const uint64_t epoch = sh.epoch;
  const uint32_t parity = (uint32_t)(epoch & 1u), flag = (uint32_t)epoch ? (uint32_t)epoch : 1u;
  const LLCall q{a.send, a.recv, a.bytes, a.coll, a.root};
  const bool gather = a.coll == LL_ALLGATHER;
  if (!gather) ll_say_here(a, epoch);
  const size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (idx * 8 < a.bytes) {
    ll_copy_push(a, q, parity, flag, idx);
    ll_copy_collect(a, q, sh, parity, flag, idx);
  }
  if (!gather) ll_wait_here(a, sh);
  ll_end(a, sh);
}

// The agent's lanes own up to kLLAgentRounds lines each (lane t: lines t, t + kLLAgentBlock, ...).  What a lane keeps between the
// phases of a collective lives in LDS, indexed by the round: its own lines, and the peers' lines of the pair of rounds in hand.
constexpr int kLLAgentRounds = (int)(kLLMaxPayload / 8 / kLLAgentBlock);
static_assert((size_t)kLLAgentRounds * kLLAgentBlock * 8 == kLLMaxPayload && kLLAgentRounds % 2 == 0, "the agent's lanes cover a full slot in whole pairs of rounds");
struct LLAgentLds {
  uint64_t mine[kLLAgentRounds][kLLAgentBlock];  // 32 KiB
  uint64_t got[2][kDsyncRanks - 1][kLLAgentBlock];  // 120 KiB: rank p's line at [p < me ? p : p - 1] (nobody gathers its own)
};
static_assert(sizeof(LLAgentLds) + 256 <= 160u * 1024, "the agent's block has one CU's LDS to itself");

// ll_gather for TWO lines of a lane at once (every wait for loads past the caches costs about a microsecond, so the lines are waited
// for in pairs); the payloads go to lds.got[0 / 1][rank, own left out][lane]
__device__ __forceinline__ bool ll_gather2(const DsyncLLArgs& a, LLShared& sh, uint32_t want, int base, uint32_t parity, uint32_t flag,
                                           size_t idx0, size_t idx1, bool two, LLAgentLds& lds) {
  // (eight ranks at a time, `base` = the first: two lines of sixteen ranks in flight would be 128 registers of loads alone)
  constexpr int G = 8;
  static_assert(kDsyncRanks % G == 0, "rank groups");
  DsyncPage* mine = a.page[a.me];
  const int t = threadIdx.x;
  pack_t v0[G], v1[G];
#pragma unroll
  for (int p = 0; p < G; p++) v0[p] = v1[p] = pack_t{0u, 0u, 0u, 0u};
  uint32_t pend0 = (want >> base) & ((1u << G) - 1u), pend1 = two ? pend0 : 0u;
  uint64_t t0 = 0;
  for (uint32_t k = 0; pend0 | pend1; k++) {
#pragma unroll
    for (int p = 0; p < G; p++)
      if (pend0 >> p & 1u) ld_sys128_issue(v0[p], reinterpret_cast<const pack_t*>(ll_slot(mine, base + p, parity) + idx0 * 16));
#pragma unroll
    for (int p = 0; p < G; p++)
      if (pend1 >> p & 1u) ld_sys128_issue(v1[p], reinterpret_cast<const pack_t*>(ll_slot(mine, base + p, parity) + idx1 * 16));
    XMPI_DRAIN();
#pragma unroll
    for (int p = 0; p < G; p += 4) {  // (the loads' results may be used from here on)
      XMPI_REGS_DEFINED4(v0[p], v0[p + 1], v0[p + 2], v0[p + 3]);
      XMPI_REGS_DEFINED4(v1[p], v1[p + 1], v1[p + 2], v1[p + 3]);
    }
#pragma unroll
    for (int p = 0; p < G; p++) {
      const int rank = base + p, at = rank < a.me ? rank : rank - 1;
      if ((pend0 >> p & 1u) && v0[p].y == flag && v0[p].w == flag) {
        lds.got[0][at][t] = ((uint64_t)v0[p].z << 32) | v0[p].x;
        pend0 &= ~(1u << p);
      }
      if ((pend1 >> p & 1u) && v1[p].y == flag && v1[p].w == flag) {
        lds.got[1][at][t] = ((uint64_t)v1[p].z << 32) | v1[p].x;
        pend1 &= ~(1u << p);
      }
    }
    if (!(pend0 | pend1)) break;
    if (k == 0) t0 = wall_clock64();
    __builtin_amdgcn_s_sleep(1);
    if ((k & 127u) == 127u) {
      uint32_t why = DSYNC_OK;
      if (a.abort_word && __hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) why = DSYNC_ABORTED;
      else if (a.spin_limit && wall_clock64() - t0 > a.spin_limit) why = DSYNC_TIMEOUT;
      if (why != DSYNC_OK) {
        atomicMax(&sh.fail, why);
        return false;
      }
    }
  }
  return true;
}

__device__ __forceinline__ uint32_t ll_valid(uint64_t bytes, size_t idx) { return bytes - idx * 8 >= 8 ? 8u : (uint32_t)(bytes - idx * 8); }

// One collective by the agent's one block.  Every dependent trip to memory past the caches is about a microsecond, so they are
// overlapped: (A) ALL of the lane's own lines are loaded at once and kept, (B) all are pushed, (C) the peers' lines are waited for
// two rounds at a time.  (One line after the other, own bytes read twice, a round more cost ~2.5 us: 8 KiB took 10.3 us, 16 KiB 16.0
// against 12.2 launched; this way a PAIR of rounds more costs ~1.7 us: 8 KiB 8.8, 16 KiB 12.1.  Keeping the results back and
// storing them all at the end -- so that no wait for loads also waits for the last pair's written-through stores -- measured
// worse, not better: 10.3 / 14.8 on a box whose launched figures were 9 % up.)
template <typename F>
__device__ __forceinline__ void ll_agent_collective(const DsyncLLArgs& a, LLShared& sh, const LLCall& q, uint32_t parity, uint32_t flag,
                                                    F fold, LLAgentLds& lds) {
  const int t = threadIdx.x, me = a.me, n = a.n;
  const size_t nlines = (size_t)((q.bytes + 7) / 8);
  const char* send = reinterpret_cast<const char*>(q.send);
  char* recv = reinterpret_cast<char*>(q.recv);
  const bool reduces = q.coll == LL_ALLREDUCE || q.coll == LL_REDUCE;
  const bool pushes = q.coll == LL_ALLREDUCE || q.coll == LL_ALLGATHER || (q.coll == LL_REDUCE && me != q.root) || (q.coll == LL_BCAST && me == q.root);
  const uint32_t everyone = n >= 32 ? 0xffffffffu : ((1u << n) - 1u);
  const uint32_t want = (q.coll == LL_ALLREDUCE || q.coll == LL_ALLGATHER || (q.coll == LL_REDUCE && me == q.root)) ? (everyone & ~(1u << me))
                        : (q.coll == LL_BCAST && me != q.root)                                                       ? (1u << q.root)
                                                                                                                     : 0u;
  // (A) this lane's own lines, all loads in flight together
  if (pushes || (reduces && want)) {
    uint64_t mine[kLLAgentRounds];
#pragma unroll
    for (int r = 0; r < kLLAgentRounds; r++) {
      const size_t idx = (size_t)t + (size_t)r * kLLAgentBlock;
      mine[r] = idx < nlines ? load8<true>(send + idx * 8, ll_valid(q.bytes, idx)) : 0;
    }
#pragma unroll
    for (int r = 0; r < kLLAgentRounds; r++) lds.mine[r][t] = mine[r];  // (read back by this lane only: no barrier)
  }
  // (B) out they go
  if (pushes) {
    for (int r = 0; r < kLLAgentRounds; r++) {
      const size_t idx = (size_t)t + (size_t)r * kLLAgentBlock;
      if (idx >= nlines) break;
      const uint64_t mine8 = lds.mine[r][t];
      if (q.coll == LL_REDUCE) {
        ll_store(ll_slot(a.page[q.root], me, parity) + idx * 16, mine8, flag);
      } else {
#pragma unroll
        for (int d = 1; d < kDsyncRanks; d++) {  // (start with the next rank: the ranks do not all hit rank 0's page first)
          const int p = (me + d) % n;
          if (d < n) ll_store(ll_slot(a.page[p], me, parity) + idx * 16, mine8, flag);
        }
      }
      if (q.coll == LL_ALLGATHER) store8<true>(recv + (size_t)me * q.bytes + idx * 8, mine8, ll_valid(q.bytes, idx));
    }
  }
  // (C) the peers' lines, two rounds per wait
  if (!want) return;
  for (int r = 0; r < kLLAgentRounds; r += 2) {
    const size_t idx0 = (size_t)t + (size_t)r * kLLAgentBlock;
    if (idx0 >= nlines) break;
    const bool two = idx0 + kLLAgentBlock < nlines;
    bool ok = true;
    for (int base = 0; base < kDsyncRanks && ok; base += 8)
      if ((want >> base) & 0xffu) ok = ll_gather2(a, sh, want, base, parity, flag, idx0, idx0 + kLLAgentBlock, two, lds);
    if (!ok) return;
    for (int j = 0; j < (two ? 2 : 1); j++) {
      const size_t idx = idx0 + (size_t)j * kLLAgentBlock;
      const uint32_t valid = ll_valid(q.bytes, idx);
      auto x = [&lds, j, t, me](int p) { return lds.got[j][p < me ? p : p - 1][t]; };
      if (reduces) {
        store8<true>(recv + idx * 8, fold(x, lds.mine[r + j][t], me, n), valid);
      } else if (q.coll == LL_ALLGATHER) {
#pragma unroll
        for (int p = 0; p < kDsyncRanks; p++)
          if (p < n && p != me) store8<true>(recv + (size_t)p * q.bytes + idx * 8, x(p), valid);
      } else {
        const uint64_t got = x(q.root);
        store8<true>(recv + idx * 8, got, valid);
      }
    }
  }
}

// The LL AGENT.  A blocking call -- the only kind the reference's API has (mpi.go:47-48) -- is launch + kernel + completion
// word, and the launch is about half of it (1 KiB, 2 processes: 10.9 us blocking against 4.9 us enqueued).  So the kernel that
// served a blocking small collective does not end at once: it watches a command record in pinned host memory for `patience`
// (tens of microseconds), and the host thread of the NEXT blocking small collective writes {send, recv, bytes, what} there
// instead of launching (kernels.h LLAgentArgs; the idea and the record are the receive agent's, sched.hip p2p_agent_kernel).
// One block of kLLAgentBlock lanes: lane t owns lines t, t + 512, ... (dsync.cpp dsync_ll sends payloads up to agent_ll_bytes this
// way; a round of lines costs a load from the flag allocation past the caches, ~2 us -- 4 KiB is one round, and beyond that the
// launched kernel's many blocks are faster: 16 KiB took 24 us with 256 lanes against 11.8 launched).  It waits for
// the PEERS inside a collective, as the launched kernel would, within the same no-progress limit; between collectives it waits
// for the HOST only, and only for its patience.  dtype and operation come with the command: one kernel, the fold chosen by a
// uniform branch per line.
__global__ __launch_bounds__(kLLAgentBlock) void ll_agent_kernel(LLAgentArgs a) {
  XMPI_SHARED(LLShared, sh);
  XMPI_SHARED(uint64_t, s_send);
  XMPI_SHARED(uint64_t, s_recv);
  XMPI_SHARED(uint64_t, s_bytes);
  XMPI_SHARED(uint32_t, s_meta);
  XMPI_SHARED(uint32_t, s_go);
  XMPI_SHARED(LLAgentLds, s_lds);  // 152 KiB: the lanes' own lines of the call in hand, the peers' lines of two rounds
  const int t = threadIdx.x;
  uint64_t seq = a.seq0, prev_epoch = 0;  // (lane 0's: the epoch of the last collective this launch ran)
  for (;;) {
    if (t == 0) {
      uint64_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
      bool have = false;
      const uint64_t t0 = wall_clock64();
      for (;;) {
        pack_t v[2];
        ld_sys128_issue(v[0], reinterpret_cast<const pack_t*>(a.cmd));
        ld_sys128_issue(v[1], reinterpret_cast<const pack_t*>(a.cmd) + 1);
        sys128_wait<2>(v);
        w0 = ((uint64_t)v[0].y << 32) | v[0].x;
        w1 = ((uint64_t)v[0].w << 32) | v[0].z;
        w2 = ((uint64_t)v[1].y << 32) | v[1].x;
        w3 = ((uint64_t)v[1].w << 32) | v[1].z;
        have = (w0 & 3u) != 0 && (w0 >> 24) == (seq & 0xffffffffffull) && (w3 >> 32) == (seq & 0xffffffffull);
        if (have || wall_clock64() - t0 > a.patience_ticks) break;
        __builtin_amdgcn_s_sleep(2);
      }
      if (have && (w0 & 3u) == 1) {
        s_send = w1;
        s_recv = w2;
        s_bytes = (w0 >> 2) & 0x3fffffu;
        s_meta = (uint32_t)w3;
        s_go = 1;
      } else {  // told to stop, or nothing came in time: say so -- the next blocking collective launches the agent again
        s_go = 0;
        __hip_atomic_store(&a.cmd[7], seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    __syncthreads();
    if (!s_go) return;
    if (t == 0) {  // (ll_begin, but for the epoch of a call that follows one of this agent's directly: no load from the page)
      uint64_t e;
      if ((s_meta >> kAgentLLConsecutiveShift & 1u) && prev_epoch) {
        e = prev_epoch + 1;
      } else {
        const uint64_t seen = ld_sys64(&a.ll.page[a.ll.me]->epoch_now);
        e = (seen > a.ll.epoch_floor ? seen : a.ll.epoch_floor) + 1;
      }
      prev_epoch = e;
      sh.epoch = e;
      sh.fail = DSYNC_OK;
    }
    __syncthreads();
    // (no fence here and none at the end: this call's buffers are read and written past the caches -- load8 / store8 <true>)
    const uint32_t meta = s_meta;
    const LLCall q{reinterpret_cast<const void*>(s_send), reinterpret_cast<void*>(s_recv), s_bytes, (int32_t)(meta & 3u),
                   (int32_t)((meta >> kAgentLLRootShift) & 15u)};
    const LLFoldRuntime fold{(int)((meta >> kAgentLLDtypeShift) & 7u), (int)((meta >> kAgentLLOpShift) & 3u)};
    const uint64_t epoch = sh.epoch;
    const uint32_t parity = (uint32_t)(epoch & 1u), flag = (uint32_t)epoch ? (uint32_t)epoch : 1u;
    const bool to_all = q.coll == LL_ALLREDUCE || q.coll == LL_ALLGATHER;
    if (!to_all) ll_say_here(a.ll, epoch);
    ll_agent_collective(a.ll, sh, q, parity, flag, fold, s_lds);
    if (!to_all) ll_wait_here(a.ll, sh);
    XMPI_DRAIN();
    __syncthreads();
    if (t == 0) {  // what ll_end does for a launched kernel of one block; the answer goes where the host polls: cmd[6]
      st_sys64(&a.ll.page[a.ll.me]->epoch_now, epoch);  // (every lane's written-through stores were acknowledged before the barrier above)
      if (a.ll.host_epoch) st_sys64(a.ll.host_epoch, epoch);
      if (sh.fail != DSYNC_OK && a.ll.status) __hip_atomic_store(a.ll.status, sh.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&a.cmd[6], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    seq++;
    __syncthreads();
  }
}

template <typename T>
hipError_t ll_op(const DsyncLLArgs& a, int op, dim3 grid, hipStream_t s, hipEvent_t es, hipEvent_t ee) {
  switch (op) {
    case OP_SUM: XMPI_LAUNCH((ll_reduce_kernel<T, OP_SUM>), grid, dim3(kBlock), s, es, ee, a); break;
    case OP_PROD: XMPI_LAUNCH((ll_reduce_kernel<T, OP_PROD>), grid, dim3(kBlock), s, es, ee, a); break;
    case OP_MIN: XMPI_LAUNCH((ll_reduce_kernel<T, OP_MIN>), grid, dim3(kBlock), s, es, ee, a); break;
    case OP_MAX: XMPI_LAUNCH((ll_reduce_kernel<T, OP_MAX>), grid, dim3(kBlock), s, es, ee, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace

hipError_t launch_dsync_ll(const DsyncLLArgs& a, int dtype, int op, hipStream_t s, hipEvent_t es, hipEvent_t ee) {
  if (a.n < 2 || a.n > kDsyncRanks || a.me < 0 || a.me >= a.n || a.root < 0 || a.root >= a.n || a.bytes == 0 ||
      a.bytes > kLLMaxPayload || !a.send || !a.recv)
    return hipErrorInvalidValue;
  const dim3 grid((unsigned)((a.bytes + 8 * kBlock - 1) / (8 * kBlock)));
  if (a.coll == LL_BCAST || a.coll == LL_ALLGATHER) {
    XMPI_LAUNCH(ll_copy_kernel, grid, dim3(kBlock), s, es, ee, a);
    return hipGetLastError();
  }
  if (a.coll != LL_ALLREDUCE && a.coll != LL_REDUCE) return hipErrorInvalidValue;
  switch (dtype) {
    case DT_U8: return ll_op<uint8_t>(a, op, grid, s, es, ee);
    case DT_I32: return ll_op<int32_t>(a, op, grid, s, es, ee);
    case DT_I64: return ll_op<int64_t>(a, op, grid, s, es, ee);
    case DT_F16: return ll_op<_Float16>(a, op, grid, s, es, ee);
    case DT_F32: return ll_op<float>(a, op, grid, s, es, ee);
    case DT_F64: return ll_op<double>(a, op, grid, s, es, ee);
    case DT_BF16: return ll_op<bf16_t>(a, op, grid, s, es, ee);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_ll_agent(const LLAgentArgs& a, hipStream_t s) {
  if (!a.cmd || a.ll.n < 2 || a.ll.n > kDsyncRanks || a.ll.me < 0 || a.ll.me >= a.ll.n) return hipErrorInvalidValue;
  hipLaunchKernelGGL(ll_agent_kernel, dim3(1), dim3(kLLAgentBlock), 0, s, a);
  return hipGetLastError();
}

}  // namespace xmpi
