// sched.hip -- gfx950 kernels of the device-synchronised schedules that are more than one fold:
//
//   * the SPLIT form of the zero-copy collectives (meet / body / done: three launches, only two blocks ever wait);
//   * the STEPPED schedules north_star names -- ring allreduce (reduce-scatter + allgather over several ring
//     channels), recursive halving + doubling, ring allgather, binary-tree broadcast and reduce -- each as ONE
//     kernel per rank that runs every step itself, step k of a worker released by a flag word the peer's worker
//     wrote after its step k-1 (no host between the steps: what engine.cpp does with a host progress loop, an
//     event and a counter in /dev/shm per step); each in a PULL form (a step loads the peer's memory) and a PUSH
//     form (a step stores into the peer's memory: the reference's one-way message, network.go:562-571) --
//     sched_steps.h;
//   * stream-ordered Send / Receive: the reference's message + ack (network.go:562-571, 616-624) as two
//     64-byte records in the flag allocations and one pull of the payload.
//
// The reference has none of this (mpi.go:130 is a stub; a user composes collectives from Send / Receive as
// examples/helloworld/helloworld.go:53-81 does).  Protocol and layouts: kernels.h.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstring>

#include "kdev.h"
#include "kernels.h"
#include "sched_steps.h"

namespace xmpi {
namespace {

#define XMPI_LAUNCH(kern, grid, block, stream, es, ee, ...)                                  \
  do {                                                                                       \
    if ((es) || (ee)) hipExtLaunchKernelGGL(kern, grid, block, 0, stream, es, ee, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kern, grid, block, 0, stream, __VA_ARGS__);                      \
  } while (0)

// =====================================================================================================================
// split form: meet
// =====================================================================================================================

// one-wave blocks of the meet and done kernels (kXcdBlocks = 16): consecutive blocks are dealt round the 8 XCDs, two rounds so
// that every XCD's L2 is acquired / released even if one block lands beside another.  HIP promises nothing about where a block
// runs, so every block records the XCD it found itself on (HW_REG_XCC_ID) and the done kernel's closing block checks that
// both kernels reached all of them (DsyncArgs::xcc_need; DSYNC_XCD otherwise).

// which XCDs a grid of one-wave blocks reaches
__global__ __launch_bounds__(64) void xcc_probe_kernel(uint32_t* mask) {
  if (threadIdx.x == 0) __hip_atomic_fetch_or(mask, 1u << xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct FlagSelftestArgs {
  DsyncPage* page[kDsyncRanks];
  int32_t me, n;
  uint64_t token, spin_limit;
  const int32_t* abort_word;
  uint32_t* seen;
};
__global__ __launch_bounds__(64) void flag_selftest_kernel(FlagSelftestArgs a) {
  XMPI_SHARED(uint32_t, s_seen);
  const int t = threadIdx.x;
  if (t == 0) s_seen = 1u << a.me;
  __syncthreads();
  if (t < a.n && t != a.me) {
    st_sys64(&a.page[t]->done[a.me][1], a.token);
    if (spin_until(&a.page[a.me]->done[t][1], a.token, a.abort_word, a.spin_limit) == DSYNC_OK)
      __hip_atomic_fetch_or(&s_seen, 1u << t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  if (t == 0) __hip_atomic_store(a.seen, s_seen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// A few blocks (at least one per XCD): announce (block 0), wait for every peer, acquire at system scope -- each block for the L2
// of the XCD it runs on, so that the data kernel behind this one cannot be served a stale line of a peer's input -- and
// block 0 translates the peers' buffers and leaves what the data kernel needs in ordinary device memory (the kernel
// boundary publishes it).  No completion exchange here: the done kernel does it.
__global__ __launch_bounds__(64) void dsync_meet_kernel(DsyncArgs a, DsyncResolved* out) {
  XMPI_SHARED(DsyncShared, sh);
  dsync_begin(a, sh);
  // this block has acquired the L2 of the XCD it runs on
  if (threadIdx.x == 0) __hip_atomic_fetch_or(&a.page[a.me]->xcc_meet, 1u << xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const int me = a.me, n = a.n;
    out->epoch = sh.epoch;
    out->fail = sh.fail;
    out->nseg = (uint32_t)a.nseg;
    for (int y = 0; y < a.nseg; y++) {
      const DsyncSeg& g = a.seg[y];
      DsyncResolvedSeg& o = out->seg[y];
      int ns = 0, nd = 0;
      uint64_t all = 0;
      for (int r = 0; r < n; r++)
        if (g.src_mask >> r & 1u)
          all |= (o.src[ns++] = g.src_from_recv == 2 ? (r == me ? sh.send[me] + g.src_off : sh.land[me] + (uint64_t)r * g.stage_stride)
                                                     : (g.src_from_recv ? sh.recv[r] : sh.send[r]) + g.src_off);
      for (int d = 0; d < n; d++) {
        const int r = (me + d) % n;
        if (g.dst_mask >> r & 1u) all |= (o.dst[nd++] = (g.dst_to_land ? sh.land[r] : sh.recv[r]) + g.dst_off);
      }
      for (int k = ns; k < kDsyncRanks; k++) o.src[k] = o.src[0];
      for (int k = nd; k < kDsyncRanks; k++) o.dst[k] = o.dst[0];
      o.count = g.count;
      o.nsrc = ns;
      o.ndst = nd;
      o.vec = (all & 15u) == 0 ? 1u : 0u;
    }
  }
}

// =====================================================================================================================
// split form: body -- the fold of reduce_n_multi_kernel, pointers from the table the meet kernel left
// =====================================================================================================================

template <typename T, int OP, int NSRC, int MODE>
__global__ __launch_bounds__(kBlock) void dsync_body_kernel(const DsyncResolved* __restrict__ res) {
  // the table is this GPU's own memory, written by the kernel before this one: uniform loads, before any fence
  const DsyncResolvedSeg* __restrict__ g = &res->seg[blockIdx.y];
  const uint32_t fail = res->fail;
  const int nsrc = (NSRC > 0) ? NSRC : g->nsrc, ndst = g->ndst;
  const size_t count = g->count;
  const uint32_t vec = g->vec;
  uint64_t dp[kDsyncRanks];
#pragma unroll
  for (int k = 0; k < kDsyncRanks; k++) dp[k] = g->dst[k];
  uint64_t sp[NSRC > 0 ? NSRC : 1];
  if constexpr (NSRC > 0) {
#pragma unroll
    for (int k = 0; k < NSRC; k++) sp[k] = g->src[k];
  }
  // No fence here: the meet kernel's blocks acquired at system scope (one per XCD) after the peers had announced
  // themselves, and the kernel boundary orders this kernel behind them.  (A per-block acquire -- an L2 invalidate per
  // block of an unbounded grid -- cost more than the fold itself: 1.66 ms instead of 0.8 for 8 x 256 MiB, r03 session 1.)
  // Nor does this kernel close the collective itself: a variant whose stores were written through at system scope, with
  // the last block exchanging "done" (two launches instead of three), made every one-tile block wait for HBM to
  // acknowledge its stores before it could give its wave slots back -- 4.2 ms instead of 0.95 (r03 session 3).  The done
  // kernel's few blocks release once per XCD instead.
  // MODE 3 is the form that assumes none of this: every load and store at system scope (sc0 sc1 -- never served from a line
  // of an L2 that predates the peer's store, written through), for a machine whose dispatcher does not deal the meet / done
  // kernels' blocks round the XCDs (xmpi_set_param "body_sys"; chosen by itself when the probe at start-up says so).
  if (fail != DSYNC_OK) return;
  const int t = threadIdx.x;
  constexpr size_t N = 16 / sizeof(T);
  if constexpr (MODE == 3) {
    if (vec) {
      const size_t npack = count / N;
      const size_t stride = (size_t)gridDim.x * kBlock;
      for (size_t i = (size_t)blockIdx.x * kBlock + t; i < npack; i += stride) {
        pack_t acc;
        if constexpr (NSRC > 0) {
          pack_t v[NSRC];
#pragma unroll
          for (int k = 0; k < NSRC; k++) ld_sys128_issue(v[k], reinterpret_cast<const pack_t*>(sp[k]) + i);
          sys128_wait_n<NSRC>(v);
          acc = v[0];
#pragma unroll
          for (int k = 1; k < NSRC; k++) acc = combine16<T, OP>(acc, v[k]);
        } else {
          pack_t v[1];
          ld_sys128_issue(v[0], reinterpret_cast<const pack_t*>(g->src[0]) + i);
          sys128_wait_n<1>(v);
          acc = v[0];
          for (int k = 1; k < nsrc; k++) {
            ld_sys128_issue(v[0], reinterpret_cast<const pack_t*>(g->src[k]) + i);
            sys128_wait_n<1>(v);
            acc = combine16<T, OP>(acc, v[0]);
          }
        }
#pragma unroll
        for (int k = 0; k < kDsyncRanks; k++)
          if (k < ndst) st_sys128(reinterpret_cast<pack_t*>(dp[k]) + i, acc);
      }
    }
    const size_t first = vec ? (count / N) * N : 0;  // ragged tail, or everything when a buffer is not 16-byte aligned
    const size_t lanes = vec ? (size_t)kBlock : (size_t)gridDim.x * kBlock;
    if (!vec || blockIdx.x == 0)
      for (size_t i = first + (vec ? 0 : (size_t)blockIdx.x * kBlock) + t; i < count; i += lanes) {
        T acc = ld_sys_elem(reinterpret_cast<const T*>(g->src[0]) + i);
        for (int k = 1; k < nsrc; k++) acc = combine_any<T, OP>(acc, ld_sys_elem(reinterpret_cast<const T*>(g->src[k]) + i));
        for (int k = 0; k < ndst; k++) st_sys_elem(reinterpret_cast<T*>(g->dst[k]) + i, acc);
      }
    return;
  }
  if (vec) {
    const size_t npack = count / N;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + t; i < npack; i += stride) {
      pack_t acc;
      if constexpr (NSRC > 0) {
        pack_t v[NSRC];
#pragma unroll
        for (int k = 0; k < NSRC; k++) v[k] = ldp<MODE>(reinterpret_cast<const pack_t*>(sp[k]) + i);
        acc = v[0];
#pragma unroll
        for (int k = 1; k < NSRC; k++) acc = combine16<T, OP>(acc, v[k]);
      } else {
        acc = ldp<MODE>(reinterpret_cast<const pack_t*>(g->src[0]) + i);
        for (int k = 1; k < nsrc; k++) acc = combine16<T, OP>(acc, ldp<MODE>(reinterpret_cast<const pack_t*>(g->src[k]) + i));
      }
#pragma unroll
      for (int k = 0; k < kDsyncRanks; k++)
        if (k < ndst) stp<(MODE != 0) ? 1 : 0>(reinterpret_cast<pack_t*>(dp[k]) + i, acc);
    }
    const size_t done = npack * N;  // ragged tail (< 16 bytes): the first lanes of the segment's block 0
    if (blockIdx.x == 0 && done + t < count) {
      const size_t i = done + t;
      T acc = reinterpret_cast<const T*>(g->src[0])[i];
      for (int k = 1; k < nsrc; k++) acc = combine_any<T, OP>(acc, reinterpret_cast<const T*>(g->src[k])[i]);
      for (int k = 0; k < ndst; k++) reinterpret_cast<T*>(g->dst[k])[i] = acc;
    }
  } else {  // some buffer is not 16-byte aligned: one element per lane
    for (size_t i = (size_t)blockIdx.x * kBlock + t; i < count; i += (size_t)gridDim.x * kBlock) {
      T acc = reinterpret_cast<const T*>(g->src[0])[i];
      for (int k = 1; k < nsrc; k++) acc = combine_any<T, OP>(acc, reinterpret_cast<const T*>(g->src[k])[i]);
      for (int k = 0; k < ndst; k++) reinterpret_cast<T*>(g->dst[k])[i] = acc;
    }
  }
}

// =====================================================================================================================
// split form: done -- a few blocks (one per XCD: each releases at system scope, i.e. writes back the L2 it runs on),
// the last of them exchanges "done" with every peer and advances the epoch
// =====================================================================================================================

__global__ __launch_bounds__(64) void dsync_done_kernel(DsyncArgs a, const DsyncResolved* res) {
  XMPI_SHARED(DsyncShared, sh);
  if (threadIdx.x == 0) {
    sh.epoch = res->epoch;
    sh.fail = res->fail;
    // (dsync_end releases: the L2 of the XCD this block runs on is written back)
    __hip_atomic_fetch_or(&a.page[a.me]->xcc_done, 1u << xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  dsync_end(a, sh, true);
}

// =====================================================================================================================
// stepped schedules
// =====================================================================================================================

// One tile: D[x] (and D2[x], when there is a second destination) = A[x] (NS == 1), A[x] op B[x] (NS == 2) or
// C[x] op (A[x] op B[x]) (NS == 3) for the byte offsets x in [lo, hi) -- multiples of sizeof(T); every base is addressed with
// the SAME offset.  vec: all bases are 16-byte aligned.
// A and C are what a peer's kernel may have written a moment ago -- into ITS memory (pull form) or into this rank's (push
// form) -- and D / D2 are what a peer's kernel, or a later step of this one, will read next: all of them go straight to memory
// at system scope (kdev.h ld_sys128_issue / st_sys128: no fence, no cache maintenance per step).  B is this rank's own data (its
// input, or what THIS block stored in an earlier step): an ordinary non-temporal load.
template <typename T, int OP, int NS>
__device__ __forceinline__ void tile_apply(char* D, char* D2, const char* A, const char* B, const char* C, size_t lo, size_t hi, bool vec) {
  const int t = threadIdx.x;
  constexpr size_t ES = sizeof(T);
  auto one = [&](size_t x) {
    T v = ld_sys_elem(reinterpret_cast<const T*>(A + x));
    if constexpr (NS >= 2) v = combine_any<T, OP>(v, *reinterpret_cast<const T*>(B + x));
    if constexpr (NS == 3) v = combine_any<T, OP>(ld_sys_elem(reinterpret_cast<const T*>(C + x)), v);
    st_sys_elem(reinterpret_cast<T*>(D + x), v);
    if (D2) st_sys_elem(reinterpret_cast<T*>(D2 + x), v);
  };
  if (vec) {
    const size_t plo = (lo + 15) & ~(size_t)15, phi = hi & ~(size_t)15;
    if (plo < phi) {
      if (phi - plo == kSchedTileBytes) {  // a whole tile: every load issued before the first use
        constexpr int U = (int)(kSchedTileBytes / 16 / kBlock);
        const pack_t* pa = reinterpret_cast<const pack_t*>(A + plo) + t;
        const pack_t* pb = NS >= 2 ? reinterpret_cast<const pack_t*>(B + plo) + t : nullptr;
        const pack_t* pc = NS == 3 ? reinterpret_cast<const pack_t*>(C + plo) + t : nullptr;  // (no arithmetic on a null base: C, D2 may be)
        pack_t* pd = reinterpret_cast<pack_t*>(D + plo) + t;
        pack_t* pd2 = D2 ? reinterpret_cast<pack_t*>(D2 + plo) + t : nullptr;
        pack_t va[U], vb[U], vc[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          ld_sys128_issue(va[u], pa + u * kBlock);
          if constexpr (NS == 3) ld_sys128_issue(vc[u], pc + u * kBlock);
          if constexpr (NS >= 2) vb[u] = ldp<2>(pb + u * kBlock);
        }
        sys128_wait<U>(va);
        if constexpr (NS == 3) sys128_wait<U>(vc);
#pragma unroll
        for (int u = 0; u < U; u++) {
          if constexpr (NS >= 2) va[u] = combine16<T, OP>(va[u], vb[u]);
          if constexpr (NS == 3) va[u] = combine16<T, OP>(vc[u], va[u]);
          st_sys128(pd + u * kBlock, va[u]);
          if (D2) st_sys128(pd2 + u * kBlock, va[u]);
        }
      } else {
        for (size_t x = plo + (size_t)t * 16; x < phi; x += (size_t)kBlock * 16) {
          pack_t v[1], c[1];
          ld_sys128_issue(v[0], reinterpret_cast<const pack_t*>(A + x));
          if constexpr (NS == 3) ld_sys128_issue(c[0], reinterpret_cast<const pack_t*>(C + x));
          pack_t w;
          if constexpr (NS >= 2) w = ldp<2>(reinterpret_cast<const pack_t*>(B + x));
          sys128_wait<1>(v);
          if constexpr (NS == 3) sys128_wait<1>(c);
          if constexpr (NS >= 2) v[0] = combine16<T, OP>(v[0], w);
          if constexpr (NS == 3) v[0] = combine16<T, OP>(c[0], v[0]);
          st_sys128(reinterpret_cast<pack_t*>(D + x), v[0]);
          if (D2) st_sys128(reinterpret_cast<pack_t*>(D2 + x), v[0]);
        }
      }
      // the elements before the first and after the last whole packet (fewer than 16 bytes each)
      size_t x = lo + (size_t)t * ES;
      if (x < plo) one(x);
      x = phi + (size_t)t * ES;
      if (x < hi) one(x);
      return;
    }
  }
  for (size_t x = lo + (size_t)t * ES; x < hi; x += (size_t)kBlock * ES) one(x);
}

// worker w of W: the tiles ti of [rlo, rhi) with ti % W == w (tile = kSchedTileBytes of the buffer, counted from
// the start of the BUFFER, not of the range: the same bytes belong to the same worker on every rank at every step)
template <typename T, int OP, int NS>
__device__ void range_apply(const SchedMove& m, uint32_t w, uint32_t W) {
  const size_t rlo = m.lo, rhi = m.hi;
  if (rlo >= rhi) return;
  const bool vec = ((m.D | m.D2 | m.A | (NS >= 2 ? m.B : 0) | (NS == 3 ? m.C : 0)) & 15u) == 0;
  constexpr size_t TB = kSchedTileBytes;
  const size_t t0 = rlo / TB;
  size_t ti = t0 + (size_t)((w + W - (uint32_t)(t0 % W)) % W);
  for (; ti * TB < rhi; ti += W) {
    const size_t lo = ti * TB > rlo ? ti * TB : rlo;
    const size_t hi = (ti + 1) * TB < rhi ? (ti + 1) * TB : rhi;
    tile_apply<T, OP, NS>(reinterpret_cast<char*>(m.D), reinterpret_cast<char*>(m.D2), reinterpret_cast<const char*>(m.A),
                          reinterpret_cast<const char*>(m.B), reinterpret_cast<const char*>(m.C), lo, hi, vec);
  }
}

// one move of a step as a kernel of its own (xmpi_link_probe engine 2)
__global__ __launch_bounds__(kBlock) void sys_copy_kernel(SchedMove m) { range_apply<uint8_t, OP_SUM, 1>(m, blockIdx.x, gridDim.x); }

template <typename T, int OP>
__global__ __launch_bounds__(kBlock) void dsync_sched_kernel(DsyncSchedArgs a) {
  XMPI_SHARED(DsyncShared, sh);
  XMPI_SHARED(SchedStep, st);
  dsync_begin(a.d, sh);
  const int t = threadIdx.x, me = a.d.me;
  // worker numbers run through the channels first (blockIdx.y = the channel): the tiles that do not divide by W go to the lowest
  // workers, and those must not all belong to channel 0 -- its links would carry 3 ... 8 % more than the others' (counted:
  // tests/devsim traffic, N = 8: busiest link direction 0.314 S -> 0.292 S)
  const uint32_t W = gridDim.x * gridDim.y, w = blockIdx.x * gridDim.y + blockIdx.y;
  DsyncPage* mine = a.d.page[me];
  if (sh.fail == DSYNC_OK) {
    const int nsteps = sched_nsteps(a);
    for (int g = 1; g <= nsteps; g++) {
      if (t == 0) sched_step(a, sh.send, sh.recv, sh.land, g, (int)blockIdx.y, &st);
      __syncthreads();
      const int wait_rank = st.wait_rank;
      if (wait_rank >= 0) {
        if (t == 0) {
          const uint32_t why = dsync_spin(step_flags(mine) + (size_t)wait_rank * kStepSlots + w, (sh.epoch << 8) | st.wait_val, a.d);
          if (why != DSYNC_OK) atomicMax(&sh.fail, why);
        }
        __syncthreads();
        if (sh.fail != DSYNC_OK) break;
        // (no fence: what the peer's step stored was written through before its flag, and is loaded past the caches)
      }
      for (int k = 0; k < st.nmv; k++) {
        const SchedMove& m = st.mv[k];
        if (m.ns == 3) range_apply<T, OP, 3>(m, w, W);
        else if (m.ns == 2) range_apply<T, OP, 2>(m, w, W);
        else range_apply<uint8_t, OP_SUM, 1>(m, w, W);
      }
      XMPI_DRAIN();
      __syncthreads();
      if (t == 0 && (st.sig[0] >= 0 || st.sig[1] >= 0)) {
        // every wave has waited for the acknowledgements of its (written-through) stores -- into this rank's memory or, push
        // form, into the peer's: the flag may follow them
        const uint64_t val = (sh.epoch << 8) | st.sig_val;
        for (int k = 0; k < 2; k++)
          if (st.sig[k] >= 0) st_sys64(step_flags(a.d.page[st.sig[k]]) + (size_t)me * kStepSlots + w, val);
      }
      __syncthreads();  // st is rewritten by the next step
    }
  }
  dsync_end(a.d, sh);
}

// =====================================================================================================================
// stream-ordered Send / Receive
// =====================================================================================================================

__device__ __forceinline__ P2PBox* p2p_boxes(DsyncPage* page) {
  return reinterpret_cast<P2PBox*>(reinterpret_cast<char*>(page) + kBoxOff);
}
__device__ __forceinline__ P2PAck* p2p_acks(DsyncPage* page) {
  return reinterpret_cast<P2PAck*>(reinterpret_cast<char*>(page) + kAckOff);
}
__device__ __forceinline__ uint64_t* p2p_taken(DsyncPage* page) {
  return reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(page) + kTakenOff);
}
__device__ __forceinline__ P2PGo* p2p_go(DsyncPage* page) {
  return reinterpret_cast<P2PGo*>(reinterpret_cast<char*>(page) + kGoOff);
}

__device__ __forceinline__ uint32_t p2p_spin(const uint64_t* p, uint64_t want, const P2PArgs& a) {
  return spin_until(p, want, a.abort_word, a.spin_limit);
}

__device__ __forceinline__ void p2p_host_done(const P2PArgs& a, uint64_t status, uint64_t bytes) {
  if (!a.host_done) return;
  st_sys64(a.host_done + 1, status);
  st_sys64(a.host_done + 2, bytes);
  __hip_atomic_store(a.host_done, a.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One wave.  Message `seq` of the pair goes into box (seq - 1) % kP2PBoxes of the receiver's allocation -- once the
// message that used the box before has been answered -- and the kernel ends when the receiver has answered this one:
// the payload has been consumed, the caller's buffer is free (the rendezvous of network.go:569).
__global__ __launch_bounds__(64) void p2p_send_kernel(P2PArgs a) {
  if (threadIdx.x != 0) return;
  // a.seq = (communicator number << 32) | n: pages are pooled and never cleared, the communicator number keeps the
  // message numbers of successive communicators apart (and growing)
  const uint64_t n = a.seq & 0xffffffffull;
  const int b = (int)((n - 1) % kP2PBoxes);
  P2PAck* ack = p2p_acks(a.my_page) + (size_t)a.peer * kP2PBoxes + b;  // written by the receiver
  uint32_t why = DSYNC_OK;
  if (n > (uint64_t)kP2PBoxes) why = p2p_spin(&ack->seq, a.seq - kP2PBoxes, a);
  uint64_t status = 0;
  if (why == DSYNC_OK) {
    P2PBox* box = p2p_boxes(a.peer_page) + (size_t)a.me * kP2PBoxes + b;
    st_sys64(&box->tag, (uint64_t)(uint32_t)a.tag | ((uint64_t)(uint32_t)a.dtype << 32));
    st_sys64(&box->bytes, a.bytes);
    st_sys64(&box->gen, a.gen);
    st_sys64(&box->slot, a.slot);
    st_sys64(&box->off, a.off);
    // the payload was written by earlier work on this stream (possibly on another XCD: the kernel boundary wrote
    // that back); the release orders the record before its number
    __hip_atomic_store(&box->seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    why = p2p_spin(&ack->seq, a.seq, a);
    if (why == DSYNC_OK) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
      status = ld_sys64(&ack->status);
    }
  }
  if (why != DSYNC_OK) status = 0x100u + why;  // (xmpi codes are small: 0x100 + DsyncStatus marks a wait that was cut short)
  p2p_host_done(a, status, a.bytes);
}

// Block 0 finds the box of (peer -> me) that carries this tag, checks it, translates where the payload lives and
// tells the other blocks (go record); everybody copies; the block that finishes last answers the sender and the host.
__global__ __launch_bounds__(kBlock) void p2p_recv_kernel(P2PArgs a) {
  XMPI_SHARED(uint64_t, s_src);
  XMPI_SHARED(uint64_t, s_bytes);
  XMPI_SHARED(uint64_t, s_status);
  XMPI_SHARED(uint64_t, s_seq);
  XMPI_SHARED(int, s_box);
  XMPI_SHARED(uint32_t, s_last);
  const int t = threadIdx.x;
  P2PGo* go = p2p_go(a.my_page) + (a.op_id % kP2PGoSlots);
  if (t == 0) {
    if (blockIdx.x == 0) {
      P2PBox* boxes = p2p_boxes(a.my_page) + (size_t)a.peer * kP2PBoxes;
      uint64_t* taken = p2p_taken(a.my_page) + (size_t)a.peer * kP2PBoxes;
      uint64_t status = 0, src = 0, bytes = 0, seq = 0;
      int found = -1;
      const uint64_t t0 = wall_clock64();
      for (uint32_t k = 1;; k++) {
        uint64_t best = ~0ull, best_taken = 0;
        for (int b = 0; b < kP2PBoxes; b++) {  // the oldest unconsumed message with this tag
          const uint64_t s = __hip_atomic_load(&boxes[b].seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
          const uint64_t tk = ld_sys64(&taken[b]);
          if ((s >> 32) != (a.comm_tag & 0xffffffffull) || s <= tk || s >= best) continue;  // not this communicator's / consumed
          if ((int32_t)(uint32_t)ld_sys64(&boxes[b].tag) != a.tag) continue;
          best = s;
          best_taken = tk;
          found = b;
        }
        if (found >= 0) {
          // claim it: receives for the same (source, tag) enqueued on DIFFERENT streams run concurrently and may both have
          // picked this box -- the compare-and-swap lets one of them have it, the other looks again
          uint64_t expect = best_taken;
          if (__hip_atomic_compare_exchange_strong(&taken[found], &expect, best, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
          found = -1;
          continue;
        }
        __builtin_amdgcn_s_sleep(1);
        if ((k & 63u) == 0) {
          if (a.abort_word && __hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) {
            status = 0x100u + DSYNC_ABORTED;
            break;
          }
          if (a.spin_limit && wall_clock64() - t0 > a.spin_limit) {
            status = 0x100u + DSYNC_TIMEOUT;
            break;
          }
        }
      }
      if (found >= 0) {
        const P2PBox* box = &boxes[found];
        seq = ld_sys64(&box->seq);
        bytes = ld_sys64(&box->bytes);
        const int dt = (int)(ld_sys64(&box->tag) >> 32) & 0xff;
        if (dt != a.dtype) status = 1;         // -> XMPI_ERR_ARG
        else if (bytes > a.bytes) status = 6;  // -> XMPI_ERR_TRUNCATE
        else if (bytes > 0) {
          src = translate(a.comm_tag, a.table, a.my_page, a.peer, ld_sys64(&box->slot), ld_sys64(&box->gen), ld_sys64(&box->off));
          if (!src) status = 0x100u + DSYNC_UNMAPPED;
        }
      }
      s_src = src;
      s_bytes = bytes;
      s_status = status;
      s_seq = seq;
      s_box = found;
      if (gridDim.x > 1) {
        st_sys64(&go->src, src);
        st_sys64(&go->bytes, bytes);
        st_sys64(&go->status, status);
        st_sys64(&go->seq, seq);
        st_sys64(&go->box, (uint64_t)(int64_t)found);
        __hip_atomic_store(&go->id, a.op_id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      // (block 0 always writes the record -- also when it gave up: this wait needs no clock of its own, only the job's abort
      // flag in case block 0 never became resident)
      uint32_t why = DSYNC_OK;
      for (uint32_t k = 1; __hip_atomic_load(&go->id, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != a.op_id; k++) {
        __builtin_amdgcn_s_sleep(1);
        if ((k & 1023u) == 0 && a.abort_word && __hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) {
          why = DSYNC_ABORTED;
          break;
        }
      }
      if (why == DSYNC_OK) {
        s_src = ld_sys64(&go->src);
        s_bytes = ld_sys64(&go->bytes);
        s_status = ld_sys64(&go->status);
        s_seq = ld_sys64(&go->seq);
        s_box = (int)(int64_t)ld_sys64(&go->box);
      } else {  // copy nothing; the ticket below is still taken (block 0, if it ever runs, must not wait for this one)
        s_src = 0;
        s_bytes = 0;
        s_status = 0x100u + why;
        s_seq = 0;
        s_box = -1;
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // the payload as the sender left it, not a stale line
  const uint64_t src = s_src, bytes = s_status ? 0 : s_bytes;
  if (bytes) copy_span(reinterpret_cast<char*>(a.buf), reinterpret_cast<const char*>(src), bytes);
  XMPI_DRAIN();
  __syncthreads();
  if (t == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    s_last = gridDim.x == 1 || __hip_atomic_fetch_add(&go->ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  }
  __syncthreads();
  if (!s_last || t != 0) return;
  if (gridDim.x > 1) __hip_atomic_store(&go->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (s_box >= 0) {  // the answer: the message has been consumed (or why not) -- network.go:616-624's ack
    P2PAck* ack = p2p_acks(a.peer_page) + (size_t)a.me * kP2PBoxes + s_box;
    st_sys64(&ack->status, s_status);
    __hip_atomic_store(&ack->seq, s_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  p2p_host_done(a, s_status, s_bytes);
}

// The blocking Receive's copy (engine.cpp p2p_recv): the host has matched the message, so nothing here waits for a
// peer -- the payload comes straight out of the sender's buffer, and the block that finishes last writes the ack
// where the SENDER's host thread polls (its mail entry in the shared control block, over PCIe) and the completion word
// where this rank's host thread polls.  No event, no host hop between the copy and the ack (network.go:616-624).
__global__ __launch_bounds__(kBlock) void p2p_pull_kernel(P2PPullArgs a) {
  XMPI_SHARED(uint32_t, s_last);
  const int t = threadIdx.x;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // the payload as the sender left it, not a stale line
  copy_span(reinterpret_cast<char*>(a.dst), reinterpret_cast<const char*>(a.src), a.bytes);
  XMPI_DRAIN();
  __syncthreads();
  if (t == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    s_last = gridDim.x == 1 || __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  }
  __syncthreads();
  if (!s_last || t != 0) return;
  if (gridDim.x > 1) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // Every block has drained its stores and released them (above): the two words go out back to back, no further
  // round trip between them (the entry's status is already XMPI_OK -- the sender wrote it when it posted the message).
  if (a.mail_state) __hip_atomic_store(a.mail_state, a.mail_done_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (a.host_done) __hip_atomic_store(a.host_done, a.done_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The receive AGENT (engine.cpp p2p_recv / agent_submit): the same copy-and-ack, by a kernel that STAYS for a while.
// A blocking Receive pays one kernel launch per message -- about 5 us before the first wave runs, most of an 8 us half
// round trip -- and a ping-pong cannot hide it (the Receive is called when the message is already on its way).  So the
// kernel that served a message does not end at once: block 0 goes back to watching the command record in pinned host
// memory, where the host thread of the NEXT Receive -- which does the matching, as always -- writes {where the payload
// is, where it goes, how long}; the copy starts a PCIe read later instead of a launch later.  The agent waits for the
// HOST's word only, and only for `patience` (tens of microseconds) after its last message; then it says so and ends, and
// the next Receive launches it again.  Nothing on the GPU ever waits for a peer, and nothing waits for the host longer
// than that bound: a program blocked in Receives whose messages arrive in any order (helloworld.go:53-81) cannot
// deadlock, the streams that share the agent's hardware queue are delayed by the patience at most.
//
// The command is 32 bytes read with two 16-byte loads issued together (a load over PCIe is a microsecond or two):
//   w0 = doorbell (1 = go, 2 = stop) | bytes << 2 (22 bits) | seq << 24     w1 = where the payload is
//   w2 = where it goes                                                     w3 = mail entry's offset in the control block | seq << 32
// written by the host w1, w2, w3, then w0 (x86 keeps the order; both halves lie in one cache line), accepted when BOTH
// halves carry the expected number.  cmd[6] = number of the last command served; cmd[7] = "gone" (the number the agent was
// waiting for, plus one).
__global__ __launch_bounds__(kBlock) void p2p_agent_kernel(P2PAgentArgs a) {
  XMPI_SHARED(uint64_t, s_src);
  XMPI_SHARED(uint64_t, s_dst);
  XMPI_SHARED(uint64_t, s_bytes);
  XMPI_SHARED(uint64_t, s_mail);
  XMPI_SHARED(uint64_t, s_seq);
  XMPI_SHARED(uint32_t, s_go);
  XMPI_SHARED(uint32_t, s_last);
  const int t = threadIdx.x;
  // device memory: [0] what block 0 tells the other blocks: (launch << 40 | n) << 1 | go for its n-th word to them (the record
  // outlives launches: the launch number keeps a stale word of the launch before from being taken for this one's),
  // [1] src, [2] dst, [3] bytes, [5] ticket, [6] the word all blocks have finished with
  uint64_t* rec = a.rec;
  uint64_t seq = a.seq0, told = 0;  // next command number (block 0); words to the other blocks so far (every block counts)
  for (;;) {
    bool wide = true;  // all blocks copy (block 0 alone takes a short message: waking the others costs more than they save)
    if (t == 0) {
      if (blockIdx.x == 0) {
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
          s_src = w1;
          s_dst = w2;
          s_bytes = (w0 >> 2) & 0x3fffffu;
          s_mail = a.ctl_dev + (w3 & 0xffffffffull);
          s_go = 1;
        } else {  // told to stop, or nothing came in time: say so -- the next Receive launches the agent again
          s_go = 0;
          __hip_atomic_store(&a.cmd[7], seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        s_seq = seq;
        if (gridDim.x > 1 && (!s_go || s_bytes > a.alone_bytes)) {
          st_sys64(&rec[1], s_src);
          st_sys64(&rec[2], s_dst);
          st_sys64(&rec[3], s_bytes);
          __hip_atomic_store(&rec[0], ((((uint64_t)a.launch << 40) | (told + 1)) << 1) | s_go, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else {
        const uint64_t key = ((uint64_t)a.launch << 40) | (told + 1);
        uint64_t v;
        while (((v = __hip_atomic_load(&rec[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) >> 1) != key) __builtin_amdgcn_s_sleep(8);
        s_go = (uint32_t)(v & 1);
        s_src = ld_sys64(&rec[1]);
        s_dst = ld_sys64(&rec[2]);
        s_bytes = ld_sys64(&rec[3]);
      }
    }
    __syncthreads();
    if (!s_go) return;
    wide = gridDim.x > 1 && s_bytes > a.alone_bytes;
    if (wide || blockIdx.x != 0) told++;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // the payload as the sender left it, not a stale line
    if (wide) copy_span(reinterpret_cast<char*>(s_dst), reinterpret_cast<const char*>(s_src), s_bytes);
    else copy_span(reinterpret_cast<char*>(s_dst), reinterpret_cast<const char*>(s_src), s_bytes, 0, 1);
    XMPI_DRAIN();
    __syncthreads();
    if (t == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
      const uint64_t key = ((uint64_t)a.launch << 40) | told;
      uint32_t* ticket = reinterpret_cast<uint32_t*>(&rec[5]);
      s_last = !wide || __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
      if (s_last && wide) {
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&rec[6], key, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (blockIdx.x == 0) {
        // every block has finished this message (its stores drained and released) before the acks go out -- and before the
        // record for the other blocks is rewritten for the next one
        if (wide)
          while (__hip_atomic_load(&rec[6], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != key) __builtin_amdgcn_s_sleep(1);
        __hip_atomic_store(reinterpret_cast<uint32_t*>(s_mail), a.mail_done_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&a.cmd[6], s_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        seq++;
      }
    }
    __syncthreads();
  }
}

// =====================================================================================================================
// launchers
// =====================================================================================================================

template <typename T, int OP>
hipError_t body_go(const DsyncResolved* res, dim3 grid, int nsrc_hint, int mode, hipStream_t s, hipEvent_t es, hipEvent_t ee) {
#define XMPI_BODY(NS)                                                                                         \
  do {                                                                                                        \
    if (mode == 3) XMPI_LAUNCH((dsync_body_kernel<T, OP, NS, 3>), grid, dim3(kBlock), s, es, ee, res);        \
    else if (mode != 0) XMPI_LAUNCH((dsync_body_kernel<T, OP, NS, 2>), grid, dim3(kBlock), s, es, ee, res);   \
    else XMPI_LAUNCH((dsync_body_kernel<T, OP, NS, 0>), grid, dim3(kBlock), s, es, ee, res);                  \
    return hipGetLastError();                                                                                 \
  } while (0)
  if constexpr (OP == OP_SUM) {
    if (nsrc_hint == 2) XMPI_BODY(2);
    if (nsrc_hint == 4) XMPI_BODY(4);
    if (nsrc_hint == 8) XMPI_BODY(8);
  }
  XMPI_BODY(0);
#undef XMPI_BODY
}

template <typename T>
hipError_t body_op(const DsyncResolved* res, dim3 grid, int nsrc_hint, int op, int mode, hipStream_t s, hipEvent_t es, hipEvent_t ee) {
  switch (op) {
    case OP_SUM: return body_go<T, OP_SUM>(res, grid, nsrc_hint, mode, s, es, ee);
    case OP_PROD: return body_go<T, OP_PROD>(res, grid, nsrc_hint, mode, s, es, ee);
    case OP_MIN: return body_go<T, OP_MIN>(res, grid, nsrc_hint, mode, s, es, ee);
    case OP_MAX: return body_go<T, OP_MAX>(res, grid, nsrc_hint, mode, s, es, ee);
    default: return hipErrorInvalidValue;
  }
}

template <typename T>
hipError_t sched_op(const DsyncSchedArgs& a, int op, dim3 grid, hipStream_t s, hipEvent_t es, hipEvent_t ee) {
  switch (op) {
    case OP_SUM: XMPI_LAUNCH((dsync_sched_kernel<T, OP_SUM>), grid, dim3(kBlock), s, es, ee, a); break;
    case OP_PROD: XMPI_LAUNCH((dsync_sched_kernel<T, OP_PROD>), grid, dim3(kBlock), s, es, ee, a); break;
    case OP_MIN: XMPI_LAUNCH((dsync_sched_kernel<T, OP_MIN>), grid, dim3(kBlock), s, es, ee, a); break;
    case OP_MAX: XMPI_LAUNCH((dsync_sched_kernel<T, OP_MAX>), grid, dim3(kBlock), s, es, ee, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace

hipError_t launch_dsync_meet(const DsyncArgs& a, DsyncResolved* out, hipStream_t s) {
  if (a.n < 1 || a.n > kDsyncRanks || a.nseg < 0 || a.nseg > kDsyncRanks || !out) return hipErrorInvalidValue;
  hipLaunchKernelGGL(dsync_meet_kernel, dim3(kXcdBlocks), dim3(64), 0, s, a, out);
  return hipGetLastError();
}

hipError_t launch_dsync_body(const DsyncResolved* res, int nseg, size_t max_packets, int nsrc_hint, int dtype, int op,
                             size_t traffic_bytes, bool sys, hipStream_t s, hipEvent_t es, hipEvent_t ee) {
  if (nseg < 1 || nseg > kDsyncRanks || !res) return hipErrorInvalidValue;
  size_t gx = (max_packets + kBlock - 1) / kBlock;
  if (gx < 1) gx = 1;
  if (gx > 0x3fffffu) gx = 0x3fffffu;
  const dim3 grid((unsigned)gx, (unsigned)nseg);
  // launches that stream more than the caches hold use non-temporal loads and stores (kernels.hip kernel_mode_for)
  const int mode = sys ? 3 : get_kernel_mode() >= 0 ? get_kernel_mode() : (traffic_bytes >= (size_t)(48u << 20) ? 2 : 0);
  switch (dtype) {
    case DT_U8: return body_op<uint8_t>(res, grid, nsrc_hint, op, mode, s, es, ee);
    case DT_I32: return body_op<int32_t>(res, grid, nsrc_hint, op, mode, s, es, ee);
    case DT_I64: return body_op<int64_t>(res, grid, nsrc_hint, op, mode, s, es, ee);
    case DT_F16: return body_op<_Float16>(res, grid, nsrc_hint, op, mode, s, es, ee);
    case DT_F32: return body_op<float>(res, grid, nsrc_hint, op, mode, s, es, ee);
    case DT_F64: return body_op<double>(res, grid, nsrc_hint, op, mode, s, es, ee);
    case DT_BF16: return body_op<bf16_t>(res, grid, nsrc_hint, op, mode, s, es, ee);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_xcc_probe(uint32_t* mask_out, int blocks, hipStream_t s) {
  if (!mask_out || blocks < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(xcc_probe_kernel, dim3((unsigned)blocks), dim3(64), 0, s, mask_out);
  return hipGetLastError();
}

hipError_t launch_flag_selftest(DsyncPage* const* page, int me, int n, uint64_t token, uint64_t spin_limit, const int32_t* abort_word,
                                uint32_t* seen_out, hipStream_t s) {
  if (!page || !seen_out || n < 1 || n > kDsyncRanks || me < 0 || me >= n) return hipErrorInvalidValue;
  FlagSelftestArgs a;
  memset(&a, 0, sizeof a);
  for (int p = 0; p < n; p++) a.page[p] = page[p];
  a.me = me;
  a.n = n;
  a.token = token;
  a.spin_limit = spin_limit;
  a.abort_word = abort_word;
  a.seen = seen_out;
  hipLaunchKernelGGL(flag_selftest_kernel, dim3(1), dim3(64), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_dsync_done(const DsyncArgs& a, const DsyncResolved* res, hipStream_t s) {
  if (!res) return hipErrorInvalidValue;
  // at least one block per XCD (the dispatcher deals consecutive blocks round the 8 XCDs): each writes back the L2 it runs on
  hipLaunchKernelGGL(dsync_done_kernel, dim3(kXcdBlocks), dim3(64), 0, s, a, res);
  return hipGetLastError();
}

hipError_t launch_dsync_sched(const DsyncSchedArgs& a, int dtype, int op, int grid_x, hipStream_t s, hipEvent_t es, hipEvent_t ee) {
  if (a.d.n < 1 || a.d.n > kDsyncRanks || a.nchan < 1 || a.nchan > kMaxSchedChannels || grid_x < 1 ||
      (size_t)grid_x * (size_t)a.nchan > (size_t)kStepSlots)
    return hipErrorInvalidValue;
  const dim3 grid((unsigned)grid_x, (unsigned)a.nchan);
  if (a.sched == SCHED_RING_ALLGATHER || a.sched == SCHED_TREE_BCAST) return sched_op<uint8_t>(a, OP_SUM, grid, s, es, ee);
  switch (dtype) {
    case DT_U8: return sched_op<uint8_t>(a, op, grid, s, es, ee);
    case DT_I32: return sched_op<int32_t>(a, op, grid, s, es, ee);
    case DT_I64: return sched_op<int64_t>(a, op, grid, s, es, ee);
    case DT_F16: return sched_op<_Float16>(a, op, grid, s, es, ee);
    case DT_F32: return sched_op<float>(a, op, grid, s, es, ee);
    case DT_F64: return sched_op<double>(a, op, grid, s, es, ee);
    case DT_BF16: return sched_op<bf16_t>(a, op, grid, s, es, ee);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_sys_copy(void* dst, const void* src, size_t bytes, int grid_x, hipStream_t s) {
  if (!dst || !src || grid_x < 1) return hipErrorInvalidValue;
  SchedMove m;
  memset(&m, 0, sizeof m);
  m.ns = 1;
  m.D = (uint64_t)(uintptr_t)dst;
  m.A = (uint64_t)(uintptr_t)src;
  m.lo = 0;
  m.hi = bytes;
  hipLaunchKernelGGL(sys_copy_kernel, dim3((unsigned)grid_x), dim3(kBlock), 0, s, m);
  return hipGetLastError();
}

hipError_t launch_p2p_pull(const P2PPullArgs& a, int grid_x, hipStream_t s) {
  if (grid_x < 1) grid_x = 1;
  hipLaunchKernelGGL(p2p_pull_kernel, dim3((unsigned)grid_x), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_p2p_agent(const P2PAgentArgs& a, int grid_x, hipStream_t s) {
  if (grid_x < 1) grid_x = 1;
  hipLaunchKernelGGL(p2p_agent_kernel, dim3((unsigned)grid_x), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_p2p_send(const P2PArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(p2p_send_kernel, dim3(1), dim3(64), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_p2p_recv(const P2PArgs& a, int grid_x, hipStream_t s) {
  if (grid_x < 1) grid_x = 1;
  hipLaunchKernelGGL(p2p_recv_kernel, dim3((unsigned)grid_x), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}

}  // namespace xmpi
