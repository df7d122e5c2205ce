// kdev.h -- device-side building blocks shared by the kernel files (kernels.hip, sched.hip): element / packet
// arithmetic, cache-policy loads and stores, and the rendezvous of the device-synchronised collectives.
// Included by HIP translation units only.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"

// ---- the few constructs a host compiler cannot take, behind names ------------------------------------------------------
// tests/devsim/ compiles THESE kernel sources for the CPU (lanes as threads, under ThreadSanitizer) to race the flag protocols
// of the code that ships instead of a model of them; it defines XMPI_DEVSIM and supplies its own versions of the four macros
// and of the system-scope packet accessors further down.  Nothing else in the kernels knows about it.
#ifndef XMPI_DEVSIM
#define XMPI_SHARED(T, name) __shared__ T name
// this wave's memory operations have completed (stores: acknowledged by where they were written to)
#define XMPI_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define XMPI_REGS_DEFINED4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#define XMPI_REG_DEFINED(a) asm volatile("" : "+v"(a)::"memory")
#endif

namespace xmpi {
namespace {

constexpr int kBlock = 256;      // 4 waves: one per SIMD
constexpr int kUnroll = 4;       // 16-byte packets per lane per operand in flight

enum { DT_U8 = 0, DT_I32 = 1, DT_I64 = 2, DT_F16 = 3, DT_F32 = 4, DT_F64 = 5, DT_BF16 = 6 };
enum { OP_SUM = 0, OP_PROD = 1, OP_MIN = 2, OP_MAX = 3 };

struct bf16_t {
  uint16_t bits;
};

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
// 16-byte packet: a native vector (not HIP's pack_t struct) so it always lives in 4 VGPRs
typedef unsigned int pack_t __attribute__((ext_vector_type(4)));

// ---- scalar combine: one rounding per operation, min/max spelled as in oracle/xmpi_oracle.c ---

template <typename T, int OP>
__device__ __forceinline__ T combine(T a, T b) {
  if constexpr (OP == OP_SUM) return a + b;
  else if constexpr (OP == OP_PROD) return a * b;
  else if constexpr (OP == OP_MIN) return (b < a) ? b : a;
  else return (a < b) ? b : a;
}

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40u);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

template <int OP>
__device__ __forceinline__ bf16_t combine_bf16(bf16_t a, bf16_t b) {
  float x = bf16_to_f32(a.bits), y = bf16_to_f32(b.bits);
  if constexpr (OP == OP_SUM) return bf16_t{f32_to_bf16(x + y)};
  else if constexpr (OP == OP_PROD) return bf16_t{f32_to_bf16(x * y)};
  else if constexpr (OP == OP_MIN) return (y < x) ? b : a;
  else return (x < y) ? b : a;
}

// wrapping integer arithmetic (Go semantics) without signed-overflow UB
template <int OP>
__device__ __forceinline__ int32_t combine_i32(int32_t a, int32_t b) {
  if constexpr (OP == OP_SUM) return (int32_t)((uint32_t)a + (uint32_t)b);
  else if constexpr (OP == OP_PROD) return (int32_t)((uint32_t)a * (uint32_t)b);
  else return combine<int32_t, OP>(a, b);
}
template <int OP>
__device__ __forceinline__ int64_t combine_i64(int64_t a, int64_t b) {
  if constexpr (OP == OP_SUM) return (int64_t)((uint64_t)a + (uint64_t)b);
  else if constexpr (OP == OP_PROD) return (int64_t)((uint64_t)a * (uint64_t)b);
  else return combine<int64_t, OP>(a, b);
}

template <typename T, int OP>
__device__ __forceinline__ T combine_any(T a, T b) {
  if constexpr (sizeof(T) == 2 && !__is_same(T, _Float16)) return combine_bf16<OP>(a, b);
  else if constexpr (__is_same(T, int32_t)) return combine_i32<OP>(a, b);
  else if constexpr (__is_same(T, int64_t)) return combine_i64<OP>(a, b);
  else if constexpr (__is_same(T, uint8_t)) {
    if constexpr (OP == OP_SUM) return (uint8_t)(a + b);
    else if constexpr (OP == OP_PROD) return (uint8_t)(a * b);
    else return combine<uint8_t, OP>(a, b);
  } else return combine<T, OP>(a, b);
}

// ---- 16-byte packet combine ------------------------------------------------------------------

template <typename T, int OP>
__device__ __forceinline__ pack_t combine16(pack_t a, pack_t b) {
  constexpr int N = 16 / sizeof(T);
  if constexpr (__is_same(T, _Float16) && (OP == OP_SUM || OP == OP_PROD)) {
    // packed halves: v_pk_add_f16 / v_pk_mul_f16, 2 elements per VALU lane-op
    union { pack_t u; half2_t h[4]; } x, y, r;
    x.u = a; y.u = b;
#pragma unroll
    for (int i = 0; i < 4; i++) r.h[i] = (OP == OP_SUM) ? (x.h[i] + y.h[i]) : (x.h[i] * y.h[i]);
    return r.u;
  } else {
    union { pack_t u; T e[N]; } x, y, r;
    x.u = a; y.u = b;
#pragma unroll
    for (int i = 0; i < N; i++) r.e[i] = combine_any<T, OP>(x.e[i], y.e[i]);
    return r.u;
  }
}

// ---- cache policy of the streaming accesses ----------------------------------------------------
// MODE 0: plain loads and stores.  MODE 1: non-temporal loads and stores.  MODE 2: non-temporal
// loads, plain stores.  Data that is touched once should not displace lines in L2 / the Infinity
// Cache: measured on MI355X at 256 MiB operands (beyond the 256 MiB Infinity Cache) nt loads take
// reduce2<f32> from 5.45 to 6.6-6.8 TB/s; which mode wins at a given size is a launch-time choice
// (set_kernel_mode / XMPI_KERNEL_MODE), the arithmetic is identical in all three.
template <int MODE>
__device__ __forceinline__ pack_t ldp(const pack_t* p) {
  if constexpr (MODE != 0) return __builtin_nontemporal_load(p);
  else return *p;
}
template <int MODE>
__device__ __forceinline__ void stp(pack_t* p, pack_t v) {
  if constexpr (MODE == 1) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// ---- device-synchronised collectives: system-scope flag words, rendezvous, completion --------------------
__device__ __forceinline__ uint64_t ld_sys64(const uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_sys64(uint64_t* p, uint64_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ uint64_t uniform64(uint64_t v) {  // the value is the same in every lane: keep it in SGPRs
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

// the XCD (accelerator complex die) this wave runs on: HW_REG_XCC_ID (hardware register 20), bits 3..0
__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 15u; }

// Wait until *p >= want.  0 = it did; otherwise why not (the job was aborted / the wait outlasted spin_limit).
__device__ uint32_t spin_until(const uint64_t* p, uint64_t want, const int32_t* abort_word, uint64_t spin_limit) {
  if (ld_sys64(p) >= want) return DSYNC_OK;
  const uint64_t t0 = wall_clock64();
  for (uint32_t k = 1;; k++) {
    __builtin_amdgcn_s_sleep(1);
    if (ld_sys64(p) >= want) return DSYNC_OK;
    if ((k & 127u) == 0) {  // the expensive checks (a load over PCIe, the clock) now and then only
      if (abort_word && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return DSYNC_ABORTED;
      if (spin_limit && wall_clock64() - t0 > spin_limit) return DSYNC_TIMEOUT;
    }
  }
}
__device__ __forceinline__ uint32_t dsync_spin(const uint64_t* p, uint64_t want, const DsyncArgs& a) {
  return spin_until(p, want, a.abort_word, a.spin_limit);
}

struct DsyncShared {
  uint64_t epoch;                                 // of this kernel
  uint64_t send[kDsyncRanks], recv[kDsyncRanks];  // every rank's buffers as addressable from here
  uint64_t land[kDsyncRanks];                     // ... and its landing block (push forms of the stepped kernels); 0 = it lends none
  uint64_t src[kDsyncRanks], dst[kDsyncRanks];    // this block's segment: sources in rank order, destinations local first
  int nsrc, ndst;
  uint32_t fail, last;
};

// Where this process mapped registration `gen` (table slot `slot`) of `peer`: from the cache in this rank's page, or
// -- the first time, and after the peer re-used the slot -- from the host's table (a few loads over PCIe).
__device__ uint64_t translate(uint64_t comm_tag, const DsyncEntry* table, DsyncPage* mine, int peer, uint64_t slot, uint64_t gen, uint64_t off) {
  if (gen == 0 || slot >= (uint64_t)kDsyncArenas) return 0;
  DsyncEntry* c = &mine->cache[peer][slot];
  uint64_t base, bytes;
  if (__hip_atomic_load(&c->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == gen &&
      __hip_atomic_load(&c->tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == comm_tag) {
    base = __hip_atomic_load(&c->base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bytes = __hip_atomic_load(&c->bytes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    const DsyncEntry* h = &table[peer * kDsyncArenas + (int)slot];
    if (__hip_atomic_load(&h->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != gen) return 0;
    base = __hip_atomic_load(&h->base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    bytes = __hip_atomic_load(&h->bytes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // several blocks may do this at once: they all write the same values, the number last
    __hip_atomic_store(&c->gen, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_store(&c->base, base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&c->bytes, bytes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&c->tag, comm_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&c->gen, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  return (off <= bytes) ? base + off : 0;
}

__device__ __forceinline__ uint64_t dsync_translate(const DsyncArgs& a, DsyncPage* mine, int peer, uint64_t slot, uint64_t gen, uint64_t off) {
  return translate(a.tag, a.table, mine, peer, slot, gen, off);
}

__device__ void dsync_begin(const DsyncArgs& a, DsyncShared& sh) {
  const int t = threadIdx.x, me = a.me, n = a.n;
  DsyncPage* mine = a.page[me];
  if (t == 0) {
    sh.fail = DSYNC_OK;
    sh.send[me] = (uint64_t)(uintptr_t)a.my_send;
    sh.recv[me] = (uint64_t)(uintptr_t)a.my_recv;
    sh.land[me] = (uint64_t)(uintptr_t)a.my_land;
    // the epoch is counted on the device (every block reads the same value: only the closing block of a kernel
    // advances it, after all the others have finished), so replaying a captured launch counts on
    const uint64_t seen = ld_sys64(&mine->epoch_now);
    sh.epoch = (seen > a.epoch_floor ? seen : a.epoch_floor) + 1;
  }
  __syncthreads();
  const uint64_t epoch = sh.epoch;
  if (t < n && t != me) {
    if (blockIdx.x == 0 && blockIdx.y == 0) {  // one block announces this rank
      DsyncSlot* out = &a.page[t]->ready[me];
      st_sys64(&out->send_gen, a.send_gen);
      st_sys64(&out->send_off, a.send_off);
      st_sys64(&out->recv_gen, a.recv_gen);
      st_sys64(&out->recv_off, a.recv_off);
      st_sys64(&out->slots, a.send_slot | (a.recv_slot << 8) | (a.land_slot << 16) | (a.land_gen ? 1ull << 24 : 0) | (a.sig << 32));
      if (a.land_gen) {  // (bit 24 of `slots` says whether these two mean anything)
        st_sys64(&out->land_gen, a.land_gen);
        st_sys64(&out->land_off, a.land_off);
      }
      __hip_atomic_store(&out->epoch, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const DsyncSlot* in = &mine->ready[t];
    uint32_t why = dsync_spin(&in->epoch, epoch, a);
    if (why == DSYNC_OK) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
      const uint64_t slots = ld_sys64(&in->slots), lgen = (slots >> 24 & 1u) ? ld_sys64(&in->land_gen) : 0;
      const uint64_t s = dsync_translate(a, mine, t, slots & 0xffu, ld_sys64(&in->send_gen), ld_sys64(&in->send_off));
      const uint64_t r = dsync_translate(a, mine, t, (slots >> 8) & 0xffu, ld_sys64(&in->recv_gen), ld_sys64(&in->recv_off));
      const uint64_t l = lgen ? dsync_translate(a, mine, t, (slots >> 16) & 0xffu, lgen, ld_sys64(&in->land_off)) : 0;
      sh.send[t] = s;
      sh.recv[t] = r;
      sh.land[t] = l;
      if (!s || !r || (lgen && !l)) why = DSYNC_UNMAPPED;
      // the peer is in another call (another collective, schedule, length, type, operation or root): nobody moves anything
      if ((uint32_t)a.sig && (uint32_t)(slots >> 32) && (uint32_t)(slots >> 32) != (uint32_t)a.sig) why = DSYNC_MISMATCH;
    }
    if (why != DSYNC_OK) atomicMax(&sh.fail, why);
  }
  // every lane acquires: what the peers wrote before they announced themselves (their inputs) must not be
  // served from a stale line of this CU's L1 / this XCD's L2
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  __syncthreads();
}

// The block that finishes last exchanges "done" with every peer.  xcd_guard (the done kernel of the split form): the masks of
// XCDs the meet and the done kernel's blocks ran on are collected, reported (status[6], status[7]) and -- xcc_need -- checked.
__device__ void dsync_end(const DsyncArgs& a, DsyncShared& sh, bool xcd_guard = false) {
  const int t = threadIdx.x, me = a.me, n = a.n;
  DsyncPage* mine = a.page[me];
  XMPI_DRAIN();  // this wave's stores have left
  __syncthreads();
  if (t == 0) {
    // a block that gave up (its own wait timed out, the job was aborted) says so where the closing block looks:
    // a kernel whose closing block happened to see nothing wrong must not report success for it
    if (sh.fail != DSYNC_OK) __hip_atomic_fetch_max(&mine->failword, sh.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // ... and are visible system-wide (writes back this XCD's L2)
    const uint32_t total = gridDim.x * gridDim.y;
    sh.last = (__hip_atomic_fetch_add(&mine->ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == total - 1) ? 1u : 0u;
  }
  __syncthreads();
  if (!sh.last) return;
  if (t == 0) {
    __hip_atomic_store(&mine->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t f = __hip_atomic_exchange(&mine->failword, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (f > sh.fail) sh.fail = f;
    if (xcd_guard) {
      const uint32_t m = __hip_atomic_exchange(&mine->xcc_meet, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t d = __hip_atomic_exchange(&mine->xcc_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a.status) {
        __hip_atomic_store(a.status + 6, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(a.status + 7, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      if (a.xcc_need > 0 && (__builtin_popcount(m) < a.xcc_need || __builtin_popcount(d) < a.xcc_need) && sh.fail == DSYNC_OK)
        sh.fail = DSYNC_XCD;
    }
  }
  __syncthreads();
  uint32_t why = DSYNC_OK;
  if (t < n && t != me) {
    __hip_atomic_store(&a.page[t]->done[me][0], sh.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (sh.fail == DSYNC_OK) why = dsync_spin(&mine->done[t][0], sh.epoch, a);
    if (why != DSYNC_OK) atomicMax(&sh.fail, why);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  __syncthreads();
  if (t == 0) {
    st_sys64(&mine->epoch_now, sh.epoch);  // this kernel is over: the next one (an ordinary launch or a graph replay) counts from here
    if (a.host_epoch) st_sys64(a.host_epoch, sh.epoch);
    if (sh.fail != DSYNC_OK && a.status) __hip_atomic_store(a.status, sh.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // the blocking caller's word, last: every rank's stores into this rank's buffers were released before its "done"
    if (a.host_done) __hip_atomic_store(a.host_done, a.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}


// ---- data that another rank's kernel writes WHILE this kernel runs (the steps of ring / halving / tree) -------------
// 16 bytes straight from / to memory at system scope (sc0 sc1): a load is never served from a line of this XCD's L2 (or
// this CU's L1) that predates the peer's store, a store is written through -- so a step needs no cache maintenance at all:
// the writer waits for its stores' acknowledgements (s_waitcnt vmcnt(0)) and raises the flag, the reader sees the flag and
// loads.  (The alternative -- ordinary accesses bracketed by release / acquire fences -- writes back and invalidates a
// whole L2 per block per step: 2.64 ms for the 8 x 256 MiB ring where the data alone needs 1.7, r03 session 1.)
// The loads are asynchronous inline assembly: sys128_wait() is what makes their results usable.
#ifndef XMPI_DEVSIM
__device__ __forceinline__ void ld_sys128_issue(pack_t& v, const pack_t* p) {
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=&v"(v) : "v"(p) : "memory");
}
__device__ __forceinline__ void st_sys128(pack_t* p, pack_t v) {
  // (s_nop: the two wait states the hardware needs before the data registers of a store of more than 8 bytes may be
  // overwritten -- the compiler inserts them for its own stores, it cannot see into this one)
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
template <int U>
__device__ __forceinline__ void sys128_wait(pack_t (&v)[U]) {
  static_assert(U == 1 || U == 2 || U == 4, "unroll");
  if constexpr (U == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0])::"memory");
  else if constexpr (U == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1])::"memory");
  else asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])::"memory");
}
#else
#include <devsim/sys128.h>  // tests/devsim: the same three accessors for a host compiler
#endif
// any number of packets: one wait, then every register passes through an (empty) volatile statement behind it
template <int U>
__device__ __forceinline__ void sys128_wait_n(pack_t* v) {
  XMPI_DRAIN();
#pragma unroll
  for (int k = 0; k < U; k++) XMPI_REG_DEFINED(v[k]);
  (void)v;
}
// one element, same scope (the ragged ends of a tile, buffers at odd alignments)
template <typename T>
__device__ __forceinline__ T ld_sys_elem(const T* p) {
  T out;
  if constexpr (sizeof(T) == 1) {
    const uint8_t u = __hip_atomic_load(reinterpret_cast<const uint8_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_memcpy(&out, &u, 1);
  } else if constexpr (sizeof(T) == 2) {
    const uint16_t u = __hip_atomic_load(reinterpret_cast<const uint16_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_memcpy(&out, &u, 2);
  } else if constexpr (sizeof(T) == 4) {
    const uint32_t u = __hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_memcpy(&out, &u, 4);
  } else {
    const uint64_t u = __hip_atomic_load(reinterpret_cast<const uint64_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_memcpy(&out, &u, 8);
  }
  return out;
}
template <typename T>
__device__ __forceinline__ void st_sys_elem(T* p, T v) {
  if constexpr (sizeof(T) == 1) {
    uint8_t u;
    __builtin_memcpy(&u, &v, 1);
    __hip_atomic_store(reinterpret_cast<uint8_t*>(p), u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  } else if constexpr (sizeof(T) == 2) {
    uint16_t u;
    __builtin_memcpy(&u, &v, 2);
    __hip_atomic_store(reinterpret_cast<uint16_t*>(p), u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  } else if constexpr (sizeof(T) == 4) {
    uint32_t u;
    __builtin_memcpy(&u, &v, 4);
    __hip_atomic_store(reinterpret_cast<uint32_t*>(p), u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  } else {
    uint64_t u;
    __builtin_memcpy(&u, &v, 8);
    __hip_atomic_store(reinterpret_cast<uint64_t*>(p), u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// dst = src, all blocks of the grid together: 16 KiB tiles, four 16-byte loads per lane in flight (a message is pulled
// over a link: round trips), bytes at odd alignments one by one
__device__ __forceinline__ void copy_span(char* dst, const char* src, size_t bytes, unsigned bid = blockIdx.x, unsigned nb = gridDim.x) {
  const int t = threadIdx.x;
  if ((((uintptr_t)src | (uintptr_t)dst) & 15u) == 0) {
    const size_t npack = bytes / 16;
    constexpr size_t kTile = (size_t)kBlock * 4;
    const pack_t* ps = reinterpret_cast<const pack_t*>(src);
    pack_t* pd = reinterpret_cast<pack_t*>(dst);
    for (size_t base = (size_t)bid * kTile; base < npack; base += (size_t)nb * kTile) {
      if (base + kTile <= npack) {
        pack_t v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = ldp<2>(ps + base + (size_t)u * kBlock + t);
#pragma unroll
        for (int u = 0; u < 4; u++) pd[base + (size_t)u * kBlock + t] = v[u];
      } else {
        for (size_t i = base + t; i < npack; i += kBlock) pd[i] = ldp<2>(ps + i);
      }
    }
    if (bid == 0 && npack * 16 + t < bytes) dst[npack * 16 + t] = src[npack * 16 + t];
  } else {
    for (size_t i = (size_t)bid * kBlock + t; i < bytes; i += (size_t)nb * kBlock) dst[i] = src[i];
  }
}

__device__ __forceinline__ uint64_t* step_flags(DsyncPage* page) {
  return reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(page) + kStepOff);
}

}  // namespace
}  // namespace xmpi
