// tests/devsim -- the DEVICE-side vocabulary of the kernel sources for a host compiler (TEST INFRASTRUCTURE; see
// hip_runtime_api.h).  mpi_amd/csrc/{kernels,sched,ll}.hip + kdev.h are compiled unchanged as C++:
//
//   * a kernel launch = one OS thread per resident block; the block's lanes are fibers of that thread, switched at
//     __syncthreads() and wherever a lane waits (s_sleep) -- a block of 256 lanes that never waits costs 256 function calls;
//   * __shared__ / XMPI_SHARED = a thread_local of the block's thread;
//   * every system- or agent-scope atomic of the kernels is a C++ atomic (relaxed loads become acquire, relaxed stores
//     release: the sanitizer does not model fences, and what a GPU needs beyond the ORDER of these accesses -- s_waitcnt
//     before a flag, write-through stores, cache maintenance -- cannot be checked on a CPU and stays with the GPU suite);
//   * the written-through data stores of the stepped kernels (st_sys128) are PLAIN stores: a reader that no chain of flag
//     words, barriers and kernel boundaries has ordered behind them is a data race the sanitizer reports.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <tuple>
#include <type_traits>
#include <utility>

#include "hip_runtime_api.h"

#ifndef XMPI_DEVSIM
#error "tests/devsim/include is for -DXMPI_DEVSIM builds only"
#endif

namespace devsim {

struct idx3 {
  unsigned x, y, z;
};
struct LaneRegs {  // the built-in variables of the lane this OS thread is running right now
  idx3 tid, bid, bdim, gdim;
};
extern thread_local LaneRegs tl_lane;

void syncthreads();          // every live lane of the block has arrived
void yield_lane();           // this lane waits: the block's other lanes run, then other threads
void sync_point();           // before every system-scope access: where DEVSIM_FUZZ perturbs the schedule
uint64_t wall_clock_ticks(); // 100 MHz, like wall_clock64()
unsigned xcc_of_block();     // HW_REG_XCC_ID of the block's wave(s): DEVSIM_XCD_MAP
uint64_t shfl_down_bits(uint64_t bits, unsigned delta, unsigned width);  // all live lanes of the block must call it together

struct KernelLaunch {
  dim3 grid, block;
  std::function<void()> lane;  // what ONE lane executes
  const char* name;
  hipEvent_t ev_start, ev_stop;
  uint64_t gen_start = 0, gen_stop = 0;  // which record of the two events this launch is (0: the next one, when it runs)
  unsigned tag = 0;  // kernels whose argument names a schedule and a form (the stepped kernels): 0x100 | schedule << 1 | push
};
void enqueue_kernel(hipStream_t stream, KernelLaunch&& k);

// ---- DEVSIM_CORRUPT_FORM (runtime.cpp): a node that gets ONE kind of data access of ONE kernel wrong -------------------------------
// The data accesses of the kernels that pass through a name a host compiler can redefine -- non-temporal stores and loads (kdev.h
// stp<1> / ldp<2>: the fold kernels, the Receive's copy kernels), the system-scope packet accessors (devsim/sys128.h: the stepped kernels, the split form's system-scope body) and
// the 8-byte system-scope stores (LL lines) -- hand their value to corrupt_bits() first, which flips a bit when the running kernel,
// the kind of access and the owner of the address are what the test asked for.  Off (one load of a flag) otherwise.
enum { ACC_NT_STORE = 1, ACC_SYS_STORE = 2, ACC_SYS_LOAD = 4, ACC_FLAG_STORE = 8, ACC_NT_LOAD = 16 };
extern bool g_corrupt_armed;
void corrupt_bits(const void* addr, void* value, unsigned bytes, int kind);
template <typename T>
inline void nt_store(T* p, T v) {
  if (g_corrupt_armed) corrupt_bits(p, &v, (unsigned)sizeof(T), ACC_NT_STORE);
  *p = v;
}
template <typename T>
inline T nt_load(const T* p) {
  T v = *p;
  if (g_corrupt_armed) corrupt_bits(p, &v, (unsigned)sizeof(T), ACC_NT_LOAD);
  return v;
}
template <typename A>
inline auto sched_tag(const A& a, int) -> decltype((void)a.sched, (void)a.push, 0u) {
  return 0x100u | ((unsigned)a.sched << 1) | ((unsigned)a.push & 1u);
}
template <typename A>
inline unsigned sched_tag(const A&, long) {
  return 0u;
}

template <typename... KArgs, typename... Args>
inline void launch(void (*kern)(KArgs...), dim3 grid, dim3 block, hipStream_t stream, hipEvent_t es, hipEvent_t ee, const char* name,
                   Args&&... args) {
  std::tuple<std::decay_t<KArgs>...> held{std::decay_t<KArgs>(std::forward<Args>(args))...};  // by value, as a dispatch packet holds them
  KernelLaunch k;
  k.grid = grid;
  k.block = block;
  k.name = name;
  k.ev_start = es;
  k.ev_stop = ee;
  k.lane = [kern, held]() { std::apply(kern, held); };
  k.tag = (0u | ... | sched_tag(args, 0));
  enqueue_kernel(stream, std::move(k));
}

// ---- atomics ------------------------------------------------------------------------------------------------------------
constexpr int ld_order(int o) { return o == __ATOMIC_RELAXED ? __ATOMIC_ACQUIRE : o; }
constexpr int st_order(int o) { return o == __ATOMIC_RELAXED ? __ATOMIC_RELEASE : o; }
// build.py --traffic (-DDEVSIM_TRACED): the kernels' plain loads and stores are traced by the compiler; the atomics below are
// kept out of that trace (it sees atomic loads and stores, not read-modify-writes) and reported by hand, every one once
void flag_touch(const void* p, unsigned bytes, int store);
#ifdef DEVSIM_TRACED
#define DEVSIM_FLAG_FN inline __attribute__((noinline, no_sanitize("coverage")))
#define DEVSIM_FLAG_TOUCH(p, bytes, store) ::devsim::flag_touch(p, bytes, store)
#else
#define DEVSIM_FLAG_FN inline
#define DEVSIM_FLAG_TOUCH(p, bytes, store) ((void)0)
#endif
template <typename T>
DEVSIM_FLAG_FN T a_load(const T* p, int order) {
  sync_point();
  DEVSIM_FLAG_TOUCH(p, sizeof(T), 0);
  return __atomic_load_n(p, ld_order(order));
}
template <typename T, typename V>
DEVSIM_FLAG_FN void a_store(T* p, V v, int order) {
  sync_point();
  DEVSIM_FLAG_TOUCH(p, sizeof(T), 1);
  T w = (T)v;
  if (sizeof(T) == 8 && g_corrupt_armed) corrupt_bits(p, &w, 8, ACC_FLAG_STORE);
  __atomic_store_n(p, w, st_order(order));
}
template <typename T, typename V>
DEVSIM_FLAG_FN T a_fetch_add(T* p, V v) {
  sync_point();
  DEVSIM_FLAG_TOUCH(p, sizeof(T), 1);
  return __atomic_fetch_add(p, (T)v, __ATOMIC_SEQ_CST);
}
template <typename T, typename V>
DEVSIM_FLAG_FN T a_fetch_or(T* p, V v) {
  sync_point();
  DEVSIM_FLAG_TOUCH(p, sizeof(T), 1);
  return __atomic_fetch_or(p, (T)v, __ATOMIC_SEQ_CST);
}
template <typename T, typename V>
DEVSIM_FLAG_FN T a_exchange(T* p, V v) {
  sync_point();
  DEVSIM_FLAG_TOUCH(p, sizeof(T), 1);
  return __atomic_exchange_n(p, (T)v, __ATOMIC_SEQ_CST);
}
template <typename T, typename V>
DEVSIM_FLAG_FN T a_fetch_max(T* p, V v) {
  sync_point();
  DEVSIM_FLAG_TOUCH(p, sizeof(T), 1);
  T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while (old < (T)v && !__atomic_compare_exchange_n(p, &old, (T)v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {
  }
  return old;
}
template <typename T, typename V>
DEVSIM_FLAG_FN bool a_cas(T* p, T* expect, V desired) {
  sync_point();
  DEVSIM_FLAG_TOUCH(p, sizeof(T), 1);
  return __atomic_compare_exchange_n(p, expect, (T)desired, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
}
DEVSIM_FLAG_FN double a_add_double(double* p, double v) {
  DEVSIM_FLAG_TOUCH(p, 8, 1);
  uint64_t* q = reinterpret_cast<uint64_t*>(p);
  uint64_t old = __atomic_load_n(q, __ATOMIC_SEQ_CST);
  for (;;) {
    double d;
    memcpy(&d, &old, 8);
    d += v;
    uint64_t nu;
    memcpy(&nu, &d, 8);
    if (__atomic_compare_exchange_n(q, &old, nu, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {
      memcpy(&d, &old, 8);
      return d;
    }
  }
}

}  // namespace devsim

// ---- function / variable qualifiers ---------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local

#define threadIdx (::devsim::tl_lane.tid)
#define blockIdx (::devsim::tl_lane.bid)
#define blockDim (::devsim::tl_lane.bdim)
#define gridDim (::devsim::tl_lane.gdim)

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
  ::devsim::launch(kern, grid, block, stream, nullptr, nullptr, #kern, __VA_ARGS__)

// ---- the four constructs of kdev.h a host compiler cannot take (see there) ------------------------------------------------
#define XMPI_SHARED(T, name) static thread_local T name
#define XMPI_DRAIN() __atomic_thread_fence(__ATOMIC_RELEASE)
#define XMPI_REGS_DEFINED4(a, b, c, d) ((void)0)
#define XMPI_REG_DEFINED(a) ((void)0)

// ---- builtins ---------------------------------------------------------------------------------------------------------------
#ifndef __HIP_MEMORY_SCOPE_SYSTEM
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#endif
#define __hip_atomic_load(p, order, scope) ::devsim::a_load(p, order)
#define __hip_atomic_store(p, v, order, scope) ::devsim::a_store(p, v, order)
#define __hip_atomic_fetch_add(p, v, order, scope) ::devsim::a_fetch_add(p, v)
#define __hip_atomic_fetch_or(p, v, order, scope) ::devsim::a_fetch_or(p, v)
#define __hip_atomic_fetch_max(p, v, order, scope) ::devsim::a_fetch_max(p, v)
#define __hip_atomic_exchange(p, v, order, scope) ::devsim::a_exchange(p, v)
#define __hip_atomic_compare_exchange_strong(p, expect, desired, so, fo, scope) ::devsim::a_cas(p, expect, desired)

#define __builtin_nontemporal_store(v, p) ::devsim::nt_store(p, v)
#define __builtin_nontemporal_load(p) ::devsim::nt_load(p)
#define __builtin_amdgcn_s_sleep(n) ::devsim::yield_lane()
#define __builtin_amdgcn_fence(order, scope) __atomic_thread_fence(order)
#define __builtin_amdgcn_readfirstlane(v) (v)
#define __builtin_amdgcn_s_getreg(r) ::devsim::xcc_of_block()
#define __syncthreads() ::devsim::syncthreads()
#define wall_clock64() ::devsim::wall_clock_ticks()

inline unsigned atomicMax(unsigned* p, unsigned v) { return ::devsim::a_fetch_max(p, v); }
inline unsigned long long atomicMax(unsigned long long* p, unsigned long long v) { return ::devsim::a_fetch_max(p, v); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return ::devsim::a_fetch_add(p, v); }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return ::devsim::a_fetch_add(p, v); }
inline double atomicAdd(double* p, double v) { return ::devsim::a_add_double(p, v); }

inline float __uint_as_float(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}
inline uint32_t __float_as_uint(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
inline long long __double_as_longlong(double d) {
  long long u;
  memcpy(&u, &d, 8);
  return u;
}
inline double __longlong_as_double(long long u) {
  double d;
  memcpy(&d, &u, 8);
  return d;
}
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }

template <typename T>
inline T __shfl_down(T v, unsigned delta, int width = 64) {
  static_assert(sizeof(T) <= 8, "shuffle");
  uint64_t bits = 0;
  memcpy(&bits, &v, sizeof(T));
  bits = ::devsim::shfl_down_bits(bits, delta, (unsigned)width);
  T out;
  memcpy(&out, &bits, sizeof(T));
  return out;
}
