// kernels.hip -- gfx950 (CDNA4, wave64) kernels of the collectives hot path.
//
// The reference has no kernels at all (it is pure Go).  These replace what a
// reference user does on the host after exchanging buffers with Send/Receive
// (examples/helloworld/helloworld.go:53-81): the elementwise combine of two (ring / halving
// step) or N (full-mesh step) rank buffers, the copy out of a receive window, and the
// bytes.Equal / floats.Equal checks of examples/bounce/bounce.go:105,133.
//
// All of them are HBM-bound streaming kernels (1 flop per 12 bytes for an f32 add), so the
// design rules are the memory ones: 16 B per lane per access (global_load_dwordx4, 1 KiB per
// wave instruction), every load of an unrolled tile issued before the first use, <=64 VGPRs
// so 8 waves/SIMD stay resident, one tile per block by default (the dispatcher balances the grid
// over the XCDs; `grid_cap` > 0 bounds the grid and the kernels loop grid-stride).  No MFMA
// (nothing here is a contraction) and no LDS in the streaming kernels (no cross-lane reuse);
// LDS + wavefront shuffles are used where a cross-lane reduction really exists: the
// verification kernels at the bottom.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "kdev.h"
#include "kernels.h"

namespace xmpi {
namespace {

// ---- dst = a op b  (the per-chunk reduction of every ring / halving step) --------------------
// Lane l of block B touches packets  base + k*256 + l  (k < kUnroll): each wave instruction
// covers one contiguous KiB; 2 x kUnroll loads are in flight before the first VALU op.

template <typename T, int OP, int MODE>
__global__ __launch_bounds__(kBlock) void reduce2_kernel(T* dst, const T* a, const T* b,
                                                         size_t npack, size_t count) {
  const pack_t* pa = reinterpret_cast<const pack_t*>(a);
  const pack_t* pb = reinterpret_cast<const pack_t*>(b);
  pack_t* pd = reinterpret_cast<pack_t*>(dst);
  constexpr size_t kTile = (size_t)kBlock * kUnroll;
  const size_t stride = (size_t)gridDim.x * kTile;
  // wave w of the block owns kUnroll consecutive KiB: lane l reads packets first + k*64, so the
  // k-th access is the same address register + an immediate offset of k KiB
  const size_t lane_off = (size_t)(threadIdx.x >> 6) * (64 * kUnroll) + (threadIdx.x & 63);
  for (size_t base = (size_t)blockIdx.x * kTile; base < npack; base += stride) {
    const size_t first = base + lane_off;
    if (base + kTile <= npack) {  // full tile: every load issued before the first use
      pack_t va[kUnroll], vb[kUnroll];
#pragma unroll
      for (int k = 0; k < kUnroll; k++) {
        va[k] = ldp<MODE>(pa + first + k * 64);
        vb[k] = ldp<MODE>(pb + first + k * 64);
      }
#pragma unroll
      for (int k = 0; k < kUnroll; k++) stp<MODE>(pd + first + k * 64, combine16<T, OP>(va[k], vb[k]));
    } else {  // last, partial tile of the buffer
      for (int k = 0; k < kUnroll; k++) {
        const size_t i = first + k * 64;
        if (i < npack) pd[i] = combine16<T, OP>(pa[i], pb[i]);
      }
    }
  }
  // ragged tail (< 16 bytes): the first lanes of block 0
  constexpr size_t N = 16 / sizeof(T);
  const size_t done = npack * N;
  if (blockIdx.x == 0 && done + threadIdx.x < count) {
    const size_t i = done + threadIdx.x;
    dst[i] = combine_any<T, OP>(a[i], b[i]);
  }
}

// Several independent reductions in ONE launch (the ring channels of one step): blockIdx.y picks the
// segment.  Same tile body as reduce2_kernel.  Each segment may store its result twice: dst is the
// local copy (null for the partial sums of a fused ring step, which are only forwarded) and dst2 the
// next rank's receive slot (xGMI write) -- receive-reduce-send in one pass over the data.
struct Reduce2Batch {
  void* dst[kMaxBatch];
  void* dst2[kMaxBatch];
  const void* a[kMaxBatch];
  const void* b[kMaxBatch];
  size_t count[kMaxBatch];
};

template <typename T, int OP, int MODE>
__global__ __launch_bounds__(kBlock) void reduce2_batch_kernel(Reduce2Batch q) {
  const int j = blockIdx.y;
  T* dst = reinterpret_cast<T*>(q.dst[j]);
  T* dst2 = reinterpret_cast<T*>(q.dst2[j]);
  const T* a = reinterpret_cast<const T*>(q.a[j]);
  const T* b = reinterpret_cast<const T*>(q.b[j]);
  constexpr size_t N = 16 / sizeof(T);
  const size_t count = q.count[j], npack = count / N;
  const pack_t* pa = reinterpret_cast<const pack_t*>(a);
  const pack_t* pb = reinterpret_cast<const pack_t*>(b);
  pack_t* pd = reinterpret_cast<pack_t*>(dst);
  pack_t* pd2 = reinterpret_cast<pack_t*>(dst2);
  constexpr size_t kTile = (size_t)kBlock * kUnroll;
  const size_t stride = (size_t)gridDim.x * kTile;
  const size_t lane_off = (size_t)(threadIdx.x >> 6) * (64 * kUnroll) + (threadIdx.x & 63);
  for (size_t base = (size_t)blockIdx.x * kTile; base < npack; base += stride) {
    const size_t first = base + lane_off;
    if (base + kTile <= npack) {
      pack_t va[kUnroll], vb[kUnroll];
#pragma unroll
      for (int k = 0; k < kUnroll; k++) {
        va[k] = ldp<MODE>(pa + first + k * 64);
        vb[k] = ldp<MODE>(pb + first + k * 64);
      }
#pragma unroll
      for (int k = 0; k < kUnroll; k++) va[k] = combine16<T, OP>(va[k], vb[k]);
      if (pd) {
#pragma unroll
        for (int k = 0; k < kUnroll; k++) stp<MODE>(pd + first + k * 64, va[k]);
      }
      if (pd2) {
#pragma unroll
        for (int k = 0; k < kUnroll; k++) stp<MODE>(pd2 + first + k * 64, va[k]);
      }
    } else {
      for (int k = 0; k < kUnroll; k++) {
        const size_t i = first + k * 64;
        if (i < npack) {
          const pack_t v = combine16<T, OP>(pa[i], pb[i]);
          if (pd) pd[i] = v;
          if (pd2) pd2[i] = v;
        }
      }
    }
  }
  const size_t done = npack * N;
  if (blockIdx.x == 0 && done + threadIdx.x < count) {
    const size_t i = done + threadIdx.x;
    const T v = combine_any<T, OP>(a[i], b[i]);
    if (dst) dst[i] = v;
    if (dst2) dst2[i] = v;
  }
}

// any alignment: one element per lane per iteration (correctness path for odd offsets)
template <typename T, int OP>
__global__ __launch_bounds__(kBlock) void reduce2_elem_kernel(T* dst, const T* a, const T* b,
                                                              size_t count) {
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += stride)
    dst[i] = combine_any<T, OP>(a[i], b[i]);
}

// ---- dst = ((s0 op s1) op s2) ...  strictly left to right (full-mesh step, rank order) -------

struct SrcPtrs {
  const void* p[kMaxReduceSrcs];
};

template <typename T, int OP, int NSRC, int MODE>
__global__ __launch_bounds__(kBlock) void reduce_n_kernel(T* dst, SrcPtrs srcs, int nsrc_rt,
                                                          size_t npack, size_t count) {
  const int nsrc = (NSRC > 0) ? NSRC : nsrc_rt;
  pack_t* pd = reinterpret_cast<pack_t*>(dst);
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < npack; i += stride) {
    if constexpr (NSRC > 0) {
      pack_t v[NSRC];
#pragma unroll
      for (int s = 0; s < NSRC; s++) v[s] = ldp<MODE>(reinterpret_cast<const pack_t*>(srcs.p[s]) + i);
      pack_t acc = v[0];
#pragma unroll
      for (int s = 1; s < NSRC; s++) acc = combine16<T, OP>(acc, v[s]);
      stp<MODE>(pd + i, acc);
    } else {
      pack_t acc = reinterpret_cast<const pack_t*>(srcs.p[0])[i];
      for (int s = 1; s < nsrc; s++)
        acc = combine16<T, OP>(acc, reinterpret_cast<const pack_t*>(srcs.p[s])[i]);
      pd[i] = acc;
    }
  }
  constexpr size_t N = 16 / sizeof(T);
  const size_t done = npack * N;
  if (blockIdx.x == 0 && done + threadIdx.x < count) {
    const size_t i = done + threadIdx.x;
    T acc = reinterpret_cast<const T*>(srcs.p[0])[i];
    for (int s = 1; s < nsrc; s++) acc = combine_any<T, OP>(acc, reinterpret_cast<const T*>(srcs.p[s])[i]);
    dst[i] = acc;
  }
}

// ---- zero-copy collective kernels: fold N rank buffers once, store the result M times ----------
// The sources are the ranks' own send buffers (one local, the others read over xGMI), the
// destinations the ranks' receive buffers (one local, the others written over xGMI): rank j runs
// this on chunk j, which is the whole allreduce for that chunk -- N reads + N writes per element and
// no staging copy anywhere.  Same lane reads element i of every source before it stores element i
// anywhere, so a destination may alias a source (in-place collectives).

struct MultiPtrs {
  const void* src[kMaxReduceSrcs];
  void* dst[kMaxReduceSrcs];
};

// The destination loop is unrolled over the maximum with a uniform guard, so all destination pointers
// sit in SGPRs before the loop (a runtime-indexed kernarg array costs a scalar load + wait per store).
// Result data is written once and not read again by this GPU: non-temporal stores (MODE != 0).
template <typename T, int OP, int NSRC, int MODE>
__global__ __launch_bounds__(kBlock) void reduce_n_multi_kernel(MultiPtrs q, int nsrc_rt, int ndst,
                                                                size_t npack, size_t count) {
  const int nsrc = (NSRC > 0) ? NSRC : nsrc_rt;
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < npack; i += stride) {
    pack_t acc;
    if constexpr (NSRC > 0) {
      pack_t v[NSRC];
#pragma unroll
      for (int s = 0; s < NSRC; s++) v[s] = ldp<MODE>(reinterpret_cast<const pack_t*>(q.src[s]) + i);
      acc = v[0];
#pragma unroll
      for (int s = 1; s < NSRC; s++) acc = combine16<T, OP>(acc, v[s]);
    } else {
      acc = reinterpret_cast<const pack_t*>(q.src[0])[i];
      for (int s = 1; s < nsrc; s++) acc = combine16<T, OP>(acc, reinterpret_cast<const pack_t*>(q.src[s])[i]);
    }
#pragma unroll
    for (int k = 0; k < kMaxReduceSrcs; k++)
      if (k < ndst) stp<(MODE != 0) ? 1 : 0>(reinterpret_cast<pack_t*>(q.dst[k]) + i, acc);
  }
  constexpr size_t N = 16 / sizeof(T);
  const size_t done = npack * N;
  if (blockIdx.x == 0 && done + threadIdx.x < count) {
    const size_t i = done + threadIdx.x;
    T acc = reinterpret_cast<const T*>(q.src[0])[i];
    for (int s = 1; s < nsrc; s++) acc = combine_any<T, OP>(acc, reinterpret_cast<const T*>(q.src[s])[i]);
    for (int k = 0; k < ndst; k++) reinterpret_cast<T*>(q.dst[k])[i] = acc;
  }
}

// The fold above with the arithmetic taken out: dst[k][i] = src[k][i] for the same N sources and N destinations -- the same
// loads, the same stores, the same grid, the same cache policy, no adds.  What THIS box's memory system gives the fold's access
// pattern on THESE buffers: bench.py runs it right behind the timed region (roofline.box_copy_us) so that a reader can tell the
// box from the code (the same binary ran the fold in 669 ... 763 us on different boxes).
template <int NSRC, int MODE>
__global__ __launch_bounds__(kBlock) void copy_pairs_kernel(MultiPtrs q, int n_rt, size_t npack, size_t bytes) {
  const int n = (NSRC > 0) ? NSRC : n_rt;
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < npack; i += stride) {
    if constexpr (NSRC > 0) {
      pack_t v[NSRC];
#pragma unroll
      for (int s = 0; s < NSRC; s++) v[s] = ldp<MODE>(reinterpret_cast<const pack_t*>(q.src[s]) + i);
#pragma unroll
      for (int k = 0; k < NSRC; k++) stp<(MODE != 0) ? 1 : 0>(reinterpret_cast<pack_t*>(q.dst[k]) + i, v[k]);
    } else {
      for (int k = 0; k < n; k++) reinterpret_cast<pack_t*>(q.dst[k])[i] = reinterpret_cast<const pack_t*>(q.src[k])[i];
    }
  }
  const size_t done = npack * 16;
  if (blockIdx.x == 0 && done + threadIdx.x < bytes)
    for (int k = 0; k < n; k++) reinterpret_cast<uint8_t*>(q.dst[k])[done + threadIdx.x] = reinterpret_cast<const uint8_t*>(q.src[k])[done + threadIdx.x];
}

// any alignment: one element per lane per iteration
template <typename T, int OP>
__global__ __launch_bounds__(kBlock) void reduce_n_multi_elem_kernel(MultiPtrs q, int nsrc, int ndst, size_t count) {
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += stride) {
    T acc = reinterpret_cast<const T*>(q.src[0])[i];
    for (int s = 1; s < nsrc; s++) acc = combine_any<T, OP>(acc, reinterpret_cast<const T*>(q.src[s])[i]);
    for (int k = 0; k < ndst; k++) reinterpret_cast<T*>(q.dst[k])[i] = acc;
  }
}

// every dst[k] = src[0]: one read, ndst writes (allgather / bcast push to all peers at once)
template <int MODE>
__global__ __launch_bounds__(kBlock) void copy_multi_kernel(MultiPtrs q, int ndst, size_t npack, size_t bytes) {
  const pack_t* src = reinterpret_cast<const pack_t*>(q.src[0]);
  constexpr size_t kTile = (size_t)kBlock * kUnroll;
  const size_t stride = (size_t)gridDim.x * kTile;
  const size_t lane_off = (size_t)(threadIdx.x >> 6) * (64 * kUnroll) + (threadIdx.x & 63);
  for (size_t base = (size_t)blockIdx.x * kTile; base < npack; base += stride) {
    const size_t first = base + lane_off;
    if (base + kTile <= npack) {
      pack_t v[kUnroll];
#pragma unroll
      for (int k = 0; k < kUnroll; k++) v[k] = ldp<MODE>(src + first + k * 64);
#pragma unroll
      for (int d = 0; d < kMaxReduceSrcs; d++)
        if (d < ndst) {
          pack_t* pd = reinterpret_cast<pack_t*>(q.dst[d]);
#pragma unroll
          for (int k = 0; k < kUnroll; k++) stp<(MODE != 0) ? 1 : 0>(pd + first + k * 64, v[k]);
        }
    } else {
      for (int k = 0; k < kUnroll; k++) {
        const size_t i = first + k * 64;
        if (i < npack) {
          const pack_t v = src[i];
          for (int d = 0; d < ndst; d++) reinterpret_cast<pack_t*>(q.dst[d])[i] = v;
        }
      }
    }
  }
  const size_t done = npack * 16;
  if (blockIdx.x == 0 && done + threadIdx.x < bytes) {
    const size_t i = done + threadIdx.x;
    const uint8_t v = reinterpret_cast<const uint8_t*>(src)[i];
    for (int d = 0; d < ndst; d++) reinterpret_cast<uint8_t*>(q.dst[d])[i] = v;
  }
}

__global__ __launch_bounds__(kBlock) void copy_multi_elem_kernel(MultiPtrs q, int ndst, size_t bytes) {
  const uint8_t* src = reinterpret_cast<const uint8_t*>(q.src[0]);
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < bytes; i += stride) {
    const uint8_t v = src[i];
    for (int d = 0; d < ndst; d++) reinterpret_cast<uint8_t*>(q.dst[d])[i] = v;
  }
}

// ---- streaming copy --------------------------------------------------------------------------

template <int MODE>
__global__ __launch_bounds__(kBlock) void copy16_kernel(pack_t* dst, const pack_t* src, size_t npack,
                                                        size_t bytes) {
  constexpr size_t kTile = (size_t)kBlock * kUnroll;
  const size_t stride = (size_t)gridDim.x * kTile;
  const size_t lane_off = (size_t)(threadIdx.x >> 6) * (64 * kUnroll) + (threadIdx.x & 63);
  for (size_t base = (size_t)blockIdx.x * kTile; base < npack; base += stride) {
    const size_t first = base + lane_off;
    if (base + kTile <= npack) {
      pack_t v[kUnroll];
#pragma unroll
      for (int k = 0; k < kUnroll; k++) v[k] = ldp<MODE>(src + first + k * 64);
#pragma unroll
      for (int k = 0; k < kUnroll; k++) stp<MODE>(dst + first + k * 64, v[k]);
    } else {
      for (int k = 0; k < kUnroll; k++) {
        const size_t i = first + k * 64;
        if (i < npack) dst[i] = src[i];
      }
    }
  }
  const size_t done = npack * 16;
  if (blockIdx.x == 0 && done + threadIdx.x < bytes) {
    const size_t i = done + threadIdx.x;
    reinterpret_cast<uint8_t*>(dst)[i] = reinterpret_cast<const uint8_t*>(src)[i];
  }
}

// Several independent copies in ONE launch: blockIdx.y picks the copy, blockIdx.x strides over its
// tiles.  This is how a rank pushes a piece to all of its peers at once (each destination is a
// different xGMI link, all driven from one grid) and how it drains the slots of all peers at once.
struct CopyBatch {
  void* dst[kMaxBatch];
  void* dst2[kMaxBatch];  // optional second destination (receive-copy-send: store locally and forward)
  const void* src[kMaxBatch];
  size_t bytes[kMaxBatch];
};

template <int MODE>
__global__ __launch_bounds__(kBlock) void copy_batch_kernel(CopyBatch b) {
  const int j = blockIdx.y;
  pack_t* dst = reinterpret_cast<pack_t*>(b.dst[j]);
  pack_t* dst2 = reinterpret_cast<pack_t*>(b.dst2[j]);
  const pack_t* src = reinterpret_cast<const pack_t*>(b.src[j]);
  const size_t bytes = b.bytes[j], npack = bytes / 16;
  constexpr size_t kTile = (size_t)kBlock * kUnroll;
  const size_t stride = (size_t)gridDim.x * kTile;
  const size_t lane_off = (size_t)(threadIdx.x >> 6) * (64 * kUnroll) + (threadIdx.x & 63);
  for (size_t base = (size_t)blockIdx.x * kTile; base < npack; base += stride) {
    const size_t first = base + lane_off;
    if (base + kTile <= npack) {
      pack_t v[kUnroll];
#pragma unroll
      for (int k = 0; k < kUnroll; k++) v[k] = ldp<MODE>(src + first + k * 64);
#pragma unroll
      for (int k = 0; k < kUnroll; k++) stp<MODE>(dst + first + k * 64, v[k]);
      if (dst2) {
#pragma unroll
        for (int k = 0; k < kUnroll; k++) stp<MODE>(dst2 + first + k * 64, v[k]);
      }
    } else {
      for (int k = 0; k < kUnroll; k++) {
        const size_t i = first + k * 64;
        if (i < npack) {
          const pack_t v = src[i];
          dst[i] = v;
          if (dst2) dst2[i] = v;
        }
      }
    }
  }
  const size_t done = npack * 16;
  if (blockIdx.x == 0 && done + threadIdx.x < bytes) {
    const size_t i = done + threadIdx.x;
    const uint8_t v = reinterpret_cast<const uint8_t*>(src)[i];
    reinterpret_cast<uint8_t*>(dst)[i] = v;
    if (dst2) reinterpret_cast<uint8_t*>(dst2)[i] = v;
  }
}

__global__ __launch_bounds__(kBlock) void copy1_kernel(uint8_t* dst, const uint8_t* src, size_t bytes) {
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < bytes; i += stride) dst[i] = src[i];
}

// ---- verification kernels: wavefront __shfl_down + LDS block reduction -----------------------

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    double o = __shfl_down(v, off, 64);
    v = (o > v) ? o : v;
  }
  return v;
}

// block-wide sum of one u64 per lane; result valid in thread 0
__device__ __forceinline__ uint64_t block_sum_u64(uint64_t v, uint64_t* lds /*[kBlock/64]*/) {
  v = wave_sum_u64(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) lds[wave] = v;
  __syncthreads();
  uint64_t r = 0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < kBlock / 64; w++) r += lds[w];
  }
  __syncthreads();
  return r;
}

__device__ __forceinline__ uint32_t diff_bytes_u32(uint32_t x, uint32_t y) {
  uint32_t d = x ^ y;  // a byte differs iff any of its bits is set
  d |= d >> 4;
  d |= d >> 2;
  d |= d >> 1;
  return __popc(d & 0x01010101u);
}

__global__ __launch_bounds__(kBlock) void count_mismatch_kernel(const uint8_t* a, const uint8_t* b,
                                                                size_t bytes, int vec_ok,
                                                                unsigned long long* out) {
  __shared__ uint64_t lds[kBlock / 64];
  uint64_t n = 0;
  const size_t stride = (size_t)gridDim.x * kBlock;
  size_t done = 0;
  if (vec_ok) {
    const size_t npack = bytes / 16;
    const pack_t* pa = reinterpret_cast<const pack_t*>(a);
    const pack_t* pb = reinterpret_cast<const pack_t*>(b);
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < npack; i += stride) {
      const pack_t x = pa[i], y = pb[i];
      n += diff_bytes_u32(x.x, y.x) + diff_bytes_u32(x.y, y.y) + diff_bytes_u32(x.z, y.z) +
           diff_bytes_u32(x.w, y.w);
    }
    done = npack * 16;
  }
  for (size_t i = done + (size_t)blockIdx.x * kBlock + threadIdx.x; i < bytes; i += stride)
    n += (a[i] != b[i]);
  const uint64_t tot = block_sum_u64(n, lds);
  if (threadIdx.x == 0 && tot) atomicAdd(out, (unsigned long long)tot);
}

__global__ __launch_bounds__(kBlock) void checksum_kernel(const uint8_t* buf, size_t bytes, int vec_ok,
                                                          unsigned long long* out) {
  __shared__ uint64_t lds[kBlock / 64];
  uint64_t s = 0;
  const size_t stride = (size_t)gridDim.x * kBlock;
  const size_t nword = bytes / 4;
  size_t wdone = 0;
  if (vec_ok) {
    const size_t npack = bytes / 16;
    const pack_t* p = reinterpret_cast<const pack_t*>(buf);
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < npack; i += stride) {
      const pack_t v = p[i];
      s += (uint64_t)v.x + v.y + v.z + v.w;
    }
    wdone = npack * 4;
  }
  for (size_t w = wdone + (size_t)blockIdx.x * kBlock + threadIdx.x; w < nword; w += stride) {
    const uint8_t* q = buf + 4 * w;  // byte loads: any alignment
    s += (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
  }
  for (size_t i = 4 * nword + (size_t)blockIdx.x * kBlock + threadIdx.x; i < bytes; i += stride)
    s += buf[i];
  const uint64_t tot = block_sum_u64(s, lds);
  if (threadIdx.x == 0 && tot) atomicAdd(out, (unsigned long long)tot);
}

template <typename T>
__device__ __forceinline__ double as_double(T v) {
  if constexpr (sizeof(T) == 2 && !__is_same(T, _Float16)) return (double)bf16_to_f32(v.bits);
  else return (double)v;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void diff_stats_kernel(const T* a, const T* b, size_t count,
                                                            unsigned long long* out_max,
                                                            double* out_sum,
                                                            unsigned long long* out_nan,
                                                            unsigned long long* out_rel) {
  __shared__ double lds_max[kBlock / 64];
  __shared__ double lds_sum[kBlock / 64];
  __shared__ uint64_t lds_nan[kBlock / 64];
  __shared__ double lds_rel[kBlock / 64];
  double mx = 0.0, sb = 0.0, rl = 0.0;
  uint64_t nn = 0;
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += stride) {
    const double x = as_double(a[i]), y = as_double(b[i]);
    const bool nx = (x != x), ny = (y != y);
    if (nx || ny) {
      nn += (nx != ny);
    } else {
      const double d = fabs(x - y);
      mx = (d > mx) ? d : mx;
      sb += fabs(y);
      // per-element relative deviation |a_i - b_i| / |b_i| (0/0 = 0, x/0 = inf): with non-negative inputs
      // b_i = sum_r |x_r,i|, so this IS the per-element bound of BASELINE.md, evaluated on the device
      const double r = (d == 0.0) ? 0.0 : d / fabs(y);
      rl = (r > rl) ? r : rl;
    }
  }
  rl = wave_max_f64(rl);
  mx = wave_max_f64(mx);
  sb = wave_sum_f64(sb);
  nn = wave_sum_u64(nn);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    lds_max[wave] = mx;
    lds_sum[wave] = sb;
    lds_nan[wave] = nn;
    lds_rel[wave] = rl;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double m = 0.0, s = 0.0, r = 0.0;
    uint64_t n = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; w++) {
      r = (lds_rel[w] > r) ? lds_rel[w] : r;
      m = (lds_max[w] > m) ? lds_max[w] : m;
      s += lds_sum[w];
      n += lds_nan[w];
    }
    // non-negative doubles order like their bit patterns
    atomicMax(out_max, (unsigned long long)__double_as_longlong(m));
    atomicMax(out_rel, (unsigned long long)__double_as_longlong(r));
    atomicAdd(out_sum, s);
    if (n) atomicAdd(out_nan, (unsigned long long)n);
  }
}

// ---- deterministic inputs: mirrors oracle_fill (oracle/xmpi_oracle.c) bit for bit ------------

__device__ __forceinline__ uint64_t hash64(uint64_t seed, uint64_t i) {
  uint64_t z = seed * 0xD1342543DE82EF95ULL + i * 0x9E3779B97F4A7C15ULL + 0x2545F4914F6CDD1DULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

// value of element i as a double exactly representable in the target type
__device__ __forceinline__ double pattern_real(int dtype, int pattern, uint64_t seed, uint64_t i) {
  const uint64_t h = hash64(seed, i);
  switch (pattern) {
    case 0:
      if (dtype == DT_F64) return (double)(h >> 11) * 0x1p-53;
      if (dtype == DT_F32) return (double)(h >> 40) * 0x1p-24;
      if (dtype == DT_F16) return (double)(h & 63u) * 0x1p-6;
      return (double)(h & 15u) * 0x1p-4;
    case 1:
      if (dtype == DT_F16) return (double)(i & 63u) * 0x1p-6 + (double)(seed & 7u);
      if (dtype == DT_BF16) return (double)(i & 15u) * 0x1p-4 + (double)(seed & 7u);
      return (double)(i % 251u) * 0x1p-8 + (double)(seed & 0xFFu);
    case 2:
      return (double)((seed & 0xFFu) + 1u);
    default: {
      const double m = (double)((h >> 40) & 0xFFu) * 0x1p-8 - 0.5;
      const int sh = (int)((h >> 8) & 7u) - 4;
      return ldexp(m, sh);
    }
  }
}

__global__ __launch_bounds__(kBlock) void fill_kernel(void* buf, size_t count, int dtype, int pattern,
                                                      uint64_t seed) {
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += stride) {
    switch (dtype) {
      case DT_U8: {
        uint8_t v = (pattern == 1)   ? (uint8_t)(seed * 31u + i)
                    : (pattern == 2) ? (uint8_t)(seed + 1u)
                                     : (uint8_t)(hash64(seed, i) >> 56);
        reinterpret_cast<uint8_t*>(buf)[i] = v;
        break;
      }
      case DT_I32: {
        int32_t v = (pattern == 1)   ? (int32_t)(((uint32_t)seed << 24) | ((uint32_t)i & 0xFFFFFFu))
                    : (pattern == 2) ? (int32_t)(seed + 1u)
                                     : (int32_t)(uint32_t)(hash64(seed, i) >> 32);
        reinterpret_cast<int32_t*>(buf)[i] = v;
        break;
      }
      case DT_I64: {
        int64_t v = (pattern == 1)   ? (int64_t)((seed << 40) | (uint64_t)i)
                    : (pattern == 2) ? (int64_t)(seed + 1u)
                                     : (int64_t)hash64(seed, i);
        reinterpret_cast<int64_t*>(buf)[i] = v;
        break;
      }
      case DT_F16:  // exactly representable => both conversions are exact
        reinterpret_cast<_Float16*>(buf)[i] = (_Float16)(float)pattern_real(dtype, pattern, seed, i);
        break;
      case DT_BF16:
        reinterpret_cast<uint16_t*>(buf)[i] = f32_to_bf16((float)pattern_real(dtype, pattern, seed, i));
        break;
      case DT_F32:
        reinterpret_cast<float*>(buf)[i] = (float)pattern_real(dtype, pattern, seed, i);
        break;
      default:
        reinterpret_cast<double*>(buf)[i] = pattern_real(dtype, pattern, seed, i);
        break;
    }
  }
}

// ---- cross-process completion flag -----------------------------------------------------------

// one device word -> a word the host can read (pinned, mapped), in stream order: how a verification kernel's count reaches the host
// WITHOUT a device-to-host copy (a copy engine's queue is one more hardware queue per process: with eight processes on one GPU the
// scheduler then time-slices all of them -- seconds per collective while kernels wait for each other)
__global__ void word_to_host_kernel(uint64_t* dst, const uint64_t* src) {
  __hip_atomic_store(dst, *src, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void signal_kernel(uint64_t* flag, uint64_t value) {
  // everything earlier on this stream has completed (stream order); publish system-wide
  __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- device-synchronised collective: rendezvous + data movement + completion in ONE kernel ------
// (protocol and layout: kernels.h, dsync.cpp).  Prologue: block 0 writes {where my buffers are, epoch} into slot
// `me` of every peer's flag page (system-scope release store over xGMI); every block waits until its own page
// holds the peers' slots for this epoch (uncached HBM, polled with system-scope loads), acquires, and
// translates the peers' buffer references into this process's mappings.  Body: the fold of reduce_n_multi_kernel.
// Epilogue: every wave drains its stores, one lane per block releases at system scope and takes a ticket; the
// block that takes the last ticket tells every peer "done" and waits until every peer said so: when the
// kernel ends, nobody reads this rank's input or writes its output any more.

// NSRC > 0: the number of sources is a compile-time constant and all their loads are issued before the first
// use (U packets per lane per source: U x NSRC 16-byte loads in flight -- remote loads over xGMI are round
// trips, so the deeper variant is for links); NSRC == 0: any number, one load at a time.
template <typename T, int OP, int NSRC, int U>
__global__ __launch_bounds__(kBlock) void dsync_fold_kernel(DsyncArgs a) {
  XMPI_SHARED(DsyncShared, sh);
  dsync_begin(a, sh);
  if (sh.fail == DSYNC_OK && a.nseg > 0) {
    const DsyncSeg& g = a.seg[blockIdx.y];
    const int t = threadIdx.x, me = a.me, n = a.n;
    if (t == 0) {  // this block's pointer lists
      int ns = 0, nd = 0;
      for (int r = 0; r < n; r++)
        if (g.src_mask >> r & 1u)
          sh.src[ns++] = g.src_from_recv == 2 ? (r == me ? sh.send[me] + g.src_off : sh.land[me] + (uint64_t)r * g.stage_stride)
                                              : (g.src_from_recv ? sh.recv[r] : sh.send[r]) + g.src_off;
      for (int d = 0; d < n; d++) {
        const int r = (me + d) % n;
        if (g.dst_mask >> r & 1u) sh.dst[nd++] = (g.dst_to_land ? sh.land[r] : sh.recv[r]) + g.dst_off;
      }
      sh.nsrc = ns;
      sh.ndst = nd;
    }
    __syncthreads();
    const int nsrc = (NSRC > 0) ? NSRC : sh.nsrc, ndst = sh.ndst;
    const size_t count = g.count;
    uint64_t all = 0;
    for (int k = 0; k < nsrc; k++) all |= sh.src[k];
    for (int k = 0; k < ndst; k++) all |= sh.dst[k];
    constexpr size_t N = 16 / sizeof(T);
    if ((all & 15u) == 0) {
      const size_t npack = count / N;
      uint64_t dp[kDsyncRanks];
#pragma unroll
      for (int k = 0; k < kDsyncRanks; k++) dp[k] = uniform64(sh.dst[k < ndst ? k : 0]);
      constexpr size_t kTile = (size_t)kBlock * U;
      const size_t stride = (size_t)gridDim.x * kTile;
      if constexpr (NSRC > 0) {
        uint64_t sp[NSRC];
#pragma unroll
        for (int k = 0; k < NSRC; k++) sp[k] = uniform64(sh.src[k]);
        for (size_t base = (size_t)blockIdx.x * kTile; base < npack; base += stride) {
          if (base + kTile <= npack) {
            pack_t v[U][NSRC];
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
              for (int k = 0; k < NSRC; k++)
                v[u][k] = ldp<2>(reinterpret_cast<const pack_t*>(sp[k]) + base + (size_t)u * kBlock + t);
#pragma unroll
            for (int u = 0; u < U; u++) {
              pack_t acc = v[u][0];
#pragma unroll
              for (int k = 1; k < NSRC; k++) acc = combine16<T, OP>(acc, v[u][k]);
#pragma unroll
              for (int k = 0; k < kDsyncRanks; k++)
                if (k < ndst) stp<1>(reinterpret_cast<pack_t*>(dp[k]) + base + (size_t)u * kBlock + t, acc);
            }
          } else {
            for (int u = 0; u < U; u++) {
              const size_t i = base + (size_t)u * kBlock + t;
              if (i >= npack) break;
              pack_t acc = ldp<2>(reinterpret_cast<const pack_t*>(sp[0]) + i);
#pragma unroll
              for (int k = 1; k < NSRC; k++) acc = combine16<T, OP>(acc, ldp<2>(reinterpret_cast<const pack_t*>(sp[k]) + i));
#pragma unroll
              for (int k = 0; k < kDsyncRanks; k++)
                if (k < ndst) stp<1>(reinterpret_cast<pack_t*>(dp[k]) + i, acc);
            }
          }
        }
      } else {
        for (size_t i = (size_t)blockIdx.x * kBlock + t; i < npack; i += (size_t)gridDim.x * kBlock) {
          pack_t acc = ldp<2>(reinterpret_cast<const pack_t*>(sh.src[0]) + i);
          for (int k = 1; k < nsrc; k++) acc = combine16<T, OP>(acc, ldp<2>(reinterpret_cast<const pack_t*>(sh.src[k]) + i));
#pragma unroll
          for (int k = 0; k < kDsyncRanks; k++)
            if (k < ndst) stp<1>(reinterpret_cast<pack_t*>(dp[k]) + i, acc);
        }
      }
      const size_t done = npack * N;  // ragged tail (< 16 bytes): the first lanes of the segment's block 0
      if (blockIdx.x == 0 && done + t < count) {
        const size_t i = done + t;
        T acc = reinterpret_cast<const T*>(sh.src[0])[i];
        for (int k = 1; k < nsrc; k++) acc = combine_any<T, OP>(acc, reinterpret_cast<const T*>(sh.src[k])[i]);
        for (int k = 0; k < ndst; k++) reinterpret_cast<T*>(sh.dst[k])[i] = acc;
      }
    } else {  // some buffer is not 16-byte aligned: one element per lane
      for (size_t i = (size_t)blockIdx.x * kBlock + t; i < count; i += (size_t)gridDim.x * kBlock) {
        T acc = reinterpret_cast<const T*>(sh.src[0])[i];
        for (int k = 1; k < nsrc; k++) acc = combine_any<T, OP>(acc, reinterpret_cast<const T*>(sh.src[k])[i]);
        for (int k = 0; k < ndst; k++) reinterpret_cast<T*>(sh.dst[k])[i] = acc;
      }
    }
  }
  dsync_end(a, sh);
}

// ---- launch helpers --------------------------------------------------------------------------

// plain launch, or a launch that carries its own begin / end events
#define XMPI_LAUNCH(kern, grid, block, stream, es, ee, ...)                                  \
  do {                                                                                       \
    if ((es) || (ee)) hipExtLaunchKernelGGL(kern, grid, block, 0, stream, es, ee, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kern, grid, block, 0, stream, __VA_ARGS__);                      \
  } while (0)

int g_kernel_mode = -1;   // -1 = by size, else forced 0 / 1 / 2
// 0 = one tile per block, the hardware dispatcher balances (measured better than a 2048-block
// grid-stride loop inside the collective: reduce_n 60 -> 56 us per 288 MiB); > 0 caps the grid
int g_grid_cap = 0;

inline int kernel_mode_for(size_t traffic_bytes) {
  if (g_kernel_mode >= 0) return g_kernel_mode;
  // a launch whose traffic exceeds what the caches can hold streams through them: keep its loads
  // from displacing anything (nt); small launches are served from L2 / Infinity Cache as they are
  return traffic_bytes >= (size_t)(48u << 20) ? 2 : 0;
}

inline int grid_for(size_t work_items, size_t per_block) {
  size_t g = (work_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g_grid_cap > 0 && g > (size_t)g_grid_cap) g = (size_t)g_grid_cap;
  if (g > 0x7fffffffu) g = 0x7fffffffu;
  return (int)g;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <typename T, int OP>
hipError_t reduce2_typed(void* dst, const void* a, const void* b, size_t count, hipStream_t s, hipEvent_t es,
                         hipEvent_t ee) {
  if (count == 0) return hipSuccess;
  if (aligned16(dst) && aligned16(a) && aligned16(b)) {
    constexpr size_t N = 16 / sizeof(T);
    const size_t npack = count / N;
    const int grid = grid_for(npack, (size_t)kBlock * kUnroll);
    const int mode = kernel_mode_for(3 * count * sizeof(T));
    if (mode == 1)
      XMPI_LAUNCH((reduce2_kernel<T, OP, 1>), dim3(grid), dim3(kBlock), s, es, ee, (T*)dst, (const T*)a, (const T*)b,
                  npack, count);
    else if (mode == 2)
      XMPI_LAUNCH((reduce2_kernel<T, OP, 2>), dim3(grid), dim3(kBlock), s, es, ee, (T*)dst, (const T*)a, (const T*)b,
                  npack, count);
    else
      XMPI_LAUNCH((reduce2_kernel<T, OP, 0>), dim3(grid), dim3(kBlock), s, es, ee, (T*)dst, (const T*)a, (const T*)b,
                  npack, count);
  } else {
    const int grid = grid_for(count, kBlock);
    XMPI_LAUNCH((reduce2_elem_kernel<T, OP>), dim3(grid), dim3(kBlock), s, es, ee, (T*)dst, (const T*)a,
                (const T*)b, count);
  }
  return hipGetLastError();
}

template <typename T>
hipError_t reduce2_op(void* dst, const void* a, const void* b, size_t count, int op, hipStream_t s, hipEvent_t es,
                      hipEvent_t ee) {
  switch (op) {
    case OP_SUM: return reduce2_typed<T, OP_SUM>(dst, a, b, count, s, es, ee);
    case OP_PROD: return reduce2_typed<T, OP_PROD>(dst, a, b, count, s, es, ee);
    case OP_MIN: return reduce2_typed<T, OP_MIN>(dst, a, b, count, s, es, ee);
    case OP_MAX: return reduce2_typed<T, OP_MAX>(dst, a, b, count, s, es, ee);
    default: return hipErrorInvalidValue;
  }
}

template <typename T, int OP>
hipError_t reduce_n_typed(void* dst, const SrcPtrs& srcs, int nsrc, size_t count, hipStream_t s, hipEvent_t es,
                          hipEvent_t ee) {
  constexpr size_t N = 16 / sizeof(T);
  const size_t npack = count / N;
  const int grid = grid_for(npack, kBlock);
  const int mode = kernel_mode_for((size_t)(nsrc + 1) * count * sizeof(T));
#define XMPI_RN(NS)                                                                                            \
  case NS:                                                                                                     \
    if (mode != 0)                                                                                             \
      XMPI_LAUNCH((reduce_n_kernel<T, OP, NS, 2>), dim3(grid), dim3(kBlock), s, es, ee, (T*)dst, srcs, nsrc, npack, \
                  count);                                                                                      \
    else                                                                                                       \
      XMPI_LAUNCH((reduce_n_kernel<T, OP, NS, 0>), dim3(grid), dim3(kBlock), s, es, ee, (T*)dst, srcs, nsrc, npack, \
                  count);                                                                                      \
    break;
  switch (nsrc) {
    XMPI_RN(2) XMPI_RN(3) XMPI_RN(4) XMPI_RN(5) XMPI_RN(6) XMPI_RN(7) XMPI_RN(8)
    default:
      XMPI_LAUNCH((reduce_n_kernel<T, OP, 0, 0>), dim3(grid), dim3(kBlock), s, es, ee, (T*)dst, srcs, nsrc, npack,
                  count);
      break;
  }
#undef XMPI_RN
  return hipGetLastError();
}

template <typename T>
hipError_t reduce_n_op(void* dst, const SrcPtrs& srcs, int nsrc, size_t count, int op, hipStream_t s, hipEvent_t es,
                       hipEvent_t ee) {
  switch (op) {
    case OP_SUM: return reduce_n_typed<T, OP_SUM>(dst, srcs, nsrc, count, s, es, ee);
    case OP_PROD: return reduce_n_typed<T, OP_PROD>(dst, srcs, nsrc, count, s, es, ee);
    case OP_MIN: return reduce_n_typed<T, OP_MIN>(dst, srcs, nsrc, count, s, es, ee);
    case OP_MAX: return reduce_n_typed<T, OP_MAX>(dst, srcs, nsrc, count, s, es, ee);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

hipError_t launch_reduce2(void* dst, const void* a, const void* b, size_t count, int dtype, int op,
                          hipStream_t s, hipEvent_t es, hipEvent_t ee) {
  switch (dtype) {
    case DT_U8: return reduce2_op<uint8_t>(dst, a, b, count, op, s, es, ee);
    case DT_I32: return reduce2_op<int32_t>(dst, a, b, count, op, s, es, ee);
    case DT_I64: return reduce2_op<int64_t>(dst, a, b, count, op, s, es, ee);
    case DT_F16: return reduce2_op<_Float16>(dst, a, b, count, op, s, es, ee);
    case DT_F32: return reduce2_op<float>(dst, a, b, count, op, s, es, ee);
    case DT_F64: return reduce2_op<double>(dst, a, b, count, op, s, es, ee);
    case DT_BF16: return reduce2_op<bf16_t>(dst, a, b, count, op, s, es, ee);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_reduce_n(void* dst, const void* const* srcs, int nsrc, size_t count, int dtype,
                           int op, hipStream_t s, hipEvent_t es, hipEvent_t ee) {
  if (nsrc < 1 || nsrc > kMaxReduceSrcs) return hipErrorInvalidValue;
  if (count == 0) return hipSuccess;
  if (nsrc == 1) {
    static const size_t esz[] = {1, 4, 8, 2, 4, 8, 2};
    if (dtype < 0 || dtype > DT_BF16) return hipErrorInvalidValue;
    return launch_copy(dst, srcs[0], count * esz[dtype], s, es, ee);
  }
  bool ok = aligned16(dst);
  SrcPtrs p;
  for (int i = 0; i < kMaxReduceSrcs; i++) p.p[i] = (i < nsrc) ? srcs[i] : nullptr;
  for (int i = 0; i < nsrc; i++) ok = ok && aligned16(srcs[i]);
  if (!ok) {  // odd alignment: the element kernel (same left-to-right order; dst may alias any source)
    void* d1[1] = {dst};
    return launch_reduce_n_multi(d1, 1, srcs, nsrc, count, dtype, op, s, es, ee);
  }
  switch (dtype) {
    case DT_U8: return reduce_n_op<uint8_t>(dst, p, nsrc, count, op, s, es, ee);
    case DT_I32: return reduce_n_op<int32_t>(dst, p, nsrc, count, op, s, es, ee);
    case DT_I64: return reduce_n_op<int64_t>(dst, p, nsrc, count, op, s, es, ee);
    case DT_F16: return reduce_n_op<_Float16>(dst, p, nsrc, count, op, s, es, ee);
    case DT_F32: return reduce_n_op<float>(dst, p, nsrc, count, op, s, es, ee);
    case DT_F64: return reduce_n_op<double>(dst, p, nsrc, count, op, s, es, ee);
    case DT_BF16: return reduce_n_op<bf16_t>(dst, p, nsrc, count, op, s, es, ee);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_copy(void* dst, const void* src, size_t bytes, hipStream_t s, hipEvent_t es, hipEvent_t ee) {
  if (bytes == 0 || dst == src) {  // nothing to launch: still honour the events
    if (es) (void)hipEventRecord(es, s);
    if (ee) (void)hipEventRecord(ee, s);
    return hipSuccess;
  }
  if (aligned16(dst) && aligned16(src)) {
    const size_t npack = bytes / 16;
    const int grid = grid_for(npack, (size_t)kBlock * kUnroll);
    const int mode = kernel_mode_for(2 * bytes);
    if (mode == 1)
      XMPI_LAUNCH(copy16_kernel<1>, dim3(grid), dim3(kBlock), s, es, ee, (pack_t*)dst, (const pack_t*)src, npack, bytes);
    else if (mode == 2)
      XMPI_LAUNCH(copy16_kernel<2>, dim3(grid), dim3(kBlock), s, es, ee, (pack_t*)dst, (const pack_t*)src, npack, bytes);
    else
      XMPI_LAUNCH(copy16_kernel<0>, dim3(grid), dim3(kBlock), s, es, ee, (pack_t*)dst, (const pack_t*)src, npack, bytes);
  } else {
    XMPI_LAUNCH(copy1_kernel, dim3(grid_for(bytes, kBlock)), dim3(kBlock), s, es, ee, (uint8_t*)dst,
                (const uint8_t*)src, bytes);
  }
  return hipGetLastError();
}

namespace {
template <typename T, int OP>
hipError_t reduce2_batch_typed(const Reduce2Batch& q, int n, size_t maxcount, size_t total, hipStream_t s,
                               hipEvent_t es, hipEvent_t ee) {
  constexpr size_t N = 16 / sizeof(T);
  int gx = grid_for(maxcount / N + 1, (size_t)kBlock * kUnroll);
  const int cap = g_grid_cap > 0 ? (g_grid_cap + n - 1) / n : 0;
  if (cap > 0 && gx > cap) gx = cap;
  const int mode = kernel_mode_for(3 * total * sizeof(T));
  if (mode == 1) XMPI_LAUNCH((reduce2_batch_kernel<T, OP, 1>), dim3(gx, n), dim3(kBlock), s, es, ee, q);
  else if (mode == 2) XMPI_LAUNCH((reduce2_batch_kernel<T, OP, 2>), dim3(gx, n), dim3(kBlock), s, es, ee, q);
  else XMPI_LAUNCH((reduce2_batch_kernel<T, OP, 0>), dim3(gx, n), dim3(kBlock), s, es, ee, q);
  return hipGetLastError();
}

template <typename T>
hipError_t reduce2_batch_op(const Reduce2Batch& q, int n, size_t maxcount, size_t total, int op, hipStream_t s,
                            hipEvent_t es, hipEvent_t ee) {
  switch (op) {
    case OP_SUM: return reduce2_batch_typed<T, OP_SUM>(q, n, maxcount, total, s, es, ee);
    case OP_PROD: return reduce2_batch_typed<T, OP_PROD>(q, n, maxcount, total, s, es, ee);
    case OP_MIN: return reduce2_batch_typed<T, OP_MIN>(q, n, maxcount, total, s, es, ee);
    case OP_MAX: return reduce2_batch_typed<T, OP_MAX>(q, n, maxcount, total, s, es, ee);
    default: return hipErrorInvalidValue;
  }
}
}  // namespace

hipError_t launch_reduce2_batch(void* const* dst, void* const* dst2, const void* const* a, const void* const* b,
                                const size_t* counts, int n, int dtype, int op, hipStream_t s, hipEvent_t es,
                                hipEvent_t ee) {
  if (n < 1 || n > kMaxBatch) return hipErrorInvalidValue;
  Reduce2Batch q;
  size_t maxc = 0, total = 0;
  bool ok = true, fused = false;
  for (int i = 0; i < n; i++) {
    q.dst[i] = dst[i];
    q.dst2[i] = dst2 ? dst2[i] : nullptr;
    q.a[i] = a[i];
    q.b[i] = b[i];
    q.count[i] = counts[i];
    maxc = counts[i] > maxc ? counts[i] : maxc;
    total += counts[i];
    fused = fused || q.dst2[i] != nullptr || dst[i] == nullptr;
    ok = ok && aligned16(dst[i]) && aligned16(q.dst2[i]) && aligned16(a[i]) && aligned16(b[i]);
  }
  if (!ok || (n == 1 && !fused) || maxc == 0) {  // odd alignment / nothing to fuse: plain launches
    // The local destination may alias an operand (in-place ring step: dst == a): the forwarded copy is
    // computed FIRST, from the untouched operands, and the aliasing store comes last.
    bool first = true;
    for (int i = 0; i < n; i++)
      for (int w = 0; w < 2; w++) {
        void* d = w == 0 ? q.dst2[i] : dst[i];
        if (!d) continue;
        const bool last = (i == n - 1) && (w == 1 || !dst[i]);
        hipError_t e = launch_reduce2(d, a[i], b[i], counts[i], dtype, op, s, first ? es : nullptr, last ? ee : nullptr);
        if (e != hipSuccess) return e;
        first = false;
      }
    if (maxc == 0) {  // nothing launched: still honour the events
      if (es) (void)hipEventRecord(es, s);
      if (ee) (void)hipEventRecord(ee, s);
    }
    return hipSuccess;
  }
  switch (dtype) {
    case DT_U8: return reduce2_batch_op<uint8_t>(q, n, maxc, total, op, s, es, ee);
    case DT_I32: return reduce2_batch_op<int32_t>(q, n, maxc, total, op, s, es, ee);
    case DT_I64: return reduce2_batch_op<int64_t>(q, n, maxc, total, op, s, es, ee);
    case DT_F16: return reduce2_batch_op<_Float16>(q, n, maxc, total, op, s, es, ee);
    case DT_F32: return reduce2_batch_op<float>(q, n, maxc, total, op, s, es, ee);
    case DT_F64: return reduce2_batch_op<double>(q, n, maxc, total, op, s, es, ee);
    case DT_BF16: return reduce2_batch_op<bf16_t>(q, n, maxc, total, op, s, es, ee);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_copy_batch(void* const* dst, void* const* dst2, const void* const* src, const size_t* bytes, int n,
                             hipStream_t s, hipEvent_t es, hipEvent_t ee) {
  if (n < 1 || n > kMaxBatch) return hipErrorInvalidValue;
  CopyBatch b;
  size_t maxb = 0, total = 0;
  bool ok = true, fused = false;
  for (int i = 0; i < n; i++) {
    b.dst[i] = dst[i];
    b.dst2[i] = dst2 ? dst2[i] : nullptr;
    b.src[i] = src[i];
    b.bytes[i] = bytes[i];
    maxb = bytes[i] > maxb ? bytes[i] : maxb;
    total += bytes[i];
    fused = fused || b.dst2[i] != nullptr;
    ok = ok && aligned16(dst[i]) && aligned16(b.dst2[i]) && aligned16(src[i]);
  }
  if (!ok || (n == 1 && !fused)) {  // odd alignment: one launch per copy, the events span the group
    bool first = true;
    for (int i = 0; i < n; i++)
      for (int w = 0; w < 2; w++) {
        void* d = w == 0 ? dst[i] : b.dst2[i];
        if (!d) continue;
        const bool last = (i == n - 1) && (w == 1 || !b.dst2[i]);
        hipError_t e = launch_copy(d, src[i], bytes[i], s, first ? es : nullptr, last ? ee : nullptr);
        if (e != hipSuccess) return e;
        first = false;
      }
    return hipSuccess;
  }
  if (maxb == 0) {
    if (es) (void)hipEventRecord(es, s);
    if (ee) (void)hipEventRecord(ee, s);
    return hipSuccess;
  }
  int gx = grid_for(maxb / 16 + 1, (size_t)kBlock * kUnroll);
  const int cap = g_grid_cap > 0 ? (g_grid_cap + n - 1) / n : 0;
  if (cap > 0 && gx > cap) gx = cap;
  const int mode = kernel_mode_for(2 * total);
  if (mode == 1) XMPI_LAUNCH(copy_batch_kernel<1>, dim3(gx, n), dim3(kBlock), s, es, ee, b);
  else if (mode == 2) XMPI_LAUNCH(copy_batch_kernel<2>, dim3(gx, n), dim3(kBlock), s, es, ee, b);
  else XMPI_LAUNCH(copy_batch_kernel<0>, dim3(gx, n), dim3(kBlock), s, es, ee, b);
  return hipGetLastError();
}

namespace {
template <typename T, int OP>
hipError_t reduce_n_multi_typed(const MultiPtrs& q, int nsrc, int ndst, size_t count, bool vec, hipStream_t s,
                                hipEvent_t es, hipEvent_t ee) {
  if (!vec) {
    XMPI_LAUNCH((reduce_n_multi_elem_kernel<T, OP>), dim3(grid_for(count, kBlock)), dim3(kBlock), s, es, ee, q, nsrc,
                ndst, count);
    return hipGetLastError();
  }
  constexpr size_t N = 16 / sizeof(T);
  const size_t npack = count / N;
  const int grid = grid_for(npack, kBlock);
  const int mode = kernel_mode_for((size_t)(nsrc + ndst) * count * sizeof(T));
#define XMPI_RNM(NS)                                                                                              \
  case NS:                                                                                                        \
    if (mode != 0)                                                                                                \
      XMPI_LAUNCH((reduce_n_multi_kernel<T, OP, NS, 2>), dim3(grid), dim3(kBlock), s, es, ee, q, nsrc, ndst, npack, \
                  count);                                                                                         \
    else                                                                                                          \
      XMPI_LAUNCH((reduce_n_multi_kernel<T, OP, NS, 0>), dim3(grid), dim3(kBlock), s, es, ee, q, nsrc, ndst, npack, \
                  count);                                                                                         \
    break;
  switch (nsrc) {
    XMPI_RNM(1) XMPI_RNM(2) XMPI_RNM(3) XMPI_RNM(4) XMPI_RNM(5) XMPI_RNM(6) XMPI_RNM(7) XMPI_RNM(8)
    default:
      XMPI_LAUNCH((reduce_n_multi_kernel<T, OP, 0, 0>), dim3(grid), dim3(kBlock), s, es, ee, q, nsrc, ndst, npack,
                  count);
      break;
  }
#undef XMPI_RNM
  return hipGetLastError();
}

template <typename T>
hipError_t reduce_n_multi_op(const MultiPtrs& q, int nsrc, int ndst, size_t count, bool vec, int op, hipStream_t s,
                             hipEvent_t es, hipEvent_t ee) {
  switch (op) {
    case OP_SUM: return reduce_n_multi_typed<T, OP_SUM>(q, nsrc, ndst, count, vec, s, es, ee);
    case OP_PROD: return reduce_n_multi_typed<T, OP_PROD>(q, nsrc, ndst, count, vec, s, es, ee);
    case OP_MIN: return reduce_n_multi_typed<T, OP_MIN>(q, nsrc, ndst, count, vec, s, es, ee);
    case OP_MAX: return reduce_n_multi_typed<T, OP_MAX>(q, nsrc, ndst, count, vec, s, es, ee);
    default: return hipErrorInvalidValue;
  }
}
}  // namespace

hipError_t launch_reduce_n_multi(void* const* dsts, int ndst, const void* const* srcs, int nsrc, size_t count,
                                 int dtype, int op, hipStream_t s, hipEvent_t es, hipEvent_t ee) {
  if (nsrc < 1 || nsrc > kMaxReduceSrcs || ndst < 0 || ndst > kMaxReduceSrcs) return hipErrorInvalidValue;
  if (count == 0 || ndst == 0) {
    if (es) (void)hipEventRecord(es, s);
    if (ee) (void)hipEventRecord(ee, s);
    return hipSuccess;
  }
  MultiPtrs q;
  bool vec = true;
  for (int i = 0; i < kMaxReduceSrcs; i++) {
    q.src[i] = (i < nsrc) ? srcs[i] : nullptr;
    q.dst[i] = (i < ndst) ? dsts[i] : nullptr;
    vec = vec && aligned16(q.src[i]) && aligned16(q.dst[i]);
  }
  switch (dtype) {
    case DT_U8: return reduce_n_multi_op<uint8_t>(q, nsrc, ndst, count, vec, op, s, es, ee);
    case DT_I32: return reduce_n_multi_op<int32_t>(q, nsrc, ndst, count, vec, op, s, es, ee);
    case DT_I64: return reduce_n_multi_op<int64_t>(q, nsrc, ndst, count, vec, op, s, es, ee);
    case DT_F16: return reduce_n_multi_op<_Float16>(q, nsrc, ndst, count, vec, op, s, es, ee);
    case DT_F32: return reduce_n_multi_op<float>(q, nsrc, ndst, count, vec, op, s, es, ee);
    case DT_F64: return reduce_n_multi_op<double>(q, nsrc, ndst, count, vec, op, s, es, ee);
    case DT_BF16: return reduce_n_multi_op<bf16_t>(q, nsrc, ndst, count, vec, op, s, es, ee);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_copy_pairs(void* const* dsts, const void* const* srcs, int n, size_t bytes, hipStream_t s, hipEvent_t es, hipEvent_t ee) {
  if (n < 1 || n > kMaxReduceSrcs) return hipErrorInvalidValue;
  MultiPtrs q;
  for (int i = 0; i < kMaxReduceSrcs; i++) {
    q.src[i] = (i < n) ? srcs[i] : nullptr;
    q.dst[i] = (i < n) ? dsts[i] : nullptr;
    if (i < n && (!aligned16(q.src[i]) || !aligned16(q.dst[i]))) return hipErrorInvalidValue;
  }
  if (bytes == 0) {
    if (es) (void)hipEventRecord(es, s);
    if (ee) (void)hipEventRecord(ee, s);
    return hipSuccess;
  }
  const size_t npack = bytes / 16;
  const int grid = grid_for(npack, kBlock);                       // (as reduce_n_multi_typed: one packet per lane)
  const int mode = kernel_mode_for((size_t)(2 * n) * bytes);      // ... and its cache policy for that many bytes
#define XMPI_CPP(NS)                                                                                                  \
  case NS:                                                                                                            \
    if (mode != 0) XMPI_LAUNCH((copy_pairs_kernel<NS, 2>), dim3(grid), dim3(kBlock), s, es, ee, q, n, npack, bytes);  \
    else XMPI_LAUNCH((copy_pairs_kernel<NS, 0>), dim3(grid), dim3(kBlock), s, es, ee, q, n, npack, bytes);            \
    break;
  switch (n) {
    XMPI_CPP(1) XMPI_CPP(2) XMPI_CPP(3) XMPI_CPP(4) XMPI_CPP(5) XMPI_CPP(6) XMPI_CPP(7) XMPI_CPP(8)
    default: XMPI_LAUNCH((copy_pairs_kernel<0, 0>), dim3(grid), dim3(kBlock), s, es, ee, q, n, npack, bytes); break;
  }
#undef XMPI_CPP
  return hipGetLastError();
}

hipError_t launch_copy_multi(void* const* dsts, int ndst, const void* src, size_t bytes, hipStream_t s,
                             hipEvent_t es, hipEvent_t ee) {
  if (ndst < 0 || ndst > kMaxReduceSrcs) return hipErrorInvalidValue;
  MultiPtrs q;
  for (int i = 0; i < kMaxReduceSrcs; i++) q.src[i] = q.dst[i] = nullptr;
  q.src[0] = src;
  int n = 0;
  bool vec = aligned16(src);
  for (int i = 0; i < ndst; i++)
    if (dsts[i] != src) {  // in-place entry: nothing to move
      q.dst[n++] = dsts[i];
      vec = vec && aligned16(dsts[i]);
    }
  if (bytes == 0 || n == 0) {
    if (es) (void)hipEventRecord(es, s);
    if (ee) (void)hipEventRecord(ee, s);
    return hipSuccess;
  }
  if (!vec) {
    XMPI_LAUNCH(copy_multi_elem_kernel, dim3(grid_for(bytes, kBlock)), dim3(kBlock), s, es, ee, q, n, bytes);
    return hipGetLastError();
  }
  const size_t npack = bytes / 16;
  const int grid = grid_for(npack + 1, (size_t)kBlock * kUnroll);
  const int mode = kernel_mode_for((size_t)(1 + n) * bytes);
  if (mode != 0) XMPI_LAUNCH(copy_multi_kernel<2>, dim3(grid), dim3(kBlock), s, es, ee, q, n, npack, bytes);
  else XMPI_LAUNCH(copy_multi_kernel<0>, dim3(grid), dim3(kBlock), s, es, ee, q, n, npack, bytes);
  return hipGetLastError();
}

hipError_t launch_count_mismatch(const void* a, const void* b, size_t bytes, uint64_t* d_out,
                                 hipStream_t s) {
  if (bytes == 0) return hipSuccess;
  const int vec_ok = aligned16(a) && aligned16(b);
  hipLaunchKernelGGL(count_mismatch_kernel, dim3(grid_for(bytes / 16 + 1, kBlock * 4)), dim3(kBlock), 0,
                     s, (const uint8_t*)a, (const uint8_t*)b, bytes, vec_ok, (unsigned long long*)d_out);
  return hipGetLastError();
}

hipError_t launch_checksum(const void* buf, size_t bytes, uint64_t* d_out, hipStream_t s) {
  if (bytes == 0) return hipSuccess;
  const int vec_ok = aligned16(buf);
  hipLaunchKernelGGL(checksum_kernel, dim3(grid_for(bytes / 16 + 1, kBlock * 4)), dim3(kBlock), 0, s,
                     (const uint8_t*)buf, bytes, vec_ok, (unsigned long long*)d_out);
  return hipGetLastError();
}

hipError_t launch_diff_stats(const void* a, const void* b, size_t count, int dtype, void* d_out,
                             hipStream_t s) {
  if (count == 0) return hipSuccess;
  unsigned long long* o_max = (unsigned long long*)d_out;
  double* o_sum = (double*)d_out + 1;
  unsigned long long* o_nan = (unsigned long long*)d_out + 2;
  unsigned long long* o_rel = (unsigned long long*)d_out + 3;
  const dim3 grid(grid_for(count, kBlock * 8)), block(kBlock);
  switch (dtype) {
    case DT_F16:
      hipLaunchKernelGGL(diff_stats_kernel<_Float16>, grid, block, 0, s, (const _Float16*)a,
                         (const _Float16*)b, count, o_max, o_sum, o_nan, o_rel);
      break;
    case DT_BF16:
      hipLaunchKernelGGL(diff_stats_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)a,
                         (const bf16_t*)b, count, o_max, o_sum, o_nan, o_rel);
      break;
    case DT_F32:
      hipLaunchKernelGGL(diff_stats_kernel<float>, grid, block, 0, s, (const float*)a, (const float*)b,
                         count, o_max, o_sum, o_nan, o_rel);
      break;
    case DT_F64:
      hipLaunchKernelGGL(diff_stats_kernel<double>, grid, block, 0, s, (const double*)a,
                         (const double*)b, count, o_max, o_sum, o_nan, o_rel);
      break;
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_fill(void* buf, size_t count, int dtype, int pattern, uint64_t seed, hipStream_t s) {
  if (count == 0) return hipSuccess;
  if (dtype < 0 || dtype > DT_BF16 || pattern < 0 || pattern > 3) return hipErrorInvalidValue;
  hipLaunchKernelGGL(fill_kernel, dim3(grid_for(count, kBlock * 4)), dim3(kBlock), 0, s, buf, count,
                     dtype, pattern, seed);
  return hipGetLastError();
}

void set_kernel_mode(int mode) { g_kernel_mode = (mode < 0 || mode > 2) ? -1 : mode; }
int get_kernel_mode() { return g_kernel_mode; }
void set_grid_cap(int cap) { g_grid_cap = cap < 0 ? 0 : cap; }

hipError_t launch_word_to_host(uint64_t* dst, const uint64_t* src, hipStream_t s) {
  hipLaunchKernelGGL(word_to_host_kernel, dim3(1), dim3(1), 0, s, dst, src);
  return hipGetLastError();
}

hipError_t launch_signal(uint64_t* flag, uint64_t value, hipStream_t s) {
  hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(1), 0, s, flag, value);
  return hipGetLastError();
}

namespace {
template <typename T, int OP, int NSRC, int U>
hipError_t dsync_go(const DsyncArgs& a, int grid_x, hipStream_t s, hipEvent_t es, hipEvent_t ee) {
  const dim3 grid(grid_x < 1 ? 1 : grid_x, a.nseg < 1 ? 1 : a.nseg);
  XMPI_LAUNCH((dsync_fold_kernel<T, OP, NSRC, U>), grid, dim3(kBlock), s, es, ee, a);
  return hipGetLastError();
}

template <typename T>
hipError_t dsync_typed(const DsyncArgs& a, int nsrc, int op, int grid_x, int unroll, hipStream_t s, hipEvent_t es,
                       hipEvent_t ee) {
  if (op == OP_SUM) {  // the hot operator: sources unrolled, one or two packets per lane per source in flight
#define XMPI_DS(NS)                                                              \
  case NS:                                                                       \
    return unroll >= 2 ? dsync_go<T, OP_SUM, NS, 2>(a, grid_x, s, es, ee) : dsync_go<T, OP_SUM, NS, 1>(a, grid_x, s, es, ee);
    switch (nsrc) {
      XMPI_DS(1) XMPI_DS(2) XMPI_DS(3) XMPI_DS(4) XMPI_DS(5) XMPI_DS(6) XMPI_DS(7) XMPI_DS(8)
      default: return dsync_go<T, OP_SUM, 0, 1>(a, grid_x, s, es, ee);
    }
#undef XMPI_DS
  }
  switch (op) {
    case OP_PROD: return dsync_go<T, OP_PROD, 0, 1>(a, grid_x, s, es, ee);
    case OP_MIN: return dsync_go<T, OP_MIN, 0, 1>(a, grid_x, s, es, ee);
    case OP_MAX: return dsync_go<T, OP_MAX, 0, 1>(a, grid_x, s, es, ee);
    default: return hipErrorInvalidValue;
  }
}
}  // namespace

hipError_t launch_dsync_fold(const DsyncArgs& a, int nsrc, int dtype, int op, int grid_x, int unroll, hipStream_t s,
                             hipEvent_t es, hipEvent_t ee) {
  if (a.n < 1 || a.n > kDsyncRanks || a.nseg < 0 || a.nseg > kDsyncRanks || nsrc < 0 || nsrc > kDsyncRanks)
    return hipErrorInvalidValue;
  switch (dtype) {
    case DT_U8: return dsync_typed<uint8_t>(a, nsrc, op, grid_x, unroll, s, es, ee);
    case DT_I32: return dsync_typed<int32_t>(a, nsrc, op, grid_x, unroll, s, es, ee);
    case DT_I64: return dsync_typed<int64_t>(a, nsrc, op, grid_x, unroll, s, es, ee);
    case DT_F16: return dsync_typed<_Float16>(a, nsrc, op, grid_x, unroll, s, es, ee);
    case DT_F32: return dsync_typed<float>(a, nsrc, op, grid_x, unroll, s, es, ee);
    case DT_F64: return dsync_typed<double>(a, nsrc, op, grid_x, unroll, s, es, ee);
    case DT_BF16: return dsync_typed<bf16_t>(a, nsrc, op, grid_x, unroll, s, es, ee);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace xmpi
