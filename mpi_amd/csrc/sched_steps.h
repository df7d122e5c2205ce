// sched_steps.h -- the step program of the stepped kernels (sched.hip), as a function both the device and the host
// compile: the kernel calls sched_step() for every step it runs, and xmpi_sched_dump (api.cpp) prints the very same steps
// for the CPU simulator (tests/sched_sim.py), which executes all ranks' programs under random interleavings and checks
// the data flow and its hazards without a GPU.
#pragma once
#include <cstdint>

#include "kernels.h"

#if defined(__HIPCC__)
#define XMPI_HD __host__ __device__ inline
#else
#define XMPI_HD inline
#endif

namespace xmpi {

struct SchedStep {
  int32_t wait_rank;   // whose step this one needs (-1: nobody's -- the rendezvous was enough)
  uint32_t wait_val;   // ... and which
  int32_t sig[2];      // who is told when this step is done (-1: nobody)
  uint32_t sig_val;
  int32_t ns;          // 0 = nothing to move, 1 = copy, 2 = combine
  uint64_t D, A, B;    // bases: byte offset x of the buffer is at base + x
  uint64_t lo, hi;     // byte range of the buffer this step covers
};

XMPI_HD void chunk_bytes(uint64_t count, uint32_t es, int parts, int j, uint64_t* lo, uint64_t* hi) {
  const uint64_t al = es >= 16 ? 1 : 16 / es;
  uint64_t base = (count + (uint64_t)parts - 1) / (uint64_t)parts;
  base = (base + al - 1) / al * al;
  const uint64_t a = (uint64_t)j * base, b = (uint64_t)(j + 1) * base;
  *lo = (a < count ? a : count) * es;
  *hi = (b < count ? b : count) * es;
}

XMPI_HD int sched_nsteps(const DsyncSchedArgs& a) {
  const int n = a.d.n;
  switch (a.sched) {
    case SCHED_RING_ALLREDUCE: return 2 * (n - 1);
    case SCHED_RHD_ALLREDUCE: {
      int l = 0;
      while ((1 << l) < n) l++;
      return 2 * l;
    }
    case SCHED_RING_ALLGATHER: return n;
    default: return a.pieces;
  }
}

// step g (1-based) of this rank on ring channel `ch`
XMPI_HD void sched_step(const DsyncSchedArgs& a, const uint64_t* send, const uint64_t* recv, int g, int ch, SchedStep* st) {
  const int n = a.d.n, me = a.d.me;
  const uint32_t es = a.elem_size;
  st->wait_rank = -1;
  st->wait_val = 0;
  st->sig[0] = st->sig[1] = -1;
  st->sig_val = (uint32_t)g;
  st->ns = 0;
  st->D = recv[me];
  st->A = st->B = 0;
  st->lo = st->hi = 0;
  if (a.sched == SCHED_RING_ALLREDUCE || a.sched == SCHED_RING_ALLGATHER) {
    int pos = 0;
    for (int i = 0; i < n; i++)
      if (a.order[ch][i] == me) pos = i;
    const int prev = a.order[ch][(pos + n - 1) % n], next = a.order[ch][(pos + 1) % n];
    if (g >= 2) {
      st->wait_rank = prev;
      st->wait_val = (uint32_t)(g - 1);
    }
    if (a.sched == SCHED_RING_ALLREDUCE) {
      if (g < 2 * (n - 1)) st->sig[0] = next;
      if (g <= n - 1) {  // reduce-scatter: my partial of chunk (pos - g) = the previous rank's partial + my contribution
        const int c = (pos + n - g) % n;
        chunk_bytes(a.count, es, n, c, &st->lo, &st->hi);
        st->ns = 2;
        st->A = g == 1 ? send[prev] : recv[prev];
        st->B = send[me];
      } else {  // allgather: the finished chunk (pos + 1 - t) travels on
        const int t = g - (n - 1);
        const int c = (pos + 1 + n - t) % n;
        chunk_bytes(a.count, es, n, c, &st->lo, &st->hi);
        st->ns = 1;
        st->A = recv[prev];
      }
    } else {
      const uint64_t blk = a.count * es;
      if (g < n) st->sig[0] = next;
      if (g == 1) {  // my own block into its place
        st->lo = (uint64_t)me * blk;
        st->hi = st->lo + blk;
        st->A = send[me] - st->lo;
        st->ns = st->A == st->D ? 0 : 1;
      } else {  // the block that reached the previous rank one step ago
        const int r = a.order[ch][(pos + n - (g - 1)) % n];
        st->lo = (uint64_t)r * blk;
        st->hi = st->lo + blk;
        st->A = recv[prev];
        st->ns = 1;
      }
    }
    return;
  }
  if (a.sched == SCHED_RHD_ALLREDUCE) {
    int l = 0;
    while ((1 << l) < n) l++;
    // the ranges: R_0 = the buffer, R_{k+1} = the half of R_k this rank keeps at halving step k
    const int level = g <= l ? g - 1 : 2 * l - g;  // halving step k = g-1; doubling undoes level 2l-g
    uint64_t lo = 0, hi = a.count * es;
    uint64_t klo = 0, khi = 0, olo = 0, ohi = 0;  // kept half / other half at `level`
    for (int k = 0; k <= level; k++) {
      const int d = n >> (k + 1);
      const uint64_t mid = lo + (((hi - lo) / 2) & ~(uint64_t)15);
      if (me & d) {
        klo = mid, khi = hi, olo = lo, ohi = mid;
      } else {
        klo = lo, khi = mid, olo = mid, ohi = hi;
      }
      lo = klo;
      hi = khi;
    }
    const int p = me ^ (n >> (level + 1));
    if (g >= 2) {
      st->wait_rank = p;
      st->wait_val = (uint32_t)(g - 1);
    }
    if (g < 2 * l) {
      const int nlevel = g + 1 <= l ? g : 2 * l - g - 1;
      st->sig[0] = me ^ (n >> (nlevel + 1));
    }
    if (g <= l) {  // halving: my half of the partner's accumulator joins mine
      st->ns = 2;
      st->lo = klo;
      st->hi = khi;
      st->A = g == 1 ? send[p] : recv[p];
      st->B = g == 1 ? send[me] : recv[me];
    } else {  // doubling: the partner's finished half
      st->ns = 1;
      st->lo = olo;
      st->hi = ohi;
      st->A = recv[p];
    }
    return;
  }
  // SCHED_TREE_BCAST: piece g of the buffer comes from the parent and is announced to the children
  const int v = (me - a.root + n) % n;
  const int c1 = 2 * v + 1, c2 = 2 * v + 2;
  if (c1 < n) st->sig[0] = (c1 + a.root) % n;
  if (c2 < n) st->sig[1] = (c2 + a.root) % n;
  uint64_t lo, hi;
  chunk_bytes(a.count * es, 1, a.pieces, g - 1, &lo, &hi);
  if (v != 0) {
    const int parent = ((v - 1) / 2 + a.root) % n;
    if ((v - 1) / 2 != 0) {  // the root's buffer is complete when it announces itself; anybody else's piece by piece
      st->wait_rank = parent;
      st->wait_val = (uint32_t)g;
    }
    st->ns = 1;
    st->lo = lo;
    st->hi = hi;
    st->A = recv[parent];
  }
}

}  // namespace xmpi
