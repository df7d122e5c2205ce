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

// recursive halving + doubling runs among the P = 2^l <= n ranks that are left when the first 2 (n - P) ranks have paired up
XMPI_HD int rhd_levels(int n) {
  int l = 0;
  while ((2 << l) <= n) l++;
  return l;
}

XMPI_HD int sched_nsteps(const DsyncSchedArgs& a) {
  const int n = a.d.n;
  switch (a.sched) {
    case SCHED_RING_ALLREDUCE: return 2 * (n - 1);
    case SCHED_RHD_ALLREDUCE: {
      const int l = rhd_levels(n);
      return 2 * l + ((1 << l) == n ? 0 : 2);  // no power of two: a fold-in step in front, a fold-out step behind
    }
    case SCHED_RING_ALLGATHER: return n;
    case SCHED_TREE_REDUCE: return 2 * a.pieces;  // per piece: one sub-step per child
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
    // n = P + R with P = 2^l: the first 2R ranks pair up (2j, 2j+1).  Step 1 (only when R > 0): the odd one folds its even
    // neighbour's input into its own -- (x_2j op x_2j+1), lower rank first -- and the even one sits out; the P ranks that are
    // left (virtual rank v: 2v+1 for v < R, v + R otherwise) halve and double as a power of two does; in a last step the even
    // ranks fetch the result from their neighbours.  With R > 0 every rank's accumulator is its RECEIVE buffer from step 1 on
    // (an unpaired rank copies its input there), so the halving steps never read a send buffer.
    const int l = rhd_levels(n);
    const int P = 1 << l, R = n - P, off = R ? 1 : 0;
    const bool idle = me < 2 * R && (me & 1) == 0;
    const int vr = me < 2 * R ? me / 2 : me - R;
    const uint64_t whole = a.count * es;
    if (R && g == 1) {
      if (idle) return;
      st->lo = 0;
      st->hi = whole;
      if (me < 2 * R) {
        st->ns = 2;
        st->A = send[me - 1];
        st->B = send[me];
      } else {
        st->A = send[me];
        st->ns = st->A == st->D ? 0 : 1;
      }
      const int v1 = vr ^ (P >> 1);
      st->sig[0] = v1 < R ? 2 * v1 + 1 : v1 + R;
      return;
    }
    if (R && g == 2 * l + 2) {
      if (!idle) return;
      st->wait_rank = me + 1;
      st->wait_val = (uint32_t)(2 * l + 1);
      st->ns = 1;
      st->A = recv[me + 1];
      st->lo = 0;
      st->hi = whole;
      return;
    }
    if (idle) return;
    const int k = g - off;  // 1 ... 2l: the step among the P ranks
    // the ranges: R_0 = the buffer, R_{j+1} = the half of R_j this rank keeps at halving step j
    const int level = k <= l ? k - 1 : 2 * l - k;  // halving step j = k-1; doubling undoes level 2l-k
    uint64_t lo = 0, hi = whole;
    uint64_t klo = 0, khi = 0, olo = 0, ohi = 0;  // kept half / other half at `level`
    for (int j = 0; j <= level; j++) {
      const int d = P >> (j + 1);
      const uint64_t mid = lo + (((hi - lo) / 2) & ~(uint64_t)15);
      if (vr & d) {
        klo = mid, khi = hi, olo = lo, ohi = mid;
      } else {
        klo = lo, khi = mid, olo = mid, ohi = hi;
      }
      lo = klo;
      hi = khi;
    }
    const int vp = vr ^ (P >> (level + 1));
    const int p = vp < R ? 2 * vp + 1 : vp + R;
    if (g >= 2) {
      st->wait_rank = p;
      st->wait_val = (uint32_t)(g - 1);
    }
    if (k < 2 * l) {
      const int nlevel = k + 1 <= l ? k : 2 * l - k - 1;
      const int vn = vr ^ (P >> (nlevel + 1));
      st->sig[0] = vn < R ? 2 * vn + 1 : vn + R;
    } else if (me < 2 * R) {
      st->sig[0] = me - 1;  // my even neighbour fetches the result
    }
    if (k <= l) {  // halving: my half of the partner's accumulator joins mine
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
  if (a.sched == SCHED_TREE_REDUCE) {
    // The binary tree of the broadcast, upwards: node v (relative to the root) folds its children's partial results into its
    // own -- (x_v op T(2v+1)) op T(2v+2) -- piece by piece: sub-step 1 of a piece takes the first child, sub-step 2 the second.
    // A leaf's partial result is its input, complete when it announces itself; an inner node's lies in its receive buffer (the
    // root's: the result; anybody else's: a registered block dsync.cpp lends it) and is announced piece by piece, after the
    // second sub-step.  Step numbers: 2 * piece + sub-step (1-based).
    const int v = (me - a.root + n) % n;
    const int piece = (g - 1) / 2, sub = (g - 1) % 2;
    const int cv = 2 * v + 1 + sub;
    uint64_t lo, hi;
    chunk_bytes(a.count * es, 1, a.pieces, piece, &lo, &hi);
    if (cv < n) {
      const int c = (cv + a.root) % n;
      const bool child_inner = 2 * cv + 1 < n;
      if (child_inner) {
        st->wait_rank = c;
        st->wait_val = (uint32_t)(2 * (piece + 1));
      }
      st->ns = 2;
      st->lo = lo;
      st->hi = hi;
      st->A = child_inner ? recv[c] : send[c];
      st->B = sub == 0 ? send[me] : recv[me];
    }
    if (sub == 1 && v != 0 && 2 * v + 1 < n) st->sig[0] = ((v - 1) / 2 + a.root) % n;
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
