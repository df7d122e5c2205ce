// sched_steps.h -- the step program of the stepped kernels (sched.hip), as a function both the device and the host
// compile: the kernel calls sched_step() for every step it runs, and xmpi_sched_dump (api.cpp) prints the very same steps
// for the CPU simulator (tests/sched_sim.py), which executes all ranks' programs under random interleavings and checks
// the data flow and its hazards without a GPU.
//
// Every schedule has a PULL and a PUSH form (DsyncSchedArgs::push).  A step is: wait for one peer's flag, up to two MOVES
// (D[x] (and D2[x]) = A[x], A[x] op B[x] or C[x] op (A[x] op B[x]) over a byte range of the buffer), signal up to two peers.
// In the pull form A is the peer's memory and D this rank's; in the push form every operand is this rank's memory -- its own
// input, and what peers have stored here -- and D is the PEER's: its receive buffer, or its LANDING block (a registered block
// the peer lends, dsync.cpp) where the receive buffer still holds an operand nobody has combined yet.  The operands, their
// order and their association are the same in both forms, so the results are the same bits.
#pragma once
#include <cstdint>

#include "kernels.h"

#if defined(__HIPCC__)
#define XMPI_HD __host__ __device__ inline
#else
#define XMPI_HD inline
#endif

namespace xmpi {

struct SchedMove {
  int32_t ns;        // 1 = copy A, 2 = A op B, 3 = C op (A op B)
  int32_t pad;
  uint64_t D, D2;    // bases: byte offset x of the buffer is at base + x; D2 = 0: one destination
  uint64_t A, B, C;
  uint64_t lo, hi;   // byte range of the buffer this move covers
};

struct SchedStep {
  int32_t wait_rank;   // whose step this one needs (-1: nobody's -- the rendezvous was enough)
  uint32_t wait_val;   // ... and which
  int32_t sig[2];      // who is told when this step is done (-1: nobody)
  uint32_t sig_val;
  int32_t nmv;         // moves of this step (0: a wait and / or a signal only)
  SchedMove mv[2];
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

// ---- landing blocks of the push forms ---------------------------------------------------------------------------------------
// a whole buffer's slot in a landing block (addressed by the buffer's own byte offsets)
XMPI_HD uint64_t land_stride(uint64_t whole) { return ((whole + 15) & ~(uint64_t)15) + 16; }
// halving: what a rank keeps after j halvings is at most whole / 2^j + 32 bytes (the cut is rounded down to 16 bytes at every
// level); level j's landing region lies behind the fold-in slot (present when n is no power of two) and the levels before it
XMPI_HD uint64_t rhd_level_bytes(uint64_t whole, int j) { return (((whole >> j) + 15) & ~(uint64_t)15) + 64; }
XMPI_HD uint64_t rhd_level_off(uint64_t whole, int n, int k) {
  uint64_t off = ((1 << rhd_levels(n)) == n) ? 0 : land_stride(whole);
  for (int j = 1; j < k; j++) off += rhd_level_bytes(whole, j);
  return off;
}
// bytes of the landing block rank a.d.me must lend (0: none).  in_place: its send and receive buffers are the same memory.
XMPI_HD uint64_t sched_land_bytes(const DsyncSchedArgs& a, bool in_place) {
  if (!a.push) return 0;
  const int n = a.d.n, me = a.d.me;
  const uint64_t whole = a.count * a.elem_size;
  switch (a.sched) {
    case SCHED_RING_ALLREDUCE: return in_place ? land_stride(whole) : 0;  // (out of place the receive buffer is the landing)
    case SCHED_RHD_ALLREDUCE: {
      const int l = rhd_levels(n), R = n - (1 << l);
      if (me < 2 * R && (me & 1) == 0) return 0;  // sits the halving out
      return rhd_level_off(whole, n, l + 1);
    }
    case SCHED_TREE_REDUCE: {
      const int v = (me - a.root + n) % n;
      return land_stride(whole) * (uint64_t)((2 * v + 1 < n) + (2 * v + 2 < n));  // one slot per child
    }
    default: return 0;
  }
}

XMPI_HD int sched_nsteps(const DsyncSchedArgs& a) {
  const int n = a.d.n;
  switch (a.sched) {
    case SCHED_RING_ALLREDUCE: return a.push ? 2 * n - 1 : 2 * (n - 1);  // (push: a last step that only waits for the last chunk)
    case SCHED_RHD_ALLREDUCE: {
      const int l = rhd_levels(n);
      // no power of two: a fold-in step in front, a fold-out step behind
      return 2 * l + (a.push ? 1 : 0) + ((1 << l) == n ? 0 : 2);
    }
    case SCHED_RING_ALLGATHER: return n;
    case SCHED_TREE_REDUCE: return 2 * a.pieces;  // per piece: one sub-step per child
    default: return a.pieces;
  }
}

XMPI_HD void sched_move(SchedStep* st, int ns, uint64_t D, uint64_t D2, uint64_t A, uint64_t B, uint64_t C, uint64_t lo, uint64_t hi) {
  if (lo >= hi || ns < 1) return;
  if (!D) {  // (the only destination is the second one)
    D = D2;
    D2 = 0;
  }
  if (!D) return;
  SchedMove& m = st->mv[st->nmv++];
  m.ns = ns;
  m.pad = 0;
  m.D = D;
  m.D2 = D2;
  m.A = A;
  m.B = ns >= 2 ? B : 0;
  m.C = ns >= 3 ? C : 0;
  m.lo = lo;
  m.hi = hi;
}

// what virtual rank vr of the P = 2^l halving ranks keeps (klo, khi) and gives away (olo, ohi) at halving level `level` (1-based);
// level 0: keeps everything
XMPI_HD void rhd_ranges(uint64_t whole, int P, int vr, int level, uint64_t* klo, uint64_t* khi, uint64_t* olo, uint64_t* ohi) {
  uint64_t lo = 0, hi = whole;
  *klo = 0, *khi = whole, *olo = *ohi = 0;
  for (int j = 0; j < level; j++) {
    const int d = P >> (j + 1);
    const uint64_t mid = lo + (((hi - lo) / 2) & ~(uint64_t)15);
    if (vr & d) {
      *klo = mid, *khi = hi, *olo = lo, *ohi = mid;
    } else {
      *klo = lo, *khi = mid, *olo = mid, *ohi = hi;
    }
    lo = *klo;
    hi = *khi;
  }
}

// ---- the push forms ------------------------------------------------------------------------------------------------------------
// land[r]: rank r's landing block as addressable from here (0: it lends none)
XMPI_HD void sched_step_push(const DsyncSchedArgs& a, const uint64_t* send, const uint64_t* recv, const uint64_t* land, int g, int ch,
                             SchedStep* st) {
  const int n = a.d.n, me = a.d.me;
  const uint32_t es = a.elem_size;
  const uint64_t whole = a.count * es;
  if (a.sched == SCHED_RING_ALLREDUCE || a.sched == SCHED_RING_ALLGATHER) {
    int pos = 0;
    for (int i = 0; i < n; i++)
      if (a.order[ch][i] == me) pos = i;
    const int prev = a.order[ch][(pos + n - 1) % n], next = a.order[ch][(pos + 1) % n];
    if (g >= 2) {
      st->wait_rank = prev;
      st->wait_val = (uint32_t)(g - 1);
    }
    uint64_t lo, hi;
    if (a.sched == SCHED_RING_ALLREDUCE) {
      // Chunk c starts at position c (step 1: its raw input goes to the next rank), every rank on the way folds what arrived
      // into its own contribution -- (what arrived) op (mine), the pull form's order -- and stores the result onwards; the rank
      // at position c - 1 (step n) holds the finished chunk, keeps it and sends it on; n - 2 forwards; a last step waits for
      // the last chunk to arrive.  Partial results land in the next rank's RECEIVE buffer -- free until the finished chunk
      // arrives, which the chain of flags puts behind the partial's consumption -- unless that buffer is also its input (in
      // place: it lends a landing block).
      const int last = 2 * n - 1;
      if (g < last) st->sig[0] = next;
      const uint64_t Lme = land[me] ? land[me] : recv[me], Lnext = land[next] ? land[next] : recv[next];
      if (g <= n - 1) {
        chunk_bytes(a.count, es, n, (pos + n - g + 1) % n, &lo, &hi);
        if (g == 1) sched_move(st, 1, Lnext, 0, send[me], 0, 0, lo, hi);
        else sched_move(st, 2, Lnext, 0, Lme, send[me], 0, lo, hi);
      } else if (g == n) {
        chunk_bytes(a.count, es, n, (pos + 1) % n, &lo, &hi);
        sched_move(st, 2, recv[next], recv[me], Lme, send[me], 0, lo, hi);
      } else if (g < last) {
        chunk_bytes(a.count, es, n, (pos + 1 + 2 * n - (g - n)) % n, &lo, &hi);
        sched_move(st, 1, recv[next], 0, recv[me], 0, 0, lo, hi);
      }
    } else {
      const uint64_t blk = a.count * es;
      if (g < n) st->sig[0] = next;
      if (g == 1) {  // my own block: into the next rank's buffer and into its place in mine
        lo = (uint64_t)me * blk;
        const uint64_t A = send[me] - lo;
        sched_move(st, 1, recv[next], A == recv[me] ? 0 : recv[me], A, 0, 0, lo, lo + blk);
      } else if (g < n) {  // the block that arrived one step ago travels on
        const int r = a.order[ch][(pos + n - (g - 1)) % n];
        lo = (uint64_t)r * blk;
        sched_move(st, 1, recv[next], 0, recv[me], 0, 0, lo, lo + blk);
      }
    }
    return;
  }
  if (a.sched == SCHED_RHD_ALLREDUCE) {
    // The pull form's pairs, ranges and association (sched_step below), the data going the other way: at halving level k a rank
    // stores the half it gives away into its partner's landing region of that level -- a region of its own per level, written
    // once per collective -- and the partner folds it into its accumulator WHILE it cuts that for level k + 1: the half it keeps
    // goes to its own receive buffer, the half it gives away straight into the next partner's landing region.  The last halving
    // step stores the finished range into the partner's receive buffer as well (the first doubling step); l - 1 forwards; a last
    // step waits.  No power of two: the even rank of a pair stores its input into the odd one's fold-in slot and receives the
    // result at the end.
    const int l = rhd_levels(n);
    const int P = 1 << l, R = n - P, off = R ? 1 : 0;
    const bool idle = me < 2 * R && (me & 1) == 0, paired = me < 2 * R && (me & 1) == 1;
    const int vr = me < 2 * R ? me / 2 : me - R;
    const int last = 2 * l + 1 + off;  // the step in which the halving ranks wait for their last range
    auto real = [&](int v) { return v < R ? 2 * v + 1 : v + R; };
    if (idle) {
      if (g == 1) {
        sched_move(st, 1, land[me + 1], 0, send[me], 0, 0, 0, whole);
        st->sig[0] = me + 1;
      } else if (g == last + 1) {
        st->wait_rank = me + 1;
        st->wait_val = (uint32_t)last;
      }
      return;
    }
    if (R && (g == 1 || g == last + 1)) return;
    const int k = g - off;  // 1 ... 2l + 1: the step among the P ranks
    uint64_t klo, khi, olo, ohi;
    if (k == 1) {
      rhd_ranges(whole, P, vr, 1, &klo, &khi, &olo, &ohi);
      const int p = real(vr ^ (P >> 1));
      const uint64_t Lp = land[p] + rhd_level_off(whole, n, 1) - olo;
      if (paired) {  // (my even neighbour's input) op (mine): the half I keep stays here, the other half goes to the partner
        st->wait_rank = me - 1;
        st->wait_val = 1;
        sched_move(st, 2, recv[me], 0, land[me], send[me], 0, klo, khi);
        sched_move(st, 2, Lp, 0, land[me], send[me], 0, olo, ohi);
      } else {
        sched_move(st, 1, Lp, 0, send[me], 0, 0, olo, ohi);
      }
      st->sig[0] = p;
      return;
    }
    if (k <= l + 1) {
      const int j = k - 1;  // the level whose landing region is folded now
      const int pj = real(vr ^ (P >> j));
      st->wait_rank = pj;
      st->wait_val = (uint32_t)(g - 1);
      rhd_ranges(whole, P, vr, j, &klo, &khi, &olo, &ohi);
      const uint64_t A = land[me] + rhd_level_off(whole, n, j) - klo;
      // my accumulator: my input until something has been stored into my receive buffer (a paired rank: by the fold-in step)
      const uint64_t B = (j == 1 && !paired) ? send[me] : recv[me];
      if (k <= l) {
        uint64_t k2lo, k2hi, o2lo, o2hi;
        rhd_ranges(whole, P, vr, k, &k2lo, &k2hi, &o2lo, &o2hi);
        const int p = real(vr ^ (P >> k));
        sched_move(st, 2, recv[me], 0, A, B, 0, k2lo, k2hi);
        sched_move(st, 2, land[p] + rhd_level_off(whole, n, k) - o2lo, 0, A, B, 0, o2lo, o2hi);
        st->sig[0] = p;
      } else {  // the last halving step: the finished range, to my receive buffer and to the partner's (the first doubling step)
        sched_move(st, 2, recv[pj], recv[me], A, B, 0, klo, khi);
        st->sig[0] = pj;
      }
      return;
    }
    if (k <= 2 * l) {  // doubling: everything I hold by now goes to the partner of the next level up
      const int i = k - (l + 1);  // 1 ... l - 1
      const int p = real(vr ^ (P >> (l - i))), pw = real(vr ^ (P >> (l - i + 1)));
      st->wait_rank = pw;
      st->wait_val = (uint32_t)(g - 1);
      rhd_ranges(whole, P, vr, l - i, &klo, &khi, &olo, &ohi);
      sched_move(st, 1, recv[p], 0, recv[me], 0, 0, klo, khi);
      st->sig[0] = p;
      return;
    }
    // k == 2l + 1: the last range arrives; a paired rank hands the result to its even neighbour
    st->wait_rank = real(vr ^ (P >> 1));
    st->wait_val = (uint32_t)(g - 1);
    if (paired) {
      sched_move(st, 1, recv[me - 1], 0, recv[me], 0, 0, 0, whole);
      st->sig[0] = me - 1;
    }
    return;
  }
  const int v = (me - a.root + n) % n;
  const int c1 = 2 * v + 1, c2 = 2 * v + 2;
  uint64_t lo, hi;
  if (a.sched == SCHED_TREE_REDUCE) {
    // every node stores its partial result -- a leaf: its input -- into its parent's landing slot (one per child), piece by
    // piece; an inner node folds (first child's) op (mine), then (second child's) op (that), the pull form's order, in the
    // second sub-step of a piece, when both have arrived, and stores the result upwards (the root: into its receive buffer)
    const int piece = (g - 1) / 2, sub = (g - 1) % 2;
    chunk_bytes(whole, 1, a.pieces, piece, &lo, &hi);
    const uint64_t stride = land_stride(whole);
    const int parent = v ? ((v - 1) / 2 + a.root) % n : -1;
    const uint64_t up = v ? land[parent] + (uint64_t)((v - 1) % 2) * stride : recv[me];
    if (c1 >= n) {  // a leaf
      if (sub == 1) {
        sched_move(st, 1, up, 0, send[me], 0, 0, lo, hi);
        st->sig[0] = parent;
      }
      return;
    }
    if (sub == 0) {
      st->wait_rank = (c1 + a.root) % n;
      st->wait_val = (uint32_t)(2 * (piece + 1));
      return;
    }
    if (c2 < n) {
      st->wait_rank = (c2 + a.root) % n;
      st->wait_val = (uint32_t)(2 * (piece + 1));
      sched_move(st, 3, up, 0, land[me], send[me], land[me] + stride, lo, hi);
    } else {
      sched_move(st, 2, up, 0, land[me], send[me], 0, lo, hi);
    }
    if (v) st->sig[0] = parent;
    return;
  }
  // SCHED_TREE_BCAST: piece g arrives from the parent (the root: is there) and is stored into both children's buffers
  chunk_bytes(whole, 1, a.pieces, g - 1, &lo, &hi);
  if (v != 0) {
    st->wait_rank = ((v - 1) / 2 + a.root) % n;
    st->wait_val = (uint32_t)g;
  }
  const uint64_t d1 = c1 < n ? recv[(c1 + a.root) % n] : 0, d2 = c2 < n ? recv[(c2 + a.root) % n] : 0;
  sched_move(st, 1, d1, d2, recv[me], 0, 0, lo, hi);
  if (c1 < n) st->sig[0] = (c1 + a.root) % n;
  if (c2 < n) st->sig[1] = (c2 + a.root) % n;
}

// step g (1-based) of this rank on ring channel `ch`
XMPI_HD void sched_step(const DsyncSchedArgs& a, const uint64_t* send, const uint64_t* recv, const uint64_t* land, int g, int ch,
                        SchedStep* st) {
  const int n = a.d.n, me = a.d.me;
  const uint32_t es = a.elem_size;
  st->wait_rank = -1;
  st->wait_val = 0;
  st->sig[0] = st->sig[1] = -1;
  st->sig_val = (uint32_t)g;
  st->nmv = 0;
  if (a.push) {
    sched_step_push(a, send, recv, land, g, ch, st);
    return;
  }
  const uint64_t D = recv[me];
  uint64_t lo, hi;
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
        chunk_bytes(a.count, es, n, (pos + n - g) % n, &lo, &hi);
        sched_move(st, 2, D, 0, g == 1 ? send[prev] : recv[prev], send[me], 0, lo, hi);
      } else {  // allgather: the finished chunk (pos + 1 - t) travels on
        const int t = g - (n - 1);
        chunk_bytes(a.count, es, n, (pos + 1 + n - t) % n, &lo, &hi);
        sched_move(st, 1, D, 0, recv[prev], 0, 0, lo, hi);
      }
    } else {
      const uint64_t blk = a.count * es;
      if (g < n) st->sig[0] = next;
      if (g == 1) {  // my own block into its place
        lo = (uint64_t)me * blk;
        const uint64_t A = send[me] - lo;
        if (A != D) sched_move(st, 1, D, 0, A, 0, 0, lo, lo + blk);
      } else {  // the block that reached the previous rank one step ago
        const int r = a.order[ch][(pos + n - (g - 1)) % n];
        lo = (uint64_t)r * blk;
        sched_move(st, 1, D, 0, recv[prev], 0, 0, lo, lo + blk);
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
      if (me < 2 * R) sched_move(st, 2, D, 0, send[me - 1], send[me], 0, 0, whole);
      else if (send[me] != D) sched_move(st, 1, D, 0, send[me], 0, 0, 0, whole);
      const int v1 = vr ^ (P >> 1);
      st->sig[0] = v1 < R ? 2 * v1 + 1 : v1 + R;
      return;
    }
    if (R && g == 2 * l + 2) {
      if (!idle) return;
      st->wait_rank = me + 1;
      st->wait_val = (uint32_t)(2 * l + 1);
      sched_move(st, 1, D, 0, recv[me + 1], 0, 0, 0, whole);
      return;
    }
    if (idle) return;
    const int k = g - off;  // 1 ... 2l: the step among the P ranks
    // the ranges: R_0 = the buffer, R_{j+1} = the half of R_j this rank keeps at halving step j
    const int level = k <= l ? k - 1 : 2 * l - k;  // halving step j = k-1; doubling undoes level 2l-k
    uint64_t klo, khi, olo, ohi;  // kept half / other half at `level`
    rhd_ranges(whole, P, vr, level + 1, &klo, &khi, &olo, &ohi);
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
      sched_move(st, 2, D, 0, g == 1 ? send[p] : recv[p], g == 1 ? send[me] : recv[me], 0, klo, khi);
    } else {  // doubling: the partner's finished half
      sched_move(st, 1, D, 0, recv[p], 0, 0, olo, ohi);
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
    chunk_bytes(a.count * es, 1, a.pieces, piece, &lo, &hi);
    if (cv < n) {
      const int c = (cv + a.root) % n;
      const bool child_inner = 2 * cv + 1 < n;
      if (child_inner) {
        st->wait_rank = c;
        st->wait_val = (uint32_t)(2 * (piece + 1));
      }
      sched_move(st, 2, D, 0, child_inner ? recv[c] : send[c], sub == 0 ? send[me] : recv[me], 0, lo, hi);
    }
    if (sub == 1 && v != 0 && 2 * v + 1 < n) st->sig[0] = ((v - 1) / 2 + a.root) % n;
    return;
  }
  // SCHED_TREE_BCAST: piece g of the buffer comes from the parent and is announced to the children
  const int v = (me - a.root + n) % n;
  const int c1 = 2 * v + 1, c2 = 2 * v + 2;
  if (c1 < n) st->sig[0] = (c1 + a.root) % n;
  if (c2 < n) st->sig[1] = (c2 + a.root) % n;
  chunk_bytes(a.count * es, 1, a.pieces, g - 1, &lo, &hi);
  if (v != 0) {
    const int parent = ((v - 1) / 2 + a.root) % n;
    if ((v - 1) / 2 != 0) {  // the root's buffer is complete when it announces itself; anybody else's piece by piece
      st->wait_rank = parent;
      st->wait_val = (uint32_t)g;
    }
    sched_move(st, 1, D, 0, recv[parent], 0, 0, lo, hi);
  }
}

}  // namespace xmpi
