// plan.cpp -- per-rank step tables for the collectives (see plan.h).  Host logic only.
#include "plan.h"

#include <algorithm>
#include <cstdio>
#include <unordered_map>

#include "../../include/xmpi.h"
#include "../../include/xmpi_test.h"

namespace xmpi {
namespace {

size_t gcd_sz(size_t a, size_t b) { return b ? gcd_sz(b, a % b) : a; }

// chunk boundaries (in elements): `parts` contiguous chunks of [lo,hi), every interior boundary a
// multiple of `align` elements from lo; trailing chunks may be empty for tiny inputs
std::vector<size_t> split_even(size_t lo, size_t hi, int parts, size_t align) {
  std::vector<size_t> b((size_t)parts + 1);
  const size_t n = hi - lo;
  size_t base = (n + (size_t)parts - 1) / (size_t)parts;
  base = (base + align - 1) / align * align;
  for (int i = 0; i <= parts; i++) b[(size_t)i] = lo + std::min(n, (size_t)i * base);
  return b;
}

struct Atom {
  size_t off;    // elements
  size_t count;  // elements
};

// pieces of one chunk [lo,hi): each at most piece_elems
std::vector<Atom> pieces_of(size_t lo, size_t hi, size_t piece_elems) {
  std::vector<Atom> v;
  for (size_t o = lo; o < hi; o += piece_elems) v.push_back({o, std::min(piece_elems, hi - o)});
  return v;
}

struct Hazard {
  int last_writer = -1;
  std::vector<int> readers;
};

struct Builder {
  const PlanParams& P;
  Plan* plan;
  size_t es;
  size_t send_shift = 0;  // BUF_SEND offset x aliases BUF_RECV offset x + send_shift when in place
  std::unordered_map<uint64_t, Hazard> haz[2];  // [0] user buffers (recv layout), [1] temp

  Builder(const PlanParams& p, Plan* out) : P(p), plan(out), es(p.elem_size) {}

  Hazard& rec(int buf, size_t off) {
    if (buf == BUF_TEMP) return haz[1][off];
    return haz[0][buf == BUF_SEND ? off + send_shift : off];
  }

  static void add_dep(Step& s, int d, int self) {
    if (d < 0 || d == self) return;
    for (int i = 0; i < s.ndeps; i++)
      if (s.deps[i] == d) return;
    if (s.ndeps < kMaxDeps) {
      s.deps[s.ndeps++] = d;
    } else {  // keep the latest ones: earlier steps on the same hazard chain are implied
      int mi = 0;
      for (int i = 1; i < kMaxDeps; i++)
        if (s.deps[i] < s.deps[mi]) mi = i;
      if (s.deps[mi] < d) s.deps[mi] = d;
    }
  }

  int emit(Step s, bool reads_src, bool writes_dst) {
    const int idx = (int)plan->steps.size();
    if (reads_src) {
      Hazard& h = rec(s.src_buf, s.src_off);
      add_dep(s, h.last_writer, idx);
    }
    if (writes_dst) {
      Hazard& h = rec(s.dst_buf, s.dst_off);
      add_dep(s, h.last_writer, idx);
      for (int r : h.readers) add_dep(s, r, idx);
      h.readers.clear();
      h.last_writer = idx;
    }
    if (reads_src) {
      Hazard& h = rec(s.src_buf, s.src_off);
      if (h.last_writer != idx) h.readers.push_back(idx);
    }
    plan->steps.push_back(s);
    return idx;
  }

  int send(int peer, int lane, int buf, const Atom& a) {
    Step s;
    s.kind = STEP_SEND; s.peer = peer; s.lane = lane;
    s.src_buf = buf; s.src_off = a.off * es; s.bytes = a.count * es;
    return emit(s, true, false);
  }
  int recv_reduce(int peer, int lane, int a_buf, int dst_buf, const Atom& a) {
    Step s;
    s.kind = STEP_RECV_REDUCE; s.peer = peer; s.lane = lane;
    s.src_buf = a_buf; s.src_off = a.off * es; s.dst_buf = dst_buf; s.dst_off = a.off * es;
    s.bytes = a.count * es;
    return emit(s, true, true);
  }
  int recv_copy(int peer, int lane, int dst_buf, size_t dst_elem_off, size_t count) {
    Step s;
    s.kind = STEP_RECV_COPY; s.peer = peer; s.lane = lane;
    s.dst_buf = dst_buf; s.dst_off = dst_elem_off * es; s.bytes = count * es;
    return emit(s, false, true);
  }
  int recv_reduce_send(int peer, int lane, int peer2, int lane2, int a_buf, int dst_buf, const Atom& a, bool keep) {
    Step s;
    s.kind = STEP_RECV_REDUCE_SEND; s.peer = peer; s.lane = lane; s.peer2 = peer2; s.lane2 = lane2;
    s.src_buf = a_buf; s.src_off = a.off * es; s.dst_buf = dst_buf; s.dst_off = a.off * es;
    s.bytes = a.count * es; s.keep_local = keep ? 1 : 0;
    return emit(s, true, keep);
  }
  int recv_copy_send(int peer, int lane, int peer2, int lane2, int dst_buf, const Atom& a) {
    Step s;
    s.kind = STEP_RECV_COPY_SEND; s.peer = peer; s.lane = lane; s.peer2 = peer2; s.lane2 = lane2;
    s.dst_buf = dst_buf; s.dst_off = a.off * es; s.bytes = a.count * es;
    return emit(s, false, true);
  }
  int recv_hold(int peer, int lane, size_t count) {
    Step s;
    s.kind = STEP_RECV_HOLD; s.peer = peer; s.lane = lane; s.bytes = count * es;
    return emit(s, false, false);
  }
  int reduce_n(int local_buf, int dst_buf, const Atom& a, const std::vector<int>& srcs) {
    Step s;
    s.kind = STEP_REDUCE_N;
    s.src_buf = local_buf; s.src_off = a.off * es; s.dst_buf = dst_buf; s.dst_off = a.off * es;
    s.bytes = a.count * es;
    s.nsrcs = (int)srcs.size();
    for (int i = 0; i < s.nsrcs; i++) s.srcs[i] = srcs[(size_t)i];
    return emit(s, true, true);
  }
  int local_copy(int src_buf, size_t src_elem_off, int dst_buf, size_t dst_elem_off, size_t count) {
    Step s;
    s.kind = STEP_LOCAL_COPY;
    s.src_buf = src_buf; s.src_off = src_elem_off * es; s.dst_buf = dst_buf; s.dst_off = dst_elem_off * es;
    s.bytes = count * es;
    return emit(s, true, true);
  }
};

size_t align_elems(size_t es) { return std::max<size_t>(1, 16 / es); }

size_t piece_elems_of(const PlanParams& p) {
  const size_t al = align_elems(p.elem_size);
  size_t pe = p.piece_bytes / p.elem_size / al * al;
  return std::max(pe, al);
}

int effective_channels(const PlanParams& p, size_t total_elems) {
  int c = std::max(1, p.channels);
  c = std::min(c, ring_channel_count(p.size) * std::max(1, p.lanes));
  // do not shred small messages: at least 64 KiB per chunk per channel
  const size_t bytes = total_elems * p.elem_size;
  const size_t min_per_channel = (size_t)p.size * (64u << 10);
  while (c > 1 && bytes / (size_t)c < min_per_channel) c--;
  return c;
}

// ---- allreduce: multi-channel ring -------------------------------------------------------------

void build_allreduce_ring(Builder& b) {
  const PlanParams& P = b.P;
  const int N = P.size;
  const size_t al = align_elems(P.elem_size), pe = piece_elems_of(P);
  const int C = effective_channels(P, P.count);
  b.plan->channels = C;
  const int nstr = ring_channel_count(N);
  const std::vector<size_t> slice = split_even(0, P.count, C, al);

  struct Chan {
    int pos, next, prev, lane;
    std::vector<size_t> cb;  // chunk bounds
  };
  std::vector<Chan> ch((size_t)C);
  for (int c = 0; c < C; c++) {
    std::vector<int> ord;
    ring_order(N, c % nstr, &ord);
    Chan& k = ch[(size_t)c];
    k.pos = (int)(std::find(ord.begin(), ord.end(), P.rank) - ord.begin());
    k.next = ord[(size_t)((k.pos + 1) % N)];
    k.prev = ord[(size_t)((k.pos + N - 1) % N)];
    k.lane = (c / nstr) % std::max(1, P.lanes);
    k.cb = split_even(slice[(size_t)c], slice[(size_t)c + 1], N, al);
  }
  if (P.fuse && P.fifo_depth >= 2) {
    // Fused ring: a chunk that arrives is combined and forwarded by ONE kernel (receive-reduce-send),
    // partial sums are never stored locally; only the last reduce-scatter step keeps its result (the
    // rank's final chunk) while already forwarding it as the first allgather hop; allgather hops
    // store and forward in one kernel (receive-copy-send).  2N-1 launches per piece instead of 4(N-1).
    //
    // Emission order.  Hop h = 0 .. 2N-3 of piece p is "pop what prev pushed at hop h, push hop h+1".
    // A fused step keeps its incoming slot until it can push, and all ranks run the same sequence, so
    // a piece in flight occupies one slot of EVERY pipe of its ring: at most W = depth-1 pieces may be
    // in flight or every pipe is full and nobody can push (a circular wait).  Steps are therefore
    // emitted by the key (start(p) + h, h, p) with start(p) = max(p, start(p-W) + 2N-1): a sliding
    // window of W pieces, each advancing one hop per time unit.  The same key orders the pushes and
    // the pops of every pipe identically on both ends.  Full rate needs W >= 2N-1 (depth >= 2N);
    // real messages have only a few pieces per chunk, so the window rarely binds.
    // (tests/test_plan_property.py runs depths 1..5; depth 1 uses the unfused schedule.)
    std::vector<std::vector<std::vector<Atom>>> pcs((size_t)C);  // [channel][chunk] -> pieces
    size_t NP = 0;  // pieces per chunk (max over chunks)
    for (int c = 0; c < C; c++) {
      pcs[(size_t)c].resize((size_t)N);
      for (int j = 0; j < N; j++) {
        pcs[(size_t)c][(size_t)j] = pieces_of(ch[(size_t)c].cb[(size_t)j], ch[(size_t)c].cb[(size_t)j + 1], pe);
        NP = std::max(NP, pcs[(size_t)c][(size_t)j].size());
      }
    }
    const int H = 2 * N - 2;  // hops (pushes) per piece
    const size_t W = (size_t)std::max(1, P.fifo_depth - 1);
    std::vector<size_t> start(NP);
    for (size_t p = 0; p < NP; p++) start[p] = p < W ? p : std::max(p, start[p - W] + (size_t)H + 1);
    // ops: (time, hop, piece); hop -1 = the initial SEND (pushes hop 0), hop h >= 0 pops hop h
    struct Op { size_t time; int hop; size_t piece; };
    std::vector<Op> ops;
    for (size_t p = 0; p < NP; p++) {
      ops.push_back({start[p], -1, p});
      for (int h = 0; h < H; h++) ops.push_back({start[p] + (size_t)h + 1, h, p});
    }
    std::sort(ops.begin(), ops.end(), [](const Op& x, const Op& y) {
      if (x.time != y.time) return x.time < y.time;
      if (x.hop != y.hop) return x.hop < y.hop;
      return x.piece < y.piece;
    });
    for (const Op& o : ops)
      for (int c = 0; c < C; c++) {
        const Chan& k = ch[(size_t)c];
        const int h = o.hop;
        if (h < 0) {  // hop 0: the own chunk starts its journey
          const std::vector<Atom>& v = pcs[(size_t)c][(size_t)k.pos];
          if (o.piece < v.size()) b.send(k.next, k.lane, BUF_SEND, v[o.piece]);
          continue;
        }
        // reduce-scatter hops 0..N-2 carry chunk (pos-h-1); allgather hops N-1..2N-3 chunk (pos-(h-(N-1)))
        const int chunk = h < N - 1 ? ((k.pos - h - 1) % N + N) % N : ((k.pos - (h - (N - 1))) % N + N) % N;
        const std::vector<Atom>& v = pcs[(size_t)c][(size_t)chunk];
        if (o.piece >= v.size()) continue;
        if (h < N - 1) b.recv_reduce_send(k.prev, k.lane, k.next, k.lane, BUF_SEND, BUF_RECV, v[o.piece], /*keep_local=*/h == N - 2);
        else if (h < H - 1) b.recv_copy_send(k.prev, k.lane, k.next, k.lane, BUF_RECV, v[o.piece]);
        else b.recv_copy(k.prev, k.lane, BUF_RECV, v[o.piece].off, v[o.piece].count);
      }
    return;
  }
  // reduce-scatter: step s sends chunk (pos-s), receives chunk (pos-s-1) and folds it in
  for (int s = 0; s < N - 1; s++) {
    for (int c = 0; c < C; c++) {
      const Chan& k = ch[(size_t)c];
      const int cs = ((k.pos - s) % N + N) % N, cr = ((k.pos - s - 1) % N + N) % N;
      for (const Atom& a : pieces_of(k.cb[(size_t)cs], k.cb[(size_t)cs + 1], pe))
        b.send(k.next, k.lane, s == 0 ? BUF_SEND : BUF_RECV, a);
      for (const Atom& a : pieces_of(k.cb[(size_t)cr], k.cb[(size_t)cr + 1], pe))
        b.recv_reduce(k.prev, k.lane, BUF_SEND, BUF_RECV, a);  // each chunk is folded once
    }
  }
  // allgather: step s sends chunk (pos+1-s), receives chunk (pos-s)
  for (int s = 0; s < N - 1; s++) {
    for (int c = 0; c < C; c++) {
      const Chan& k = ch[(size_t)c];
      const int cs = ((k.pos + 1 - s) % N + N) % N, cr = ((k.pos - s) % N + N) % N;
      for (const Atom& a : pieces_of(k.cb[(size_t)cs], k.cb[(size_t)cs + 1], pe))
        b.send(k.next, k.lane, BUF_RECV, a);
      for (const Atom& a : pieces_of(k.cb[(size_t)cr], k.cb[(size_t)cr + 1], pe))
        b.recv_copy(k.prev, k.lane, BUF_RECV, a.off, a.count);
    }
  }
}

// ---- allreduce: recursive halving (reduce-scatter) + recursive doubling (allgather) ------------

void build_allreduce_rhd(Builder& b) {
  const PlanParams& P = b.P;
  const int N = P.size, r = P.rank;
  const size_t al = align_elems(P.elem_size), pe = piece_elems_of(P);
  const std::vector<size_t> cb = split_even(0, P.count, N, al);
  struct Level {
    int peer, klo, khi, glo, ghi;  // partner, kept chunk range, given-away chunk range
  };
  std::vector<Level> lv;
  int lo = 0, hi = N;  // chunk range this rank is still responsible for
  for (int mask = N >> 1; mask >= 1; mask >>= 1) {
    const int mid = (lo + hi) / 2;
    const bool keep_low = (r & mask) == 0;
    Level L;
    L.peer = r ^ mask;
    L.klo = keep_low ? lo : mid;
    L.khi = keep_low ? mid : hi;
    L.glo = keep_low ? mid : lo;
    L.ghi = keep_low ? hi : mid;
    const bool first = lv.empty();
    for (int c = L.glo; c < L.ghi; c++)
      for (const Atom& a : pieces_of(cb[(size_t)c], cb[(size_t)c + 1], pe))
        b.send(L.peer, 0, first ? BUF_SEND : BUF_RECV, a);
    for (int c = L.klo; c < L.khi; c++)
      for (const Atom& a : pieces_of(cb[(size_t)c], cb[(size_t)c + 1], pe))
        b.recv_reduce(L.peer, 0, first ? BUF_SEND : BUF_RECV, BUF_RECV, a);
    lv.push_back(L);
    lo = L.klo;
    hi = L.khi;
  }
  // doubling: undo the halving levels in reverse; at level k this rank holds its kept range of
  // that level complete and the partner holds the range given away to it
  for (int k = (int)lv.size() - 1; k >= 0; k--) {
    const Level& L = lv[(size_t)k];
    for (int c = L.klo; c < L.khi; c++)
      for (const Atom& a : pieces_of(cb[(size_t)c], cb[(size_t)c + 1], pe)) b.send(L.peer, 0, BUF_RECV, a);
    for (int c = L.glo; c < L.ghi; c++)
      for (const Atom& a : pieces_of(cb[(size_t)c], cb[(size_t)c + 1], pe))
        b.recv_copy(L.peer, 0, BUF_RECV, a.off, a.count);
  }
}

// ---- allreduce: full-mesh one-hop reduce-scatter + allgather; rank-order fold ------------------

// Small messages: one-shot.  Every rank pushes its WHOLE buffer to every peer (one batched launch
// over all links) and folds the N buffers itself, in rank order: two stages instead of four, at the
// price of (N-1)*S instead of 2(N-1)/N*S bytes per rank on the wire -- irrelevant below ~1 MiB.
void build_allreduce_oneshot(Builder& b) {
  const PlanParams& P = b.P;
  const int N = P.size, r = P.rank;
  const size_t pe = piece_elems_of(P);
  const int L = std::max(1, P.lanes);
  const std::vector<Atom> pcs = pieces_of(0, P.count, pe);
  for (size_t p = 0; p < pcs.size(); p++)
    for (int d = 1; d < N; d++) b.send((r + d) % N, (int)(p % (size_t)L), BUF_SEND, pcs[p]);
  for (size_t p = 0; p < pcs.size(); p++) {
    std::vector<int> srcs;
    for (int q = 0; q < N; q++) srcs.push_back(q == r ? -1 : b.recv_hold(q, (int)(p % (size_t)L), pcs[p].count));
    b.reduce_n(BUF_SEND, BUF_RECV, pcs[p], srcs);
  }
}

void build_allreduce_direct(Builder& b) {
  const PlanParams& P = b.P;
  const int N = P.size, r = P.rank;
  if (P.count * P.elem_size <= P.oneshot_bytes) {
    build_allreduce_oneshot(b);
    return;
  }
  const size_t al = align_elems(P.elem_size), pe = piece_elems_of(P);
  const std::vector<size_t> cb = split_even(0, P.count, N, al);
  const std::vector<Atom> mine = pieces_of(cb[(size_t)r], cb[(size_t)r + 1], pe);
  // scatter my copy of every other rank's chunk (peer order staggered so that not everybody
  // targets rank 0 first)
  size_t maxp = 0;
  std::vector<std::vector<Atom>> pcs((size_t)N);
  for (int j = 0; j < N; j++) {
    pcs[(size_t)j] = pieces_of(cb[(size_t)j], cb[(size_t)j + 1], pe);
    maxp = std::max(maxp, pcs[(size_t)j].size());
  }
  for (size_t p = 0; p < maxp; p++)
    for (int d = 1; d < N; d++) {
      const int j = (r + d) % N;
      if (p < pcs[(size_t)j].size()) b.send(j, (int)(p % (size_t)std::max(1, P.lanes)), BUF_SEND, pcs[(size_t)j][p]);
    }
  // fold the N contributions of my chunk in rank order
  std::vector<int> red((size_t)mine.size());
  for (size_t p = 0; p < mine.size(); p++) {
    std::vector<int> srcs;
    for (int q = 0; q < N; q++) {
      if (q == r) srcs.push_back(-1);
      else srcs.push_back(b.recv_hold(q, (int)(p % (size_t)std::max(1, P.lanes)), mine[p].count));
    }
    red[p] = b.reduce_n(BUF_SEND, BUF_RECV, mine[p], srcs);
  }
  // allgather the reduced chunks
  for (size_t p = 0; p < maxp; p++) {
    if (p < mine.size())
      for (int d = 1; d < N; d++) b.send((r + d) % N, (int)(p % (size_t)std::max(1, P.lanes)), BUF_RECV, mine[p]);
    for (int d = 1; d < N; d++) {
      const int j = (r + N - d) % N;
      if (p < pcs[(size_t)j].size())
        b.recv_copy(j, (int)(p % (size_t)std::max(1, P.lanes)), BUF_RECV, pcs[(size_t)j][p].off, pcs[(size_t)j][p].count);
    }
  }
}

// ---- allgather --------------------------------------------------------------------------------

void build_allgather_ring(Builder& b) {
  const PlanParams& P = b.P;
  const int N = P.size;
  const size_t al = align_elems(P.elem_size), pe = piece_elems_of(P);
  const int C = effective_channels(P, P.count * (size_t)N);
  b.plan->channels = C;
  const int nstr = ring_channel_count(N);
  const std::vector<size_t> slice = split_even(0, P.count, C, al);  // of one rank's block
  b.local_copy(BUF_SEND, 0, BUF_RECV, (size_t)P.rank * P.count, P.count);
  struct Chan { int pos, next, prev, lane; std::vector<int> ord; };
  std::vector<Chan> ch((size_t)C);
  for (int c = 0; c < C; c++) {
    Chan& k = ch[(size_t)c];
    ring_order(N, c % nstr, &k.ord);
    k.pos = (int)(std::find(k.ord.begin(), k.ord.end(), P.rank) - k.ord.begin());
    k.next = k.ord[(size_t)((k.pos + 1) % N)];
    k.prev = k.ord[(size_t)((k.pos + N - 1) % N)];
    k.lane = (c / nstr) % std::max(1, P.lanes);
  }
  if (P.fuse && P.fifo_depth >= 2 && N > 1) {
    // fused: a block piece that arrives is stored and forwarded by one kernel (receive-copy-send);
    // same sliding-window emission order as the fused allreduce ring, with N-1 hops per piece
    std::vector<std::vector<Atom>> pcs((size_t)C);
    size_t NP = 0;
    for (int c = 0; c < C; c++) {
      pcs[(size_t)c] = pieces_of(slice[(size_t)c], slice[(size_t)c + 1], pe);
      NP = std::max(NP, pcs[(size_t)c].size());
    }
    const int H = N - 1;
    const size_t W = (size_t)std::max(1, P.fifo_depth - 1);
    std::vector<size_t> start(NP);
    for (size_t p = 0; p < NP; p++) start[p] = p < W ? p : std::max(p, start[p - W] + (size_t)H + 1);
    struct Op { size_t time; int hop; size_t piece; };
    std::vector<Op> ops;
    for (size_t p = 0; p < NP; p++) {
      ops.push_back({start[p], -1, p});
      for (int h = 0; h < H; h++) ops.push_back({start[p] + (size_t)h + 1, h, p});
    }
    std::sort(ops.begin(), ops.end(), [](const Op& x, const Op& y) {
      if (x.time != y.time) return x.time < y.time;
      if (x.hop != y.hop) return x.hop < y.hop;
      return x.piece < y.piece;
    });
    for (const Op& o : ops)
      for (int c = 0; c < C; c++) {
        const Chan& k = ch[(size_t)c];
        if (o.piece >= pcs[(size_t)c].size()) continue;
        const Atom& a = pcs[(size_t)c][o.piece];
        if (o.hop < 0) {
          b.send(k.next, k.lane, BUF_SEND, a);
          continue;
        }
        const int br = k.ord[(size_t)(((k.pos - o.hop - 1) % N + N) % N)];  // whose block arrives at this hop
        Atom dst = a;
        dst.off += (size_t)br * P.count;
        if (o.hop < H - 1) b.recv_copy_send(k.prev, k.lane, k.next, k.lane, BUF_RECV, dst);
        else b.recv_copy(k.prev, k.lane, BUF_RECV, dst.off, dst.count);
      }
    return;
  }
  for (int s = 0; s < N - 1; s++)
    for (int c = 0; c < C; c++) {
      const Chan& k = ch[(size_t)c];
      // blocks travel by ring position: step s forwards the block of the rank s positions behind
      const int bs = k.ord[(size_t)(((k.pos - s) % N + N) % N)];
      const int br = k.ord[(size_t)(((k.pos - s - 1) % N + N) % N)];
      for (const Atom& a : pieces_of(slice[(size_t)c], slice[(size_t)c + 1], pe)) {
        Atom src = a;
        if (s == 0) {
          b.send(k.next, k.lane, BUF_SEND, src);
        } else {
          src.off += (size_t)bs * P.count;
          b.send(k.next, k.lane, BUF_RECV, src);
        }
      }
      for (const Atom& a : pieces_of(slice[(size_t)c], slice[(size_t)c + 1], pe))
        b.recv_copy(k.prev, k.lane, BUF_RECV, (size_t)br * P.count + a.off, a.count);
    }
}

void build_allgather_direct(Builder& b) {
  const PlanParams& P = b.P;
  const int N = P.size, r = P.rank;
  const size_t pe = piece_elems_of(P);
  const int L = std::max(1, P.lanes);
  b.local_copy(BUF_SEND, 0, BUF_RECV, (size_t)r * P.count, P.count);
  const std::vector<Atom> pcs = pieces_of(0, P.count, pe);
  for (size_t p = 0; p < pcs.size(); p++) {
    for (int d = 1; d < N; d++) b.send((r + d) % N, (int)(p % (size_t)L), BUF_SEND, pcs[p]);
    for (int d = 1; d < N; d++) {
      const int j = (r + N - d) % N;
      b.recv_copy(j, (int)(p % (size_t)L), BUF_RECV, (size_t)j * P.count + pcs[p].off, pcs[p].count);
    }
  }
}

// ---- broadcast: binary tree rooted at `root`, pipelined in pieces ------------------------------

void build_bcast_tree(Builder& b) {
  const PlanParams& P = b.P;
  const int N = P.size;
  const size_t pe = piece_elems_of(P);
  const int vr = (P.rank - P.root + N) % N;
  const int parent = (vr == 0) ? -1 : ((vr - 1) / 2 + P.root) % N;
  const int c1 = 2 * vr + 1, c2 = 2 * vr + 2;
  for (const Atom& a : pieces_of(0, P.count, pe)) {
    if (parent >= 0) b.recv_copy(parent, 0, BUF_RECV, a.off, a.count);
    if (c1 < N) b.send((c1 + P.root) % N, 0, BUF_RECV, a);
    if (c2 < N) b.send((c2 + P.root) % N, 0, BUF_RECV, a);
  }
}

// Full-mesh broadcast: the root scatters chunk j to rank j (every link of the root carries S/N), then
// every rank forwards its chunk to the ranks that lack it (every link carries S/N once more) -- two
// hops instead of the tree's log2 N, and 1/N of the bytes per link.
void build_bcast_direct(Builder& b) {
  const PlanParams& P = b.P;
  const int N = P.size, r = P.rank, R = P.root;
  const size_t al = align_elems(P.elem_size), pe = piece_elems_of(P);
  const int L = std::max(1, P.lanes);
  const std::vector<size_t> cb = split_even(0, P.count, N, al);
  auto chunk = [&](int j) { return pieces_of(cb[(size_t)j], cb[(size_t)j + 1], pe); };
  if (r == R) {
    for (int d = 1; d < N; d++) {  // scatter
      const int j = (R + d) % N;
      const std::vector<Atom> pcs = chunk(j);
      for (size_t p = 0; p < pcs.size(); p++) b.send(j, (int)(p % (size_t)L), BUF_RECV, pcs[p]);
    }
    const std::vector<Atom> mine = chunk(R);  // the root's own chunk goes to everybody
    for (size_t p = 0; p < mine.size(); p++)
      for (int d = 1; d < N; d++) b.send((R + d) % N, (int)(p % (size_t)L), BUF_RECV, mine[p]);
    return;
  }
  const std::vector<Atom> mine = chunk(r);
  for (size_t p = 0; p < mine.size(); p++) {  // my chunk arrives from the root and goes on to the others
    b.recv_copy(R, (int)(p % (size_t)L), BUF_RECV, mine[p].off, mine[p].count);
    for (int d = 1; d < N; d++) {
      const int q = (r + d) % N;
      if (q != R) b.send(q, (int)(p % (size_t)L), BUF_RECV, mine[p]);
    }
  }
  for (int d = 1; d < N; d++) {  // everybody else's chunk, from its owner (the root's from the root)
    const int q = (r + N - d) % N;
    const std::vector<Atom> pcs = chunk(q);
    for (size_t p = 0; p < pcs.size(); p++) b.recv_copy(q, (int)(p % (size_t)L), BUF_RECV, pcs[p].off, pcs[p].count);
  }
}

// ---- reduce to root ----------------------------------------------------------------------------

void build_reduce_tree(Builder& b) {
  const PlanParams& P = b.P;
  const int N = P.size;
  const size_t pe = piece_elems_of(P);
  const int vr = (P.rank - P.root + N) % N;
  const int parent = (vr == 0) ? -1 : ((vr - 1) / 2 + P.root) % N;
  const int c1 = 2 * vr + 1, c2 = 2 * vr + 2;
  const int acc = (vr == 0) ? BUF_RECV : BUF_TEMP;
  if (vr != 0 && c1 < N) b.plan->temp_bytes = P.count * P.elem_size;
  for (const Atom& a : pieces_of(0, P.count, pe)) {
    int cur = BUF_SEND;  // where the running value of this piece lives
    if (c1 < N) {
      Step s;
      s.kind = STEP_RECV_REDUCE; s.peer = (c1 + P.root) % N; s.lane = 0;
      s.src_buf = cur; s.src_off = a.off * b.es; s.dst_buf = acc; s.dst_off = a.off * b.es;
      s.bytes = a.count * b.es;
      b.emit(s, true, true);
      cur = acc;
    }
    if (c2 < N) {
      Step s;
      s.kind = STEP_RECV_REDUCE; s.peer = (c2 + P.root) % N; s.lane = 0;
      s.src_buf = cur; s.src_off = a.off * b.es; s.dst_buf = acc; s.dst_off = a.off * b.es;
      s.bytes = a.count * b.es;
      b.emit(s, true, true);
      cur = acc;
    }
    if (parent >= 0) b.send(parent, 0, cur, a);
    else if (cur == BUF_SEND) b.local_copy(BUF_SEND, a.off, BUF_RECV, a.off, a.count);  // N == 1
  }
}

// Large messages: reduce-scatter over the full mesh (rank j folds chunk j of everybody, in rank order),
// then the folded chunks are gathered at the root: every link carries S/N instead of the root's links
// carrying S each, and the fold is spread over all ranks.
void build_reduce_scatter_gather(Builder& b) {
  const PlanParams& P = b.P;
  const int N = P.size, r = P.rank, R = P.root;
  const size_t al = align_elems(P.elem_size), pe = piece_elems_of(P);
  const int L = std::max(1, P.lanes);
  const std::vector<size_t> cb = split_even(0, P.count, N, al);
  auto chunk = [&](int j) { return pieces_of(cb[(size_t)j], cb[(size_t)j + 1], pe); };
  const int acc = (r == R) ? BUF_RECV : BUF_TEMP;
  if (r != R) b.plan->temp_bytes = P.count * P.elem_size;
  for (int d = 1; d < N; d++) {  // my contribution to everybody else's chunk
    const int j = (r + d) % N;
    const std::vector<Atom> pcs = chunk(j);
    for (size_t p = 0; p < pcs.size(); p++) b.send(j, (int)(p % (size_t)L), BUF_SEND, pcs[p]);
  }
  const std::vector<Atom> mine = chunk(r);
  for (size_t p = 0; p < mine.size(); p++) {  // fold my chunk, hand it to the root
    const int lane = (int)(p % (size_t)L);
    std::vector<int> srcs;
    for (int q = 0; q < N; q++) srcs.push_back(q == r ? -1 : b.recv_hold(q, lane, mine[p].count));
    b.reduce_n(BUF_SEND, acc, mine[p], srcs);
    if (r != R) b.send(R, lane, BUF_TEMP, mine[p]);
  }
  if (r == R)
    for (int d = 1; d < N; d++) {
      const int j = (r + N - d) % N;
      const std::vector<Atom> pcs = chunk(j);
      for (size_t p = 0; p < pcs.size(); p++) b.recv_copy(j, (int)(p % (size_t)L), BUF_RECV, pcs[p].off, pcs[p].count);
    }
}

void build_reduce_direct(Builder& b) {
  const PlanParams& P = b.P;
  const int N = P.size, r = P.rank;
  if (N > 2 && P.count * P.elem_size > P.oneshot_bytes) {
    build_reduce_scatter_gather(b);
    return;
  }
  const size_t pe = piece_elems_of(P);
  const int L = std::max(1, P.lanes);
  const std::vector<Atom> pcs = pieces_of(0, P.count, pe);
  for (size_t p = 0; p < pcs.size(); p++) {
    const int lane = (int)(p % (size_t)L);
    if (r != P.root) {
      b.send(P.root, lane, BUF_SEND, pcs[p]);
    } else {
      std::vector<int> srcs;
      for (int q = 0; q < N; q++) srcs.push_back(q == r ? -1 : b.recv_hold(q, lane, pcs[p].count));
      b.reduce_n(BUF_SEND, BUF_RECV, pcs[p], srcs);
    }
  }
}

bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

}  // namespace

static std::vector<int> ring_strides(int size) {
  // 1, N-1, 3, N-3, ... over the residues coprime to N
  std::vector<int> strides;
  for (int d = 1; d <= size / 2; d++)
    if (gcd_sz((size_t)d, (size_t)size) == 1) {
      strides.push_back(d);
      if (size - d != d) strides.push_back(size - d);
    }
  if (strides.empty()) strides.push_back(1);
  return strides;
}

// Even N >= 4: Walecki's decomposition of the complete graph K_N into N/2 - 1 edge-disjoint
// Hamiltonian cycles (rank N-1 is the hub "infinity", the others are taken mod N-1):
//     H_i = (N-1, i, i-1, i+1, i-2, i+2, ...),   i = 0 .. N/2 - 2
// Each cycle is used in both directions, so N - 2 directed rings run at once and no two of them
// share a link direction: at N = 8 that is 6 channels over 6 of the 7 xGMI links of every GPU
// (the circulant strides 1, 7, 3, 5 reach only 4 channels over 2 links).  Odd N and N = 2 keep
// the circulant strides coprime to N.
static bool use_walecki(int size) { return size >= 4 && size % 2 == 0; }

void ring_order(int size, int channel, std::vector<int>* order) {
  order->resize((size_t)size);
  if (use_walecki(size)) {
    const int ncyc = size / 2 - 1, m = size - 1;
    const int ch = channel % (2 * ncyc), i = ch / 2;
    std::vector<int> seq;
    seq.push_back(size - 1);
    seq.push_back(i % m);
    for (int k = 1; k < size / 2; k++) {
      seq.push_back(((i - k) % m + m) % m);
      seq.push_back((i + k) % m);
    }
    seq.resize((size_t)size);
    if (ch % 2) std::reverse(seq.begin(), seq.end());  // the same links, the other direction
    *order = seq;
    return;
  }
  const std::vector<int> strides = ring_strides(size);
  const int d = strides[(size_t)channel % strides.size()];
  for (int i = 0; i < size; i++) (*order)[(size_t)i] = (int)(((long)i * d) % size);
}

int ring_channel_count(int size) {
  if (use_walecki(size)) return size - 2;
  return (int)ring_strides(size).size();
}

int build_plan(const PlanParams& p, Plan* out) {
  out->steps.clear();
  out->temp_bytes = 0;
  out->channels = 1;
  if (p.size < 1 || p.rank < 0 || p.rank >= p.size || p.elem_size == 0 || p.root < 0 || p.root >= p.size ||
      p.piece_bytes < p.elem_size)
    return XMPI_ERR_ARG;
  if (p.size > kMaxSrcs) return XMPI_ERR_UNSUPPORTED;
  Builder b(p, out);
  int algo = p.algo;
  switch (p.coll) {
    case COLL_ALLREDUCE:
      // a full xGMI mesh: the one-hop schedule keeps all N-1 links busy and folds in rank order
      if (algo == XMPI_ALGO_AUTO) algo = XMPI_ALGO_DIRECT;
      if (algo == XMPI_ALGO_RHD && !is_pow2(p.size)) algo = XMPI_ALGO_RING;
      if (p.size == 1) {
        b.local_copy(BUF_SEND, 0, BUF_RECV, 0, p.count);
      } else if (algo == XMPI_ALGO_RING) {
        build_allreduce_ring(b);
      } else if (algo == XMPI_ALGO_RHD) {
        build_allreduce_rhd(b);
      } else if (algo == XMPI_ALGO_DIRECT) {
        build_allreduce_direct(b);
      } else {
        return XMPI_ERR_UNSUPPORTED;
      }
      break;
    case COLL_ALLGATHER:
      if (algo == XMPI_ALGO_AUTO) algo = XMPI_ALGO_DIRECT;
      b.send_shift = (size_t)p.rank * p.count * p.elem_size;
      if (algo == XMPI_ALGO_RING) build_allgather_ring(b);
      else if (algo == XMPI_ALGO_DIRECT) build_allgather_direct(b);
      else return XMPI_ERR_UNSUPPORTED;
      break;
    case COLL_BCAST:
      // small: the tree's log2 N latency-bound hops; large: scatter + allgather over the whole mesh
      if (algo == XMPI_ALGO_AUTO)
        algo = (p.size > 2 && p.count * p.elem_size > p.oneshot_bytes) ? XMPI_ALGO_DIRECT : XMPI_ALGO_TREE;
      if (algo == XMPI_ALGO_TREE) build_bcast_tree(b);
      else if (algo == XMPI_ALGO_DIRECT) build_bcast_direct(b);
      else return XMPI_ERR_UNSUPPORTED;
      break;
    case COLL_REDUCE:
      if (algo == XMPI_ALGO_AUTO)
        algo = (p.size > 2 && p.count * p.elem_size > p.oneshot_bytes) ? XMPI_ALGO_DIRECT : XMPI_ALGO_TREE;
      if (algo == XMPI_ALGO_TREE) build_reduce_tree(b);
      else if (algo == XMPI_ALGO_DIRECT) build_reduce_direct(b);
      else return XMPI_ERR_UNSUPPORTED;
      break;
    default:
      return XMPI_ERR_ARG;
  }
  out->algo = algo;
  return XMPI_OK;
}

std::string plan_to_text(const Plan& plan) {
  static const char* kn[] = {"SEND", "RECV_REDUCE", "RECV_COPY", "RECV_HOLD", "REDUCE_N", "LOCAL_COPY",
                             "RECV_REDUCE_SEND", "RECV_COPY_SEND"};
  std::string out;
  char line[512];
  snprintf(line, sizeof line, "plan algo=%d channels=%d temp_bytes=%zu steps=%zu\n", plan.algo, plan.channels,
           plan.temp_bytes, plan.steps.size());
  out += line;
  for (size_t i = 0; i < plan.steps.size(); i++) {
    const Step& s = plan.steps[i];
    int n = snprintf(line, sizeof line, "%zu %s peer=%d lane=%d src=%d:%zu dst=%d:%zu bytes=%zu deps=", i, kn[s.kind],
                     s.peer, s.lane, s.src_buf, s.src_off, s.dst_buf, s.dst_off, s.bytes);
    for (int d = 0; d < s.ndeps; d++) n += snprintf(line + n, sizeof line - (size_t)n, "%s%d", d ? "," : "", s.deps[d]);
    n += snprintf(line + n, sizeof line - (size_t)n, " srcs=");
    for (int d = 0; d < s.nsrcs; d++) n += snprintf(line + n, sizeof line - (size_t)n, "%s%d", d ? "," : "", s.srcs[d]);
    n += snprintf(line + n, sizeof line - (size_t)n, " peer2=%d lane2=%d keep=%d", s.peer2, s.lane2, s.keep_local);
    snprintf(line + n, sizeof line - (size_t)n, "\n");
    out += line;
  }
  return out;
}

void zc_chunk(size_t count, size_t elem_size, int size, int j, size_t* elem_off, size_t* elem_cnt) {
  const std::vector<size_t> b = split_even(0, count, std::max(1, size), align_elems(std::max<size_t>(1, elem_size)));
  const size_t k = (size_t)std::min(std::max(j, 0), std::max(1, size) - 1);
  *elem_off = b[k];
  *elem_cnt = b[k + 1] - b[k];
}

}  // namespace xmpi
