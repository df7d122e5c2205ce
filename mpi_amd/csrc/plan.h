// plan.h -- collective schedules as explicit step tables (host logic only, no HIP).
//
// The reference has no collectives (mpi.go:130 is a stub); a reference user composes them from
// Send/Receive the way examples/helloworld/helloworld.go:53-81 does.  Here every collective is
// compiled, per rank, into a list of steps over one-directional FIFO pipes between ranks
// (pipe = ring of slots in the receiver's HBM window).  The executor (engine.cpp) runs the
// table; tests/test_plan.py simulates all ranks' tables on the CPU against the oracle, so the
// N = 2/4/8 data flow is checked without any GPU.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace xmpi {

enum StepKind : int {
  STEP_SEND = 0,         // push [src_buf+src_off, bytes) into the next slot of pipe(me -> peer, lane)
  STEP_RECV_REDUCE = 1,  // dst = a op slot     (a = src_buf+src_off), pops pipe(peer -> me, lane)
  STEP_RECV_COPY = 2,    // dst = slot
  STEP_RECV_HOLD = 3,    // wait for the slot and keep it for a later STEP_REDUCE_N
  STEP_REDUCE_N = 4,     // dst = fold over srcs[] in the given (rank) order; releases held slots
  STEP_LOCAL_COPY = 5,   // dst = src, both local
  // fused ring primitives: pop pipe(peer -> me, lane) and push pipe(me -> peer2, lane2) in ONE kernel
  STEP_RECV_REDUCE_SEND = 6,  // v = a op slot; next slot = v; dst = v too when keep_local
  STEP_RECV_COPY_SEND = 7,    // dst = slot; next slot = slot
};

enum BufId : int { BUF_SEND = 0, BUF_RECV = 1, BUF_TEMP = 2 };
enum CollKind : int { COLL_ALLREDUCE = 0, COLL_ALLGATHER = 1, COLL_BCAST = 2, COLL_REDUCE = 3 };

constexpr int kMaxDeps = 18;  // >= kMaxRanks + 2: a write may follow one read per peer (one-shot, in place)
constexpr int kMaxSrcs = 16;

struct Step {
  int kind = 0;
  int peer = -1;
  int lane = 0;
  int src_buf = BUF_SEND;
  size_t src_off = 0;  // bytes
  int dst_buf = BUF_RECV;
  size_t dst_off = 0;  // bytes
  size_t bytes = 0;
  int ndeps = 0;
  int deps[kMaxDeps] = {0};  // earlier steps of THIS rank that must have completed
  int nsrcs = 0;
  int srcs[kMaxSrcs] = {0};  // STEP_REDUCE_N: index of a RECV_HOLD step, or -1 = the local operand
  int peer2 = -1;            // fused steps: where the result is pushed
  int lane2 = 0;
  int keep_local = 0;        // STEP_RECV_REDUCE_SEND: also store the result in dst
};

struct PlanParams {
  int coll = COLL_ALLREDUCE;
  int algo = 1;
  int size = 1;
  int rank = 0;
  int root = 0;
  size_t count = 0;      // elements (allgather: per rank)
  size_t elem_size = 4;  // bytes
  int channels = 1;      // ring channels (each a different Hamiltonian cycle of the mesh)
  int lanes = 2;         // FIFO lanes per ordered rank pair
  size_t piece_bytes = 1 << 20;  // max bytes per step (<= slot size)
  int fuse = 1;                  // ring: receive-reduce-send / receive-copy-send in one kernel
  int fifo_depth = 8;            // slots per pipe (bounds the pieces a fused ring keeps in flight)
  size_t oneshot_bytes = 1 << 20;  // direct allreduce: messages up to this size use the one-shot form
};

struct Plan {
  std::vector<Step> steps;
  size_t temp_bytes = 0;  // BUF_TEMP the executor must provide
  int algo = 0;           // algorithm actually used (after fallbacks)
  int channels = 1;
};

// Returns 0 or a negative xmpi error code (XMPI_ERR_ARG / XMPI_ERR_UNSUPPORTED).
int build_plan(const PlanParams& p, Plan* out);

// Ring channel c of an N-rank full mesh: the cyclic order of ranks.  Channel strides are the
// residues coprime to N (1, N-1, 3, N-3, ...): distinct strides use distinct xGMI links, and
// stride d / N-d use the two directions of the same links.
int ring_channel_count(int size);
void ring_order(int size, int channel, std::vector<int>* order);

std::string plan_to_text(const Plan& plan);

// Chunk j of a count-element buffer cut for `size` ranks with 16-byte aligned boundaries (the cut of
// the zero-copy collectives: rank j folds / forwards chunk j): element offset and length.
void zc_chunk(size_t count, size_t elem_size, int size, int j, size_t* elem_off, size_t* elem_cnt);

}  // namespace xmpi
