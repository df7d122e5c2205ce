/*
 * xmpi.h -- C ABI of the MI355X-native message-passing / collectives library.
 *
 * This header is the drop-in boundary: it is what a cgo shim (see INTEGRATION.md and
 * go/xgmi/xgmi.go) binds to put an xGMI/HIP backend behind btracey/mpi's Go API.  Every
 * entry point names the reference interface it replaces (file:line are relative to the
 * reference repository root).
 *
 * Conventions
 *   - plain C: opaque handle, raw pointers, sizes in ELEMENTS of `dtype` unless stated;
 *   - every function returns 0 (XMPI_OK) or a negative xmpi error code; HIP failures are
 *     mapped to XMPI_ERR_HIP and the HIP error text is kept for xmpi_last_error();
 *   - calls are blocking (reference: mpi.go:47-48) except xmpi_send_nowait and the xmpi_i*
 *     collectives, and may be made from any OS thread (cgo moves goroutines between threads; each
 *     entry point re-selects the comm's device);
 *   - buffers may be device pointers (HBM, the hot path) or host pointers (Go slices: Send / Receive carry them through the
 *     job's shared segment, collectives stand a registered HBM block in for them);
 *   - one communicator == one rank == one process == one MI355X.
 */
#ifndef XMPI_H
#define XMPI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct xmpi_comm xmpi_comm;

/* element types (the reference types payloads by Go reflection through gob,
 * network.go:539,597; a device buffer needs an explicit tag) */
typedef enum {
  XMPI_U8 = 0,
  XMPI_I32 = 1,
  XMPI_I64 = 2,
  XMPI_F16 = 3,
  XMPI_F32 = 4,
  XMPI_F64 = 5,
  XMPI_BF16 = 6,
  XMPI_DTYPE_COUNT = 7,
  /* (every int is a value of the type: what a binding passes out of range -- cgo hands over a uint32 -- comes back as
   * XMPI_ERR_ARG instead of being undefined behaviour in the C++ that reads it; the same for the two enums below) */
  XMPI_DTYPE_ANY_INT = 0x7fffffff
} xmpi_dtype;

typedef enum { XMPI_SUM = 0, XMPI_PROD = 1, XMPI_MIN = 2, XMPI_MAX = 3, XMPI_OP_COUNT = 4, XMPI_OP_ANY_INT = 0x7fffffff } xmpi_op;

/* collective schedules */
/* With one process per GPU, RING (allreduce, allgather), RHD (allreduce) and TREE (bcast) are ONE kernel per rank that
 * runs every step of the schedule itself, the steps released by flag words between the peers' kernels (sched.hip);
 * with ranks that share a process and a GPU -- and for TREE reduce, DIRECT -- they are step tables run by the host
 * through the HBM receive windows. */
typedef enum {
  XMPI_ALGO_AUTO = 0,   /* the library's choice: its tuned table (xmpi_tune) or the zero-copy fold */
  XMPI_ALGO_RING = 1,   /* multi-channel ring: reduce-scatter + allgather            */
  XMPI_ALGO_RHD = 2,    /* recursive halving (reduce-scatter) + doubling (allgather) */
  XMPI_ALGO_DIRECT = 3, /* full-mesh one-hop reduce-scatter + allgather; rank-order sum */
  XMPI_ALGO_TREE = 4,   /* binary tree (bcast / reduce)                              */
  XMPI_ALGO_ZCOPY = 5,  /* zero-copy: one kernel folds straight out of the peers' registered
                           buffers (rank order) and stores straight into them; no staging  */
  XMPI_ALGO_ZPUSH = 6,  /* zero-copy allreduce that only WRITES over xGMI: contributions are pushed
                           to the owner of their chunk -- with one process per GPU into the block its
                           communicator keeps for such things (in place and any count included; ranks that
                           meet on the host: into its receive buffer, out of place, count divisible by
                           ranks x 16 B) --, folded locally
                           in rank order, results pushed back; 2 kernels, one hop each way (the store-only
                           counterpart of ZCOPY); reduce: the same with the results stored to the root
                           only; allgather and bcast are store-only as ZCOPY runs them        */
  XMPI_ALGO_LL = 7,     /* low latency, messages up to 32 KiB per rank with one process per GPU: every rank pushes
                           its payload as {data, flag} lines into the peers' flag allocations and folds
                           locally in rank order -- one one-way hop, nothing registered, announced or read
                           remotely (the reference: one message + one ack per Send, network.go:562-571);
                           longer messages, or ranks that meet on the host: the same as ZCOPY / AUTO.
                           A BLOCKING call of up to agent_ll_bytes (8 KiB; XMPI_AGENT_LL_BYTES) that finds its
                           stream idle is not even launched: a one-block kernel that lingers behind the one
                           before (XMPI_LL_AGENT_US, default 40) takes it from a command record in pinned memory
                           (XMPI_AGENT_LL=0: always launch) -- every call of the reference's API is blocking,
                           mpi.go:47-48                                                                      */
  /* The PUSH forms of the stepped kernels (one process per GPU): the same schedules, pairs, chunks and association as
   * RING / RHD / TREE -- bit-identical results -- but every payload byte crosses its link as a posted STORE: a step reads only
   * this rank's memory (its input and what peers stored here), combines, and stores the result into the PEER's receive buffer
   * or the landing block the peer lends (in-place ring allreduce: one buffer's worth of registered memory per rank; halving:
   * about as much; tree reduce: one per child) -- the reference's one-way message (network.go:562-571) without the ack, where
   * RING / RHD / TREE load every byte over the link (a round trip per packet).  With ranks that meet on the host: the same
   * as RING / RHD / TREE.  xmpi_tune times both forms. */
  XMPI_ALGO_RING_PUSH = 8, /* allreduce, allgather */
  XMPI_ALGO_RHD_PUSH = 9,  /* allreduce            */
  XMPI_ALGO_TREE_PUSH = 10, /* bcast, reduce       */
  XMPI_ALGO_COUNT = 11,
  XMPI_ALGO_ANY_INT = 0x7fffffff
} xmpi_algo;

/* error codes */
#define XMPI_OK 0
#define XMPI_ERR_ARG (-1)        /* bad argument                                          */
#define XMPI_ERR_HIP (-2)        /* a HIP runtime call failed (see xmpi_last_error)       */
#define XMPI_ERR_BOOTSTRAP (-3)  /* shared control block could not be created / joined    */
#define XMPI_ERR_TIMEOUT (-4)    /* a peer did not arrive within XMPI_TIMEOUT_S           */
#define XMPI_ERR_TAG_EXISTS (-5) /* {peer,tag} already in use (mpi.go:172-182 TagExists)  */
#define XMPI_ERR_TRUNCATE (-6)   /* receive buffer smaller than the message               */
#define XMPI_ERR_NOMEM (-7)
#define XMPI_ERR_STATE (-8)      /* not initialised / already finalised                   */
#define XMPI_ERR_UNSUPPORTED (-9)
#define XMPI_ERR_NOGPU (-10)     /* no usable HIP device: the product has no CPU fallback */
#define XMPI_ERR_PEER (-11)      /* a peer rank reported failure                          */

/* ---- lifecycle -------------------------------------------------------------------------- */

/* Replaces mpi.Init -> (*Network).Init (mpi.go:96-98, network.go:53-65): instead of a TCP
 * mesh, ranks meet in a POSIX shared-memory control block named after `job_key`, pin
 * `device` (rank i -> GPU i is the launcher's job), allocate their HBM receive windows and
 * exchange hipIpc handles.  rank/size come from the launcher (reference: index of -mpi-addr
 * in the sorted -mpi-alladdr list, network.go:94-109).  device < 0 selects rank % ndev. */
int xmpi_init(int rank, int size, int device, const char* job_key, xmpi_comm** out);

/* The reference's Init returns an error only when the mesh cannot be built (network.go:53-65).  Here a rank that cannot map a
 * peer's memory -- its HBM window, its (uncached) flag page -- does not fail the job either: inside xmpi_init every rank publishes
 * what it could map and all of them take the best level EVERYBODY reached: ranks meeting on the device + windows (nothing
 * degraded); ranks meeting on the host (a flag page nobody could allocate, export or map: the zero-copy collectives with a host
 * rendezvous, the staged step tables for the rest); no windows (collectives device-synchronised only, Send / Receive out of
 * registered buffers and host slices, no xmpi_send_nowait).  xmpi_init fails -- on every rank, with the reason -- only when
 * neither is left.  xmpi_get_param("degraded") is the level as bits (1: the split form's data kernel at system scope, 2: ranks
 * meet on the host, 4: no windows); this is the text: which level and the first reason a rank gave, "" when nothing is
 * degraded. */
const char* xmpi_degraded(const xmpi_comm* comm);

/* Replaces mpi.Finalize -> (*Network).close (mpi.go:102-104, network.go:354-369). */
int xmpi_finalize(xmpi_comm* comm);

/* Replace mpi.Rank / mpi.Size (mpi.go:112-119, network.go:41-50): rank is -1 and size 0 for
 * a NULL (uninitialised) communicator, exactly the reference's "not initialised" answer. */
int xmpi_rank(const xmpi_comm* comm);
int xmpi_size(const xmpi_comm* comm);
int xmpi_device(const xmpi_comm* comm);

/* Host-side rendezvous of all ranks (the reference has no barrier; needed for timing). */
int xmpi_barrier(xmpi_comm* comm);

const char* xmpi_strerror(int code);
/* Text of the last failure on this thread ("" if none): what the reference carries in its `error` values and panic messages
 * (mpi.go:20-21, network.go:555,611). */
const char* xmpi_last_error(void);
/* (no counterpart in the reference) */
const char* xmpi_version(void);

/* ---- HBM buffers ------------------------------------------------------------------------ */

/* HBM of this rank's device.  Buffers from xmpi_malloc are REGISTERED: peers may map them
 * (hipIpc) so the zero-copy collectives can read and write them in place over xGMI.  They are
 * 256-byte aligned blocks of a few large arenas that stay allocated (and mapped by the peers)
 * until the process's last communicator is finalised: allocating and freeing costs no runtime
 * call and no re-mapping (XMPI_ARENA_MIN_BYTES / XMPI_ARENA_MAX_BYTES, default 64 MiB / 1 GiB;
 * a larger request gets an arena of its own).  xmpi_free takes the pointer xmpi_malloc returned.  The
 * reference has no counterpart (its payloads are Go values, network.go:539); the Go shim wraps
 * these in its DeviceBuffer type. */
void* xmpi_malloc(xmpi_comm* comm, size_t bytes);
int xmpi_free(xmpi_comm* comm, void* dptr);
/* (No counterpart in the reference: its payloads are Go values, network.go:539.)
 * Register / forget device memory that was NOT allocated by xmpi_malloc (e.g. a framework's
 * allocator): [dptr, dptr+bytes) must lie inside one hipMalloc allocation of this rank's device.
 * Deregister before that allocation is freed.  Unregistered buffers still work with every
 * collective -- through the staged (window) path.  Memory of xmpi_malloc is registered as it is
 * (register: no-op, deregister: XMPI_ERR_ARG). */
int xmpi_register(xmpi_comm* comm, void* dptr, size_t bytes);
int xmpi_deregister(xmpi_comm* comm, void* dptr);
/* Blocking copy between any two of {host, this rank's HBM}; fill (no counterpart in the reference: a binding without a HIP
 * binding of its own -- the cgo shim -- fills and reads DeviceBuffers through these). */
int xmpi_memcpy(xmpi_comm* comm, void* dst, const void* src, size_t bytes);
int xmpi_memset(xmpi_comm* comm, void* dst, int byte, size_t bytes);
/* Wait until every stream of the communicator has drained (no counterpart in the reference). */
int xmpi_sync(xmpi_comm* comm);

/* ---- point to point --------------------------------------------------------------------- */

/* Replaces mpi.Send -> (*Network).Send (mpi.go:126-128, network.go:518-572): the payload is
 * pushed peer-to-peer into the destination's HBM window over xGMI instead of being gob-encoded
 * onto a net.Conn.  Rendezvous semantics are kept: returns only after the matching receive
 * consumed the message (network.go:569).  {dest,tag} must be unique among concurrent sends
 * (mpi.go:121-125) -> XMPI_ERR_TAG_EXISTS otherwise.  dest == own rank is allowed
 * (network.go:545-548) when a concurrent xmpi_recv is posted from another thread.
 * A non-empty message from a registered buffer (xmpi_malloc / xmpi_register; p2p_direct_bytes = 1 is
 * the smallest such message, < 0 turns this off) is not pushed at all: the matching receive copies it
 * straight out of the sender's HBM (one pass, one xGMI crossing).  Unregistered
 * device memory travels through the mail slots of the receiver's window (slot-in by the sender,
 * slot-out by the receiver); a payload in HOST memory (network.go:518 takes Go values) through the entry's host lane in the
 * shared segment (XMPI_HOST_LANES=0: staged through HBM like unregistered device memory) -- 1 us per direction for a small
 * slice, and into a device destination by DMA out of the lane.  That copy is one kernel whose last block writes both acknowledgements itself, and which stays for
 * p2p_agent_us (XMPI_P2P_AGENT_US, default 40; 0 = one launch per message) after a message to take the next one from a command
 * record in pinned host memory instead of being launched again: between two processes a small message takes 4.5-5 us per
 * direction.  Nothing on the GPU ever waits for a peer on this path. */
int xmpi_send(xmpi_comm* comm, const void* buf, size_t count, xmpi_dtype dtype, int dest, int tag);

/* The split of Send the reference's author sketched and left commented out (mpi.go:132-152):
 * xmpi_send_nowait returns once the payload has left the caller's buffer (it may be modified at
 * once) without waiting for the receiver -- the payload sits in the mail slots of the receiver's
 * window; a message longer than those slots hold (p2p_depth x p2p_slot_bytes, 8 MiB) still needs
 * the receive to drain them.  xmpi_wait blocks until the destination confirmed reception and
 * frees the {dest, tag} pair for re-use, exactly as the sketched Wait(destination, tag). */
int xmpi_send_nowait(xmpi_comm* comm, const void* buf, size_t count, xmpi_dtype dtype, int dest,
                     int tag);
int xmpi_wait(xmpi_comm* comm, int dest, int tag);

/* Replaces mpi.Receive -> (*Network).Receive + receiveReader (mpi.go:157-159,
 * network.go:575-625).  `capacity` is the room in `buf` (elements); `*got` (optional) is the
 * element count of the message.  The dtype must match the sender's. */
int xmpi_recv(xmpi_comm* comm, void* buf, size_t capacity, xmpi_dtype dtype, int src, int tag,
              size_t* got);

/* Blocks until a message {src, tag} has been posted and reports its element count and dtype without
 * consuming it.  Lets a binding size the destination the way the reference's in-place gob decode
 * re-slices / re-allocates the caller's slice (network.go:594-601, bounce.go:89,94). */
int xmpi_probe(xmpi_comm* comm, int src, int tag, size_t* count, xmpi_dtype* dtype);

/* ---- collectives (absent from the reference: mpi.go:130 is a commented-out stub, mpi.go:69-71
 *      an unused probe variable; defined here in the reference's delegate style) --------------
 * Called by every rank, in the same order, with the same count / dtype / op / root / algo.  Ranks that are not -- where they
 * meet on the device (above the LL lines' limit) or through the control block -- all get XMPI_ERR_ARG ("not in the same call")
 * at once and nothing is moved: every kernel announces a signature of its call with its buffers.
 * Blocking: the buffers must be complete when the call is made and may be reused when it returns.
 * ZCOPY and AUTO: the peers' buffers are read and written in place by ONE kernel per rank; floating-point
 * folds are in rank order 0..N-1.  With one process per GPU the ranks meet inside that kernel (flag words
 * in HBM; see the stream-ordered forms below -- the blocking call enqueues on the communicator's stream
 * and waits), and a buffer the peers cannot map (host memory, unregistered device memory) is stood in for
 * by a registered block (one local copy each way).  Ranks sharing a process and a GPU meet through the
 * control block instead, and there unregistered buffers send every rank to the staged schedules
 * (RING / RHD / DIRECT / TREE through the HBM receive windows), which any rank can also ask for by name. */

/* (absent from the reference, mpi.go:130)  root's buffer replicated to every rank, bit-exact.  algo: TREE | TREE_PUSH (binary
 * tree, pull / push form) | ZCOPY | AUTO. */
int xmpi_bcast(xmpi_comm* comm, void* buf, size_t count, xmpi_dtype dtype, int root, int algo);

/* (absent from the reference, mpi.go:130)  recvbuf (significant at root only) = op over ranks of sendbuf.
 * algo: TREE | TREE_PUSH | DIRECT | ZCOPY | AUTO. */
int xmpi_reduce(xmpi_comm* comm, const void* sendbuf, void* recvbuf, size_t count,
                xmpi_dtype dtype, xmpi_op op, int root, int algo);

/* (absent from the reference: the stub at mpi.go:130)  recvbuf = op over ranks of sendbuf, on every rank (sendbuf == recvbuf
 * allowed).  algo: RING | RHD | RING_PUSH | RHD_PUSH | DIRECT | ZCOPY | ZPUSH | LL | AUTO.  DIRECT and ZCOPY sum in rank order 0..N-1 (bit-identical
 * to the reference-user composition "gather everything, add on the host in rank order"). */
int xmpi_allreduce(xmpi_comm* comm, const void* sendbuf, void* recvbuf, size_t count,
                   xmpi_dtype dtype, xmpi_op op, int algo);

/* Non-blocking forms: the call returns at once with a request; one worker per communicator runs the
 * operations in the order they were issued (every rank must issue them in the same order, as with
 * the blocking calls; a blocking collective or xmpi_barrier issued later runs after them).  The
 * buffers belong to the operation until xmpi_request_wait returned.  This is the overlap of
 * communication with the caller's own work that the reference's sketched Send / Wait pair was after
 * (mpi.go:132-152), for the collectives. */
typedef struct xmpi_request xmpi_request;
int xmpi_iallreduce(xmpi_comm* comm, const void* sendbuf, void* recvbuf, size_t count,
                    xmpi_dtype dtype, xmpi_op op, int algo, xmpi_request** req);
int xmpi_iallgather(xmpi_comm* comm, const void* sendbuf, void* recvbuf, size_t count,
                    xmpi_dtype dtype, int algo, xmpi_request** req);
int xmpi_ibcast(xmpi_comm* comm, void* buf, size_t count, xmpi_dtype dtype, int root, int algo,
                xmpi_request** req);
int xmpi_ireduce(xmpi_comm* comm, const void* sendbuf, void* recvbuf, size_t count,
                 xmpi_dtype dtype, xmpi_op op, int root, int algo, xmpi_request** req);
/* The Wait of the sketched pair (mpi.go:132-152), for a collective.
 * *done = 1 once the operation has completed (xmpi_request_wait will not block). */
int xmpi_request_test(xmpi_request* req, int* done);
/* Blocks until the operation completed, returns ITS status and frees the request (mpi.go:146-152: "Wait blocks until ..."). */
int xmpi_request_wait(xmpi_request* req);

/* Stream-ordered forms.  The collective is ENQUEUED on `stream` (a hipStream_t passed as void*; NULL = the
 * communicator's own stream) and the call returns without waiting for any peer: like a kernel launch, it runs
 * after the work already on that stream, and work enqueued after it sees its result; the buffers belong to
 * the operation until the stream has passed it.  One kernel per rank is the whole collective -- it exchanges
 * "my buffers are ready / here they are" and "I am done" with the peers through flag words in HBM (written
 * over xGMI), no host thread polls anything (mpi_amd/csrc/dsync.cpp).  This is the
 * overlap the reference's author sketched for Send / Wait (mpi.go:132-152), on HIP streams.
 * Every rank must enqueue the same collectives in the same order (per communicator, whatever the streams).
 * Buffers must be device memory; memory that is not registered (xmpi_malloc / xmpi_register) is copied
 * through a registered block on the same stream.  The layout this is for is one process per GPU; ranks that
 * share a process AND a GPU (bench.py on a 1-GPU box) meet on the host instead (the call then waits).
 * A failure inside the kernel (a peer that never arrives within XMPI_TIMEOUT_S, an aborted job) is reported by
 * the next xmpi_stream_sync or blocking collective. */
int xmpi_allreduce_on_stream(xmpi_comm* comm, const void* sendbuf, void* recvbuf, size_t count,
                             xmpi_dtype dtype, xmpi_op op, void* stream);
int xmpi_allgather_on_stream(xmpi_comm* comm, const void* sendbuf, void* recvbuf, size_t count,
                             xmpi_dtype dtype, void* stream);
int xmpi_bcast_on_stream(xmpi_comm* comm, void* buf, size_t count, xmpi_dtype dtype, int root, void* stream);
int xmpi_reduce_on_stream(xmpi_comm* comm, const void* sendbuf, void* recvbuf, size_t count,
                          xmpi_dtype dtype, xmpi_op op, int root, void* stream);
/* Stream-ordered Send / Receive: the reference's message + ack (network.go:562-571, 616-624) as ONE kernel on each side.
 * The sender's kernel puts {message number, tag, dtype, bytes, where the payload lives} into a 64-byte box of the
 * receiver's flag allocation (a store over xGMI) and ends when the receiver has answered; the receiver's kernel waits for
 * a box carrying its tag, pulls the payload straight out of the sender's HBM into `buf` and answers with its verdict (a
 * message longer than `capacity` or of another dtype is consumed and both sides get the error).  Like the collectives
 * above they are ordered with the work on `stream` and no host thread waits; failures are reported by the next
 * xmpi_stream_sync.  Device memory only; an unregistered send buffer is copied through a registered block.  At most 8
 * messages of an ordered pair may be unanswered at once.  A kernel that waits holds its stream (and, while it waits, the
 * hardware queue behind it): enqueue matching sends and receives so that no stream waits for work queued behind it --
 * the blocking xmpi_send / xmpi_recv have no such rule and do not use these kernels.  Needs ranks that meet on the
 * device (one process per GPU); XMPI_ERR_UNSUPPORTED otherwise. */
int xmpi_send_on_stream(xmpi_comm* comm, const void* buf, size_t count, xmpi_dtype dtype, int dest, int tag, void* stream);
int xmpi_recv_on_stream(xmpi_comm* comm, void* buf, size_t capacity, xmpi_dtype dtype, int src, int tag, void* stream);

/* (No counterpart in the reference.)  Streams for callers without a HIP binding of their own (the cgo shim, ctypes): a non-blocking hipStream_t of
 * the communicator's device; xmpi_stream_sync waits for everything enqueued on it (NULL = the communicator's
 * own stream) and returns the status of the collectives that ran on it. */
void* xmpi_stream_create(xmpi_comm* comm);
int xmpi_stream_destroy(xmpi_comm* comm, void* stream);
int xmpi_stream_sync(xmpi_comm* comm, void* stream);

/* (No counterpart in the reference.)  hipGraph capture: the stream-ordered collectives enqueued on `stream` between xmpi_graph_begin and xmpi_graph_end
 * (registered device buffers; call each once before capturing so that everything is mapped) become an executable
 * graph that xmpi_graph_launch replays -- one launch for the whole sequence, the caller's own kernels captured on
 * that stream included.  Every rank captures the same sequence and replays it equally often.  A graph cannot hold a
 * block the library lends per call: where the schedule table names a push form, its pull form is captured (the same bits);
 * where it names push-only or the tree reduce, the fold.  Needs ranks that meet on the device (one process per GPU);
 * XMPI_ERR_UNSUPPORTED otherwise. */
int xmpi_graph_begin(xmpi_comm* comm, void* stream);
int xmpi_graph_end(xmpi_comm* comm, void* stream, void** graph);
int xmpi_graph_launch(xmpi_comm* comm, void* graph, void* stream);
int xmpi_graph_destroy(xmpi_comm* comm, void* graph);

/* (No counterpart in the reference; bounce.go:83-151 loops in Go.)  The same allreduce `iters` times back to back: the step loop of a benchmark without per-call
 * host-language overhead (bench.py hosts several ranks as Python threads, which would otherwise
 * queue for the interpreter lock between steps; a Go or C++ caller has no such cost).  With one process
 * per GPU the steps are enqueued on the communicator's stream and waited for once at the end. */
int xmpi_allreduce_repeat(xmpi_comm* comm, const void* sendbuf, void* recvbuf, size_t count,
                          xmpi_dtype dtype, xmpi_op op, int algo, int iters);

/* (absent from the reference, mpi.go:130; its nearest idiom is the all-to-all of helloworld.go:53-81)
 * recvbuf[r*count : (r+1)*count] = rank r's sendbuf, bit-exact.  algo: RING | RING_PUSH | DIRECT | ZCOPY | AUTO. */
int xmpi_allgather(xmpi_comm* comm, const void* sendbuf, void* recvbuf, size_t count,
                   xmpi_dtype dtype, int algo);

/* ---- local kernels (the HBM-bound pieces, exposed for parity tests and rooflines) --------- */

/* (No counterpart in the reference: a reference user adds on the host what Receive delivered, the helloworld.go:53-81 idiom.)
 * dst[i] = a[i] op b[i]  (the per-chunk reduction every ring / halving step runs). */
int xmpi_reduce_local(xmpi_comm* comm, void* dst, const void* a, const void* b, size_t count,
                      xmpi_dtype dtype, xmpi_op op);
/* dst[i] = ((src[0][i] op src[1][i]) op src[2][i]) ... strictly left to right, nsrc <= 16 -- the order in which a reference
 * user who received every rank's buffer (helloworld.go:53-81) would add them. */
int xmpi_reduce_local_n(xmpi_comm* comm, void* dst, const void* const* srcs, int nsrc,
                        size_t count, xmpi_dtype dtype, xmpi_op op);
/* dst = src through the library's streaming copy kernel (no counterpart in the reference). */
int xmpi_copy_local(xmpi_comm* comm, void* dst, const void* src, size_t bytes);
/* (No counterpart in the reference.)  The kernels of the zero-copy collectives, on local buffers: every dsts[k][i] = left-to-right fold
 * of srcs[.][i] (one pass: nsrc reads + ndst writes per element), and dsts[k] = src. */
int xmpi_reduce_local_multi(xmpi_comm* comm, void* const* dsts, int ndst, const void* const* srcs,
                            int nsrc, size_t count, xmpi_dtype dtype, xmpi_op op);
int xmpi_copy_local_multi(xmpi_comm* comm, void* const* dsts, int ndst, const void* src, size_t bytes);
/* (No counterpart in the reference.)  dsts[k] = srcs[k], k < n <= 16, in ONE launch with the access pattern of xmpi_reduce_local_multi on
 * the same pointers (same grid and cache policy, n loads + n stores per 16-byte packet): the fold without its arithmetic -- what this
 * machine's memory gives that pattern on those buffers (bench.py roofline.box_copy_us).  16-byte aligned pointers. */
int xmpi_copy_local_pairs(xmpi_comm* comm, void* const* dsts, const void* const* srcs, int n, size_t bytes);

/* Verification kernels (LDS + wavefront-shuffle reductions; replace bytes.Equal /
 * floats.Equal of examples/bounce/bounce.go:105,133 for HBM-resident data). */
int xmpi_count_mismatch(xmpi_comm* comm, const void* a, const void* b, size_t bytes,
                        uint64_t* mismatching_bytes);
/* sum of the buffer read as little-endian u32 words (mod 2^64) + trailing bytes (bounce.go:105's bytes.Equal, as a number two
 * ranks can compare). */
int xmpi_checksum(xmpi_comm* comm, const void* buf, size_t bytes, uint64_t* sum);
/* (bounce.go:133's floats.Equal with the differences spelled out)
 * stats[0] = max_i |a_i - b_i|, stats[1] = sum_i |b_i|, stats[2] = count of i with a NaN
 * mismatch; a, b of dtype F16/BF16/F32/F64; accumulated in double. */
int xmpi_diff_stats(xmpi_comm* comm, const void* a, const void* b, size_t count,
                    xmpi_dtype dtype, double stats[3]);
/* (bounce.go:133 again, relative)  *max_rel = max_i |a_i - b_i| / |b_i| (0/0 = 0, x/0 = inf; inf if a NaN sits on one side only), in double.  With
 * non-negative inputs b_i (a rank-order sum) equals sum_r |x_r,i|, so this evaluates the per-element tolerance
 * rule |delta_i| <= tol * sum_r |x_r,i| of BASELINE.md for a whole buffer on the device. */
int xmpi_diff_rel(xmpi_comm* comm, const void* a, const void* b, size_t count, xmpi_dtype dtype,
                  double* max_rel);
/* ---- tuning / introspection ------------------------------------------------------------- */

/* The reference's only knobs are its flags (flags.go:44-50: addresses, -mpi-inittimeout, protocol, password).  Every name, its
 * default and who is meant to set it: INTEGRATION.md section 5 (checked against the code by the CPU suite).  xmpi_set_param must
 * be called identically on every rank; xmpi_get_param also reads the diagnostics listed there. */
int xmpi_set_param(xmpi_comm* comm, const char* name, long value);
long xmpi_get_param(const xmpi_comm* comm, const char* name);

/* (No counterpart in the reference: it has one transport and one schedule.)  The library's own schedule table.  xmpi_tune times, on this job's real layout and links, the schedules it offers
 * for an allreduce (one zero-copy kernel with 1 or 2 packets in flight, the meet / body / done form, the push-only
 * form, the ring and the halving kernel each in its pull and its push form, LL lines), an allgather (fold, ring kernel in both forms,
 * LL lines), a bcast (fold, tree kernel in both forms, LL lines) and a reduce (the fold's forms, push-only, tree kernel in both
 * forms, LL lines), for message sizes 1 KiB ... max_bytes (x4 steps), lets
 * every rank see the slowest rank's figures and keeps the winner per size class: XMPI_ALGO_AUTO (and the
 * stream-ordered forms) consult that table from then on, so a Go or C caller gets the schedule a benchmark would pick.
 * Collective (same max_bytes on every rank); a few hundred milliseconds.  The table is readable through
 * xmpi_get_param("tune_algo_<collective>_<class>") (collective 0 = allreduce, 1 = allgather, 2 = bcast, 3 = reduce; class k = messages of
 * [2^(k+8), 2^(k+9)) bytes per rank; -1 = built-in rule), "tune_split_..." (1 = meet / body / done), "tune_unroll_...".
 * Per row of (max-over-ranks) mean times the fastest candidate wins, except that the default stays unless beaten by more
 * than 3 % (noise must not flip the schedule). */
int xmpi_tune(xmpi_comm* comm, size_t max_bytes);

/* (No counterpart in the reference; bounce.go:140-151 prints wall-clock means.)
 * Kernel profiling: when enabled every reduction / copy kernel launch is bracketed by HIP
 * events on the stream it runs on.  kind: 0 = reduce2, 1 = reduceN, 2 = copy kernel,
 * 3 = peer copy (hipMemcpyAsync or kernel push), 4 = zero-copy fold / multi-destination copy. */
int xmpi_prof_enable(xmpi_comm* comm, int on);
int xmpi_prof_reset(xmpi_comm* comm);
int xmpi_prof_get(xmpi_comm* comm, int kind, uint64_t* launches, double* total_ms,
                  uint64_t* bytes);

/* Link diagnostic -- what examples/bounce is for in the reference (bounce.go:83-151), without a second program: times `iters`
 * back-to-back copies of `bytes` between this rank's window and the
 * peer's (direction 0 = write to the peer, 1 = read from the peer; engine 0 = hipMemcpyAsync, 1 = the library's copy kernel,
 * 2 = a kernel that moves data the way a step of the ring / halving / tree kernels does: system-scope loads, written-through stores).  The
 * peer must not be inside a collective; call it on both ranks of a pair for the bidirectional rate. */
int xmpi_link_probe(xmpi_comm* comm, int peer, size_t bytes, int engine, int iters, int direction,
                    double* gbps);

/* bytes per element of a dtype, 0 for an unknown one (the reference's payloads carry their type through gob, network.go:539) */
size_t xmpi_dtype_size(xmpi_dtype dtype);

#ifdef __cplusplus
}
#endif
#endif /* XMPI_H */
