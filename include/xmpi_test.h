/*
 * xmpi_test.h -- entry points of libxmpi.so that exist for the test suites, the verification programs under examples/ and
 * bench.py's parity leg: deterministic inputs shared with the CPU oracle, host-only self-tests and introspection of the
 * library's host logic (step tables, step programs, chunk cuts, the tuner's decision rule).  They are NOT part of the drop-in
 * boundary: the cgo shim (go/xgmi) binds include/xmpi.h only, and nothing here replaces a reference interface.
 */
#ifndef XMPI_TEST_H
#define XMPI_TEST_H

#include "xmpi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Fill with the deterministic test pattern shared with the CPU oracle (oracle/xmpi_oracle.c
 * `oracle_fill`): see DESIGN.md "synthetic inputs". */
int xmpi_fill_pattern(xmpi_comm* comm, void* buf, size_t count, xmpi_dtype dtype, int pattern,
                      uint64_t seed);

/* The decision xmpi_tune applies to one row of (max-over-ranks) mean times in microseconds, <= 0 = not run: the fastest,
 * except that candidate 0 (the default) stays unless beaten by more than `margin` (host logic only). */
int xmpi_tune_decide(const double* mean_us, int n, double margin);

/* Host-only self-test of the control plane shared by the ranks of a job (no GPU call): every rank
 * of `size` calls it with the same key; exercises join, barriers, pipe counters, the mail-entry
 * states (with the direct-pull offer), the zero-copy buffer descriptors and the retire logs for
 * `rounds` rounds.  Used by the CPU test-suite with plain OS processes. */
int xmpi_ctl_selftest(const char* job_key, int rank, int size, int rounds);

/* Schedule introspection (host logic only, no GPU needed): writes the step table the executor
 * would run for (coll, algo, size, rank, count) as text into out; returns needed length.  fifo_depth 0 / oneshot_bytes
 * (size_t)-1: the library's defaults (8 slots per pipe; one-shot schedules up to 1 MiB). */
int xmpi_plan_dump(int coll, int algo, int size, int rank, int root, size_t count,
                   size_t elem_size, int channels, size_t piece_elems, int fifo_depth, size_t oneshot_bytes, char* out, size_t cap);

/* The step program of a stepped kernel (ring allreduce = 1, recursive halving + doubling = 2, ring allgather = 3,
 * binary-tree bcast = 4, binary-tree reduce = 5; form 0 = pull, 1 = push; in_place: the ranks' send buffers are their receive
 * buffers) for one rank and ring channel, as text, from the very function the kernel runs (mpi_amd/csrc/sched_steps.h; host
 * logic only, no GPU needed: the CPU test-suite executes all ranks' programs under random interleavings).  Returns the needed
 * length. */
int xmpi_sched_dump(int sched, int form, int in_place, int size, int rank, int root, int pieces, size_t count, size_t elem_size,
                    int nchan, int channel, char* out, size_t cap);
/* Bytes of the landing block `rank` lends to the push form of that schedule (0: none). */
size_t xmpi_sched_land_bytes(int sched, int in_place, int size, int rank, int root, size_t count, size_t elem_size);

/* Chunk j of a count-element buffer cut for `size` ranks the way the zero-copy collectives cut it
 * (16-byte aligned boundaries): element offset and length.  Host logic only. */
int xmpi_zc_chunk(size_t count, size_t elem_size, int size, int j, size_t* elem_off, size_t* elem_cnt);

/* Host-only self-test of xmpi_malloc's block bookkeeping (no GPU call): `rounds` random allocate /
 * free operations on a synthetic arena; 0 = blocks never overlapped, stayed aligned and coalesced
 * back into one free block, otherwise the number of the failed check. */
int xmpi_heap_selftest(uint64_t seed, int rounds);

#ifdef __cplusplus
}
#endif
#endif /* XMPI_TEST_H */
