/*
 * xmpi_oracle.h -- CPU oracle for the hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the
 * product (mpi_amd/, include/xmpi.h) never links, imports or calls anything under oracle/.
 *
 * What it restates
 *   - point-to-point: the reference's Send/Receive is a lossless hand-off (gob is a lossless
 *     codec; bounce.go:105,133 and helloworld.go:78 are the only checks the reference holds)
 *     => oracle = "receiver's bytes == sender's bytes".
 *   - collectives: ABSENT from the reference (mpi.go:130 is a commented-out stub, mpi.go:69-71
 *     an unused variable).  PARITY UNPINNED: there is no reference code, test, fixture or
 *     golden vector for Bcast/Reduce/Allreduce/Allgather.  The oracle is therefore DEFINED as
 *     what a reference user obtains by exchanging whole buffers with the lossless
 *     Send/Receive (the all-to-all idiom of examples/helloworld/helloworld.go:53-81) and
 *     combining on the host in rank order 0..N-1, in the element type, IEEE round-to-nearest-
 *     even, one rounding per operation (Go rounds every float32 op to float32; no FMA).
 *   - the gob wire codec and the loopback-TCP Send/Receive path live in oracle/gob_codec.h and
 *     oracle/refpath.cpp (pinned against the known-answer vectors of gob's format document).
 */
#ifndef XMPI_ORACLE_H
#define XMPI_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* dtype / op numbering is the C ABI's (include/xmpi.h) */
enum { OR_U8 = 0, OR_I32 = 1, OR_I64 = 2, OR_F16 = 3, OR_F32 = 4, OR_F64 = 5, OR_BF16 = 6 };
enum { OR_SUM = 0, OR_PROD = 1, OR_MIN = 2, OR_MAX = 3 };
enum { OR_PAT_UNIFORM = 0, OR_PAT_INDEX = 1, OR_PAT_CONST = 2, OR_PAT_SIGNED = 3 };

size_t oracle_dtype_size(int dtype);

/* counter-based generator shared with the device-side xmpi_fill_pattern */
uint64_t oracle_hash(uint64_t seed, uint64_t i);
int oracle_fill(void* buf, size_t count, int dtype, int pattern, uint64_t seed);
/* elements [start, start + count) of the same sequence, without generating the prefix */
int oracle_fill_range(void* buf, size_t start, size_t count, int dtype, int pattern, uint64_t seed);
/* elements [start, start + n) of an allreduce result against the rank-order fold of oracle_fill(pattern, seed0 + r),
 * r = 0..nranks-1, regenerated block by block: the WHOLE buffer of a full-size configuration can be checked against the
 * oracle.  Returns the number of elements whose bits differ (~0 on bad arguments); *first_bad = the first such index. */
uint64_t oracle_check_allreduce(const void* got, size_t start, size_t n, int dtype, int pattern, uint64_t seed0,
                                int nranks, int op, uint64_t* first_bad);

/* dst[i] = a[i] op b[i] */
int oracle_reduce2(void* dst, const void* a, const void* b, size_t count, int dtype, int op);

/* out[i] = ((in[0][i] op in[1][i]) op in[2][i]) ... strict rank order, one rounding per op */
int oracle_reduce_ranks(void* out, const void* const* in, int nranks, size_t count, int dtype,
                        int op);
/* out[r*count + i] = in[r][i] */
int oracle_allgather(void* out, const void* const* in, int nranks, size_t count, int dtype);

/* half / bfloat16 helpers (bit patterns) */
float oracle_half_to_float(uint16_t h);
uint16_t oracle_double_to_half(double d);
float oracle_bf16_to_float(uint16_t h);
uint16_t oracle_float_to_bf16(float f);

/* comparison helpers used by the tests */
uint64_t oracle_count_mismatch(const void* a, const void* b, size_t bytes);
uint64_t oracle_checksum(const void* buf, size_t bytes);
/* stats[0]=max|a-b|, stats[1]=sum|b|, stats[2]=#NaN mismatches */
int oracle_diff_stats(const void* a, const void* b, size_t count, int dtype, double stats[3]);

#ifdef __cplusplus
}
#endif
#endif
