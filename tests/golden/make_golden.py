"""Generates tests/golden/collectives_kat.json.

The reference holds NO golden vectors for this path (no tests at all) and its
collectives do not exist (mpi.go:130), so these known answers are produced by an independent
pure-numpy restatement of the project's oracle definition ("exchange whole buffers losslessly,
fold on the host in rank order, one rounding per operation") -- NOT by oracle/xmpi_oracle.c and
not by the GPU path, both of which are checked against this file.

    python tests/golden/make_golden.py        # rewrites collectives_kat.json
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
U = np.uint64


def hash64(seed, idx):
    with np.errstate(over="ignore"):
        z = U(seed) * U(0xD1342543DE82EF95) + idx.astype(np.uint64) * U(0x9E3779B97F4A7C15) + U(0x2545F4914F6CDD1D)
        z = (z ^ (z >> U(30))) * U(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> U(27))) * U(0x94D049BB133111EB)
        return z ^ (z >> U(31))


def uniform(dtype, seed, n):
    h = hash64(seed, np.arange(n, dtype=np.uint64))
    if dtype == "f32":
        return ((h >> U(40)).astype(np.float64) * 2.0 ** -24).astype(np.float32)
    if dtype == "f64":
        return (h >> U(11)).astype(np.float64) * 2.0 ** -53
    if dtype == "f16":
        return ((h & U(63)).astype(np.float64) / 64.0).astype(np.float16)
    if dtype == "i64":
        return h.view(np.int64)
    if dtype == "i32":
        return (h >> U(32)).astype(np.uint32).view(np.int32)
    raise ValueError(dtype)


def fold(ins, np_dtype):
    acc = ins[0].astype(np_dtype).copy()
    with np.errstate(over="ignore"):
        for x in ins[1:]:
            acc = (acc + x.astype(np_dtype)).astype(np_dtype)  # one rounding per rank, in rank order
    return acc


def hexbytes(a):
    return np.ascontiguousarray(a).tobytes().hex()


def main():
    cases = []
    for dtype, npdt in (("f32", np.float32), ("f64", np.float64), ("f16", np.float16), ("i64", np.int64),
                        ("i32", np.int32)):
        for n_ranks in (2, 4, 8):
            count = 24
            ins = [uniform(dtype, 1000 + r, count) for r in range(n_ranks)]
            cases.append({"kind": "allreduce_sum", "dtype": dtype, "ranks": n_ranks, "count": count, "seed0": 1000,
                          "pattern": "uniform", "inputs_rank0_hex": hexbytes(ins[0]),
                          "result_hex": hexbytes(fold(ins, npdt))})
    # x_r[i] = r + 1  =>  N(N+1)/2 everywhere
    for n_ranks in (2, 4, 8):
        cases.append({"kind": "allreduce_sum", "dtype": "f32", "ranks": n_ranks, "count": 8, "seed0": 0,
                      "pattern": "const", "result_hex": hexbytes(np.full(8, n_ranks * (n_ranks + 1) / 2, np.float32))})
    # BASELINE cfg 3 layout: x[i] = (r << 40) | i, gathered in rank order
    n_ranks, count = 4, 16
    blocks = [(np.int64(r) << np.int64(40)) | np.arange(count, dtype=np.int64) for r in range(n_ranks)]
    cases.append({"kind": "allgather", "dtype": "i64", "ranks": n_ranks, "count": count, "seed0": 0,
                  "pattern": "index", "result_hex": hexbytes(np.concatenate(blocks))})
    # catastrophic-cancellation vector: shows that the fold order is observable in f32
    big = [np.float32(1e8), np.float32(1.0), np.float32(-1e8), np.float32(1.0)]
    acc = np.float32(0)
    seq = big[0]
    for x in big[1:]:
        seq = np.float32(seq + x)
    cases.append({"kind": "fold_order", "dtype": "f32", "values": [float(x) for x in big], "rank_order_result": float(seq)})
    with open(os.path.join(HERE, "collectives_kat.json"), "w") as f:
        json.dump({"note": "independent numpy restatement; see make_golden.py", "cases": cases}, f, indent=1)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
