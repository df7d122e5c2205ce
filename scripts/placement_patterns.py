#!/usr/bin/env python3
"""Which offsets should 16 separately allocated 256 MiB buffers have so that the 8 + 8 fold streams fastest?  Every buffer is its
own (uncoloured) arena block + slack; buffer k is used at base_k + off_k for a family of patterns, in the two role orders a
program produces (all sends then all receives; send, receive per rank).   python scripts/placement_patterns.py"""
import json
import os
import random
import sys
import time

os.environ["XMPI_HEAP_COLOUR"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpi_amd import xmpi  # noqa: E402

N, BYTES, SLACK = 8, 256 << 20, 40 << 20
COUNT = BYTES // 4
K, M = 1024, 1 << 20


def timed(comm, dsts, srcs, reps=10):
    for _ in range(2):
        comm.reduce_local_multi(dsts, srcs, COUNT, xmpi.F32, xmpi.SUM)
    comm.sync()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        comm.reduce_local_multi(dsts, srcs, COUNT, xmpi.F32, xmpi.SUM)
        comm.sync()
        best = min(best, time.perf_counter() - t0)
    return best * 1e6


def tri(k):
    return k * (k + 1) // 2


def bitrev4(k):
    return int(f"{k:04b}"[::-1], 2)


def main():
    comm = xmpi.Comm(0, 1, 0, f"patterns-{os.getpid()}")
    bufs = [comm.alloc(BYTES + SLACK) for _ in range(16)]
    for b in bufs:
        comm.memset(b, 0, BYTES + SLACK)
    rng = random.Random(7)
    pats = {
        "none": lambda k: 0,
        "k*4K": lambda k: k * 4 * K,
        "k*12K": lambda k: k * 12 * K,
        "k*36K": lambda k: k * 36 * K,
        "k*68K": lambda k: k * 68 * K,
        "k*132K": lambda k: k * 132 * K,
        "k*516K": lambda k: k * 516 * K,
        "k*(1M+4K)": lambda k: k * (M + 4 * K),
        "k*(2M+4K)": lambda k: k * (2 * M + 4 * K),
        "tri*4K": lambda k: tri(k) * 4 * K,
        "tri*(2M+4K)": lambda k: tri(k) % 16 * 2 * M + tri(k) * 4 * K,
        "bitrev*4K": lambda k: bitrev4(k) * 4 * K,
        "bitrev*(2M+4K)": lambda k: bitrev4(k) * (2 * M + 4 * K),
        "k*4K+(k%4)*2M": lambda k: k * 4 * K + (k % 4) * 2 * M,
        "k*4K+(k//4)*2M": lambda k: k * 4 * K + (k // 4) * 2 * M,
        "k*256": lambda k: k * 256,
        "k*512": lambda k: k * 512,
        "k*1K": lambda k: k * K,
        "k*(4K+256)": lambda k: k * (4 * K + 256),
        "k*(4K+512)": lambda k: k * (4 * K + 512),
        "k*(2M+4K+512)": lambda k: k * (2 * M + 4 * K + 512),
    }
    for i in range(6):
        offs = [rng.randrange(0, SLACK // 4096) * 4096 for _ in range(16)]
        pats[f"random4K#{i}"] = (lambda o: (lambda k: o[k]))(offs)
    for i in range(4):
        offs = [rng.randrange(0, SLACK // 256) * 256 for _ in range(16)]
        pats[f"random256#{i}"] = (lambda o: (lambda k: o[k]))(offs)
    rows = []
    for name, f in pats.items():
        ptrs = [bufs[k].at(f(k)) for k in range(16)]
        assert all(f(k) + BYTES <= BYTES + SLACK for k in range(16)), name
        a = timed(comm, ptrs[N:], ptrs[:N])  # all sends, then all receives
        b = timed(comm, ptrs[1::2], ptrs[0::2])  # send, receive per rank
        c = timed(comm, ptrs[:N], ptrs[N:])  # roles swapped
        rows.append((name, a, b, c))
        print(json.dumps({"pattern": name, "sends_then_recvs_us": round(a, 1), "per_rank_pairs_us": round(b, 1), "swapped_us": round(c, 1)}), flush=True)
    rows.sort(key=lambda r: max(r[1:]))
    print("best by worst case:", [(r[0], round(max(r[1:]), 1)) for r in rows[:6]])
    comm.finalize()


if __name__ == "__main__":
    main()
