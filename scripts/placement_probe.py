#!/usr/bin/env python3
"""Does the N = 1 kernel's speed depend on WHERE its 16 buffers lie?  (The round's N = 1 lines range from 670 to 810 us for
the same launch, per process rather than per box.)  One process, one rank: reduce_n_multi over 8 sources and 8 destinations
of 256 MiB each, carved out of one allocation at base + k * (256 MiB + pad) for several pads; then 16 separate allocations.
    python scripts/placement_probe.py            (on the GPU box; prints one JSON line per layout)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpi_amd import xmpi  # noqa: E402

N = 8
BYTES = 256 << 20
COUNT = BYTES // 4


def timed(comm, dsts, srcs, reps=12):
    for _ in range(3):
        comm.reduce_local_multi(dsts, srcs, COUNT, xmpi.F32, xmpi.SUM)
    comm.sync()
    best, total = 1e9, 0.0
    for _ in range(reps):
        t0 = time.perf_counter()
        comm.reduce_local_multi(dsts, srcs, COUNT, xmpi.F32, xmpi.SUM)
        comm.sync()
        dt = time.perf_counter() - t0
        best, total = min(best, dt), total + dt
    return best * 1e6, total / reps * 1e6


def main():
    comm = xmpi.Comm(0, 1, 0, f"placement-{os.getpid()}")
    pads = [0, 256, 4096, 4096 + 256, 65536, 65536 + 4096, 1 << 20, (1 << 20) + 4096, (2 << 20) + 65536, 3 << 20]
    if len(sys.argv) > 1:
        pads = [int(eval(x, {"K": 1024, "M": 1 << 20})) for x in sys.argv[1].split(",")]
    slack = 16 * max(pads)
    big = comm.alloc(16 * BYTES + slack + 4096)
    comm.memset(big, 0, 16 * BYTES + slack)
    for pad in pads:
        ptrs = [big.at(k * (BYTES + pad)) for k in range(16)]
        best, avg = timed(comm, ptrs[N:], ptrs[:N])
        print(json.dumps({"layout": "one allocation", "pad_bytes": pad, "best_us": round(best, 1), "avg_us": round(avg, 1),
                          "TBps_best": round(2 * N * BYTES / best / 1e6, 3)}), flush=True)
    # interleaved: source k next to destination k
    ptrs = [big.at(k * BYTES) for k in range(16)]
    best, avg = timed(comm, ptrs[1::2], ptrs[0::2])
    print(json.dumps({"layout": "one allocation, sources and destinations alternate", "best_us": round(best, 1), "avg_us": round(avg, 1)}), flush=True)
    big.free()
    for trial in range(3 if len(sys.argv) < 3 or sys.argv[2] == "allocs" else 0):
        bufs = [comm.alloc(BYTES) for _ in range(16)]
        if trial == 1:
            bufs = bufs[::-1]
        if trial == 2:
            bufs = bufs[0::2] + bufs[1::2]
        for b in bufs:
            comm.memset(b, 0, BYTES)
        best, avg = timed(comm, bufs[N:], bufs[:N])
        print(json.dumps({"layout": f"16 allocations, order {trial}", "best_us": round(best, 1), "avg_us": round(avg, 1),
                          "addr_mod_1GiB_MiB": [(b.ptr % (1 << 30)) >> 20 for b in bufs], "addr_mod_2MiB_KiB": [(b.ptr % (2 << 20)) >> 10 for b in bufs]}), flush=True)
        for b in bufs:
            b.free()
    comm.finalize()


if __name__ == "__main__":
    main()
