#!/usr/bin/env python3
"""Is the fold's speed on a buffer a matter of how its allocation is ALIGNED in virtual memory (the GPU's page tables can use
large fragments only where virtual and physical address agree modulo the fragment)?  Raw hipMalloc blocks of several sizes; for
each: its base, the lowest set bit of the base, and the kernel time (dispatch events) of 8 reads out of it + 1 write elsewhere,
and of 1 read elsewhere + 8 writes into it.   python scripts/r04_tlb.py"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpi_amd import xmpi  # noqa: E402

hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
GIB, MIB = 1 << 30, 1 << 20
SLICE = 128 * MIB
COUNT = SLICE // 4


def kernel_us(comm, dsts, srcs, reps=6):
    comm.reduce_local_multi(dsts, srcs, COUNT, xmpi.F32, xmpi.SUM)
    comm.prof_reset()
    comm.prof_enable(True)
    comm.set_param("prof_every", 1)
    for _ in range(reps):
        comm.reduce_local_multi(dsts, srcs, COUNT, xmpi.F32, xmpi.SUM)
    comm.sync()
    n, ms, _ = comm.prof_get(xmpi.PROF_ZCOPY)
    comm.prof_enable(False)
    return round(ms * 1e3 / max(1, n), 1)


def main():
    comm = xmpi.Comm(0, 1, 0, f"tlb-{os.getpid()}")
    ref = comm.alloc(GIB)  # the fixed partner
    comm.memset(ref, 0, GIB)
    rows = []
    for size in [GIB] * 6 + [GIB - 2 * MIB] * 6 + [GIB + 62 * MIB] * 4 + [2 * GIB] * 3:
        p = C.c_void_p()
        if hip.hipMalloc(C.byref(p), size) != 0:
            break
        base = p.value
        comm.memset(base, 0, min(size, GIB))
        slices = [base + k * SLICE for k in range(min(size, GIB) // SLICE)][:8]
        if len(slices) < 8:
            slices = (slices * 8)[:8]
        r = kernel_us(comm, [ref.ptr], slices)                       # 8 reads from the block, 1 write elsewhere
        w = kernel_us(comm, slices, [ref.ptr + 512 * MIB])           # 1 read elsewhere, 8 writes into the block
        rows.append({"size_MiB": size // MIB, "base": hex(base), "align_MiB": (base & -base) / MIB, "reads_us": r, "writes_us": w})
        print(json.dumps(rows[-1]), flush=True)
    comm.finalize()


if __name__ == "__main__":
    main()
