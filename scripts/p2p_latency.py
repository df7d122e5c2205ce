"""Ping-pong latency of xmpi_send / xmpi_recv between two ranks (threads of this process, or run twice as
processes via tests/gpu_harness) for a few message sizes and both copy engines (development tool)."""
import os
import sys
import threading
import time
import uuid

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpi_amd import xmpi  # noqa: E402

SIZES = [1, 1024, 65536, 1 << 20, 16 << 20]


def body(rank, key, out):
    comm = xmpi.Comm(rank, 2, 0, key)
    peer = 1 - rank
    a, b = comm.alloc(SIZES[-1]), comm.alloc(SIZES[-1])
    comm.fill(a, SIZES[-1], xmpi.U8, 0, 1 + rank)
    for direct in (1, -1):
        for engine in (0, 1):
            comm.barrier()
            comm.set_param("p2p_direct_bytes", direct)
            comm.set_param("copy_engine", engine)
            for n in SIZES:
                iters = 200 if n <= (1 << 20) else 40
                for w in range(10 + iters):
                    if w == 10:
                        t0 = time.perf_counter()
                    if rank == 0:
                        comm.send(a, n, xmpi.U8, peer, 1)
                        comm.recv(b, n, xmpi.U8, peer, 1)
                    else:
                        comm.recv(b, n, xmpi.U8, peer, 1)
                        comm.send(b, n, xmpi.U8, peer, 1)
                half = (time.perf_counter() - t0) / iters / 2
                if rank == 0:
                    out.append(f"direct={direct:>2} engine={engine} {n:>9} B : half round trip {half * 1e6:8.1f} us  {n / half / 1e9:8.2f} GB/s")
    comm.barrier()
    a.free()
    b.free()
    comm.finalize()


if __name__ == "__main__":
    key = f"p2plat-{os.getpid()}-{uuid.uuid4().hex[:6]}"
    out = []
    ts = [threading.Thread(target=body, args=(r, key, out)) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    print("\n".join(out))
