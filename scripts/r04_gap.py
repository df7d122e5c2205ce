#!/usr/bin/env python3
"""Why is the fold slower inside the N = 1 bench's timed region (750-800 us per 4 GiB launch) than on fresh buffers right behind
it (680-690 us)?  Same kernel, same launch shape, same process.  Eight rank threads like bench.py; the kernel's own dispatch
events throughout (prof_every = 1).  Legs:
  A  the collective itself (allreduce_repeat, 10 launches)
  B  rank 0 alone, reduce_local_multi on THE SAME sixteen buffers (sources = the ranks' send, destinations = their recv buffers)
  C  rank 0 alone on sixteen fresh buffers allocated one after the other (what bench.py's roofline_isolated does)
  D  as C, but allocated in the order the ranks allocate theirs (source, destination, source, ...)
  E  the timed buffers again, destinations in the order 0, 1, 2 ... vs the collective's own (me, me+1, ...)
prints one JSON object;   python scripts/r04_gap.py [MiB per rank]"""
import json
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpi_amd import xmpi  # noqa: E402

R = 8
NBYTES = (int(sys.argv[1]) if len(sys.argv) > 1 else 256) << 20
COUNT = NBYTES // 4
KEY = f"gap-{os.getpid()}"
out, bufs, lock = {}, {}, threading.Lock()
errors = []


def slot(b):
    return (b.ptr >> 12) & 15


def kernel_us(comm, fn, reps):
    comm.prof_reset()
    comm.prof_enable(True)
    comm.set_param("prof_every", 1)
    fn(reps)
    comm.sync()
    n, ms, by = comm.prof_get(xmpi.PROF_ZCOPY)
    comm.prof_enable(False)
    return {"launches": n, "avg_us": round(ms * 1e3 / max(1, n), 1), "bytes_per_launch": by // max(1, n)}


def rank_main(r):
    try:
        comm = xmpi.Comm(r, R, 0, KEY)
        send, recv = comm.alloc(NBYTES), comm.alloc(NBYTES)
        comm.fill(send, COUNT, xmpi.F32, xmpi.PAT_UNIFORM, 1000 + r)
        comm.memset(recv, 0, NBYTES)
        with lock:
            bufs[r] = (send, recv)
        comm.barrier()
        for _ in range(3):
            comm.allreduce(send, recv, COUNT, xmpi.F32, xmpi.SUM, xmpi.ALGO_AUTO)
        comm.barrier()
        a = kernel_us(comm, lambda k: comm.allreduce_repeat(send, recv, COUNT, xmpi.F32, xmpi.SUM, xmpi.ALGO_AUTO, k), 10)
        comm.barrier()
        if r == 0:
            out["slots_timed"] = {"send": [slot(bufs[k][0]) for k in range(R)], "recv": [slot(bufs[k][1]) for k in range(R)]}
            out["A_collective"] = a
            srcs = [bufs[k][0] for k in range(R)]
            dsts = [bufs[k][1] for k in range(R)]
            run = lambda d, s: kernel_us(comm, lambda k: [comm.reduce_local_multi(d, s, COUNT, xmpi.F32, xmpi.SUM) for _ in range(k)], 8)  # noqa: E731
            out["B_same_buffers_rank0_alone"] = run(dsts, srcs)
            zs = [comm.alloc(NBYTES) for _ in range(R)]
            zd = [comm.alloc(NBYTES) for _ in range(R)]
            for k, b in enumerate(zs):
                comm.fill(b, COUNT, xmpi.F32, xmpi.PAT_UNIFORM, 77 + k)
            out["slots_C"] = {"src": [slot(b) for b in zs], "dst": [slot(b) for b in zd]}
            out["C_fresh_sources_then_destinations"] = run(zd, zs)
            out["C_roles_swapped"] = run(zs, zd)
            for b in zs + zd:
                b.free()
            pairs = [comm.alloc(NBYTES) for _ in range(2 * R)]
            ps, pd = pairs[0::2], pairs[1::2]
            out["slots_D"] = {"src": [slot(b) for b in ps], "dst": [slot(b) for b in pd]}
            out["D_fresh_allocated_in_pairs"] = run(pd, ps)
            for b in pairs:
                b.free()
            # which side decides: the timed set's sources with fresh destinations and the other way round; 8 reads + 1 write; 1 read + 8 writes
            fs = [comm.alloc(NBYTES) for _ in range(R)]
            fd = [comm.alloc(NBYTES) for _ in range(R)]
            for k, b in enumerate(fs):
                comm.fill(b, COUNT, xmpi.F32, xmpi.PAT_UNIFORM, 177 + k)
            out["ptrs"] = {"timed_send": [hex(b.ptr) for b in srcs], "timed_recv": [hex(b.ptr) for b in dsts],
                           "fresh_src": [hex(b.ptr) for b in fs], "fresh_dst": [hex(b.ptr) for b in fd]}
            out["F_fresh"] = run(fd, fs)
            out["F_timed_sources_fresh_destinations"] = run(fd, srcs)
            out["F_fresh_sources_timed_destinations"] = run(dsts, fs)
            out["F_reads_timed_sources_one_write"] = run(fd[:1], srcs)
            out["F_reads_fresh_sources_one_write"] = run(fd[:1], fs)
            out["F_reads_timed_recv_as_sources_one_write"] = run(fd[:1], dsts)
            out["F_reads_fresh_dst_as_sources_one_write"] = run(fd[1:2], fd[:1] * 0 + fd[2:] + fs[:2])
            out["F_two_sources_timed_destinations"] = run(dsts, fs[:2])
            out["F_two_sources_fresh_destinations"] = run(fd, fs[:2])
            out["F_two_sources_timed_send_as_destinations"] = run(srcs, fs[:2])
            out["F_two_sources_fresh_src_as_destinations"] = run(fs[2:] + fd[:2], fs[:2])
            for b in fs + fd:
                b.free()
            out["E_same_buffers_again"] = run(dsts, srcs)
            out["E_destinations_rotated"] = run(dsts[3:] + dsts[:3], srcs)
            out["E_sources_as_destinations"] = run(srcs, dsts)
        comm.barrier()
        a2 = kernel_us(comm, lambda k: comm.allreduce_repeat(send, recv, COUNT, xmpi.F32, xmpi.SUM, xmpi.ALGO_AUTO, k), 10)
        if r == 0:
            out["A_collective_again"] = a2
        comm.barrier()
        comm.finalize()
    except BaseException:  # noqa: BLE001
        import traceback
        errors.append((r, traceback.format_exc()))


ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(R)]
for t in ts:
    t.start()
for t in ts:
    t.join()
for r, tb in errors:
    sys.stderr.write(f"rank {r}:\n{tb}\n")
print(json.dumps(out))
sys.exit(1 if errors else 0)
