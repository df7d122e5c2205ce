"""Multi-rank GPU scenarios.  Each function runs on ONE rank (a process or a thread) against the
C ABI through mpi_amd.xmpi and checks its own results against the CPU oracle.  On the 1-GPU
test box all ranks share device 0 (functional test of the real multi-process hipIpc path)."""
from __future__ import annotations

import ctypes
import os
import sys
import threading

import numpy as np

from mpi_amd import xmpi
from oracle import oracle

FLOATS = (xmpi.F16, xmpi.F32, xmpi.F64, xmpi.BF16)
# per-element tolerance factor on sum_i |x_i| when the summation order differs from rank order
TOL = {xmpi.F32: 1e-6, xmpi.F64: 2e-15, xmpi.F16: 2.0 ** -10, xmpi.BF16: 2.0 ** -7}


def check_reduced(got: np.ndarray, ins, dtype: int, op: int, exact: bool, what: str):
    want = oracle.reduce_ranks(ins, dtype, op)
    if got.size == 0:
        return
    if exact or dtype not in FLOATS or op in (xmpi.MIN, xmpi.MAX):
        bad = oracle.count_mismatch(got, want)
        if bad:
            idx = np.nonzero(got.view(np.uint8).reshape(got.size, -1) != want.view(np.uint8).reshape(want.size, -1))[0]
            raise AssertionError(f"{what}: {bad} bytes differ from the rank-order oracle; elements {idx.min()}..{idx.max()} "
                                 f"of {got.size} ({np.unique(idx).size} elements), e.g. got {got[idx[0]]!r} want {want[idx[0]]!r}")
        return
    g, w = oracle.as_float64(got, dtype), oracle.as_float64(want, dtype)
    if op == xmpi.SUM:
        scale = np.sum([np.abs(oracle.as_float64(x, dtype)) for x in ins], axis=0)
    else:
        scale = np.abs(w)
    n = len(ins)
    # f32/f64: BASELINE.md bound (two orderings, each within (N-1)*eps*sum|x|, N <= 8);
    # 16-bit floats round every partial sum to 11 / 8 bits: (N-1) * 2 * eps
    bound = TOL[dtype] * scale * (max(1, n - 1) if dtype in (xmpi.F16, xmpi.BF16) else 1.0)
    err = np.abs(g - w)
    worst = int(np.argmax(err - bound))
    assert np.all(err <= bound), f"{what}: |delta|={err[worst]:.3e} > bound {bound[worst]:.3e} at {worst}"


def allreduce_case(comm, dtype, count, algo, op=xmpi.SUM, pattern=xmpi.PAT_UNIFORM, inplace=False, seed0=1000,
                   exact=None, misalign=0):
    """misalign: the buffers start `misalign` elements into their (256-byte aligned) allocations"""
    rank, size = comm.rank(), comm.size()
    es = xmpi.DTYPE_SIZE[dtype]
    shift = misalign * es
    send = comm.alloc(count * es + shift)
    recv = send if inplace else comm.alloc(count * es + shift)
    comm.fill(send.at(shift), count, dtype, pattern, seed0 + rank)
    ins = [oracle.fill(count, dtype, pattern, seed0 + r) for r in range(size)]
    mine = send.download(xmpi.NUMPY_DTYPE[dtype], count, byte_offset=shift)
    assert mine.tobytes() == ins[rank].tobytes(), "device fill differs from oracle fill"
    if not inplace:
        comm.memset(recv, 0xA5, count * es + shift)
    comm.allreduce(send.at(shift), recv.at(shift), count, dtype, op, algo)
    got = recv.download(xmpi.NUMPY_DTYPE[dtype], count, byte_offset=shift)
    if exact is None:
        # rank-order algorithms and exactly-summable inputs must be bit-identical
        exact = algo in (xmpi.ALGO_DIRECT, xmpi.ALGO_ZCOPY, xmpi.ALGO_AUTO) or size <= 2 or pattern in (xmpi.PAT_CONST,) or (
            dtype in (xmpi.F16, xmpi.BF16) and pattern in (xmpi.PAT_UNIFORM, xmpi.PAT_INDEX) and op == xmpi.SUM)
    check_reduced(got, ins, dtype, op, exact,
                  f"allreduce {xmpi.DTYPE_NAME[dtype]} n={count} algo={algo} op={op} pat={pattern} inplace={inplace}")
    if not inplace:
        again = send.download(xmpi.NUMPY_DTYPE[dtype], count, byte_offset=shift)
        assert again.tobytes() == ins[rank].tobytes(), "sendbuf was modified"
        recv.free()
    send.free()


def _hip_runtime():
    """the HIP runtime the library under test allocates from: ROCm's -- or, in the CPU suite's tests/devsim runs, the stand-in
    library itself (it carries the simulated runtime; memory from the real one would mean nothing to it)"""
    return ctypes.CDLL(os.environ.get("XMPI_DEVSIM_LIB") or "libamdhip64.so")


def sc_allreduce_small(comm, args):
    counts = args.get("counts", [0, 1, 3, 17, 1000, 4099, 65536 + 5])
    for dtype in args.get("dtypes", [xmpi.F32, xmpi.I64, xmpi.F16, xmpi.F64, xmpi.I32, xmpi.U8, xmpi.BF16]):
        for algo in (xmpi.ALGO_RING, xmpi.ALGO_RHD, xmpi.ALGO_DIRECT, xmpi.ALGO_AUTO):
            for count in counts:
                allreduce_case(comm, dtype, count, algo)
    # other operators, signed data, in place
    for op in (xmpi.PROD, xmpi.MIN, xmpi.MAX):
        for dtype in (xmpi.F32, xmpi.I32, xmpi.F16):
            # a float product in another order may underflow differently: compare it only where the
            # fold is in rank order (DIRECT, bit-exact)
            if not (op == xmpi.PROD and dtype in FLOATS):
                allreduce_case(comm, dtype, 3001, xmpi.ALGO_RING, op=op, pattern=xmpi.PAT_SIGNED)
            allreduce_case(comm, dtype, 3001, xmpi.ALGO_DIRECT, op=op, pattern=xmpi.PAT_SIGNED)
    for algo in (xmpi.ALGO_RING, xmpi.ALGO_RHD, xmpi.ALGO_DIRECT):
        allreduce_case(comm, xmpi.F32, 100003, algo, inplace=True)
        allreduce_case(comm, xmpi.F32, 100003, algo, pattern=xmpi.PAT_SIGNED)
        allreduce_case(comm, xmpi.F64, 50001, algo, pattern=xmpi.PAT_SIGNED)
        allreduce_case(comm, xmpi.I64, 4097, algo, pattern=xmpi.PAT_CONST)  # x_r = r+1 -> N(N+1)/2
    # buffers that start one element into their allocation (no 16-byte packets: the element kernels), in place:
    # a fused ring step stores the same sum twice (locally and into the next rank's slot) -- both from the
    # untouched operands, also when the local destination IS an operand
    for algo in (xmpi.ALGO_RING, xmpi.ALGO_RHD, xmpi.ALGO_DIRECT, xmpi.ALGO_AUTO):
        allreduce_case(comm, xmpi.F32, 40001, algo, pattern=xmpi.PAT_SIGNED, inplace=True, misalign=1)
        allreduce_case(comm, xmpi.I64, 5003, algo, pattern=xmpi.PAT_UNIFORM, inplace=True, misalign=1, op=xmpi.PROD)
        allreduce_case(comm, xmpi.F16, 30011, algo, misalign=3)
    # push-only (only stores cross a link): out of place with equal chunks the receive buffers are the staging area, otherwise --
    # in place, ragged counts -- the communicators' own blocks are; rank order either way: bit-exact
    for count in (1, 17, 4099, 100003, 8 * 4096):
        for inplace in (False, True):
            allreduce_case(comm, xmpi.F32, count, xmpi.ALGO_ZPUSH, pattern=xmpi.PAT_SIGNED, inplace=inplace, exact=True)
    allreduce_case(comm, xmpi.F16, 30011, xmpi.ALGO_ZPUSH, inplace=True, misalign=3, exact=True)
    allreduce_case(comm, xmpi.I64, 5003, xmpi.ALGO_ZPUSH, op=xmpi.MAX, pattern=xmpi.PAT_SIGNED, inplace=True, exact=True)
    for root in sorted({0, comm.size() - 1}):  # ... and reduce: the folded chunks stored to the root only
        reduce_case(comm, xmpi.F32, 100003, root, xmpi.ALGO_ZPUSH, exact=True, what="push-only reduce")
        reduce_case(comm, xmpi.I64, 17, root, xmpi.ALGO_ZPUSH, pat=xmpi.PAT_UNIFORM, what="push-only reduce")
    # known answer: x_r[i] = r + 1  =>  every element N(N+1)/2
    n = comm.size()
    buf = comm.alloc(4 * 1024)
    comm.fill(buf, 1024, xmpi.F32, xmpi.PAT_CONST, comm.rank())
    comm.allreduce(buf, buf, 1024, xmpi.F32, xmpi.SUM, xmpi.ALGO_RING)
    assert np.all(buf.download(np.float32, 1024) == np.float32(n * (n + 1) / 2))
    buf.free()


def sc_allreduce_medium(comm, args):
    """multi-piece, multi-channel pipelines (several MiB per rank)"""
    for piece in (0, 64 << 10):
        comm.set_param("piece_bytes", piece)
        for channels in (1, 4):
            comm.set_param("channels", channels)
            for algo in (xmpi.ALGO_RING, xmpi.ALGO_RHD, xmpi.ALGO_DIRECT):
                allreduce_case(comm, xmpi.F32, (3 << 20) + 7, algo)
                allreduce_case(comm, xmpi.F16, (2 << 20) + 3, algo)
    comm.set_param("piece_bytes", 0)
    comm.set_param("channels", 4)
    for engine in (1, 0):
        comm.set_param("copy_engine", engine)
        allreduce_case(comm, xmpi.I64, (1 << 20) + 1, xmpi.ALGO_RING)
        allreduce_case(comm, xmpi.F32, (4 << 20) + 5, xmpi.ALGO_DIRECT, pattern=xmpi.PAT_SIGNED)


def allgather_case(comm, dtype, count, algo, inplace=False):
    rank, size = comm.rank(), comm.size()
    es = xmpi.DTYPE_SIZE[dtype]
    recv = comm.alloc(count * es * size)
    comm.memset(recv, 0x5A, count * es * size)
    if inplace:
        send_ptr = recv.at(rank * count * es)
        comm.fill(send_ptr, count, dtype, xmpi.PAT_INDEX, rank)
        send = None
    else:
        send = comm.alloc(count * es)
        comm.fill(send, count, dtype, xmpi.PAT_INDEX, rank)
        send_ptr = send.ptr
    comm.allgather(send_ptr, recv, count, dtype, algo)
    got = recv.download(xmpi.NUMPY_DTYPE[dtype], count * size)
    want = oracle.allgather([oracle.fill(count, dtype, xmpi.PAT_INDEX, r) for r in range(size)], dtype)
    assert got.tobytes() == want.tobytes(), f"allgather {xmpi.DTYPE_NAME[dtype]} n={count} algo={algo} inplace={inplace}"
    recv.free()
    if send:
        send.free()


def sc_allgather(comm, args):
    for dtype in (xmpi.I64, xmpi.U8, xmpi.F32):
        for algo in (xmpi.ALGO_RING, xmpi.ALGO_DIRECT, xmpi.ALGO_AUTO):
            for count in args.get("counts", (0, 1, 5, 1000, 4099, (1 << 20) + 3)):
                allgather_case(comm, dtype, count, algo)
    allgather_case(comm, xmpi.I64, 70001, xmpi.ALGO_RING, inplace=True)
    allgather_case(comm, xmpi.I64, 70001, xmpi.ALGO_DIRECT, inplace=True)


def sc_bcast_reduce(comm, args):
    rank, size = comm.rank(), comm.size()
    # the full-mesh staged forms (bcast: scatter + forward, reduce: reduce-scatter + gather) were validated on the
    # GPU with the default transport only; with the copy kernel as transport this scenario keeps to tree / one-hop
    mesh_forms = comm.get_param("copy_engine") == 0
    for root in sorted({0, size - 1, size // 2}):
        for dtype, count in ((xmpi.U8, 1), (xmpi.I64, 4099), (xmpi.F32, (1 << 20) + 9), (xmpi.F16, 0)):
            es = xmpi.DTYPE_SIZE[dtype]
            for algo in (xmpi.ALGO_TREE, xmpi.ALGO_DIRECT, xmpi.ALGO_AUTO) if mesh_forms else (xmpi.ALGO_TREE, xmpi.ALGO_AUTO):
                buf = comm.alloc(count * es)
                comm.fill(buf, count, dtype, xmpi.PAT_UNIFORM, 40 + rank)
                comm.bcast(buf, count, dtype, root, algo)
                got = buf.download(xmpi.NUMPY_DTYPE[dtype], count)
                want = oracle.fill(count, dtype, xmpi.PAT_UNIFORM, 40 + root)
                assert got.tobytes() == want.tobytes(), f"bcast root={root} {xmpi.DTYPE_NAME[dtype]} n={count} algo={algo}"
                buf.free()
        for algo in (xmpi.ALGO_TREE, xmpi.ALGO_DIRECT):
            # (the 4 MiB case takes DIRECT's reduce-scatter + gather form; the others send everything to the root)
            for dtype, count, pat in ((xmpi.F32, 100003, xmpi.PAT_SIGNED), (xmpi.I64, 4099, xmpi.PAT_UNIFORM),
                                      (xmpi.F16, 5001, xmpi.PAT_UNIFORM), (xmpi.F64, 1, xmpi.PAT_SIGNED),
                                      (xmpi.F32, (1 << 20) + 9, xmpi.PAT_SIGNED)):
                if count > (1 << 20) and not mesh_forms:
                    continue
                es = xmpi.DTYPE_SIZE[dtype]
                send = comm.alloc(count * es)
                recv = comm.alloc(count * es)
                comm.fill(send, count, dtype, pat, 70 + rank)
                comm.reduce(send, recv, count, dtype, xmpi.SUM, root, algo)
                if rank == root:
                    ins = [oracle.fill(count, dtype, pat, 70 + r) for r in range(size)]
                    got = recv.download(xmpi.NUMPY_DTYPE[dtype], count)
                    exact = algo == xmpi.ALGO_DIRECT or size <= 2 or (dtype == xmpi.F16 and pat == xmpi.PAT_UNIFORM)
                    check_reduced(got, ins, dtype, xmpi.SUM, exact, f"reduce root={root} algo={algo}")
                send.free()
                recv.free()
    # host-resident buffers go through the same collectives (staged through HBM)
    x = oracle.fill(1000, xmpi.F32, xmpi.PAT_UNIFORM, 5 + rank)
    out = np.zeros_like(x)
    comm.allreduce(x, out, 1000, xmpi.F32, xmpi.SUM, xmpi.ALGO_DIRECT)
    want = oracle.reduce_ranks([oracle.fill(1000, xmpi.F32, xmpi.PAT_UNIFORM, 5 + r) for r in range(size)], xmpi.F32, 0)
    assert out.tobytes() == want.tobytes()


def _zc_launches(comm) -> int:
    return comm.prof_get(xmpi.PROF_ZCOPY)[0]


def _zc_why(comm) -> str:
    return (f"zero-copy attempts {comm.get_param('zc_seq')}, went staged: {comm.get_param('zc_fallbacks_unregistered')} "
            f"(a buffer not registered / not exportable), {comm.get_param('zc_fallbacks_unmappable')} (a peer could not map)")


def sc_zero_copy(comm, args):
    """Zero-copy collectives (zcopy.cpp): kernels read and write the peers' registered buffers in place.
    Rank-order fold => every result is bit-identical to the oracle, for every dtype and operator."""
    rank, size = comm.rank(), comm.size()
    Z = xmpi.ALGO_ZCOPY
    comm.prof_enable(True)
    comm.set_param("prof_every", 1)
    before = _zc_launches(comm)
    for dtype in (xmpi.F32, xmpi.I64, xmpi.F16, xmpi.F64, xmpi.I32, xmpi.U8, xmpi.BF16):
        for count in args.get("counts", [0, 1, 3, 17, 1000, 4099, 65536 + 5]):
            allreduce_case(comm, dtype, count, Z, exact=True)
    allreduce_case(comm, xmpi.F32, (3 << 20) + 7, Z, pattern=xmpi.PAT_SIGNED, exact=True)
    allreduce_case(comm, xmpi.F32, (3 << 20) + 7, xmpi.ALGO_AUTO, pattern=xmpi.PAT_SIGNED, inplace=True, exact=True)
    for op in (xmpi.PROD, xmpi.MIN, xmpi.MAX):
        for dtype in (xmpi.F32, xmpi.I32, xmpi.F16, xmpi.BF16):
            allreduce_case(comm, dtype, 3001, Z, op=op, pattern=xmpi.PAT_SIGNED, exact=True)
    allreduce_case(comm, xmpi.F64, 50001, Z, pattern=xmpi.PAT_SIGNED, inplace=True, exact=True)
    # the write-only variant (contributions pushed into the peers' receive buffers, folded locally, results pushed
    # back): equal chunks and out-of-place buffers take it, anything else runs the read-based form; same bits
    ZP = xmpi.ALGO_ZPUSH
    for dtype, count in ((xmpi.F32, 4 * size * 1024), (xmpi.F16, 8 * size * 4099), (xmpi.I64, 2 * size * 3), (xmpi.F64, 2 * size * 50001),
                         (xmpi.BF16, 8 * size), (xmpi.F32, 100003), (xmpi.U8, 16 * size * 7 + 5)):
        allreduce_case(comm, dtype, count, ZP, pattern=xmpi.PAT_SIGNED, exact=True)
    allreduce_case(comm, xmpi.F32, 4 * size * 65536, ZP, op=xmpi.MAX, pattern=xmpi.PAT_SIGNED, exact=True)
    allreduce_case(comm, xmpi.F32, 4 * size * 1024, ZP, pattern=xmpi.PAT_SIGNED, inplace=True, exact=True)
    # the big allreduce gave every rank a chunk: the zero-copy kernel ran here (not a silent fallback)
    # (rank 0 always launches; ranks hosted by threads of its process leave their chunks to it)
    assert rank != 0 or _zc_launches(comm) > before, "zero-copy path did not run: " + _zc_why(comm)
    assert comm.get_param("zc_fallbacks_unregistered") == 0 and comm.get_param("zc_fallbacks_unmappable") == 0, _zc_why(comm)

    # buffers at odd element offsets: the element-wise kernel, still in place in the peers' memory
    for dtype, count in ((xmpi.F32, 10007), (xmpi.F16, 4099), (xmpi.U8, 1001)):
        es = xmpi.DTYPE_SIZE[dtype]
        send, recv = comm.alloc((count + 3) * es), comm.alloc((count + 3) * es)
        comm.fill(send.at(es), count, dtype, xmpi.PAT_SIGNED, 300 + rank)
        comm.allreduce(send.at(es), recv.at(2 * es), count, dtype, xmpi.SUM, Z)
        ins = [oracle.fill(count, dtype, xmpi.PAT_SIGNED, 300 + r) for r in range(size)]
        got = recv.download(xmpi.NUMPY_DTYPE[dtype], count, byte_offset=2 * es)
        check_reduced(got, ins, dtype, xmpi.SUM, True, f"zero-copy unaligned {xmpi.DTYPE_NAME[dtype]}")
        send.free()
        recv.free()

    for dtype in (xmpi.I64, xmpi.U8, xmpi.F32):
        for count in (0, 1, 5, 1000, 4099, (1 << 20) + 3):
            allgather_case(comm, dtype, count, Z)
    allgather_case(comm, xmpi.I64, 70001, Z, inplace=True)

    for push_bytes in (256 << 10, 0):  # root pushes to everyone / scatter + allgather of the chunks
        comm.set_param("zc_bcast_push_bytes", push_bytes)
        for root in sorted({0, size - 1, size // 2}):
            for dtype, count in ((xmpi.U8, 1), (xmpi.U8, 37), (xmpi.I64, 4099), (xmpi.F32, (1 << 20) + 9), (xmpi.F16, 0)):
                es = xmpi.DTYPE_SIZE[dtype]
                buf = comm.alloc(count * es)
                comm.fill(buf, count, dtype, xmpi.PAT_UNIFORM, 40 + rank)
                comm.bcast(buf, count, dtype, root, Z)
                got = buf.download(xmpi.NUMPY_DTYPE[dtype], count)
                want = oracle.fill(count, dtype, xmpi.PAT_UNIFORM, 40 + root)
                assert got.tobytes() == want.tobytes(), f"zero-copy bcast root={root} {xmpi.DTYPE_NAME[dtype]} n={count}"
                buf.free()
    comm.set_param("zc_bcast_push_bytes", 256 << 10)

    for root in sorted({0, size - 1, size // 2}):
        for dtype, count, pat in ((xmpi.F32, 100003, xmpi.PAT_SIGNED), (xmpi.I64, 4099, xmpi.PAT_UNIFORM),
                                  (xmpi.F16, 5001, xmpi.PAT_SIGNED), (xmpi.F64, 1, xmpi.PAT_SIGNED)):
            for inplace_root in (False, True):
                es = xmpi.DTYPE_SIZE[dtype]
                send = comm.alloc(count * es)
                recv = send if (inplace_root and rank == root) else comm.alloc(count * es)
                comm.fill(send, count, dtype, pat, 70 + rank)
                comm.reduce(send, recv if rank == root else None, count, dtype, xmpi.SUM, root, Z)
                if rank == root:
                    ins = [oracle.fill(count, dtype, pat, 70 + r) for r in range(size)]
                    got = recv.download(xmpi.NUMPY_DTYPE[dtype], count)
                    check_reduced(got, ins, dtype, xmpi.SUM, True,
                                  f"zero-copy reduce root={root} {xmpi.DTYPE_NAME[dtype]} n={count} inplace={inplace_root}")
                if recv is not send:
                    recv.free()
                send.free()

    # buffers the peers cannot map.  HOST slices (what a caller of the reference passes): a registered arena block stands in for
    # each and the same kernel runs -- with ranks that meet on the device (dsync.cpp: one process per rank) and with ranks that meet
    # through the control block (api.cpp collective) alike: nobody falls back to the staged schedule
    dsync = comm.get_param("dsync") == 1
    bounced0 = comm.get_param("dsync_bounced")
    staged0, zc0 = comm.get_param("zc_fallbacks_unregistered"), comm.get_param("zc_seq")
    x = oracle.fill(1000, xmpi.F32, xmpi.PAT_UNIFORM, 5 + rank)
    out = np.zeros_like(x)
    comm.allreduce(x, out, 1000, xmpi.F32, xmpi.SUM, Z)  # host memory
    want = oracle.reduce_ranks([oracle.fill(1000, xmpi.F32, xmpi.PAT_UNIFORM, 5 + r) for r in range(size)], xmpi.F32, 0)
    assert out.tobytes() == want.tobytes()
    assert not dsync or comm.get_param("dsync_bounced") == bounced0 + 2, "host send + receive buffer: two stand-ins"
    if not dsync and comm.get_param("zero_copy") == 1 and size > 1:
        assert comm.get_param("zc_seq") > zc0 and comm.get_param("zc_fallbacks_unregistered") == staged0, _zc_why(comm)
    y = x.copy()  # ... in place, and a host operand with a result in HBM
    comm.allreduce(y, y, 1000, xmpi.F32, xmpi.SUM, Z)
    assert y.tobytes() == want.tobytes(), "allreduce of a host slice in place"
    dres = comm.alloc(4000)
    comm.allreduce(x, dres, 1000, xmpi.F32, xmpi.SUM, Z)
    assert dres.download(np.float32, 1000).tobytes() == want.tobytes(), "host operand, result in HBM"
    dres.free()
    # device memory from another allocator (here: plain hipMalloc) on ONE rank: staged until it is registered,
    # zero-copy while it is, staged again after deregistration
    import ctypes
    hip = _hip_runtime()
    count = 20011
    send, mine = comm.alloc(count * 4), comm.alloc(count * 4)
    foreign = ctypes.c_void_p(0)
    if rank == size - 1:
        comm.sync()  # (selects this rank's device on this thread)
        assert hip.hipMalloc(ctypes.byref(foreign), ctypes.c_size_t(count * 4)) == 0
    recv_ptr = foreign.value if rank == size - 1 else mine.ptr
    comm.fill(send, count, xmpi.F32, xmpi.PAT_SIGNED, 900 + rank)
    ins = [oracle.fill(count, xmpi.F32, xmpi.PAT_SIGNED, 900 + r) for r in range(size)]

    def result():
        out = np.empty(count, dtype=np.float32)
        xmpi._check(xmpi.lib().xmpi_memcpy(comm.handle, out.ctypes.data, recv_ptr, count * 4), "download")
        return out

    staged0 = comm.get_param("zc_fallbacks_unregistered")
    for phase in ("unregistered", "registered", "deregistered"):
        if rank == size - 1 and phase == "registered":
            comm.register(recv_ptr, count * 4)
        if rank == size - 1 and phase == "deregistered":
            comm.deregister(recv_ptr)
        comm.memset(recv_ptr, 0, count * 4)
        before, b0 = _zc_launches(comm), comm.get_param("dsync_bounced")
        comm.allreduce(send, recv_ptr, count, xmpi.F32, xmpi.SUM, Z)
        check_reduced(result(), ins, xmpi.F32, xmpi.SUM, True, f"foreign buffer, {phase}")
        if dsync:  # the kernel runs in every phase; only the owner of the foreign buffer needs a stand-in, unless registered
            assert _zc_launches(comm) > before
            assert comm.get_param("dsync_bounced") - b0 == (1 if rank == size - 1 and phase != "registered" else 0), phase
        elif phase == "registered":
            assert rank != 0 or _zc_launches(comm) > before, _zc_why(comm)
        else:
            assert _zc_launches(comm) == before, "zero-copy kernel ran although a peer's buffer was not registered"
    assert dsync or comm.get_param("zc_fallbacks_unregistered") == staged0 + 2, _zc_why(comm)
    assert comm.get_param("zc_fallbacks_unmappable") == 0, _zc_why(comm)
    # buffers of xmpi_malloc are blocks of long-lived arenas: nothing was mapped per call
    assert comm.get_param("heap_arenas") <= 8, comm.get_param("heap_arenas")
    comm.barrier()
    if rank == size - 1:
        assert hip.hipFree(foreign) == 0
    send.free()
    mine.free()
    comm.prof_enable(False)


def sc_nonblocking(comm, args):
    """Non-blocking collectives: issued back to back, completed by the communicator's worker in issue order
    while the caller does its own work; results identical to the blocking forms."""
    rank, size = comm.rank(), comm.size()
    n = args.get("count", 300007)
    ops = []
    for k, (dtype, algo) in enumerate(((xmpi.F32, xmpi.ALGO_AUTO), (xmpi.I64, xmpi.ALGO_RING), (xmpi.F16, xmpi.ALGO_DIRECT))):
        es = xmpi.DTYPE_SIZE[dtype]
        s, r = comm.alloc(n * es), comm.alloc(n * es)
        comm.fill(s, n, dtype, xmpi.PAT_SIGNED if dtype != xmpi.I64 else xmpi.PAT_UNIFORM, 600 + 10 * k + rank)
        ops.append((dtype, algo, s, r, k))
    reqs = [comm.iallreduce(s, r, n, dtype, xmpi.SUM, algo) for dtype, algo, s, r, _ in ops]
    gs, gr = comm.alloc(n * 8), comm.alloc(n * 8 * size)
    comm.fill(gs, n, xmpi.I64, xmpi.PAT_INDEX, rank)
    reqs.append(comm.iallgather(gs, gr, n, xmpi.I64))
    b = comm.alloc(n * 4)
    comm.fill(b, n, xmpi.F32, xmpi.PAT_UNIFORM, 640 + rank)
    reqs.append(comm.ibcast(b, n, xmpi.F32, size - 1))
    # the caller's own work overlaps with all of that
    x, y, z = comm.alloc(n * 4), comm.alloc(n * 4), comm.alloc(n * 4)
    comm.fill(x, n, xmpi.F32, xmpi.PAT_UNIFORM, 1)
    comm.fill(y, n, xmpi.F32, xmpi.PAT_UNIFORM, 2)
    for _ in range(5):
        comm.reduce_local(z, x, y, n, xmpi.F32, xmpi.SUM)
    assert z.download(np.float32, n).tobytes() == oracle.reduce2(oracle.fill(n, xmpi.F32, 0, 1), oracle.fill(n, xmpi.F32, 0, 2),
                                                                  xmpi.F32, xmpi.SUM).tobytes()
    for q in reversed(reqs):  # completion order is the issue order, whatever order they are waited for in
        comm.request_wait(q)
    for dtype, algo, s, r, k in ops:
        pat = xmpi.PAT_SIGNED if dtype != xmpi.I64 else xmpi.PAT_UNIFORM
        ins = [oracle.fill(n, dtype, pat, 600 + 10 * k + q) for q in range(size)]
        exact = algo != xmpi.ALGO_RING or dtype == xmpi.I64
        check_reduced(r.download(xmpi.NUMPY_DTYPE[dtype], n), ins, dtype, xmpi.SUM, exact, f"iallreduce {k}")
    want = oracle.allgather([oracle.fill(n, xmpi.I64, xmpi.PAT_INDEX, q) for q in range(size)], xmpi.I64)
    assert gr.download(np.int64, n * size).tobytes() == want.tobytes()
    assert b.download(np.float32, n).tobytes() == oracle.fill(n, xmpi.F32, xmpi.PAT_UNIFORM, 640 + size - 1).tobytes()
    # a blocking collective issued while a non-blocking one is in flight runs after it, on every rank
    dtype, algo, s, r, k = ops[0]
    q1 = comm.iallreduce(s, r, n, dtype, xmpi.MAX, xmpi.ALGO_AUTO)
    comm.allreduce(r, z, n, dtype, xmpi.SUM, xmpi.ALGO_AUTO)  # consumes the result of q1
    assert comm.request_test(q1)
    comm.request_wait(q1)
    ins = [oracle.fill(n, dtype, xmpi.PAT_SIGNED, 600 + q) for q in range(size)]
    mx = oracle.reduce_ranks(ins, dtype, xmpi.MAX)
    check_reduced(z.download(np.float32, n), [mx] * size, dtype, xmpi.SUM, True, "blocking after non-blocking")
    # the operation's own status comes back from wait
    bad = comm.iallreduce(s, r, n, 99, xmpi.SUM, xmpi.ALGO_AUTO)
    try:
        comm.request_wait(bad)
        raise AssertionError("a bad dtype must be reported by request_wait")
    except xmpi.XmpiError as e:
        assert e.code == xmpi.ERR_ARG
    comm.barrier()
    for buf in [gs, gr, b, x, y, z] + [o[2] for o in ops] + [o[3] for o in ops]:
        buf.free()


# ---- point to point ------------------------------------------------------------------------------

BOUNCE_LENGTHS = [0, 1, 10, 100, 1000, 10**4, 10**5, 10**6, 10**7]  # examples/bounce/bounce.go:33


def sc_bounce(comm, args):
    """examples/bounce/bounce.go:83-137: even rank sends to rank+1, odd rank echoes; the even rank
    checks the echo is bit-identical (bytes.Equal / floats.Equal)."""
    rank, size = comm.rank(), comm.size()
    assert size % 2 == 0
    even = rank % 2 == 0
    peer = rank + 1 if even else rank - 1
    maxlen = BOUNCE_LENGTHS[-1]
    msg = comm.alloc(maxlen)
    rcv = comm.alloc(maxlen)
    comm.fill(msg, maxlen, xmpi.U8, xmpi.PAT_UNIFORM, 1 + rank)
    # what bounce measures, by the library's own probe: every engine (copy engine, copy kernel, the stepped kernels' system-scope
    # accesses), write to the peer and read from it
    for engine in (0, 1, 2):
        for direction in (0, 1):
            if even:
                assert comm.link_probe(peer, 4 << 20, engine, 3, direction) > 0
            comm.barrier()
    for length in BOUNCE_LENGTHS:
        for dtype, count in ((xmpi.U8, length), (xmpi.F64, length // 8)):
            nbytes = count * xmpi.DTYPE_SIZE[dtype]
            comm.memset(rcv, 0, maxlen)  # bounce.go:110-112: no false positives from the previous round
            if even:
                comm.send(msg, count, dtype, peer, 0)
                got = comm.recv(rcv, count, dtype, peer, 0)
                assert got == count
                assert comm.count_mismatch(msg, rcv, nbytes) == 0, f"echo of {nbytes} bytes differs"
            else:
                got = comm.recv(rcv, count, dtype, peer, 0)
                assert got == count
                comm.send(rcv, count, dtype, peer, 0)
    # BASELINE cfg 2: 1 MiB float32 ping-pong, bit-exact
    n = 262144
    comm.fill(msg, n, xmpi.F32, xmpi.PAT_UNIFORM, 1)
    for _ in range(5):
        comm.memset(rcv, 0, n * 4)
        if even:
            comm.send(msg, n, xmpi.F32, peer, 3)
            comm.recv(rcv, n, xmpi.F32, peer, 3)
            assert comm.count_mismatch(msg, rcv, n * 4) == 0
        else:
            comm.recv(rcv, n, xmpi.F32, peer, 3)
            comm.send(rcv, n, xmpi.F32, peer, 3)
    # registered buffers of >= p2p_direct_bytes were pulled straight out of the sender's HBM (one copy);
    # shorter ones, or everything when p2p_direct_bytes < 0, went through the mail slots
    direct, staged = comm.get_param("p2p_direct_count"), comm.get_param("p2p_staged_count")
    thr = comm.get_param("p2p_direct_bytes")
    sizes = [c * xmpi.DTYPE_SIZE[d] for length in BOUNCE_LENGTHS for d, c in ((xmpi.U8, length), (xmpi.F64, length // 8))]
    want_direct = 0 if thr < 0 else sum(1 for b in sizes if b >= max(1, thr)) + 5
    assert direct == want_direct and staged == len(sizes) + 5 - want_direct, (direct, staged, want_direct)
    # ... by the receive agent (the copy-and-ack kernel that lingers for the next message) when it is on and the message is
    # not longer than it takes: every such message was served by it, with far fewer launches than messages
    if comm.get_param("p2p_agent_us") > 0 and comm.get_param("p2p_kernel_ack") == 1 and thr >= 0:
        small = sum(1 for b in sizes if max(1, thr) <= b <= 512 << 10) + 5 * (1 if n * 4 <= 512 << 10 else 0)
        assert comm.get_param("p2p_agent_served") == small, (comm.get_param("p2p_agent_served"), small)
        assert 1 <= comm.get_param("p2p_agent_launches") <= small
    else:
        assert comm.get_param("p2p_agent_served") == 0


def sc_host_payloads(comm, args):
    """What the reference's callers hand to Send / Receive is HOST memory (Go slices: bounce.go:96, helloworld.go:58).  Here
    every mix -- host -> host (the host lanes of the shared segment), host -> device (DMA out of the lane), device -> host
    (one copy out of the sender's registered HBM), device -> device -- at lengths around the lane's piece and ring sizes,
    with odd offsets, bit for bit against the oracle's pattern; then truncation, an un-waited send, many tags at once."""
    import threading
    rank, size = comm.rank(), comm.size()
    assert size == 2
    peer = 1 - rank
    lane = comm.get_param("host_lane_bytes")
    lanes_on = lane > 0
    piece = lane // 4 if lanes_on else 65536
    lengths = [1, 8, 1000, piece - 1, piece, piece + 1, 4 * piece, 4 * piece + 1, 9 * piece + 5, (1 << 20) + 3, 10_000_000]
    cap = max(lengths) + 64
    dev_a, dev_b = comm.alloc(cap), comm.alloc(cap)
    host_a, host_b = np.zeros(cap, dtype=np.uint8), np.zeros(cap, dtype=np.uint8)
    lane0, direct0, staged0 = comm.get_param("p2p_lane_count"), comm.get_param("p2p_direct_count"), comm.get_param("p2p_staged_count")
    n_lane = n_direct = 0
    tag = 0
    for src_kind in ("host", "dev"):
        for dst_kind in ("host", "dev"):
            for i, n in enumerate(lengths):
                off = (3 * i) % 16 if src_kind == "host" or dst_kind == "host" else 0  # host spans need no alignment
                tag += 1
                want = oracle.fill(n, xmpi.U8, xmpi.PAT_UNIFORM, 100 + tag)
                if rank == 0:
                    if src_kind == "host":
                        host_a[off:off + n] = want
                        comm.send(host_a[off:off + n], n, xmpi.U8, peer, tag)
                        host_a[off:off + n] = 0
                    else:
                        dev_a.upload(want, 0)
                        comm.send(dev_a, n, xmpi.U8, peer, tag)
                else:
                    if dst_kind == "host":
                        host_b[:] = 0xEE
                        got = comm.recv(host_b[off:off + n], n, xmpi.U8, peer, tag)
                        assert got == n
                        assert host_b[off:off + n].tobytes() == want.tobytes(), f"{src_kind}->{dst_kind} {n} bytes differ"
                        assert (off == 0 or host_b[off - 1] == 0xEE) and host_b[off + n] == 0xEE, "wrote outside the destination"
                    else:
                        comm.memset(dev_b, 0xEE, cap)
                        got = comm.recv(dev_b, n, xmpi.U8, peer, tag)
                        assert got == n
                        back = dev_b.download(np.uint8, n + 1)
                        assert back[:n].tobytes() == want.tobytes(), f"{src_kind}->{dst_kind} {n} bytes differ"
                        assert back[n] == 0xEE, "wrote past the destination"
                    if src_kind == "host":
                        n_lane += 1
                    else:  # a registered source is pulled by the receiver, whatever the destination
                        n_direct += 1
    if rank == 1:
        lanes_used = comm.get_param("p2p_lane_count") - lane0
        if lanes_on:
            assert lanes_used == n_lane, (lanes_used, n_lane)
            assert comm.get_param("p2p_staged_count") == staged0, "nothing went through the HBM slots"
            assert comm.get_param("p2p_direct_count") - direct0 == n_direct
        else:
            assert lanes_used == 0
    # a message longer than the destination: both sides learn it (the payload, partly in the lane, is dropped)
    for n, capn in ((100, 10), (6 * piece, 2 * piece)):
        tag += 1
        try:
            if rank == 0:
                host_a[:n] = 7
                comm.send(host_a[:n], n, xmpi.U8, peer, tag)
            else:
                comm.recv(host_b[:capn], capn, xmpi.U8, peer, tag)
            raise AssertionError("truncation must be reported on both sides")
        except xmpi.XmpiError as e:
            assert e.code == xmpi.ERR_TRUNCATE, e
    # Send without waiting: a payload that fits the lane's ring has left the caller's slice when the call returns
    tag += 1
    k = 3 * piece
    want = oracle.fill(k, xmpi.U8, xmpi.PAT_UNIFORM, 900)
    if rank == 0:
        host_a[:k] = want
        comm.send_nowait(host_a[:k], k, xmpi.U8, peer, tag)
        host_a[:k] = 0
        comm.wait(peer, tag)
    else:
        import time
        time.sleep(0.3)
        comm.recv(host_b[:k], k, xmpi.U8, peer, tag)
        assert host_b[:k].tobytes() == want.tobytes()
    # helloworld's shape: several messages of one pair in flight at once (one mail entry and one lane each), any order
    errs = []

    def one(t, n):
        try:
            w = oracle.fill(n, xmpi.U8, xmpi.PAT_UNIFORM, 1000 + t)
            if rank == 0:
                comm.send(w.copy(), n, xmpi.U8, peer, 2000 + t)
            else:
                h = np.zeros(n, dtype=np.uint8)
                comm.recv(h, n, xmpi.U8, peer, 2000 + (3 - t))
                assert h.tobytes() == oracle.fill(n, xmpi.U8, xmpi.PAT_UNIFORM, 1000 + (3 - t)).tobytes()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    sizes = [5 * piece + 1, 17, piece, 2 * piece + 9]
    ths = [threading.Thread(target=one, args=(t, sizes[t] if rank == 0 else sizes[3 - t])) for t in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    # the ping-pong of bounce.go with slices: what a round trip costs (reported, not asserted: boxes differ)
    import time
    small = np.zeros(8, dtype=np.uint8)
    for w in range(220):
        if w == 20:
            t0 = time.perf_counter()
        if rank == 0:
            comm.send(small, 8, xmpi.U8, peer, 5)
            comm.recv(small, 8, xmpi.U8, peer, 5)
        else:
            comm.recv(small, 8, xmpi.U8, peer, 5)
            comm.send(small, 8, xmpi.U8, peer, 5)
    if rank == 0:
        print(f"host slices, 8 bytes, round trip through ctypes: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us (lanes {'on' if lanes_on else 'off'})")
    # the mixes: a slice on one side, HBM on the other (round trip = one message each way, sender's kind -> receiver's kind)
    for nb in (8, 1 << 20):
        hbuf = np.zeros(nb, dtype=np.uint8)
        for mine, theirs in (("host", "dev"), ("dev", "host")):
            a_buf = hbuf if (mine if rank == 0 else theirs) == "host" else dev_a
            for w in range(120):
                if w == 20:
                    t0 = time.perf_counter()
                if rank == 0:
                    comm.send(a_buf, nb, xmpi.U8, peer, 6)
                    comm.recv(a_buf, nb, xmpi.U8, peer, 6)
                else:
                    comm.recv(a_buf, nb, xmpi.U8, peer, 6)
                    comm.send(a_buf, nb, xmpi.U8, peer, 6)
            if rank == 0:
                print(f"host slices, {nb} bytes, rank 0 {mine} <-> rank 1 {theirs}: {(time.perf_counter() - t0) / 100 * 1e6:.1f} us per round trip")
    # ... and an allreduce of slices (collectives.go hands them to xmpi_allreduce as they are: stand-ins in HBM, the fold on the GPU)
    hb0 = comm.get_param("host_bounce_calls")
    for n in (2, 256, 65536, 65537, 262144):
        x = oracle.fill(n, xmpi.F32, xmpi.PAT_SIGNED, 40 + rank)
        out = np.zeros(n, dtype=np.float32)
        comm.allreduce(x, out, n, xmpi.F32, xmpi.SUM, xmpi.ALGO_AUTO)
        want = oracle.reduce_ranks([oracle.fill(n, xmpi.F32, xmpi.PAT_SIGNED, 40 + r) for r in range(size)], xmpi.F32, xmpi.SUM)
        assert out.tobytes() == want.tobytes(), f"allreduce of host slices, {n} elements"
        t0 = time.perf_counter()
        for _ in range(50):
            comm.allreduce(x, out, n, xmpi.F32, xmpi.SUM, xmpi.ALGO_AUTO)
        if rank == 0:
            print(f"allreduce of host slices, {n * 4} bytes: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us per call")
    if comm.get_param("dsync") == 1:  # up to 256 KiB a slice goes in and comes out through pinned memory the GPU touches itself
        assert comm.get_param("host_bounce_calls") - hb0 == 3 * 51 * 2, comm.get_param("host_bounce_calls") - hb0
    # in place, a reduce to one root, a broadcast, an allgather of slices
    x = oracle.fill(1000, xmpi.I64, xmpi.PAT_SIGNED, 70 + rank)
    ins = [oracle.fill(1000, xmpi.I64, xmpi.PAT_SIGNED, 70 + r) for r in range(size)]
    y = x.copy()
    comm.allreduce(y, y, 1000, xmpi.I64, xmpi.SUM, xmpi.ALGO_AUTO)
    assert y.tobytes() == oracle.reduce_ranks(ins, xmpi.I64, xmpi.SUM).tobytes()
    out = np.zeros(1000, dtype=np.int64)
    comm.reduce(x, out, 1000, xmpi.I64, xmpi.MAX, 1)
    if rank == 1:
        assert out.tobytes() == oracle.reduce_ranks(ins, xmpi.I64, xmpi.MAX).tobytes()
    else:
        assert not out.any(), "a rank that is not the root keeps its receive slice"
    b = ins[1].copy() if rank == 1 else np.zeros(1000, dtype=np.int64)
    comm.bcast(b, 1000, xmpi.I64, 1)
    assert b.tobytes() == ins[1].tobytes()
    g = np.zeros(1000 * size, dtype=np.int64)
    comm.allgather(x, g, 1000, xmpi.I64)
    assert g.tobytes() == np.concatenate(ins).tobytes()
    dev_a.free()
    dev_b.free()
    comm.barrier()


_COLOUR_SEEN = {}


def sc_heap_colours(comm, args):
    """heap.cpp: large blocks of a heap that serves several ranks (threads of one process) start in different 4 KiB slots of
    a 64 KiB frame -- 16 consecutive large blocks in 16 different slots (bits 12..15 of the address) -- so that the buffers of
    one fold do not meet in the same HBM banks; a heap with one rank leaves blocks where the free range starts.  Contents
    survive, blocks do not overlap, freeing gives everything back."""
    rank, size = comm.rank(), comm.size()
    n = 16 // size
    bufs = [comm.alloc(1 << 20) for _ in range(n)]
    for i, b in enumerate(bufs):
        comm.memset(b, 16 * rank + i, 1 << 20)
    _COLOUR_SEEN[rank] = [b.ptr for b in bufs]
    comm.barrier()
    ptrs = sorted(p for r in range(size) for p in _COLOUR_SEEN.get(r, [])) if args.get("threads") else sorted(_COLOUR_SEEN[rank])
    for a, b in zip(ptrs, ptrs[1:]):
        assert a + (1 << 20) <= b, "blocks overlap"
    slots = [(p >> 12) & 15 for p in ptrs]
    if args.get("threads"):
        assert len(ptrs) == 16 and sorted(slots) == list(range(16)), slots
    else:
        assert len(set(slots)) == 1, f"one rank per process: blocks are not coloured {slots}"
    for i, b in enumerate(bufs):
        back = b.download(np.uint8, 1 << 20)
        assert back[0] == 16 * rank + i and back[-1] == 16 * rank + i and int(back.min()) == int(back.max())
    comm.barrier()
    for b in bufs:
        b.free()


def sc_helloworld(comm, args):
    """examples/helloworld/helloworld.go:53-81: every rank concurrently sends a string to every rank
    (itself included) with tag 0 and receives one from every rank."""
    rank, size = comm.rank(), comm.size()
    errors = []

    def text(dst, src):
        return (f"\"I'm just node {src} talking to myself\"" if dst == src else f"\"Hello node {dst}, I'm node {src}\"").encode()

    def sender(i):
        try:
            s = np.frombuffer(text(i, rank), dtype=np.uint8).copy()
            comm.send(s, s.size, xmpi.U8, i, 0)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    def receiver(i):
        try:
            buf = np.zeros(256, dtype=np.uint8)
            got = comm.recv(buf, 256, xmpi.U8, i, 0)
            assert buf[:got].tobytes() == text(rank, i), (buf[:got].tobytes(), text(rank, i))
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    ts = [threading.Thread(target=sender, args=(i,)) for i in range(size)]
    ts += [threading.Thread(target=receiver, args=(i,)) for i in range(size)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


def sc_p2p_semantics(comm, args):
    rank, size = comm.rank(), comm.size()
    assert size >= 2
    if rank > 1:
        comm.barrier()
        return
    peer = 1 - rank
    n = 300000  # spans several window slots only with small slots; still multi-piece-safe
    a = comm.alloc(n * 4)
    b = comm.alloc(n * 4)
    # concurrent sends with distinct tags, received in the opposite order (mpi.go:121-125)
    errors = []
    if rank == 0:
        comm.fill(a, n, xmpi.F32, xmpi.PAT_UNIFORM, 11)
        comm.fill(b, n, xmpi.I32, xmpi.PAT_UNIFORM, 12)

        def snd(buf, dt, tag):
            try:
                comm.send(buf, n, dt, peer, tag)
            except Exception as e:  # noqa: BLE001
                errors.append(e)

        ts = [threading.Thread(target=snd, args=(a, xmpi.F32, 7)), threading.Thread(target=snd, args=(b, xmpi.I32, 8))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errors, errors
    else:
        comm.recv(b, n, xmpi.I32, peer, 8)
        comm.recv(a, n, xmpi.F32, peer, 7)
        assert a.download(np.float32, n).tobytes() == oracle.fill(n, xmpi.F32, 0, 11).tobytes()
        assert b.download(np.int32, n).tobytes() == oracle.fill(n, xmpi.I32, 0, 12).tobytes()
    # receive buffer larger than the message: count reported; smaller: XMPI_ERR_TRUNCATE on both sides
    if rank == 0:
        comm.send(a, 10, xmpi.F32, peer, 1)
        try:
            comm.send(a, 100, xmpi.F32, peer, 2)
            raise AssertionError("truncated send must fail")
        except xmpi.XmpiError as e:
            assert e.code == xmpi.ERR_TRUNCATE
    else:
        assert comm.recv(b, 1000, xmpi.F32, peer, 1) == 10
        try:
            comm.recv(b, 50, xmpi.F32, peer, 2)
            raise AssertionError("truncated receive must fail")
        except xmpi.XmpiError as e:
            assert e.code == xmpi.ERR_TRUNCATE
    # duplicate {peer, tag} among concurrent calls -> TagExists (the reference panics, network.go:469)
    if rank == 0:
        got = []

        def blocked():
            try:
                comm.send(a, 4, xmpi.F32, peer, 99)
            except Exception as e:  # noqa: BLE001
                got.append(e)

        t = threading.Thread(target=blocked)
        t.start()
        import time
        time.sleep(0.3)  # the first send is now waiting for its receiver
        try:
            comm.send(a, 4, xmpi.F32, peer, 99)
            raise AssertionError("duplicate tag must fail")
        except xmpi.XmpiError as e:
            assert e.code == xmpi.ERR_TAG_EXISTS
        comm.send(a, 1, xmpi.U8, peer, 100)  # release the peer
        t.join()
        assert not got, got
    else:
        comm.recv(b, 1, xmpi.U8, peer, 100)
        comm.recv(b, 4, xmpi.F32, peer, 99)
    # host-resident payloads on either side
    if rank == 0:
        h = oracle.fill(5000, xmpi.F64, 0, 3)
        comm.send(h, 5000, xmpi.F64, peer, 5)
    else:
        h = np.zeros(5000, dtype=np.float64)
        comm.recv(h, 5000, xmpi.F64, peer, 5)
        assert h.tobytes() == oracle.fill(5000, xmpi.F64, 0, 3).tobytes()
    # Send without waiting for the receiver + Wait (the reference author's sketched API, mpi.go:132-152):
    # the send returns while the receiver is still busy elsewhere, the buffer may be overwritten at once,
    # the pair {dest, tag} stays taken until wait() has collected the confirmation
    import time
    k = 30000  # 120 KB: fits the entry's two 64 KiB slots of test_p2p_semantics, so nothing has to be drained first
    want = oracle.fill(k, xmpi.F32, xmpi.PAT_SIGNED, 21)
    if rank == 0:
        comm.fill(a, k, xmpi.F32, xmpi.PAT_SIGNED, 21)
        t0 = time.perf_counter()
        comm.send_nowait(a, k, xmpi.F32, peer, 31)
        assert time.perf_counter() - t0 < 0.4, "send_nowait waited for the receiver"
        comm.memset(a, 0xFF, k * 4)  # the payload no longer lives here
        try:
            comm.send_nowait(a, 1, xmpi.F32, peer, 31)
            raise AssertionError("the pair {dest, tag} must stay reserved until wait()")
        except xmpi.XmpiError as e:
            assert e.code == xmpi.ERR_TAG_EXISTS
        hsrc = oracle.fill(1000, xmpi.I64, 0, 22)
        comm.send_nowait(hsrc, 1000, xmpi.I64, peer, 32)  # host payload
        hsrc[:] = 0
        comm.wait(peer, 31)
        comm.wait(peer, 32)
        try:
            comm.wait(peer, 31)
            raise AssertionError("nothing is outstanding any more")
        except xmpi.XmpiError as e:
            assert e.code == xmpi.ERR_ARG
        comm.send_nowait(a, 0, xmpi.F32, peer, 31)  # the pair is free again; empty message
        comm.wait(peer, 31)
    else:
        time.sleep(0.5)
        assert comm.recv(b, k, xmpi.F32, peer, 31) == k
        assert b.download(np.float32, k).tobytes() == want.tobytes()
        hd = np.zeros(1000, dtype=np.int64)
        comm.recv(hd, 1000, xmpi.I64, peer, 32)
        assert hd.tobytes() == oracle.fill(1000, xmpi.I64, 0, 22).tobytes()
        assert comm.recv(b, 0, xmpi.F32, peer, 31) == 0
    a.free()
    b.free()
    comm.barrier()


def sc_fullsize(comm, args):
    """BASELINE.json configs at full size, verified on the device (size-independent properties)."""
    rank, size = comm.rank(), comm.size()
    which = args["which"]
    if which == "cfg3":  # allgather int64 16 MiB / rank, bit-exact positions
        count = 2097152
        send = comm.alloc(count * 8)
        recv = comm.alloc(count * 8 * size)
        want = comm.alloc(count * 8 * size)
        comm.fill(send, count, xmpi.I64, xmpi.PAT_INDEX, rank)
        for r in range(size):
            comm.fill(want.at(r * count * 8), count, xmpi.I64, xmpi.PAT_INDEX, r)
        for algo in (xmpi.ALGO_RING, xmpi.ALGO_RING_PUSH, xmpi.ALGO_DIRECT, xmpi.ALGO_ZCOPY):
            comm.memset(recv, 0, count * 8 * size)
            comm.allgather(send, recv, count, xmpi.I64, algo)
            assert comm.count_mismatch(recv, want, count * 8 * size) == 0, f"cfg3 algo {algo}"
        # spot-check against the CPU oracle as well: x[i] = (r << 40) | i
        for r in (0, size - 1):
            got = recv.download(np.int64, 1024, byte_offset=(r * count + count - 1024) * 8)
            assert np.array_equal(got, (np.int64(r) << 40) | np.arange(count - 1024, count, dtype=np.int64))
        return
    if which == "cfg4":  # allreduce-sum f32 256 MiB
        count, dtype = args.get("count", 67108864), xmpi.F32
    else:  # cfg5: fp16, exactly summable inputs (k/64) -> every algorithm bit-identical
        count = args.get("count", 536870912)
        if count == "auto":  # the full 1 GiB per rank when the GPU(s) have room for every rank's three buffers, else 256 MiB
            sharers = max(1, comm.get_param("dsync_sharers")) if comm.get_param("dsync") == 1 else size
            need_mib = 3 * 1024 * sharers + 8192
            count = 536870912 if comm.get_param("hbm_free_mib") >= need_mib else 134217728
            # every rank must take the same size: agree on the smallest
            agreed = np.array([count], dtype=np.int64)
            out = np.zeros(1, dtype=np.int64)
            comm.allreduce(agreed, out, 1, xmpi.I64, xmpi.MIN, xmpi.ALGO_DIRECT)
            count = int(out[0])
        dtype = xmpi.F16
    es = xmpi.DTYPE_SIZE[dtype]
    send = comm.alloc(count * es)
    ref = comm.alloc(count * es)
    out = comm.alloc(count * es)
    seed0 = 1000 if which == "cfg4" else 2000
    comm.fill(send, count, dtype, xmpi.PAT_UNIFORM, seed0 + rank)
    # rank-order result (DIRECT) is the on-device reference; check windows of it against the CPU oracle
    comm.allreduce(send, ref, count, dtype, xmpi.SUM, xmpi.ALGO_DIRECT)
    for off in (0, count // 3, count - 65536):
        off = off // 8 * 8
        mine = send.download(xmpi.NUMPY_DTYPE[dtype], 65536, byte_offset=off * es)
        assert mine.tobytes() == oracle.fill_range(off, 65536, dtype, xmpi.PAT_UNIFORM, seed0 + rank).tobytes(), \
            "device fill window differs from the oracle's"
    # The WHOLE result against the CPU oracle (oracle_check_allreduce regenerates every rank's input block by block and
    # folds in rank order): every chunk, every chunk boundary.  f32 (the headline, 64 Mi elements): every rank checks its
    # whole buffer.  fp16 at 1 GiB (512 Mi elements x 8 inputs of software half conversions): rank r checks the r-th
    # eighth plus the 4 Ki elements either side of its ends, and all ranks' buffers are shown identical by their checksums.
    if dtype == xmpi.F32:
        lo, hi = 0, count
    else:
        per = (count + size - 1) // size
        lo, hi = max(0, rank * per - 4096), min(count, (rank + 1) * per + 4096)
    step = 1 << 25  # 32 Mi elements per download
    for start in range(lo, hi, step):
        n = min(step, hi - start)
        got = ref.download(xmpi.NUMPY_DTYPE[dtype], n, byte_offset=start * es)
        bad, first = oracle.check_allreduce(got, start, dtype, xmpi.PAT_UNIFORM, seed0, size, xmpi.SUM)
        assert bad == 0, f"{which}: DIRECT differs from the rank-order oracle in {bad} elements of [{start}, {start + n}), first at {first}"
        del got
    sums = np.zeros(size, dtype=np.int64)
    mine_sum = np.array([comm.checksum(ref, count * es) & 0x7FFFFFFFFFFFFFFF], dtype=np.int64)
    comm.allgather(mine_sum, sums, 1, xmpi.I64, xmpi.ALGO_DIRECT)
    assert np.all(sums == sums[0]), f"{which}: the ranks' results differ from each other: {sums}"
    pulled = {}
    for algo in (xmpi.ALGO_RING, xmpi.ALGO_RHD, xmpi.ALGO_RING_PUSH, xmpi.ALGO_RHD_PUSH):
        comm.memset(out, 0, count * es)
        comm.allreduce(send, out, count, dtype, xmpi.SUM, algo)
        # the push form of a schedule at full size: the same bits as its pull form (checksums of the whole buffer)
        if algo in (xmpi.ALGO_RING, xmpi.ALGO_RHD):
            pulled[algo] = comm.checksum(out, count * es)
        elif comm.get_param("dsync") == 1:
            base = xmpi.ALGO_RING if algo == xmpi.ALGO_RING_PUSH else xmpi.ALGO_RHD
            assert comm.checksum(out, count * es) == pulled[base], f"{which} algo {algo}: not the bits of algo {base}"
        if dtype == xmpi.F16:
            assert comm.count_mismatch(out, ref, count * es) == 0, f"{which} algo {algo}: not bit-identical"
        else:
            # BASELINE.md's rule, per element, over all 67 108 864 of them, on the device: the inputs are
            # non-negative, so the rank-order sum ref_i IS sum_r |x_r,i| and the rule reads
            # |out_i - ref_i| <= 1e-6 * ref_i  (two summation orders of 8 floats differ by <= 2*7*2^-24 = 8.3e-7)
            rel = comm.diff_rel(out, ref, count, dtype)
            assert rel <= 1e-6, f"{which} algo {algo}: max_i |delta_i| / sum_r|x_r,i| = {rel}"
            mx, sb, nn = comm.diff_stats(out, ref, count, dtype)
            assert nn == 0 and mx <= 1e-6 * size
            s1, s2 = comm.checksum(out, count * es), comm.checksum(ref, count * es)
            assert s1 != 0 and s2 != 0
    # idempotence of the data path: the same call again gives the same bits
    comm.allreduce(send, out, count, dtype, xmpi.SUM, xmpi.ALGO_DIRECT)
    assert comm.count_mismatch(out, ref, count * es) == 0
    # zero-copy folds in rank order too: bit-identical to DIRECT (hence to the oracle), f32 included;
    # AUTO picks it for registered buffers; in place it must not read what a peer already overwrote
    for algo in (xmpi.ALGO_ZCOPY, xmpi.ALGO_AUTO):
        comm.memset(out, 0, count * es)
        comm.allreduce(send, out, count, dtype, xmpi.SUM, algo)
        assert comm.count_mismatch(out, ref, count * es) == 0, f"{which} algo {algo}: not bit-identical to rank order"
    comm.copy_local(out, send, count * es)
    comm.allreduce(out, out, count, dtype, xmpi.SUM, xmpi.ALGO_ZCOPY)
    assert comm.count_mismatch(out, ref, count * es) == 0, f"{which}: in-place zero-copy differs"
    if comm.get_param("dsync") == 1:
        # the split form at full size: every launch's meet / done blocks reached every XCD (checked on the device), and its
        # system-scope data kernel (the form that needs no such coverage) gives the same bits
        assert comm.get_param("xcd_short") == 0
        comm.set_param("dsync_split_bytes", 1)
        comm.set_param("body_sys", 1)
        comm.memset(out, 0, count * es)
        comm.allreduce(send, out, count, dtype, xmpi.SUM, xmpi.ALGO_ZCOPY)
        assert comm.count_mismatch(out, ref, count * es) == 0, f"{which}: split form with the system-scope data kernel differs"
        comm.set_param("body_sys", 0)
        comm.set_param("dsync_split_bytes", 4 << 20)
        # binary-tree reduce (the stepped kernel) to the last rank, and LL's refusal of a long message (it names the fold then)
        root = size - 1
        comm.memset(out, 0, count * es)
        comm.reduce(send, out if rank == root else None, count, dtype, xmpi.SUM, root, xmpi.ALGO_TREE)
        if rank == root:
            if dtype == xmpi.F16:
                assert comm.count_mismatch(out, ref, count * es) == 0, f"{which}: tree reduce not bit-identical"
            else:
                assert comm.diff_rel(out, ref, count, dtype) <= 1e-6, f"{which}: tree reduce beyond 1e-6 * sum|x|"
        comm.memset(out, 0, count * es)
        comm.allreduce(send, out, count, dtype, xmpi.SUM, xmpi.ALGO_LL)
        assert comm.count_mismatch(out, ref, count * es) == 0, f"{which}: XMPI_ALGO_LL above its limit is not the fold"


def np_hash(seed: int, idx: np.ndarray) -> np.ndarray:
    """numpy restatement of oracle_hash (cross-checked against the C one in tests/test_oracle.py)"""
    u = np.uint64
    with np.errstate(over="ignore"):
        z = u(seed) * u(0xD1342543DE82EF95) + idx.astype(np.uint64) * u(0x9E3779B97F4A7C15) + u(0x2545F4914F6CDD1D)
        z = (z ^ (z >> u(30))) * u(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> u(27))) * u(0x94D049BB133111EB)
        return z ^ (z >> u(31))


def oracle_fill_window(dtype, seed, start, n):
    """elements [start, start+n) of oracle_fill(..., PAT_UNIFORM, seed) without generating the prefix"""
    h = np_hash(seed, np.arange(start, start + n, dtype=np.uint64))
    if dtype == xmpi.F32:
        return ((h >> np.uint64(40)).astype(np.float64) * 2.0 ** -24).astype(np.float32)
    if dtype == xmpi.F16:
        return ((h & np.uint64(63)).astype(np.float64) / 64.0).astype(np.float16)
    raise NotImplementedError


def sc_stream_ordered(comm, args):
    """xmpi_*_on_stream: collectives ENQUEUED on a HIP stream, several back to back without any host wait in between;
    each consumes what the previous one produced, so a result that matches the oracle proves the stream order.  With
    one process per rank the ranks meet inside the kernels (dsync.cpp); ranks hosted by threads meet on the host."""
    rank, size = comm.rank(), comm.size()
    dsync = comm.get_param("dsync") == 1
    st = comm.stream_create()
    launches0 = comm.get_param("dsync_launches")
    for count in args.get("counts", [1, 1000, 4099, (1 << 18) + 3]):
        # int64: x -> allreduce -> allreduce (in place) -> allreduce: exact whatever the order, wraps like Go
        a, b = comm.alloc(count * 8), comm.alloc(count * 8)
        comm.fill(a, count, xmpi.I64, xmpi.PAT_UNIFORM, 600 + rank)
        comm.sync()
        comm.allreduce_on_stream(a, b, count, xmpi.I64, xmpi.SUM, st)
        comm.allreduce_on_stream(b, b, count, xmpi.I64, xmpi.SUM, st)
        comm.allreduce_on_stream(b, a, count, xmpi.I64, xmpi.SUM, st)
        comm.stream_sync(st)
        ins = [oracle.fill(count, xmpi.I64, xmpi.PAT_UNIFORM, 600 + r) for r in range(size)]
        s1 = oracle.reduce_ranks(ins, xmpi.I64, xmpi.SUM)
        with np.errstate(over="ignore"):
            want = (s1.astype(np.uint64) * np.uint64(size) * np.uint64(size)).astype(np.int64)
        got = a.download(np.int64, count)
        assert got.tobytes() == want.tobytes(), f"chained allreduce on a stream, n={count}"
        # f32 in rank order: bit-identical to the oracle; then a broadcast of the result of the last rank's
        # private modification, an allgather of the broadcast, a reduce of the gathered blocks
        f, g = comm.alloc(count * 4), comm.alloc(count * 4)
        gath = comm.alloc(count * 4 * size)
        red = comm.alloc(count * 4 * size)
        comm.fill(f, count, xmpi.F32, xmpi.PAT_SIGNED, 700 + rank)
        comm.sync()
        root = size - 1
        comm.allreduce_on_stream(f, g, count, xmpi.F32, xmpi.SUM, st)
        comm.bcast_on_stream(f, count, xmpi.F32, root, st)           # everybody's f = the root's input
        comm.allgather_on_stream(g, gath, count, xmpi.F32, st)      # size copies of the sum
        comm.reduce_on_stream(gath, red if rank == 0 else None, count * size, xmpi.F32, xmpi.MAX, 0, st)
        comm.stream_sync(st)
        fin = [oracle.fill(count, xmpi.F32, xmpi.PAT_SIGNED, 700 + r) for r in range(size)]
        want_sum = oracle.reduce_ranks(fin, xmpi.F32, xmpi.SUM)
        assert g.download(np.float32, count).tobytes() == want_sum.tobytes(), f"allreduce_on_stream f32 n={count}"
        assert f.download(np.float32, count).tobytes() == fin[root].tobytes(), f"bcast_on_stream n={count}"
        assert gath.download(np.float32, count * size).tobytes() == np.tile(want_sum, size).tobytes(), "allgather_on_stream"
        if rank == 0:
            assert red.download(np.float32, count * size).tobytes() == np.tile(want_sum, size).tobytes(), "reduce_on_stream"
        for x in (a, b, f, g, gath, red):
            x.free()
    # the communicator's own stream (stream = None), then a blocking collective right behind it
    n = 50021
    a, b = comm.alloc(n * 4), comm.alloc(n * 4)
    comm.fill(a, n, xmpi.F32, xmpi.PAT_UNIFORM, 800 + rank)
    comm.allreduce_on_stream(a, b, n, xmpi.F32, xmpi.SUM, None)
    comm.allreduce(b, a, n, xmpi.F32, xmpi.MAX, xmpi.ALGO_AUTO)
    want = oracle.reduce_ranks([oracle.fill(n, xmpi.F32, xmpi.PAT_UNIFORM, 800 + r) for r in range(size)], xmpi.F32, xmpi.SUM)
    assert a.download(np.float32, n).tobytes() == want.tobytes()
    # the same allreduce several times inside one call: with ranks that meet on the device the steps are enqueued
    # back to back and waited for once (in place: every step folds the previous result, exact in int64)
    comm.fill(a, n // 2, xmpi.I64, xmpi.PAT_CONST, rank)  # x_r = r + 1
    comm.allreduce_repeat(a, a, n // 2, xmpi.I64, xmpi.SUM, xmpi.ALGO_AUTO, 4)
    want_rep = np.full(n // 2, (size * (size + 1) // 2) * size ** 3, dtype=np.int64)
    assert a.download(np.int64, n // 2).tobytes() == want_rep.tobytes(), "allreduce_repeat"
    a.free()
    b.free()
    if dsync:
        assert comm.get_param("dsync_launches") > launches0, "the device-synchronised kernels did not run"
        # hipGraph: two allreduces captured once, replayed five times (in place on int64: every replay multiplies by
        # size^2 -- exact, wraps like Go); then ordinary calls again: the epochs are counted on the device, so
        # captured and ordinary launches interleave
        m = 20011
        g1, g2 = comm.alloc(m * 8), comm.alloc(m * 8)
        comm.fill(g1, m, xmpi.I64, xmpi.PAT_CONST, 0)  # all ones
        comm.allreduce_on_stream(g1, g2, m, xmpi.I64, xmpi.SUM, st)  # warm: everything mapped; g2 = size
        comm.allreduce_on_stream(g2, g1, m, xmpi.I64, xmpi.SUM, st)  # g1 = size^2
        comm.stream_sync(st)
        comm.graph_begin(st)
        comm.allreduce_on_stream(g1, g2, m, xmpi.I64, xmpi.SUM, st)
        comm.allreduce_on_stream(g2, g1, m, xmpi.I64, xmpi.SUM, st)
        graph = comm.graph_end(st)
        for _ in range(5):
            comm.graph_launch(graph, st)
        comm.allreduce_on_stream(g1, g2, m, xmpi.I64, xmpi.SUM, st)  # an ordinary launch right behind the replays
        comm.stream_sync(st)
        want_g1 = np.uint64(size) ** np.uint64(12)
        with np.errstate(over="ignore"):
            want_g2 = want_g1 * np.uint64(size)
        assert np.all(g1.download(np.int64, m).view(np.uint64) == want_g1), "graph replays"
        assert np.all(g2.download(np.int64, m).view(np.uint64) == want_g2), "ordinary launch after graph replays"
        comm.graph_destroy(graph)
        g1.free()
        g2.free()
        # device memory of another allocator, never registered: copied through a registered block on the same stream
        import ctypes
        hip = _hip_runtime()
        n = 30011
        src, dst = ctypes.c_void_p(0), ctypes.c_void_p(0)
        comm.sync()
        assert hip.hipMalloc(ctypes.byref(src), ctypes.c_size_t(n * 4)) == 0
        assert hip.hipMalloc(ctypes.byref(dst), ctypes.c_size_t(n * 4)) == 0
        comm.fill(src.value, n, xmpi.F32, xmpi.PAT_SIGNED, 900 + rank)
        b0 = comm.get_param("dsync_bounced")
        comm.allreduce_on_stream(src.value, dst.value, n, xmpi.F32, xmpi.SUM, st)
        comm.allreduce_on_stream(dst.value, dst.value, n, xmpi.F32, xmpi.MAX, st)
        comm.stream_sync(st)
        out = np.empty(n, dtype=np.float32)
        xmpi._check(xmpi.lib().xmpi_memcpy(comm.handle, out.ctypes.data, dst.value, n * 4), "download")
        want = oracle.reduce_ranks([oracle.fill(n, xmpi.F32, xmpi.PAT_SIGNED, 900 + r) for r in range(size)], xmpi.F32, xmpi.SUM)
        assert out.tobytes() == want.tobytes(), "unregistered buffers on a stream"
        assert comm.get_param("dsync_bounced") == b0 + 3
        comm.barrier()
        assert hip.hipFree(src) == 0 and hip.hipFree(dst) == 0
    comm.stream_destroy(st)


def sc_lifecycle_stress(comm, args):
    """init -> a few collectives with the copy kernel as transport (batched pushes) -> finalize, over and over, each
    rank leaving at its own pace: nothing may still write through a mapping that a peer has already released"""
    import random
    import time
    rank, size = comm.rank(), comm.size()
    rnd = random.Random(1234 + rank)
    base = args.get("key", "life")
    t_start = time.time()
    for it in range(args.get("iters", 50)):
        c2 = xmpi.Comm(rank, size, comm.device(), f"{base}-{it}")
        c2.set_param("copy_engine", 1)
        c2.set_param("batch_copies", 1)
        for algo in (xmpi.ALGO_RING, xmpi.ALGO_DIRECT, xmpi.ALGO_AUTO):
            allreduce_case(c2, xmpi.F32, 4099 + it, algo)
        if it % 5 == 0:
            allgather_case(c2, xmpi.I64, 1000 + it, xmpi.ALGO_RING)
        time.sleep(rnd.random() * 0.004)  # ranks reach finalize at different times
        c2.finalize()
        if args.get("verbose") and rank == 0 and it % 5 == 0:
            print(f"lifetime {it}: {comm.get_param('hbm_free_mib')} MiB of HBM free, {time.time() - t_start:.2f} s so far", flush=True)


def sc_sched(comm, args):
    """The stepped kernels (sched.hip): ring allreduce / allgather, recursive halving + doubling, binary-tree broadcast as
    ONE kernel per rank -- every dtype and operator, in place, odd alignments, one to many workers and ring channels."""
    rank, size = comm.rank(), comm.size()
    if comm.get_param("dsync") != 1:
        return  # ranks that meet on the host run these names as host-driven step tables (other scenarios)
    quick = bool(args.get("quick"))  # (tests/devsim with many virtual devices: the multi-MiB cases are the GPU suite's)
    l0 = comm.get_param("dsync_sched_launches")
    for channels, grid in args.get("shapes", [(0, 0), (1, 1), (2, 3), (0, 8)]):
        comm.set_param("sched_channels", channels)
        comm.set_param("sched_grid", grid)
        for dtype in (xmpi.F32, xmpi.I64, xmpi.F16, xmpi.BF16, xmpi.U8, xmpi.F64, xmpi.I32):
            for algo in (xmpi.ALGO_RING, xmpi.ALGO_RHD, xmpi.ALGO_RING_PUSH, xmpi.ALGO_RHD_PUSH):
                for count in args.get("counts", [1, 17, 4099, 65536 + 5]):
                    allreduce_case(comm, dtype, count, algo)
        # the push form of a schedule folds the same operands in the same association as its pull form: the same bits, on data
        # where another order shows (f32 / f16 of both signs), out of place and in place
        for pull, push in ((xmpi.ALGO_RING, xmpi.ALGO_RING_PUSH), (xmpi.ALGO_RHD, xmpi.ALGO_RHD_PUSH)):
            for dtype, count in ((xmpi.F32, 100003), (xmpi.F16, 70001)) + (() if quick else ((xmpi.F32, (1 << 20) + 1),)):
                for inplace in (False, True):
                    same_bits_case(comm, dtype, count, pull, push, inplace)
        # several of them enqueued back to back, nobody waiting in between: the push forms' landing block is used again by the next
        # collective before the host has seen the last one end
        for algo in (xmpi.ALGO_RING_PUSH, xmpi.ALGO_RHD_PUSH):
            n = 200003
            send, recv = comm.alloc(n * 4), comm.alloc(n * 4)
            comm.fill(send, n, xmpi.F32, xmpi.PAT_SIGNED, 500 + rank)
            comm.allreduce_repeat(send, recv, n, xmpi.F32, xmpi.SUM, algo, 4)
            ins = [oracle.fill(n, xmpi.F32, xmpi.PAT_SIGNED, 500 + r) for r in range(size)]
            check_reduced(recv.download(np.float32, n), ins, xmpi.F32, xmpi.SUM, size <= 2, f"4 x allreduce algo={algo} enqueued back to back")
            comm.fill(recv, n, xmpi.F32, xmpi.PAT_CONST, rank)  # x_r = r + 1, in place twice: N (N + 1) / 2, then N times that
            comm.allreduce_repeat(recv, recv, n, xmpi.F32, xmpi.SUM, algo, 2)
            assert np.all(recv.download(np.float32, n) == np.float32(size * size * (size + 1) / 2)), f"2 x in-place allreduce algo={algo} back to back"
            send.free()
            recv.free()
        for algo in (xmpi.ALGO_RING, xmpi.ALGO_RHD, xmpi.ALGO_RING_PUSH, xmpi.ALGO_RHD_PUSH):
            if not quick:
                allreduce_case(comm, xmpi.F32, (3 << 20) + 7, algo, pattern=xmpi.PAT_SIGNED)
                allreduce_case(comm, xmpi.F32, (1 << 20) + 1, algo, inplace=True)
            allreduce_case(comm, xmpi.I64, 300007, algo, pattern=xmpi.PAT_UNIFORM, inplace=True, op=xmpi.PROD)
            allreduce_case(comm, xmpi.F16, 70001, algo, misalign=3)
            allreduce_case(comm, xmpi.F32, 40001, algo, pattern=xmpi.PAT_SIGNED, inplace=True, misalign=1)
            for op in (xmpi.MIN, xmpi.MAX):
                allreduce_case(comm, xmpi.F32, 30011, algo, op=op, pattern=xmpi.PAT_SIGNED)
                allreduce_case(comm, xmpi.BF16, 3001, algo, op=op, pattern=xmpi.PAT_SIGNED)
            allreduce_case(comm, xmpi.I64, 4097, algo, pattern=xmpi.PAT_CONST)
        for algo in (xmpi.ALGO_RING, xmpi.ALGO_RING_PUSH):
            for dtype in (xmpi.I64, xmpi.U8, xmpi.F32):
                for count in (1, 5, 1000, 4099) + (() if quick else ((1 << 20) + 3,)):
                    allgather_case(comm, dtype, count, algo)
            allgather_case(comm, xmpi.I64, 70001, algo, inplace=True)
        for piece in (256 << 10, 4096):
            comm.set_param("tree_piece_bytes", piece)
            for root in sorted({0, size - 1, size // 2}):
                for dtype, count in ((xmpi.U8, 1), (xmpi.U8, 37), (xmpi.I64, 4099)) + (() if quick else ((xmpi.F32, (1 << 20) + 9),)):
                    for algo in (xmpi.ALGO_TREE, xmpi.ALGO_TREE_PUSH):
                        es = xmpi.DTYPE_SIZE[dtype]
                        buf = comm.alloc(count * es)
                        comm.fill(buf, count, dtype, xmpi.PAT_UNIFORM, 40 + rank)
                        comm.bcast(buf, count, dtype, root, algo)
                        got = buf.download(xmpi.NUMPY_DTYPE[dtype], count)
                        want = oracle.fill(count, dtype, xmpi.PAT_UNIFORM, 40 + root)
                        assert got.tobytes() == want.tobytes(), f"tree bcast root={root} {xmpi.DTYPE_NAME[dtype]} n={count} piece={piece} algo={algo}"
                        buf.free()
            # the same tree upwards (SCHED_TREE_REDUCE): an inner node folds its children's partial results into its own,
            # piece by piece; a non-root's receive buffer is never written
            for root in sorted({0, size - 1, size // 2}):
                for dtype, count, pat, op in ((xmpi.F32, 100003, xmpi.PAT_SIGNED, xmpi.SUM), (xmpi.I64, 4099, xmpi.PAT_UNIFORM, xmpi.SUM),
                                             (xmpi.F16, 5001, xmpi.PAT_UNIFORM, xmpi.SUM), (xmpi.F64, 1, xmpi.PAT_SIGNED, xmpi.SUM),
                                             (xmpi.BF16, 3001, xmpi.PAT_SIGNED, xmpi.MAX), (xmpi.I32, 70001, xmpi.PAT_SIGNED, xmpi.MIN),
                                             (xmpi.U8, 37, xmpi.PAT_UNIFORM, xmpi.SUM)) + (() if quick else ((xmpi.F32, (1 << 20) + 9, xmpi.PAT_SIGNED, xmpi.SUM),)):
                    exact = size <= 2 or dtype not in FLOATS or op != xmpi.SUM or (dtype == xmpi.F16 and pat == xmpi.PAT_UNIFORM)
                    pull = reduce_case(comm, dtype, count, root, xmpi.ALGO_TREE, op=op, pat=pat, exact=exact, what=f"tree reduce piece={piece}")
                    push = reduce_case(comm, dtype, count, root, xmpi.ALGO_TREE_PUSH, op=op, pat=pat, exact=exact, what=f"tree reduce (push) piece={piece}")
                    assert pull == push, f"tree reduce root={root} {xmpi.DTYPE_NAME[dtype]} n={count}: the push form's bits differ from the pull form's"
        comm.set_param("tree_piece_bytes", 256 << 10)
        # tree reduce in place at the root, and with a misaligned root buffer
        root = size // 2
        n = 50021
        buf = comm.alloc(n * 4 + 4)
        comm.fill(buf.at(4), n, xmpi.F32, xmpi.PAT_SIGNED, 70 + rank)
        for algo in (xmpi.ALGO_TREE, xmpi.ALGO_TREE_PUSH):
            comm.fill(buf.at(4), n, xmpi.F32, xmpi.PAT_SIGNED, 70 + rank)
            comm.reduce(buf.at(4), buf.at(4) if rank == root else None, n, xmpi.F32, xmpi.SUM, root, algo)
            ins = [oracle.fill(n, xmpi.F32, xmpi.PAT_SIGNED, 70 + r) for r in range(size)]
            got = buf.download(np.float32, n, byte_offset=4)
            if rank == root:
                check_reduced(got, ins, xmpi.F32, xmpi.SUM, size <= 2, f"tree reduce in place at the root algo={algo}")
            else:
                assert got.tobytes() == ins[rank].tobytes(), "tree reduce modified a non-root's input"
        buf.free()
    comm.set_param("sched_channels", 0)
    comm.set_param("sched_grid", 0)
    assert comm.get_param("dsync_sched_launches") > l0, "the stepped kernels did not run"
    # a host buffer and a never-registered device buffer take the same kernels through registered stand-ins
    x = oracle.fill(5000, xmpi.I64, xmpi.PAT_UNIFORM, 5 + rank)
    want = oracle.reduce_ranks([oracle.fill(5000, xmpi.I64, xmpi.PAT_UNIFORM, 5 + r) for r in range(size)], xmpi.I64, 0)
    for algo in (xmpi.ALGO_RING, xmpi.ALGO_RING_PUSH, xmpi.ALGO_RHD_PUSH, xmpi.ALGO_ZPUSH):
        out = np.zeros_like(x)
        comm.allreduce(x, out, 5000, xmpi.I64, xmpi.SUM, algo)
        assert out.tobytes() == want.tobytes(), f"host slices, algo {algo}"
        both = x.copy()  # ... and in place (the stand-in is its own receive buffer: the push forms land in the communicator's block)
        comm.allreduce(both, both, 5000, xmpi.I64, xmpi.SUM, algo)
        assert both.tobytes() == want.tobytes(), f"host slice in place, algo {algo}"
    for algo in (xmpi.ALGO_TREE, xmpi.ALGO_TREE_PUSH):  # ... and the tree kernels, both directions
        for root in sorted({0, size - 1}):
            out = np.zeros_like(x)
            comm.reduce(x, out if rank == root else None, 5000, xmpi.I64, xmpi.SUM, root, algo)
            assert rank != root or out.tobytes() == want.tobytes(), f"host slices, tree reduce algo {algo} root {root}"
            y = x.copy()
            comm.bcast(y, 5000, xmpi.I64, root, algo)
            assert y.tobytes() == oracle.fill(5000, xmpi.I64, xmpi.PAT_UNIFORM, 5 + root).tobytes(), f"host slices, tree bcast algo {algo} root {root}"
    # some ranks in place, the others not (a rank's landing block is its own business: it announces one, or none)
    n = 40009
    a, b = comm.alloc(n * 4), comm.alloc(n * 4)
    ins = [oracle.fill(n, xmpi.F32, xmpi.PAT_SIGNED, 950 + r) for r in range(size)]
    for algo in (xmpi.ALGO_RING, xmpi.ALGO_RING_PUSH, xmpi.ALGO_RHD, xmpi.ALGO_RHD_PUSH, xmpi.ALGO_ZPUSH, xmpi.ALGO_ZCOPY):
        comm.fill(a, n, xmpi.F32, xmpi.PAT_SIGNED, 950 + rank)
        out = a if rank % 2 else b
        comm.allreduce(a, out, n, xmpi.F32, xmpi.SUM, algo)
        check_reduced(out.download(np.float32, n), ins, xmpi.F32, xmpi.SUM, size <= 2 or algo in (xmpi.ALGO_ZPUSH, xmpi.ALGO_ZCOPY),
                      f"allreduce algo={algo}, odd ranks in place")
    a.free()
    b.free()
    # hipGraph: the table AUTO follows names a form that lends a block per call (a push form's landing block, the tree reduce's
    # accumulator) -- which a graph cannot hold.  Captured, the call runs the pull form (same bits) / the fold; replays and the
    # eager calls around them interleave (one epoch counter), and the eager call's bits are the replay's.
    st = comm.stream_create()
    n = 30011
    a, b, e = comm.alloc(n * 4), comm.alloc(n * 4), comm.alloc(n * 4)
    ins = [oracle.fill(n, xmpi.F32, xmpi.PAT_SIGNED, 900 + r) for r in range(size)]

    def table(coll, algo):
        for k in range(24):
            comm.set_param(f"tune_algo_{coll}_{k}", algo)
        comm.set_param("tuned", 1)

    for algo, inplace in ((xmpi.ALGO_RING_PUSH, False), (xmpi.ALGO_RING_PUSH, True), (xmpi.ALGO_RHD_PUSH, False), (xmpi.ALGO_RHD_PUSH, True)):
        table(0, algo)
        l1 = comm.get_param("dsync_sched_launches")
        comm.fill(a, n, xmpi.F32, xmpi.PAT_SIGNED, 900 + rank)
        comm.allreduce(a, e, n, xmpi.F32, xmpi.SUM, xmpi.ALGO_AUTO)  # eager: the push form (and everything mapped before the capture)
        comm.allreduce_on_stream(a, b, n, xmpi.F32, xmpi.SUM, st)
        comm.stream_sync(st)
        assert comm.get_param("dsync_sched_launches") == l1 + 2, "AUTO did not follow the table"
        comm.memset(b, 0, n * 4)
        src = b if inplace else a
        if inplace:
            comm.memcpy(b, a, n * 4)
        comm.graph_begin(st)
        comm.allreduce_on_stream(src, b, n, xmpi.F32, xmpi.SUM, st)
        graph = comm.graph_end(st)
        assert comm.get_param("dsync_land_bytes") == 0, "a captured collective borrowed a block"
        comm.graph_launch(graph, st)
        comm.stream_sync(st)
        assert b.download(np.float32, n).tobytes() == e.download(np.float32, n).tobytes(), f"captured algo={algo} inplace={inplace}: not the eager call's bits"
        if not inplace:  # replay, an eager push-form call behind it, replay again
            comm.memset(b, 0, n * 4)
            comm.graph_launch(graph, st)
            comm.allreduce_on_stream(a, e, n, xmpi.F32, xmpi.SUM, st)
            comm.graph_launch(graph, st)
            comm.stream_sync(st)
            assert b.download(np.float32, n).tobytes() == e.download(np.float32, n).tobytes(), f"replays of algo={algo} around an eager call"
            check_reduced(b.download(np.float32, n), ins, xmpi.F32, xmpi.SUM, size <= 2, f"captured allreduce algo={algo}")
        comm.graph_destroy(graph)
    m = n // 2
    for algo in (xmpi.ALGO_TREE, xmpi.ALGO_TREE_PUSH):
        table(3, algo)
        table(2, algo)
        root = size - 1
        comm.fill(a, m, xmpi.I64, xmpi.PAT_UNIFORM, 300 + rank)
        comm.memset(b, 0, m * 8)
        comm.reduce(a, e if rank == root else None, m, xmpi.I64, xmpi.SUM, root, xmpi.ALGO_AUTO)  # eager: the tree
        comm.graph_begin(st)
        comm.reduce_on_stream(a, b if rank == root else None, m, xmpi.I64, xmpi.SUM, root, st)  # captured: the fold (int64: the same bits)
        comm.bcast_on_stream(b, m, xmpi.I64, root, st)  # the tree bcast lends nothing: captured as it is
        graph = comm.graph_end(st)
        for _ in range(2):
            comm.graph_launch(graph, st)
        comm.stream_sync(st)
        want64 = oracle.reduce_ranks([oracle.fill(m, xmpi.I64, xmpi.PAT_UNIFORM, 300 + r) for r in range(size)], xmpi.I64, 0)
        assert b.download(np.int64, m).tobytes() == want64.tobytes(), f"captured reduce + bcast under a table that says algo={algo}"
        if rank == root:
            assert e.download(np.int64, m).tobytes() == want64.tobytes()
        comm.graph_destroy(graph)
    comm.set_param("tuned", 0)
    for coll in (0, 2, 3):
        for k in range(24):
            comm.set_param(f"tune_algo_{coll}_{k}", -1)
    comm.stream_destroy(st)
    for x in (a, b, e):
        x.free()


def bcast_case(comm, dtype, count, root, algo, seed=40, what="bcast"):
    rank = comm.rank()
    es = xmpi.DTYPE_SIZE[dtype]
    buf = comm.alloc(count * es)
    comm.fill(buf, count, dtype, xmpi.PAT_UNIFORM, seed + rank)
    comm.bcast(buf, count, dtype, root, algo)
    got = buf.download(xmpi.NUMPY_DTYPE[dtype], count)
    want = oracle.fill(count, dtype, xmpi.PAT_UNIFORM, seed + root)
    assert got.tobytes() == want.tobytes(), f"{what} root={root} {xmpi.DTYPE_NAME[dtype]} n={count} algo={algo}"
    buf.free()


def reduce_case(comm, dtype, count, root, algo, op=xmpi.SUM, pat=xmpi.PAT_SIGNED, exact=True, what="reduce"):
    """returns the root's result as bytes (None elsewhere)"""
    rank, size = comm.rank(), comm.size()
    es = xmpi.DTYPE_SIZE[dtype]
    send, recv = comm.alloc(count * es), comm.alloc(count * es)
    comm.fill(send, count, dtype, pat, 70 + rank)
    comm.memset(recv, 0x3C, count * es)
    comm.reduce(send, recv if rank == root else None, count, dtype, op, root, algo)
    out = None
    if rank == root:
        ins = [oracle.fill(count, dtype, pat, 70 + r) for r in range(size)]
        got = recv.download(xmpi.NUMPY_DTYPE[dtype], count)
        check_reduced(got, ins, dtype, op, exact, f"{what} root={root} algo={algo}")
        out = got.tobytes()
    else:
        assert recv.download(np.uint8, count * es).tobytes() == bytes([0x3C]) * (count * es), "reduce wrote a non-root's buffer"
    send.free()
    recv.free()
    return out


def same_bits_case(comm, dtype, count, pull, push, inplace):
    """the same inputs through the pull and the push form of a schedule: bit-identical results on every rank"""
    rank = comm.rank()
    es = xmpi.DTYPE_SIZE[dtype]
    out = []
    for algo in (pull, push):
        send = comm.alloc(count * es)
        recv = send if inplace else comm.alloc(count * es)
        comm.fill(send, count, dtype, xmpi.PAT_SIGNED, 300 + rank)
        comm.allreduce(send, recv, count, dtype, xmpi.SUM, algo)
        out.append(recv.download(np.uint8, count * es).tobytes())
        if not inplace:
            recv.free()
        send.free()
    assert out[0] == out[1], f"allreduce {xmpi.DTYPE_NAME[dtype]} n={count} inplace={inplace}: algo {push} and algo {pull} differ in " \
                             f"{sum(a != b for a, b in zip(out[0], out[1]))} bytes"


def _ll_agent_section(comm, maxb):
    """The LL agent (ll.hip ll_agent_kernel): a BLOCKING small LL collective is run by a kernel that lingers behind the one
    before -- no launch.  Every dtype and operation (the fold is chosen at run time there), every collective and root, payloads
    of one line up to the slot limit (16 lines per lane), host slices; the agent gone (its patience over) and launched again;
    blocking calls between enqueued ones without a host wait; the switch off."""
    import time
    rank, size = comm.rank(), comm.size()
    L = xmpi.ALGO_LL
    if comm.get_param("agent_ll") < 1 or comm.get_param("ll_agent_us") <= 0:
        return
    ab = comm.get_param("agent_ll_bytes")
    served = lambda: comm.get_param("dsync_ll_agent")
    comm.set_param("agent_ll", 2)  # (1 and 2 are the same: 2 was "also outside bursts" while a burst rule was tried -- dsync.cpp dsync_ll)

    def reduce_once(dtype, count, op, pattern, seed, root=None, expect_agent=True):
        es = xmpi.DTYPE_SIZE[dtype]
        send, recv = comm.alloc(count * es), comm.alloc(count * es)
        comm.fill(send, count, dtype, pattern, seed + rank)
        comm.memset(recv, 0xA5, count * es)
        comm.sync()  # (the agent takes a call only when the stream it would have been enqueued on is idle)
        g0 = served()
        if root is None:
            comm.allreduce(send, recv, count, dtype, op, L)
        else:
            comm.reduce(send, recv if rank == root else None, count, dtype, op, root, L)
        assert served() == g0 + (1 if expect_agent else 0), f"LL agent: {'not ' if expect_agent else ''}taken ({xmpi.DTYPE_NAME[dtype]} n={count})"
        if root is None or rank == root:
            ins = [oracle.fill(count, dtype, pattern, seed + r) for r in range(size)]
            check_reduced(recv.download(xmpi.NUMPY_DTYPE[dtype], count), ins, dtype, op, True,
                          f"LL agent {'allreduce' if root is None else 'reduce'} {xmpi.DTYPE_NAME[dtype]} n={count} op={op}")
        send.free()
        recv.free()

    k = 0
    for dtype in (xmpi.F32, xmpi.I64, xmpi.F16, xmpi.F64, xmpi.I32, xmpi.U8, xmpi.BF16):
        for op in (xmpi.SUM, xmpi.PROD, xmpi.MIN, xmpi.MAX):
            k += 1
            reduce_once(dtype, (1, 7, 129, ab // xmpi.DTYPE_SIZE[dtype])[k % 4], op, xmpi.PAT_SIGNED, 3000 + 20 * k)
    reduce_once(xmpi.F32, ab // 4 + 1, xmpi.SUM, xmpi.PAT_UNIFORM, 4000, expect_agent=False)  # above the limit: launched
    comm.set_param("agent_ll_bytes", maxb)
    for count in (maxb // 8, maxb // 8 - 1, 257, 256):  # sixteen lines per lane; a ragged last round
        reduce_once(xmpi.I64, count, xmpi.SUM, xmpi.PAT_UNIFORM, 4100 + count)
    reduce_once(xmpi.U8, maxb - 3, xmpi.MAX, xmpi.PAT_UNIFORM, 4200)
    for root in range(size):
        reduce_once(xmpi.F32, 1001, xmpi.SUM, xmpi.PAT_SIGNED, 4300 + root, root=root)
        time.sleep(0.002 if root == rank else 0)  # one rank late, the others' agents wait in the collective; every agent's patience over
        n = 777
        b = comm.alloc(n * 8)
        comm.fill(b, n, xmpi.I64, xmpi.PAT_UNIFORM, 4400 + 10 * root + rank)
        comm.sync()
        g0 = served()
        comm.bcast(b, n, xmpi.I64, root, L)
        assert served() == g0 + 1
        assert b.download(np.int64, n).tobytes() == oracle.fill(n, xmpi.I64, xmpi.PAT_UNIFORM, 4400 + 10 * root + root).tobytes(), "LL agent bcast"
        out = comm.alloc(n * 8 * size)
        comm.sync()
        comm.allgather(b, out, n, xmpi.I64, L)  # (every rank holds the root's block now)
        assert served() == g0 + 2
        assert out.download(np.int64, n * size).tobytes() == np.tile(oracle.fill(n, xmpi.I64, xmpi.PAT_UNIFORM, 4400 + 10 * root + root), size).tobytes(), "LL agent allgather"
        b.free()
        out.free()
    # host slices: in through pinned memory the agent reads itself, out the same way
    for count in (1, 1000):
        x = oracle.fill(count, xmpi.F32, xmpi.PAT_SIGNED, 50 + rank)
        out = np.zeros_like(x)
        g0 = served()
        comm.allreduce(x, out, count, xmpi.F32, xmpi.SUM, L)
        want = oracle.reduce_ranks([oracle.fill(count, xmpi.F32, xmpi.PAT_SIGNED, 50 + r) for r in range(size)], xmpi.F32, 0)
        assert out.tobytes() == want.tobytes(), f"LL agent allreduce of host slices n={count}"
        assert served() == g0 + 1
    # blocking calls between enqueued ones, no host wait in between: whoever runs a collective, they run one at a time and in order
    st = comm.stream_create()
    m = 500
    a, b2 = comm.alloc(m * 8), comm.alloc(m * 8)
    comm.fill(a, m, xmpi.I64, xmpi.PAT_CONST, 0)  # all ones
    comm.sync()
    reps = 8
    for i in range(reps):
        comm.allreduce_on_stream(a, b2, m, xmpi.I64, xmpi.SUM, st)  # enqueued: b2 = size * a
        comm.stream_sync(st) if i % 2 else None
        comm.allreduce(b2, a, m, xmpi.I64, xmpi.SUM, L)             # blocking (the agent when st is idle, a launch behind it otherwise)
    comm.stream_sync(st)
    with np.errstate(over="ignore"):
        want = np.uint64(size) ** np.uint64(2 * reps)
    assert np.all(a.download(np.int64, m).view(np.uint64) == want), "LL agent between enqueued collectives"
    a.free()
    b2.free()
    # back to back with nothing in between: the agent is told that the epoch is its last one plus one and does not read the page;
    # a launched kernel in between moves the epoch under it -- the next call says so (the host knows what it called last)
    m = 300
    a, b2 = comm.alloc(m * 8), comm.alloc(m * 8)
    comm.fill(a, m, xmpi.I64, xmpi.PAT_CONST, 0)  # all ones
    comm.sync()
    g0 = served()
    for i in range(10):
        comm.allreduce(a, b2, m, xmpi.I64, xmpi.SUM, L)
        comm.allreduce(b2, a, m, xmpi.I64, xmpi.SUM, L)
    assert served() == g0 + 20, "back-to-back blocking collectives: all by the agent"
    comm.allreduce_on_stream(a, b2, m, xmpi.I64, xmpi.SUM, st)  # launched: size^21
    comm.stream_sync(st)
    comm.allreduce(b2, a, m, xmpi.I64, xmpi.SUM, L)              # the agent, told to read the epoch: size^22
    comm.allreduce(a, b2, m, xmpi.I64, xmpi.SUM, L)              # ... and to count again: size^23
    comm.bcast(b2, m, xmpi.I64, size - 1, L)
    comm.allreduce(b2, a, m, xmpi.I64, xmpi.SUM, L)              # size^24
    with np.errstate(over="ignore"):
        want = np.uint64(size) ** np.uint64(24)
    assert np.all(a.download(np.int64, m).view(np.uint64) == want), "LL agent: epochs counted by the agent, a launched kernel in between"
    a.free()
    b2.free()
    comm.stream_destroy(st)
    # who runs a rank's lines is that rank's business: the odd ranks launch, the even ranks' agents serve -- one protocol
    comm.set_param("agent_ll", 2 * (1 - rank % 2))
    for i in range(4):
        reduce_once(xmpi.I64, (1, 64, 500, 512)[i], xmpi.SUM, xmpi.PAT_UNIFORM, 4700 + 10 * i, expect_agent=rank % 2 == 0)
        reduce_once(xmpi.F32, 300, xmpi.MAX, xmpi.PAT_SIGNED, 4750 + 10 * i, root=i % size, expect_agent=rank % 2 == 0)
    # the switch
    comm.set_param("agent_ll", 0)
    reduce_once(xmpi.F32, 100, xmpi.SUM, xmpi.PAT_UNIFORM, 4500, expect_agent=False)
    comm.set_param("agent_ll", 2)
    comm.set_param("agent_ll_bytes", ab)
    reduce_once(xmpi.F32, 100, xmpi.SUM, xmpi.PAT_UNIFORM, 4600)
    # host slices: in and out through pinned memory, the same limit
    for count, by_agent in ((ab // 4, True), (ab // 4 + 1, False)):
        x = oracle.fill(count, xmpi.F32, xmpi.PAT_SIGNED, 70 + rank)
        out = np.zeros_like(x)
        comm.sync()
        g0 = served()
        comm.allreduce(x, out, count, xmpi.F32, xmpi.SUM, L)
        assert served() == g0 + (1 if by_agent else 0), f"host slices of {count * 4} bytes, agent_ll_bytes {ab}"
        want = oracle.reduce_ranks([oracle.fill(count, xmpi.F32, xmpi.PAT_SIGNED, 70 + r) for r in range(size)], xmpi.F32, 0)
        assert out.tobytes() == want.tobytes(), f"LL allreduce of host slices n={count}"
    # a call long after the one before: the agent has gone (its patience over) and is started again
    comm.set_param("agent_ll", 1)
    time.sleep(0.05)
    comm.sync()
    n0, g0 = comm.get_param("ll_agent_launches"), served()
    x = oracle.fill(64, xmpi.I64, xmpi.PAT_UNIFORM, 90 + rank)
    out = np.zeros_like(x)
    comm.allreduce(x, out, 64, xmpi.I64, xmpi.SUM, L)
    want = oracle.reduce_ranks([oracle.fill(64, xmpi.I64, xmpi.PAT_UNIFORM, 90 + r) for r in range(size)], xmpi.I64, 0)
    assert out.tobytes() == want.tobytes()
    assert served() == g0 + 1 and comm.get_param("ll_agent_launches") >= n0, "the agent, started again"


def sc_ll(comm, args):
    """The LL small collectives (ll.hip): {data, flag} lines pushed into the peers' flag allocations, local rank-order fold --
    every dtype and operator up to the slot limit, ragged tails, odd alignments, in place, every root; long runs of
    broadcasts from one root with ranks that dawdle (a slot must not be overwritten under a slow reader); LL and
    zero-copy collectives mixed on one stream without a host wait; graph replays; host slices; unregistered memory."""
    import time
    rank, size = comm.rank(), comm.size()
    L = xmpi.ALGO_LL
    if comm.get_param("dsync") != 1:  # ranks that meet on the host: the name means the library's own choice
        allreduce_case(comm, xmpi.F32, 1000, L, exact=True)
        allgather_case(comm, xmpi.I64, 100, L)
        bcast_case(comm, xmpi.U8, 37, size - 1, L)
        return
    maxb = comm.get_param("ll_max_bytes")
    ll_default = comm.get_param("ll_bytes")
    l0 = comm.get_param("dsync_ll_launches")
    nrun = 0
    for dtype in (xmpi.F32, xmpi.I64, xmpi.F16, xmpi.F64, xmpi.I32, xmpi.U8, xmpi.BF16):
        es = xmpi.DTYPE_SIZE[dtype]
        for count in args.get("counts", [1, 2, 3, 7, 17, 255, 1000, 4099, maxb // es - 1, maxb // es]):
            if count * es <= maxb:
                allreduce_case(comm, dtype, count, L, exact=True)
                nrun += 1
    for op in (xmpi.PROD, xmpi.MIN, xmpi.MAX):
        for dtype in (xmpi.F32, xmpi.I32, xmpi.F16, xmpi.BF16, xmpi.I64, xmpi.F64, xmpi.U8):
            allreduce_case(comm, dtype, 3001, L, op=op, pattern=xmpi.PAT_SIGNED, exact=True)
            nrun += 1
    allreduce_case(comm, xmpi.F32, 4001, L, pattern=xmpi.PAT_SIGNED, inplace=True, misalign=1, exact=True)
    allreduce_case(comm, xmpi.F16, 3011, L, misalign=3, exact=True)
    allreduce_case(comm, xmpi.U8, 13, L, misalign=1, inplace=True, exact=True)
    allreduce_case(comm, xmpi.I64, 1003, L, pattern=xmpi.PAT_UNIFORM, inplace=True, misalign=1, op=xmpi.PROD, exact=True)
    allreduce_case(comm, xmpi.I64, 4097 if 4097 * 8 <= maxb else 1000, L, pattern=xmpi.PAT_CONST, exact=True)  # x_r = r+1 -> N(N+1)/2
    nrun += 5
    assert comm.get_param("dsync_ll_launches") == l0 + nrun, "a named LL allreduce went another way"
    # named, but too long for the slots: the fold (still rank order)
    allreduce_case(comm, xmpi.F32, maxb // 4 + 1, L, exact=True)
    assert comm.get_param("dsync_ll_launches") == l0 + nrun
    for dtype in (xmpi.I64, xmpi.U8, xmpi.F32):
        es = xmpi.DTYPE_SIZE[dtype]
        for count in (1, 5, 13, 1000, 4099, maxb // es):
            if count * es <= maxb:
                allgather_case(comm, dtype, count, L)
    allgather_case(comm, xmpi.I64, 1001, L, inplace=True)
    allgather_case(comm, xmpi.U8, 1001, L, inplace=True)
    for root in range(size):
        for dtype, count in ((xmpi.U8, 1), (xmpi.U8, 37), (xmpi.I64, 4096), (xmpi.F32, 1001), (xmpi.F16, maxb // 2)):
            bcast_case(comm, dtype, count, root, L)
        reduce_case(comm, xmpi.F32, 4001, root, L)
        reduce_case(comm, xmpi.I64, 1, root, L, pat=xmpi.PAT_UNIFORM)
        reduce_case(comm, xmpi.F16, 5001, root, L, op=xmpi.MAX)
        reduce_case(comm, xmpi.BF16, 333, root, L, op=xmpi.PROD)
    _ll_agent_section(comm, maxb)
    # -- many broadcasts from one root, enqueued back to back, while one rank after the other dawdles: a root that ran more
    #    than one epoch ahead of a reader would overwrite a slot under it (the `here` words are what stops it)
    st = comm.stream_create()
    comm.set_param("ll_bytes", maxb)
    K, n = 24, 1500
    bufs = [comm.alloc(n * 8) for _ in range(K)]
    for rounds in range(2):
        root = (size - 1) if rounds else 0
        for k in range(K):
            comm.fill(bufs[k], n, xmpi.I64, xmpi.PAT_UNIFORM, 5000 + 100 * k + rank)
        comm.sync()
        for k in range(K):
            if (k % size) == rank and k % 3 == 0:
                time.sleep(0.002)
            comm.bcast_on_stream(bufs[k], n, xmpi.I64, root, st)
        comm.stream_sync(st)
        for k in range(K):
            want = oracle.fill(n, xmpi.I64, xmpi.PAT_UNIFORM, 5000 + 100 * k + root)
            assert bufs[k].download(np.int64, n).tobytes() == want.tobytes(), f"broadcast {k} of a run from root {root}"
    # -- reduces to one root likewise (the non-roots never hear from the root except through `here`)
    outs = [comm.alloc(n * 8) for _ in range(K)]
    for k in range(K):
        comm.fill(bufs[k], n, xmpi.I64, xmpi.PAT_UNIFORM, 9000 + 100 * k + rank)
    comm.sync()
    for k in range(K):
        if (k % size) == rank and k % 2 == 0:
            time.sleep(0.002)
        comm.reduce_on_stream(bufs[k], outs[k] if rank == 1 % size else None, n, xmpi.I64, xmpi.SUM, 1 % size, st)
    comm.stream_sync(st)
    if rank == 1 % size:
        for k in range(K):
            want = oracle.reduce_ranks([oracle.fill(n, xmpi.I64, xmpi.PAT_UNIFORM, 9000 + 100 * k + r) for r in range(size)], xmpi.I64, xmpi.SUM)
            assert outs[k].download(np.int64, n).tobytes() == want.tobytes(), f"reduce {k} of a run"
    # -- LL and zero-copy collectives alternate on one stream, each consuming the other's result (int64: exact, wraps like Go)
    m_small, m_big = 1000, 70001
    a, b = comm.alloc(m_big * 8), comm.alloc(m_big * 8)
    comm.fill(a, m_big, xmpi.I64, xmpi.PAT_CONST, 0)  # all ones
    comm.sync()
    e0 = comm.get_param("dsync_ll_launches")
    reps = 6
    for k in range(reps):
        comm.allreduce_on_stream(a, b, m_small, xmpi.I64, xmpi.SUM, st)   # LL: the head of b = size * head of a
        comm.allreduce_on_stream(b, a, m_big, xmpi.I64, xmpi.SUM, st)     # fold: a = size * b everywhere
        comm.bcast_on_stream(a, m_small, xmpi.I64, k % size, st)         # LL
        comm.allgather_on_stream(a, b, m_small // size, xmpi.I64, st)    # LL: b's head = a's heads (all equal)
    comm.stream_sync(st)
    assert comm.get_param("dsync_ll_launches") == e0 + 3 * reps, "AUTO did not take the LL path below ll_bytes"
    # head: h -> size*h (into b) -> size*(size*h) (into a); tail of a: t -> size * b_tail where b's tail is never written after fill
    bt = b.download(np.int64, m_big)
    at = a.download(np.int64, m_big)
    with np.errstate(over="ignore"):
        want_head = np.uint64(size) ** np.uint64(2 * reps)
    assert np.all(at[:m_small].view(np.uint64) == want_head), "LL / fold chain: head"
    assert np.all(bt[: (m_small // size) * size].view(np.uint64) == want_head), "LL allgather behind a broadcast"
    a.free()
    b.free()
    # -- a captured graph of LL collectives, replayed (the epoch -- and with it the slots' parity and the flag -- is counted
    #    on the device), ordinary launches in between
    m = 1200
    g1, g2 = comm.alloc(m * 8), comm.alloc(m * 8)
    comm.fill(g1, m, xmpi.I64, xmpi.PAT_CONST, 0)
    comm.sync()
    comm.graph_begin(st)
    comm.allreduce_on_stream(g1, g2, m, xmpi.I64, xmpi.SUM, st)
    comm.bcast_on_stream(g2, m, xmpi.I64, size - 1, st)
    comm.allreduce_on_stream(g2, g1, m, xmpi.I64, xmpi.SUM, st)
    graph = comm.graph_end(st)
    for k in range(5):
        comm.graph_launch(graph, st)
        if k == 2:
            comm.allreduce_on_stream(g1, g1, m, xmpi.I64, xmpi.MAX, st)  # an ordinary launch between two replays
    comm.stream_sync(st)
    assert np.all(g1.download(np.int64, m).view(np.uint64) == np.uint64(size) ** np.uint64(10)), "graph replays of LL collectives"
    comm.graph_destroy(graph)
    g1.free()
    g2.free()
    comm.set_param("ll_bytes", ll_default)
    # -- host slices (what the reference's callers pass) and device memory nobody registered
    for count in (1, 1000, maxb // 4):
        x = oracle.fill(count, xmpi.F32, xmpi.PAT_SIGNED, 5 + rank)
        out = np.zeros_like(x)
        comm.allreduce(x, out, count, xmpi.F32, xmpi.SUM, L)
        want = oracle.reduce_ranks([oracle.fill(count, xmpi.F32, xmpi.PAT_SIGNED, 5 + r) for r in range(size)], xmpi.F32, 0)
        assert out.tobytes() == want.tobytes(), f"LL allreduce of host slices n={count}"
        comm.allreduce(x, x, count, xmpi.F32, xmpi.SUM, L)
        assert x.tobytes() == want.tobytes(), f"LL allreduce of a host slice in place n={count}"
    y = oracle.fill(777, xmpi.I64, xmpi.PAT_UNIFORM, 11 + rank)
    comm.bcast(y, 777, xmpi.I64, size // 2, L)
    assert y.tobytes() == oracle.fill(777, xmpi.I64, xmpi.PAT_UNIFORM, 11 + size // 2).tobytes(), "LL bcast of a host slice"
    z = np.zeros(100 * size, dtype=np.int64)
    comm.allgather(y[:100].copy(), z, 100, xmpi.I64, L)
    assert z.tobytes() == np.tile(y[:100], size).tobytes(), "LL allgather of host slices"
    import ctypes
    hip = _hip_runtime()
    n = 3001
    src, dst = ctypes.c_void_p(0), ctypes.c_void_p(0)
    comm.sync()
    assert hip.hipMalloc(ctypes.byref(src), ctypes.c_size_t(n * 4)) == 0
    assert hip.hipMalloc(ctypes.byref(dst), ctypes.c_size_t(n * 4)) == 0
    comm.fill(src.value, n, xmpi.F32, xmpi.PAT_SIGNED, 900 + rank)
    comm.set_param("ll_bytes", maxb)
    b0 = comm.get_param("dsync_bounced")
    comm.allreduce_on_stream(src.value, dst.value, n, xmpi.F32, xmpi.SUM, st)
    comm.stream_sync(st)
    out = np.empty(n, dtype=np.float32)
    xmpi._check(xmpi.lib().xmpi_memcpy(comm.handle, out.ctypes.data, dst.value, n * 4), "download")
    want = oracle.reduce_ranks([oracle.fill(n, xmpi.F32, xmpi.PAT_SIGNED, 900 + r) for r in range(size)], xmpi.F32, xmpi.SUM)
    assert out.tobytes() == want.tobytes(), "LL allreduce of unregistered device memory"
    assert comm.get_param("dsync_bounced") == b0, "LL lines need no registered stand-in"
    comm.set_param("ll_bytes", ll_default)
    comm.barrier()
    assert hip.hipFree(src) == 0 and hip.hipFree(dst) == 0
    comm.stream_destroy(st)
    for x in bufs + outs:
        x.free()


def sc_split(comm, args):
    """The meet / body / done form of the zero-copy collectives, forced for every size (dsync_split_bytes = 1): the same
    bits as the one-kernel form, rank order, in place, odd alignments; mixed freely with one-kernel collectives."""
    rank, size = comm.rank(), comm.size()
    if comm.get_param("dsync") != 1:
        return
    Z = xmpi.ALGO_ZCOPY
    l0 = comm.get_param("dsync_split_launches")
    comm.set_param("dsync_split_bytes", 1)
    for dtype in (xmpi.F32, xmpi.I64, xmpi.F16, xmpi.F64, xmpi.I32, xmpi.U8, xmpi.BF16):
        for count in args.get("counts", [1, 3, 17, 1000, 4099, 65536 + 5]):
            allreduce_case(comm, dtype, count, Z, exact=True)
    allreduce_case(comm, xmpi.F32, (3 << 20) + 7, Z, pattern=xmpi.PAT_SIGNED, exact=True)
    allreduce_case(comm, xmpi.F32, (3 << 20) + 7, xmpi.ALGO_AUTO, pattern=xmpi.PAT_SIGNED, inplace=True, exact=True)
    for op in (xmpi.PROD, xmpi.MIN, xmpi.MAX):
        for dtype in (xmpi.F32, xmpi.I32, xmpi.F16, xmpi.BF16):
            allreduce_case(comm, dtype, 3001, Z, op=op, pattern=xmpi.PAT_SIGNED, exact=True)
    allreduce_case(comm, xmpi.F16, 30011, Z, misalign=3, exact=True)
    for k in range(6):  # the two forms alternate on one stream
        comm.set_param("dsync_split_bytes", 1 if k % 2 else 0)
        allreduce_case(comm, xmpi.F64, 50001 + k, Z, pattern=xmpi.PAT_SIGNED, inplace=bool(k & 2), exact=True)
    comm.set_param("dsync_split_bytes", 1)
    for dtype in (xmpi.I64, xmpi.U8):
        for count in (1, 1000, (1 << 20) + 3):
            allgather_case(comm, dtype, count, Z)
    for root in sorted({0, size - 1}):
        for dtype, count, pat in ((xmpi.F32, 100003, xmpi.PAT_SIGNED), (xmpi.I64, 4099, xmpi.PAT_UNIFORM)):
            es = xmpi.DTYPE_SIZE[dtype]
            send, recv = comm.alloc(count * es), comm.alloc(count * es)
            comm.fill(send, count, dtype, pat, 70 + rank)
            comm.reduce(send, recv if rank == root else None, count, dtype, xmpi.SUM, root, Z)
            if rank == root:
                ins = [oracle.fill(count, dtype, pat, 70 + r) for r in range(size)]
                check_reduced(recv.download(xmpi.NUMPY_DTYPE[dtype], count), ins, dtype, xmpi.SUM, True, f"split reduce root={root}")
            send.free()
            recv.free()
    # stream-ordered, several back to back, a graph replay of split collectives
    st = comm.stream_create()
    m = 20011
    g1, g2 = comm.alloc(m * 8), comm.alloc(m * 8)
    comm.fill(g1, m, xmpi.I64, xmpi.PAT_CONST, 0)
    comm.allreduce_on_stream(g1, g2, m, xmpi.I64, xmpi.SUM, st)
    comm.allreduce_on_stream(g2, g1, m, xmpi.I64, xmpi.SUM, st)
    comm.stream_sync(st)
    comm.graph_begin(st)
    comm.allreduce_on_stream(g1, g2, m, xmpi.I64, xmpi.SUM, st)
    comm.allreduce_on_stream(g2, g1, m, xmpi.I64, xmpi.SUM, st)
    graph = comm.graph_end(st)
    for _ in range(3):
        comm.graph_launch(graph, st)
    comm.stream_sync(st)
    assert np.all(g1.download(np.int64, m).view(np.uint64) == np.uint64(size) ** np.uint64(8)), "graph replays of split collectives"
    comm.graph_destroy(graph)
    comm.stream_destroy(st)
    g1.free()
    g2.free()
    assert comm.get_param("dsync_split_launches") > l0, "the split form did not run"
    # the assumption under the form: the few blocks of the meet and the done kernel reach EVERY XCD's L2.  Every launch above was
    # checked on the device (xcd_check: a miss fails the collective); these are the masks of the last one
    xc = comm.get_param("xcds")
    assert xc >= 1, "the XCD probe did not run"
    if comm.get_param("xcd_check") == 1 and comm.get_param("body_sys") == 0:
        for which in ("xcd_meet_mask", "xcd_done_mask", "xcd_probe_mask"):
            m = comm.get_param(which)
            assert bin(m).count("1") == xc, f"{which} = {m:#x}: {xc} XCDs"
    assert comm.get_param("xcd_short") == 0
    # ... and the data kernel that does not need it (system-scope loads and stores): the same bits
    comm.set_param("body_sys", 1)
    for dtype in (xmpi.F32, xmpi.I64, xmpi.F16, xmpi.F64, xmpi.I32, xmpi.U8, xmpi.BF16):
        for count in (1, 17, 4099, 65536 + 5):
            allreduce_case(comm, dtype, count, Z, exact=True)
    allreduce_case(comm, xmpi.F32, (3 << 20) + 7, Z, pattern=xmpi.PAT_SIGNED, inplace=True, exact=True)
    allreduce_case(comm, xmpi.F16, 30011, Z, misalign=3, exact=True)
    for op in (xmpi.PROD, xmpi.MIN, xmpi.MAX):
        allreduce_case(comm, xmpi.F32, 3001, Z, op=op, pattern=xmpi.PAT_SIGNED, exact=True)
    allgather_case(comm, xmpi.I64, (1 << 18) + 3, Z)
    comm.set_param("body_sys", 0)
    comm.set_param("dsync_split_bytes", 4 << 20)
    if args.get("trip"):  # a device with one XCD more than it has: the guard must refuse the result, on every rank
        comm.set_param("dsync_split_bytes", 1)
        comm.set_param("xcds", xc + 1)
        try:
            allreduce_case(comm, xmpi.F32, 4099, Z, exact=True)
        except xmpi.XmpiError as e:
            assert "XCD" in str(e), str(e)
            assert comm.get_param("xcd_short") == 1
            # the job is aborted now (a collective failed: nobody may trust the buffers): nothing to meet the peers in any more
            print(f"rank {rank}/{size} split: ok (the guard refused the launch)")
            sys.stdout.flush()
            os._exit(0)
        raise AssertionError("the XCD guard let a short launch pass")


def sc_soak(comm, args):
    """A long seeded random walk over everything the device-synchronised path offers, every result checked against the oracle:
    collectives of every kind in every form (LL lines, one-kernel fold, meet / body / done with either data kernel, push-only,
    ring / halving / tree kernels), blocking and stream-ordered on two streams, Send / Receive rings between them, parameters
    flipped between calls.  What the single-form tests cannot see is what one form leaves behind for the next on the shared,
    never-cleared flag page: epochs, slot parities, tickets, step words, boxes.  Every rank draws the same sequence."""
    import random
    rank, size = comm.rank(), comm.size()
    rng = random.Random(args.get("seed", 20260921))
    dev = comm.get_param("dsync") == 1
    steps = args.get("steps", 200)
    A = xmpi
    streams = [comm.stream_create(), comm.stream_create()] if dev else []
    pending = []  # (kind, buffers ..., expected) of stream-ordered collectives not yet looked at
    pending_send = []
    dtypes = (A.F32, A.I64, A.F16, A.F64, A.I32, A.U8, A.BF16)

    def pick_count():
        r = rng.random()
        if r < 0.35:
            return rng.choice([1, 2, 3, 17, 255, 256, 257, 1000])
        if r < 0.7:
            return rng.randrange(1, 9000)
        if r < 0.93:
            return rng.randrange(9000, 300000)
        return rng.randrange(300000, (3 << 20))

    def drain():
        for st in streams:
            comm.stream_sync(st)
        for what, recv, count, dtype, ins, op, exact in pending:
            got = recv.download(A.NUMPY_DTYPE[dtype], count)
            check_reduced(got, ins, dtype, op, exact, what)
            recv.free()
        pending.clear()

    done = {}
    # what AUTO resolves to: rank order untuned; a "table" step writes rows into the library's schedule table (what xmpi_tune would,
    # had it measured them fastest) -- then AUTO is whatever the row says, also for the stream-ordered, non-blocking and CAPTURED calls
    auto_in_order = {0: True, 3: True}  # (allreduce, reduce: is AUTO's fold in rank order?)

    def order_free(dtype, pat, op):
        return op in (A.MIN, A.MAX) or dtype in (A.I64, A.I32, A.U8) or (dtype in (A.F16, A.BF16) and pat == A.PAT_UNIFORM and op == A.SUM)

    for k in range(steps):
        kind = rng.choice(["allreduce"] * 5 + ["allgather", "bcast", "reduce", "p2p", "stream_allreduce", "stream_allreduce", "params",
                           "stream_p2p", "nonblocking", "host_slices", "graph", "table"])
        done[kind] = done.get(kind, 0) + 1
        dtype = rng.choice(dtypes)
        count = pick_count()
        es = A.DTYPE_SIZE[dtype]
        if kind == "params" and dev:
            comm.set_param("dsync_split_bytes", rng.choice([0, 1, 65536, 4 << 20]))
            comm.set_param("body_sys", rng.choice([0, 0, 1]))
            comm.set_param("ll_bytes", rng.choice([0, 1024, 8192, 32768]))
            comm.set_param("agent_ll", rng.choice([2, 1, 0]))  # blocking LL collectives: by the lingering agent (started whenever / in bursts), or launched
            comm.set_param("agent_ll_bytes", rng.choice([1024, 4096, 32768]))
            comm.set_param("dsync_unroll", rng.choice([1, 2]))
            continue
        if kind == "table":
            if not dev:
                continue
            drain()
            how = rng.choice(["none", "rank order", "anything", "anything"])
            menu = {0: [-1, A.ALGO_ZCOPY, A.ALGO_ZPUSH, A.ALGO_LL] + ([A.ALGO_RING, A.ALGO_RING_PUSH, A.ALGO_RHD, A.ALGO_RHD_PUSH] if how == "anything" else []),
                    1: [-1, A.ALGO_ZCOPY, A.ALGO_LL, A.ALGO_RING, A.ALGO_RING_PUSH],
                    2: [-1, A.ALGO_ZCOPY, A.ALGO_LL, A.ALGO_TREE, A.ALGO_TREE_PUSH],
                    3: [-1, A.ALGO_ZCOPY, A.ALGO_ZPUSH, A.ALGO_LL] + ([A.ALGO_TREE, A.ALGO_TREE_PUSH] if how == "anything" else [])}
            for coll in range(4):
                for cls in range(0, 24, 2):
                    row = rng.choice(menu[coll]) if how != "none" else -1
                    for c2 in (cls, cls + 1):
                        comm.set_param(f"tune_algo_{coll}_{c2}", row)
                        comm.set_param(f"tune_split_{coll}_{c2}", rng.choice([-1, 0, 1]) if how != "none" else -1)
            comm.set_param("tuned", 0 if how == "none" else 1)
            auto_in_order[0] = auto_in_order[3] = how != "anything"
            continue
        if kind == "allreduce":
            algos = [A.ALGO_AUTO, A.ALGO_ZCOPY, A.ALGO_ZPUSH, A.ALGO_RING, A.ALGO_LL, A.ALGO_RING_PUSH] + ([A.ALGO_RHD, A.ALGO_RHD_PUSH] if dev or size & (size - 1) == 0 else [])
            algo = rng.choice(algos)
            op = rng.choice([A.SUM, A.SUM, A.SUM, A.PROD, A.MIN, A.MAX])
            pat = rng.choice([A.PAT_UNIFORM, A.PAT_SIGNED])
            inplace = rng.random() < 0.3
            mis = rng.choice([0, 0, 0, 1, 3])
            rank_order = algo in (A.ALGO_ZCOPY, A.ALGO_ZPUSH, A.ALGO_LL) or (algo == A.ALGO_AUTO and auto_in_order[0]) or size <= 2
            if not rank_order and op == A.PROD and dtype in FLOATS:
                op = A.SUM  # (products in another order: the stated tolerance is for sums)
            exact = rank_order or op in (A.MIN, A.MAX) or dtype in (A.I64, A.I32, A.U8) or (dtype in (A.F16, A.BF16) and pat == A.PAT_UNIFORM and op == A.SUM)
            allreduce_case(comm, dtype, count, algo, op=op, pattern=pat, inplace=inplace, seed0=3000 + k, exact=exact, misalign=mis)
        elif kind == "allgather":
            allgather_case(comm, rng.choice([A.I64, A.U8, A.F32]), min(count, 200000), rng.choice([A.ALGO_AUTO, A.ALGO_RING, A.ALGO_RING_PUSH, A.ALGO_ZCOPY, A.ALGO_LL]),
                           inplace=rng.random() < 0.3)
        elif kind == "bcast":
            bcast_case(comm, dtype, count, rng.randrange(size), rng.choice([A.ALGO_AUTO, A.ALGO_TREE, A.ALGO_TREE_PUSH, A.ALGO_ZCOPY, A.ALGO_LL]), seed=40 + k)
        elif kind == "reduce":
            algo = rng.choice([A.ALGO_AUTO, A.ALGO_TREE, A.ALGO_TREE_PUSH, A.ALGO_ZCOPY, A.ALGO_LL])
            pat = rng.choice([A.PAT_UNIFORM, A.PAT_SIGNED])
            in_order = algo not in (A.ALGO_TREE, A.ALGO_TREE_PUSH) and (algo != A.ALGO_AUTO or auto_in_order[3])
            exact = in_order or size <= 2 or order_free(dtype, pat, A.SUM)
            reduce_case(comm, dtype, min(count, 500000), rng.randrange(size), algo, pat=pat, exact=exact)
        elif kind == "p2p" and size > 1:
            # a ring of blocking messages: even ranks send first, odd ranks receive first (rendezvous sends: no cycle may form)
            n = min(count, 400000)
            tag = 500 + (k % 7)
            nxt, prv = (rank + 1) % size, (rank + size - 1) % size
            out, inn = comm.alloc(n * es), comm.alloc(n * es)
            comm.fill(out, n, dtype, A.PAT_SIGNED, 9000 + k * 16 + rank)
            comm.memset(inn, 0, n * es)
            first_send = rank % 2 == 0 and not (size % 2 == 1 and rank == size - 1)
            if first_send:
                comm.send(out, n, dtype, nxt, tag)
                comm.recv(inn, n, dtype, prv, tag)
            else:
                comm.recv(inn, n, dtype, prv, tag)
                comm.send(out, n, dtype, nxt, tag)
            want = oracle.fill(n, dtype, A.PAT_SIGNED, 9000 + k * 16 + prv)
            assert inn.download(A.NUMPY_DTYPE[dtype], n).tobytes() == want.tobytes(), f"soak step {k}: message from {prv}"
            out.free()
            inn.free()
        elif kind == "stream_allreduce" and dev:
            # enqueued, not waited for: the next operations (any form, any stream) run right behind it
            st = rng.choice(streams)
            n = min(count, 250000)
            send, recv = comm.alloc(n * es), comm.alloc(n * es)
            comm.fill(send, n, dtype, A.PAT_UNIFORM, 5000 + k * 16 + rank)
            comm.allreduce_on_stream(send, recv, n, dtype, A.SUM, st)
            ins = [oracle.fill(n, dtype, A.PAT_UNIFORM, 5000 + k * 16 + r) for r in range(size)]
            pending.append((f"soak step {k}: stream-ordered allreduce {A.DTYPE_NAME[dtype]} n={n}", recv, n, dtype, ins, A.SUM,
                            auto_in_order[0] or size <= 2 or order_free(dtype, A.PAT_UNIFORM, A.SUM)))
            if len(pending) >= 6 or rng.random() < 0.25:
                drain()
            # (send is read by peers until the collective is over: freed by the drain's successor -- kept alive in the tuple's closure)
            pending_send.append(send)
        elif kind == "stream_p2p" and dev and size > 1:
            # the stream-ordered pair as a ring on ONE stream per rank: even ranks send first, odd ranks receive first -- a waiting
            # kernel holds its stream AND the hardware queue under it, and two streams of a process may share a queue (the
            # harness gives every process two: GPU_MAX_HW_QUEUES), so "the receive is on another stream" promises nothing:
            # with send and receive on two streams this very walk deadlocked, every rank's receive queued behind its own send
            drain()
            n = min(count, 100000)
            tag = 700 + (k % 5)
            nxt, prv = (rank + 1) % size, (rank + size - 1) % size
            out, inn = comm.alloc(n * es), comm.alloc(n * es)
            comm.fill(out, n, dtype, A.PAT_SIGNED, 11000 + k * 16 + rank)
            comm.memset(inn, 0, n * es)
            st = streams[k % 2]
            if rank % 2 == 0:
                comm.send_on_stream(out, n, dtype, nxt, tag, st)
                comm.recv_on_stream(inn, n, dtype, prv, tag, st)
            else:
                comm.recv_on_stream(inn, n, dtype, prv, tag, st)
                comm.send_on_stream(out, n, dtype, nxt, tag, st)
            comm.stream_sync(st)
            want = oracle.fill(n, dtype, A.PAT_SIGNED, 11000 + k * 16 + prv)
            assert inn.download(A.NUMPY_DTYPE[dtype], n).tobytes() == want.tobytes(), f"soak step {k}: stream-ordered message from {prv}"
            out.free()
            inn.free()
        elif kind == "nonblocking":
            # two collectives handed to the communicator's worker, a blocking one issued behind them (it runs after them)
            n = min(count, 120000)
            s1, r1 = comm.alloc(n * es), comm.alloc(n * es)
            g1, g2 = comm.alloc(n * 8), comm.alloc(n * 8 * size)
            comm.fill(s1, n, dtype, A.PAT_UNIFORM, 13000 + k * 16 + rank)
            comm.fill(g1, n, A.I64, A.PAT_INDEX, rank)
            q1 = comm.iallreduce(s1, r1, n, dtype, A.SUM, A.ALGO_AUTO)
            q2 = comm.iallgather(g1, g2, n, A.I64)
            allreduce_case(comm, A.I32, 1 + n // 7, A.ALGO_AUTO, seed0=13500 + k, exact=True)
            comm.request_wait(q2)
            comm.request_wait(q1)
            ins = [oracle.fill(n, dtype, A.PAT_UNIFORM, 13000 + k * 16 + r) for r in range(size)]
            check_reduced(r1.download(A.NUMPY_DTYPE[dtype], n), ins, dtype, A.SUM, auto_in_order[0] or size <= 2 or order_free(dtype, A.PAT_UNIFORM, A.SUM),
                          f"soak step {k}: iallreduce")
            want = oracle.allgather([oracle.fill(n, A.I64, A.PAT_INDEX, r) for r in range(size)], A.I64)
            assert g2.download(np.int64, n * size).tobytes() == want.tobytes(), f"soak step {k}: iallgather"
            for b in (s1, r1, g1, g2):
                b.free()
        elif kind == "host_slices":
            # what the reference's callers pass: host memory on both sides of a collective and of a message
            n = min(count, 70000)
            x = oracle.fill(n, dtype, A.PAT_UNIFORM, 15000 + k * 16 + rank)
            y = np.zeros_like(x)
            comm.allreduce(x, y, n, dtype, A.SUM, A.ALGO_AUTO)
            ins = [oracle.fill(n, dtype, A.PAT_UNIFORM, 15000 + k * 16 + r) for r in range(size)]
            check_reduced(y, ins, dtype, A.SUM, auto_in_order[0] or size <= 2 or order_free(dtype, A.PAT_UNIFORM, A.SUM), f"soak step {k}: allreduce of host slices")
            if size > 1:
                nxt, prv = (rank + 1) % size, (rank + size - 1) % size
                z = np.zeros_like(x)
                if rank % 2 == 0 and not (size % 2 == 1 and rank == size - 1):
                    comm.send(x, n, dtype, nxt, 900)
                    comm.recv(z, n, dtype, prv, 900)
                else:
                    comm.recv(z, n, dtype, prv, 900)
                    comm.send(x, n, dtype, nxt, 900)
                assert z.tobytes() == ins[prv].tobytes(), f"soak step {k}: host slice from {prv}"
        elif kind == "graph" and dev:
            # a captured pair of collectives replayed a few times between everything else (the epoch is counted on the device)
            drain()
            m = min(count, 50000)
            g1, g2 = comm.alloc(m * 8), comm.alloc(m * 8)
            comm.fill(g1, m, A.I64, A.PAT_CONST, 0)
            st = streams[k % 2]
            comm.graph_begin(st)
            comm.allreduce_on_stream(g1, g2, m, A.I64, A.SUM, st)
            comm.allreduce_on_stream(g2, g1, m, A.I64, A.SUM, st)
            graph = comm.graph_end(st)
            reps = rng.choice([1, 2, 3])
            first = int(g1.download(np.int64, 1)[0])
            for _ in range(reps):
                comm.graph_launch(graph, st)
            comm.stream_sync(st)
            want = np.uint64(first)
            for _ in range(2 * reps):
                want = want * np.uint64(size)
            assert np.all(g1.download(np.int64, m).view(np.uint64) == want), f"soak step {k}: graph replays"
            comm.graph_destroy(graph)
            g1.free()
            g2.free()
        if k % 50 == 49:
            drain()
            for b in pending_send:
                b.free()
            pending_send.clear()
    drain()
    for b in pending_send:
        b.free()
    for st in streams:
        comm.stream_destroy(st)
    if dev:
        assert comm.get_param("xcd_short") == 0
        comm.set_param("tuned", 0)
        comm.set_param("dsync_split_bytes", 4 << 20)
        comm.set_param("body_sys", 0)
    if rank == 0:
        forms = {k: comm.get_param(k) for k in ("dsync_launches", "dsync_ll_launches", "dsync_split_launches", "dsync_sched_launches",
                                               "p2p_direct_count", "p2p_agent_served", "zc_seq")}
        print(f"soak: {steps} steps {done}; {forms}")


def sc_multistream(comm, args):
    """Device-synchronised collectives of one rank on DIFFERENT streams with no host synchronisation between them (a
    stream-ordered one on a user stream, a blocking one on the communicator's stream right behind it, another user
    stream): the kernels of a rank share its page's epoch, ticket and slots, so the library orders them itself."""
    rank, size = comm.rank(), comm.size()
    if comm.get_param("dsync") != 1:
        return
    s1, s2 = comm.stream_create(), comm.stream_create()
    n = 30011
    bufs = [(comm.alloc(n * 8), comm.alloc(n * 8)) for _ in range(3)]
    for k, (a, _) in enumerate(bufs):
        comm.fill(a, n, xmpi.I64, xmpi.PAT_UNIFORM, 100 * k + rank)
    comm.sync()
    for it in range(args.get("iters", 25)):
        comm.allreduce_on_stream(bufs[0][0], bufs[0][1], n, xmpi.I64, xmpi.SUM, s1)
        comm.allreduce(bufs[1][0], bufs[1][1], n, xmpi.I64, xmpi.MAX, xmpi.ALGO_AUTO)  # blocking, the communicator's own stream
        comm.allreduce_on_stream(bufs[2][0], bufs[2][1], n, xmpi.I64, xmpi.SUM, s2)
        if it % 5 == 4:
            comm.allreduce(bufs[1][0], bufs[1][1], n, xmpi.I64, xmpi.MAX, xmpi.ALGO_RING)
    comm.stream_sync(s1)
    comm.stream_sync(s2)
    for k, op in ((0, xmpi.SUM), (1, xmpi.MAX), (2, xmpi.SUM)):
        ins = [oracle.fill(n, xmpi.I64, xmpi.PAT_UNIFORM, 100 * k + r) for r in range(size)]
        check_reduced(bufs[k][1].download(np.int64, n), ins, xmpi.I64, op, True, f"multi-stream collective {k}")
    comm.stream_destroy(s1)
    comm.stream_destroy(s2)
    for a, b in bufs:
        a.free()
        b.free()


def sc_p2p_stream(comm, args):
    """xmpi_send_on_stream / xmpi_recv_on_stream: the reference's Send / Receive as one kernel on each side (message + ack
    through the flag allocations, the payload pulled out of the sender's HBM), ordered with the other work on the stream."""
    rank, size = comm.rank(), comm.size()
    if comm.get_param("dsync") != 1:
        return
    assert size % 2 == 0
    even, peer = rank % 2 == 0, rank ^ 1
    st = comm.stream_create()
    nmax = (1 << 20) + 5
    a, b = comm.alloc(nmax * 4), comm.alloc(nmax * 4)
    # ping-pong, the echo enqueued right behind the receive on the same stream; the sizes of the reference's bounce
    for k, n in enumerate([0, 1, 3, 1000, 4099, 65536 + 3, nmax]):
        comm.fill(a, n, xmpi.F32, xmpi.PAT_SIGNED, 10 * k + rank)
        comm.memset(b, 0, nmax * 4)
        if even:
            comm.send_on_stream(a, n, xmpi.F32, peer, 5, st)
            comm.recv_on_stream(b, n, xmpi.F32, peer, 5, st)
            comm.stream_sync(st)
            assert comm.count_mismatch(a, b, n * 4) == 0, f"echo of {n} floats differs"
        else:
            comm.recv_on_stream(b, nmax, xmpi.F32, peer, 5, st)  # room for more than arrives
            comm.send_on_stream(b, n, xmpi.F32, peer, 5, st)
            comm.stream_sync(st)
            assert b.download(np.float32, n).tobytes() == oracle.fill(n, xmpi.F32, xmpi.PAT_SIGNED, 10 * k + peer).tobytes()
    # more messages than a pair has boxes, back to back, no host in between; then a collective on the same stream
    n = 5003
    comm.fill(a, n, xmpi.I64, xmpi.PAT_UNIFORM, 77 + rank)
    for k in range(20):
        if even:
            comm.send_on_stream(a, n, xmpi.I64, peer, 100 + k, st)
        else:
            comm.recv_on_stream(b.at(k * n * 8 % (nmax * 4 - n * 8) // 8 * 8), n, xmpi.I64, peer, 100 + k, st)
    comm.allreduce_on_stream(a, a, n, xmpi.I64, xmpi.SUM, st)
    comm.stream_sync(st)
    ins = [oracle.fill(n, xmpi.I64, xmpi.PAT_UNIFORM, 77 + r) for r in range(size)]
    check_reduced(a.download(np.int64, n), ins, xmpi.I64, xmpi.SUM, True, "allreduce behind 20 messages on one stream")
    if not even:
        off = 19 * n * 8 % (nmax * 4 - n * 8) // 8 * 8
        assert b.download(np.int64, n, byte_offset=off).tobytes() == ins[peer].tobytes()
    # a message that does not fit / of another dtype: consumed, and both sides learn it from stream_sync
    for tag, cap, dt in ((7, 50, xmpi.F32), (8, 100, xmpi.I32)):
        try:
            if even:
                comm.send_on_stream(a, 100, xmpi.F32, peer, tag, st)
            else:
                comm.recv_on_stream(b, cap, dt, peer, tag, st)
            comm.stream_sync(st)
            raise AssertionError("a truncated / mistyped message must fail on both sides")
        except xmpi.XmpiError as e:
            assert e.code == (xmpi.ERR_TRUNCATE if tag == 7 else xmpi.ERR_ARG), e
    # the job goes on; a payload in memory the receiver cannot map (plain hipMalloc) travels through a registered block
    import ctypes
    hip = _hip_runtime()
    src = ctypes.c_void_p(0)
    comm.sync()
    assert hip.hipMalloc(ctypes.byref(src), ctypes.c_size_t(40000)) == 0
    comm.fill(src.value, 10000, xmpi.F32, xmpi.PAT_UNIFORM, 500 + rank)
    if even:
        comm.send_on_stream(src.value, 10000, xmpi.F32, peer, 9, st)
    else:
        comm.recv_on_stream(b, 10000, xmpi.F32, peer, 9, st)
    comm.stream_sync(st)
    if not even:
        assert b.download(np.float32, 10000).tobytes() == oracle.fill(10000, xmpi.F32, xmpi.PAT_UNIFORM, 500 + peer).tobytes()
    comm.barrier()
    assert hip.hipFree(src) == 0
    comm.stream_destroy(st)
    a.free()
    b.free()


def sc_tune(comm, args):
    """xmpi_tune: the library times its own schedules and AUTO follows the table -- the same table on every rank."""
    rank, size = comm.rank(), comm.size()
    if not args.get("tuned_by_init"):  # (XMPI_AUTOTUNE_BYTES: xmpi_init has tuned already)
        comm.tune(args.get("max_bytes", 4 << 20))
    if comm.get_param("dsync") != 1:
        assert comm.get_param("tuned") == 0
        return
    assert comm.get_param("tuned") == 1
    table = np.array([comm.get_param(f"tune_{w}_{c}_{k}") for w in ("algo", "split", "unroll") for c in (0, 1, 2, 3) for k in range(24)],
                     dtype=np.int64)
    everybody = np.zeros(table.size * size, dtype=np.int64)
    comm.allgather(table, everybody, table.size, xmpi.I64, xmpi.ALGO_DIRECT)
    assert np.all(everybody.reshape(size, -1) == table), "the ranks tuned different tables"
    assert any(v >= 0 for v in table[:24]), "nothing was tuned"
    # ... after checking every candidate's ANSWER on patterned inputs: on a healthy machine nothing is rejected
    assert comm.get_param("tune_rejected") == 0 and all(comm.get_param(f"tune_rejected_{c}") == 0 for c in range(4)), comm.degraded()
    assert comm.get_param("degraded") & 8 == 0 and comm.get_param("tune_check_us") > 0
    for count in (1, 300, 4099, 100003, (1 << 20) + 1):  # whatever AUTO now takes: the right result (int64: exact in any order)
        allreduce_case(comm, xmpi.I64, count, xmpi.ALGO_AUTO, exact=True)
        allreduce_case(comm, xmpi.F32, count, xmpi.ALGO_AUTO, exact=False)
        allgather_case(comm, xmpi.I64, count, xmpi.ALGO_AUTO)
        for root in sorted({0, size - 1}):  # (the tuner measured with root 0: the table holds for any)
            bcast_case(comm, xmpi.I64, count, root, xmpi.ALGO_AUTO, what="tuned bcast")
            reduce_case(comm, xmpi.I64, count, root, xmpi.ALGO_AUTO, what="tuned reduce")
    # every collective has its own measured rows: bcast and reduce choose among the fold, the tree kernels in both forms and LL lines
    algo_of = lambda coll: table[coll * 24:(coll + 1) * 24]
    for coll, allowed in ((2, {xmpi.ALGO_ZCOPY, xmpi.ALGO_TREE, xmpi.ALGO_TREE_PUSH, xmpi.ALGO_LL}),
                          (3, {xmpi.ALGO_ZCOPY, xmpi.ALGO_ZPUSH, xmpi.ALGO_TREE, xmpi.ALGO_TREE_PUSH, xmpi.ALGO_LL})):
        assert any(v >= 0 for v in algo_of(coll)) and all(int(v) in allowed for v in algo_of(coll) if v >= 0), (coll, algo_of(coll))
    if rank == 0:
        print("tuned allreduce table:", [int(v) for v in table[:24]], "split:", [int(v) for v in table[96:120]], flush=True)
        print("tuned bcast / reduce tables:", [int(v) for v in algo_of(2)], [int(v) for v in algo_of(3)], flush=True)


def sc_devices(comm, args):
    """One process per DEVICE (north_star's layout; rank i <-> device i): what only exists there -- peer access between
    devices, cross-device mappings of the windows and flag pages, the cross-device branch of the pull kernel's grid, the link
    probe -- and every form of the collectives across it.  Runs wherever the runtime shows >= size devices: an 8-GPU node, or
    the CPU suite's tests/devsim (virtual devices; the reference's counterpart is one process per host, network.go:163-263)."""
    rank, size = comm.rank(), comm.size()
    assert comm.device() == rank, f"rank {rank} is on device {comm.device()}"
    assert comm.get_param("dsync") == 1, "ranks on different devices do not meet on the device"
    assert comm.get_param("dsync_sharers") == 1, f"{comm.get_param('dsync_sharers')} ranks think they share device {comm.device()}"
    # the link probe, both directions, copy engine and both kernels (rates mean nothing on virtual devices: that it runs and is > 0)
    for engine in (0, 1, 2):
        for direction in (0, 1):
            if rank < 2:
                assert comm.link_probe(1 - rank, 1 << 20, engine, 3, direction) > 0
            comm.barrier()
    for k, v in args.get("expect_params", {}).items():
        assert comm.get_param(k) == v, f"{k} = {comm.get_param(k)}, expected {v}"
    own_choice = comm.get_param("body_sys")  # 1: xmpi_init's probe grid missed an XCD, the L2-free data kernel is the only safe one
    comm.set_param("xcd_check", 1)
    forms = [("one kernel", {"dsync_split_bytes": 0, "ll_bytes": 0}, xmpi.ALGO_ZCOPY),
             ("push only", {"dsync_split_bytes": 0, "ll_bytes": 0}, xmpi.ALGO_ZPUSH),
             ("meet / body / done", {"dsync_split_bytes": 1, "ll_bytes": 0, "body_sys": 0}, xmpi.ALGO_ZCOPY),
             ("meet / body / done, system-scope data", {"dsync_split_bytes": 1, "ll_bytes": 0, "body_sys": 1}, xmpi.ALGO_ZCOPY),
             ("LL lines", {"dsync_split_bytes": 0, "ll_bytes": comm.get_param("ll_max_bytes")}, xmpi.ALGO_LL),
             ("ring kernel", {}, xmpi.ALGO_RING), ("halving kernel", {}, xmpi.ALGO_RHD), ("auto", {"ll_bytes": 4096}, xmpi.ALGO_AUTO)]
    for what, params, algo in forms:
        if own_choice == 1 and params.get("body_sys") == 0:
            continue
        for k, v in params.items():
            comm.set_param(k, v)
        for count in args.get("counts", (1, 300, 4099, 70001)):
            if algo == xmpi.ALGO_LL and count * 8 > comm.get_param("ll_max_bytes"):
                continue
            allreduce_case(comm, xmpi.I64, count, algo, exact=True)
            allreduce_case(comm, xmpi.F32, count, algo, exact=algo not in (xmpi.ALGO_RING, xmpi.ALGO_RHD))
    comm.set_param("body_sys", own_choice)
    comm.set_param("dsync_split_bytes", 0)
    for root in sorted({0, size // 2, size - 1}):
        for algo in (xmpi.ALGO_AUTO, xmpi.ALGO_TREE):
            bcast_case(comm, xmpi.I64, 20011, root, algo, what="bcast across devices")
            reduce_case(comm, xmpi.I64, 20011, root, algo, what="reduce across devices")
    for algo in (xmpi.ALGO_RING, xmpi.ALGO_DIRECT, xmpi.ALGO_ZCOPY):
        allgather_case(comm, xmpi.I64, 4099, algo)
    # Send / Receive across devices: short (mail slots / agent), long (the pull kernel, grid from the link rate)
    for nbytes in (8, 4096, 1 << 20, (3 << 20) + 24):
        n = nbytes // 8
        tag = 70 + (nbytes & 15)
        buf = comm.alloc(nbytes)
        right, left = (rank + 1) % size, (rank - 1) % size
        if rank % 2 == 0:
            comm.fill(buf, n, xmpi.I64, xmpi.PAT_INDEX, rank)
            comm.send(buf, n, xmpi.I64, right, tag)
            got_from = None
            if size % 2 == 1 and rank == 0:  # (odd ring: rank 0's left neighbour is even as well)
                got_from = left
        else:
            got_from = left
        if got_from is not None:
            inb = comm.alloc(nbytes)
            comm.recv(inb, n, xmpi.I64, got_from, tag)
            got = inb.download(np.int64, n)
            assert got.tobytes() == oracle.fill(n, xmpi.I64, xmpi.PAT_INDEX, got_from).tobytes(), f"{nbytes} bytes from {got_from}"
            inb.free()
        buf.free()
        comm.barrier()
    assert comm.get_param("xcd_short") == 0


def sc_xcd_flaky(comm, args):
    """A dispatcher that now and then deals a small grid round fewer XCDs than the GPU has (tests/devsim, DEVSIM_XCD_MAP=flaky;
    no MI355X we met does this): the split form's guard must refuse THAT collective -- the rank whose launch fell short says so,
    the job is aborted, its peers see a failed collective -- and never hand back a buffer nobody may trust."""
    rank, size = comm.rank(), comm.size()
    comm.set_param("xcd_check", 1)
    comm.set_param("body_sys", 0)
    comm.set_param("dsync_split_bytes", 1)
    comm.set_param("ll_bytes", 0)
    try:
        for k in range(400):
            allreduce_case(comm, xmpi.I64, 4099 + k, xmpi.ALGO_ZCOPY, exact=True)
    except xmpi.XmpiError as e:
        mine = comm.get_param("xcd_short") == 1
        assert ("XCD" in str(e)) == mine or not mine, str(e)
        print(f"rank {rank}/{size} xcd_flaky: ok ({'the guard refused the launch' if mine else 'a peer refused, the job was aborted'})")
        sys.stdout.flush()
        os._exit(0)
    raise AssertionError("400 split collectives under a dispatcher that misses an XCD one small grid in eight, and no refusal")


def _traffic(comm):
    """(reset, read) of tests/devsim's traffic counters: read() -> the matrix [executing device][owner, 16 = host][load, store] in
    bytes, gathered over the ranks (one process per device: every process counts its own kernels)"""
    rt = _hip_runtime()
    assert rt.devsim_traffic_enabled() == 1, "the traffic scenario needs tests/devsim's traced library and DEVSIM_TRAFFIC=1"
    size = comm.size()

    def reset():
        comm.barrier()
        rt.devsim_traffic_reset()
        comm.barrier()

    def read():
        rt.hipDeviceSynchronize()  # (a blocking collective returns when its closing block says so: the other blocks' counts land when the kernel ends)
        comm.barrier()
        row = np.zeros(3 * 17 * 2, dtype=np.uint64)
        rt.devsim_traffic_read(ctypes.c_int(comm.device()), row.ctypes.data_as(ctypes.c_void_p))
        everybody = np.zeros(size * row.size, dtype=np.int64)
        comm.allgather(row.view(np.int64), everybody, row.size, xmpi.I64, xmpi.ALGO_DIRECT)
        return everybody.reshape(size, 3, 17, 2)
    return reset, read


def sc_traffic(comm, args):
    """What would cross the links, counted: one process per virtual device, every load and store of the kernels traced
    (tests/devsim, build --traffic).  For every schedule of the allreduce (and allgather, bcast, reduce, Send / Receive) on S bytes
    per rank: the payload bytes device i reads from / writes to device j's memory against what the schedule's plan says that
    link carries -- to the byte -- and the bytes every device's HBM serves (its own kernels' plus what the
    peers read and write there) against section 5's algorithmic figure.  The library's flag pages (flag words, boxes, LL lines)
    and its small tables are counted apart: their remote STORES are what a link carries on top of the payload; their loads are
    lanes polling their own page.  The reference has no counterpart: its Send writes whole gob frames to one TCP connection
    per peer (network.go:518-571)."""
    import json
    rank, size = comm.rank(), comm.size()
    reset, read = _traffic(comm)
    count = args.get("count", size * 2 * 4096)  # int64: every chunk, half and quarter of it whole 16-byte packets
    S = count * 8
    c = S // size  # one rank's chunk
    pow2 = size & (size - 1) == 0
    report = {}
    pairs = [(i, j) for i in range(size) for j in range(size) if i != j]

    def run(name, fn, plan=None):
        """plan(i, j) -> (loads, stores) of payload device i is planned to do in device j's memory; None: only reported"""
        reset()
        fn()
        both = read()
        m, flags, tables = both[:, 0], both[:, 1], both[:, 2]
        if plan is not None:
            for i, j in pairs:
                want = plan(i, j)
                for k in (0, 1):
                    assert int(m[i, j, k]) == want[k], (f"{name}: device {i} {'stores to' if k else 'loads from'} device {j}: "
                                                        f"{int(m[i, j, k])} payload bytes, plan {want[k]}\n{m[:, :size, k]}")
        into = [[int(m[i, j, 0] + m[j, i, 1]) for j in range(size)] for i in range(size)]  # payload INTO device i FROM device j, by either end's doing
        hbm = [int(m[i, i].sum() + sum(m[j, i].sum() for j in range(size) if j != i)) for i in range(size)]  # served by device i's memory
        report[name] = {
            "remote_loads": int(sum(m[i, j, 0] for i, j in pairs)), "remote_stores": int(sum(m[i, j, 1] for i, j in pairs)),
            "links_used": sum(1 for i, j in pairs if into[i][j] > 0), "busiest_link_direction": max(into[i][j] for i, j in pairs),
            "hbm_per_device_max": max(hbm), "hbm_per_device_min": min(hbm), "host_bytes": int(m[:, 16].sum()),
            "flag_page_remote_stores": int(sum(flags[i, j, 1] for i, j in pairs)), "flag_page_remote_loads": int(sum(flags[i, j, 0] for i, j in pairs)),
            "flag_page_local_loads": int(sum(flags[i, i, 0] for i in range(size))), "table_loads_per_lane": int(tables[:, :, 0].sum())}
        return m

    send, recv = comm.alloc(S), comm.alloc(S * size)
    comm.fill(send, count, xmpi.I64, xmpi.PAT_INDEX, rank)

    def allreduce(algo, **params):
        def fn():
            for k, v in params.items():
                comm.set_param(k, v)
            comm.allreduce(send, recv, count, xmpi.I64, xmpi.SUM, algo)
        return fn

    one = {"dsync_split_bytes": 0, "ll_bytes": 0}
    # the zero-copy fold: rank i reduces chunk i out of everybody's send buffer and writes it into everybody's receive buffer --
    # S / N each way over EVERY link (xGMI is point to point: all seven links of a GPU carry an equal share at once)
    for name, params in (("allreduce fold, one kernel", one), ("allreduce fold, meet / body / done", {"dsync_split_bytes": 1, "ll_bytes": 0, "body_sys": 0}),
                         ("allreduce fold, system-scope data kernel", {"dsync_split_bytes": 1, "ll_bytes": 0, "body_sys": 1})):
        m = run(name, allreduce(xmpi.ALGO_ZCOPY, **params), lambda i, j: (c, c))
        for i in range(size):
            assert int(m[i, i].sum()) == 2 * c, f"{name}: device {i} moved {int(m[i, i].sum())} bytes of its own memory, plan {2 * c}"
        assert report[name]["hbm_per_device_max"] == 2 * S == report[name]["hbm_per_device_min"]  # N reads + N writes per element of a chunk (section 5)
        # what the rendezvous costs a link: the announcement (5 words + the epoch) and the "done" word -- 56 bytes per ordered pair, no remote flag LOAD
        assert report[name]["flag_page_remote_stores"] == 56 * size * (size - 1) and report[name]["flag_page_remote_loads"] == 0, report[name]
    comm.set_param("body_sys", 0)
    # push only: chunk j of my buffer into rank j's scratch, then my reduced chunk into everybody's receive buffer; no remote load
    run("allreduce push only", allreduce(xmpi.ALGO_ZPUSH, **one), lambda i, j: (0, 2 * c))
    # ... in place: the contributions land in the ranks' own blocks instead of their receive buffers; the same bytes over the same links
    comm.allreduce(send, recv, count, xmpi.I64, xmpi.SUM, xmpi.ALGO_ZCOPY)
    _hip_runtime().hipDeviceSynchronize()
    run("allreduce push only, in place", lambda: comm.allreduce(recv, recv, count, xmpi.I64, xmpi.SUM, xmpi.ALGO_ZPUSH), lambda i, j: (0, 2 * c))
    assert comm.get_param("dsync_land_bytes") >= S
    # ring, one kernel per rank: 2 (N - 1) / N x S per rank, loads only.  The channels are different Hamiltonian cycles (Walecki's
    # decomposition, plan.cpp), so that the rings together use every link instead of one neighbour's: with the default channels
    # nobody's busiest link carries more than its share of a single ring would
    m = run("allreduce ring kernel", allreduce(xmpi.ALGO_RING, **one))
    # (a channel's share of the buffer is whole 16 KiB tiles: which piece a rank never has to fetch differs by a tile per channel)
    ragged = 2 * 8 * 16384
    for i in range(size):
        assert abs(int(sum(m[i, j, 0] for j in range(size) if j != i)) - 2 * (size - 1) * c) <= ragged, f"ring: device {i} loaded {m[i, :size, 0]}"
    assert report["allreduce ring kernel"]["remote_loads"] == size * 2 * (size - 1) * c, "the rings together moved more than 2 (N - 1) / N x S per rank"
    assert report["allreduce ring kernel"]["remote_stores"] == 0
    assert report["allreduce ring kernel"]["busiest_link_direction"] <= 2 * (size - 1) * c
    # recursive halving + doubling: S / 2 + S / 4 + ... each way; the partner at distance N / 2 alone carries S
    m = run("allreduce halving kernel", allreduce(xmpi.ALGO_RHD, **one))
    if pow2:
        for i, j in pairs:
            d = i ^ j
            want = 2 * S * d // size if d & (d - 1) == 0 else 0  # distance d: a block of S * d / N, once halving, once doubling
            assert int(m[i, j, 0]) == want, f"halving: device {i} loads {int(m[i, j, 0])} from device {j}, plan {want}"
    # the PUSH forms of the stepped kernels: the same bytes over the same links, every one of them as a store
    m = run("allreduce ring kernel, push", allreduce(xmpi.ALGO_RING_PUSH, **one))
    for i in range(size):
        assert abs(int(sum(m[i, j, 1] for j in range(size) if j != i)) - 2 * (size - 1) * c) <= ragged, f"ring push: device {i} stored {m[i, :size, 1]}"
    r = report["allreduce ring kernel, push"]
    assert r["remote_stores"] == size * 2 * (size - 1) * c and r["remote_loads"] == 0 and r["flag_page_remote_loads"] == 0, r
    assert r["busiest_link_direction"] == report["allreduce ring kernel"]["busiest_link_direction"], "push and pull load the links differently"
    assert comm.get_param("dsync_land_bytes") == 0, "an out-of-place push ring borrowed a landing block"

    # in place: the receive buffer is the input, the partial results land in the landing blocks the ranks lend (one buffer's worth)
    run("allreduce ring kernel, push, in place", lambda: comm.allreduce(recv, recv, count, xmpi.I64, xmpi.SUM, xmpi.ALGO_RING_PUSH))
    r = report["allreduce ring kernel, push, in place"]
    assert r["remote_stores"] == size * 2 * (size - 1) * c and r["remote_loads"] == 0, r
    assert S <= comm.get_param("dsync_land_bytes") <= S + 32
    m = run("allreduce halving kernel, push", allreduce(xmpi.ALGO_RHD_PUSH, **one))
    assert report["allreduce halving kernel, push"]["remote_loads"] == 0
    assert report["allreduce halving kernel, push"]["remote_stores"] == report["allreduce halving kernel"]["remote_loads"]
    if pow2:
        for i, j in pairs:
            d = i ^ j
            want = 2 * S * d // size if d & (d - 1) == 0 else 0
            assert int(m[i, j, 1]) == want, f"halving push: device {i} stores {int(m[i, j, 1])} to device {j}, plan {want}"
    # allgather
    run("allgather fold", lambda: comm.allgather(send, recv, count, xmpi.I64, xmpi.ALGO_ZCOPY), lambda i, j: (0, S))
    m = run("allgather ring kernel", lambda: comm.allgather(send, recv, count, xmpi.I64, xmpi.ALGO_RING))
    for i in range(size):
        assert int(sum(m[i, j, 0] for j in range(size) if j != i)) == (size - 1) * S
    m = run("allgather ring kernel, push", lambda: comm.allgather(send, recv, count, xmpi.I64, xmpi.ALGO_RING_PUSH))
    # (which block a rank does NOT send on differs by channel -- the next rank's on that channel's ring -- and a channel's share of
    # a block is whole tiles: per device within a tile per channel, all devices together to the byte)
    for i in range(size):
        assert abs(int(sum(m[i, j, 1] for j in range(size) if j != i)) - (size - 1) * S) <= ragged and int(sum(m[i, j, 0] for j in range(size) if j != i)) == 0
    assert report["allgather ring kernel, push"]["remote_stores"] == size * (size - 1) * S
    run("allgather direct (step tables, copy engine)", lambda: comm.allgather(send, recv, count, xmpi.I64, xmpi.ALGO_DIRECT))
    # bcast / reduce, root 0
    def bcast_with(push_bytes):
        def fn():
            comm.set_param("zc_bcast_push_bytes", push_bytes)
            comm.bcast(recv, count, xmpi.I64, 0, xmpi.ALGO_AUTO)
        return fn
    run("bcast, root pushes", bcast_with(S), lambda i, j: (0, S if i == 0 else 0))
    if size > 2:  # above zc_bcast_push_bytes: the root scatters, everybody forwards its chunk -- the root's links carry 2 S / N, not S
        run("bcast, scatter + allgather", bcast_with(0), lambda i, j: (0, 0 if j == 0 else 2 * c if i == 0 else c))
    comm.set_param("zc_bcast_push_bytes", 256 << 10)
    m = run("bcast tree kernel", lambda: comm.bcast(recv, count, xmpi.I64, 0, xmpi.ALGO_TREE))
    assert report["bcast tree kernel"]["remote_loads"] == (size - 1) * S and report["bcast tree kernel"]["remote_stores"] == 0
    assert all(int(m[0, j, 0]) == 0 for j in range(1, size)), "the root of a bcast read from somebody"
    m = run("bcast tree kernel, push", lambda: comm.bcast(recv, count, xmpi.I64, 0, xmpi.ALGO_TREE_PUSH))
    assert report["bcast tree kernel, push"]["remote_stores"] == (size - 1) * S and report["bcast tree kernel, push"]["remote_loads"] == 0
    assert all(int(m[j, 0, 1]) == 0 for j in range(1, size)), "somebody stored into the root of a bcast"
    run("reduce fold (chunks, then to the root)", lambda: comm.reduce(send, recv if rank == 0 else None, count, xmpi.I64, xmpi.SUM, 0, xmpi.ALGO_AUTO),
        lambda i, j: (c, c if (j == 0 and i != 0) else 0))
    m = run("reduce tree kernel", lambda: comm.reduce(send, recv if rank == 0 else None, count, xmpi.I64, xmpi.SUM, 0, xmpi.ALGO_TREE))
    assert report["reduce tree kernel"]["remote_loads"] == (size - 1) * S and report["reduce tree kernel"]["remote_stores"] == 0
    m = run("reduce tree kernel, push", lambda: comm.reduce(send, recv if rank == 0 else None, count, xmpi.I64, xmpi.SUM, 0, xmpi.ALGO_TREE_PUSH))
    assert report["reduce tree kernel, push"]["remote_stores"] == (size - 1) * S and report["reduce tree kernel, push"]["remote_loads"] == 0

    # Send / Receive 0 -> 1: the receiver pulls the payload out of the sender's memory, once; nobody else moves a byte
    def p2p():
        if rank == 0:
            comm.send(send, count, xmpi.I64, 1, 5)
        elif rank == 1:
            comm.recv(recv, count, xmpi.I64, 0, 5)
    run("Send / Receive 0 -> 1", p2p, lambda i, j: (S if (i, j) == (1, 0) else 0, 0))
    # LL lines: 16 KiB per rank as {8 bytes of data, 8 of flag} lines into every peer's flag page -- 2 S per peer, one way, no load
    n_ll = 2048
    if n_ll * 8 <= comm.get_param("ll_max_bytes"):
        def ll():
            comm.set_param("ll_bytes", comm.get_param("ll_max_bytes"))
            comm.allreduce(send, recv, n_ll, xmpi.I64, xmpi.SUM, xmpi.ALGO_LL)
        run("allreduce LL lines (16 KiB)", ll, lambda i, j: (0, 0))
        r = report["allreduce LL lines (16 KiB)"]
        assert r["flag_page_remote_loads"] == 0, "an LL collective read a peer's memory"
        assert r["flag_page_remote_stores"] == size * (size - 1) * 2 * n_ll * 8, r
        comm.set_param("ll_bytes", 0)
    send.free()
    recv.free()
    if rank == 0:
        print("TRAFFIC " + json.dumps({"ranks": size, "bytes_per_rank": S, "chunk": c, "schedules": report}), flush=True)


def sc_linkprobe(comm, args):
    """xmpi_link_probe between ranks 0 and 1, every engine and direction, as one JSON line (scripts: what a link -- or, with both
    ranks on one GPU, its HBM -- gives hipMemcpyAsync, the copy kernel and the stepped kernels' system-scope accesses)"""
    import json
    rank = comm.rank()
    out = {}
    for engine, name in ((0, "hipMemcpyAsync"), (1, "copy_kernel"), (2, "sys_kernel")):
        for direction, dn in ((0, "write"), (1, "read")):
            comm.barrier()
            if rank == 0:
                out[f"{name}_{dn}_GBps"] = round(comm.link_probe(1, args.get("bytes", 64 << 20), engine, args.get("iters", 10), direction), 1)
            comm.barrier()
    if rank == 0:
        print("LINKPROBE " + json.dumps({"ranks_share_gpu": comm.get_param("dsync_sharers") > 1, "bytes": args.get("bytes", 64 << 20), **out}), flush=True)


def sc_rooted_bench(comm, args):
    """bcast and reduce by every name, timed (scripts/r05_tree.sh): blocking calls back to back, the slowest rank's mean, one JSON
    line; every result checked once (int64 / f32 uniform halves: exact in any order)"""
    import json
    import time
    rank, size = comm.rank(), comm.size()
    rows = []
    for nbytes in args.get("sizes", [1 << 20, 256 << 20]):
        n = nbytes // 4
        a, b = comm.alloc(nbytes), comm.alloc(nbytes)
        iters = args.get("iters", 20 if nbytes <= (16 << 20) else 8)
        row = {"bytes": nbytes}
        for coll, forms in (("bcast", (("fold", xmpi.ALGO_ZCOPY), ("tree", xmpi.ALGO_TREE), ("tree_push", xmpi.ALGO_TREE_PUSH))),
                            ("reduce", (("fold", xmpi.ALGO_ZCOPY), ("push_only", xmpi.ALGO_ZPUSH), ("tree", xmpi.ALGO_TREE), ("tree_push", xmpi.ALGO_TREE_PUSH)))):
            for name, algo in forms:
                comm.fill(a, n, xmpi.I32, xmpi.PAT_UNIFORM, 60 + rank)
                for it in range(-2, iters):
                    if it == 0:
                        comm.barrier()
                        t0 = time.perf_counter()
                    if coll == "bcast":
                        comm.bcast(a, n, xmpi.I32, 0, algo)
                    else:
                        comm.reduce(a, b, n, xmpi.I32, xmpi.SUM, 0, algo)
                us = np.array([(time.perf_counter() - t0) / iters * 1e6])
                worst = np.zeros(1)
                comm.allreduce(us, worst, 1, xmpi.F64, xmpi.MAX, xmpi.ALGO_AUTO)
                if coll == "bcast":
                    assert a.download(np.int32, n).tobytes() == oracle.fill(n, xmpi.I32, xmpi.PAT_UNIFORM, 60).tobytes(), (name, nbytes)
                elif rank == 0:
                    want = oracle.reduce_ranks([oracle.fill(n, xmpi.I32, xmpi.PAT_UNIFORM, 60 + r) for r in range(size)], xmpi.I32, 0)
                    assert b.download(np.int32, n).tobytes() == want.tobytes(), (name, nbytes)
                row[f"{coll}_{name}_us"] = round(float(worst[0]), 1)
        rows.append(row)
        a.free()
        b.free()
    if rank == 0:
        print("ROOTED " + json.dumps({"ranks": size, "ranks_share_gpu": comm.get_param("dsync_sharers") > 1, "rows": rows}), flush=True)


def sc_degraded(comm, args):
    """A job that could not map everything (tests/devsim fault injection: flag pages / windows / one rank's first open / no
    uncached memory): xmpi_init came back with a WORKING communicator at the best level every rank reached -- the level and the
    reason are readable, every collective by every name still matches the oracle, the reference's two programs still run
    (helloworld.go:53-81: host strings; a ping-pong out of registered HBM).  The reference: Init fails only when the mesh
    cannot be built (network.go:53-65)."""
    rank, size = comm.rank(), comm.size()
    level = comm.get_param("degraded")
    assert level == args["expect"], f"degraded = {level} ({comm.degraded()!r}), expected {args['expect']}"
    assert args.get("why", "") in comm.degraded(), comm.degraded()
    assert (comm.degraded() == "") == (level & 14 == 0)
    assert comm.get_param("windows_ok") == (0 if level & 4 else 1) and comm.get_param("dsync") == (0 if level & 2 else 1)
    sc_allreduce_small(comm, {"counts": [1, 17, 4099], "dtypes": [xmpi.F32, xmpi.I64, xmpi.F16]})
    allgather_case(comm, xmpi.I64, 4099, xmpi.ALGO_AUTO)
    allgather_case(comm, xmpi.I64, 1000, xmpi.ALGO_DIRECT)
    for root in (0, size - 1):
        bcast_case(comm, xmpi.F32, 5001, root, xmpi.ALGO_AUTO)
        bcast_case(comm, xmpi.U8, 37, root, xmpi.ALGO_TREE)
        reduce_case(comm, xmpi.I64, 4099, root, xmpi.ALGO_AUTO, pat=xmpi.PAT_UNIFORM)
        reduce_case(comm, xmpi.I32, 3001, root, xmpi.ALGO_DIRECT)
    x = oracle.fill(5000, xmpi.I64, xmpi.PAT_UNIFORM, 5 + rank)  # a host slice in a collective: a registered stand-in
    out = np.zeros_like(x)
    comm.allreduce(x, out, 5000, xmpi.I64, xmpi.SUM, xmpi.ALGO_AUTO)
    want = oracle.reduce_ranks([oracle.fill(5000, xmpi.I64, xmpi.PAT_UNIFORM, 5 + r) for r in range(size)], xmpi.I64, 0)
    assert out.tobytes() == want.tobytes()
    sc_helloworld(comm, args)
    # Send / Receive out of registered HBM (the receiver pulls), into HBM and into a host slice; and out of device memory the
    # library never registered (without windows: through a registered stand-in on the sender's side)
    n = 70001
    peer = rank ^ 1
    if peer < size:
        a, b = comm.alloc(n * 4), comm.alloc(n * 4)
        comm.fill(a, n, xmpi.F32, xmpi.PAT_SIGNED, 900 + rank)
        host = np.zeros(n, dtype=np.float32)
        rt = _hip_runtime()
        raw = ctypes.c_void_p()
        assert rt.hipMalloc(ctypes.byref(raw), ctypes.c_size_t(n * 4)) == 0
        comm.memcpy(raw.value, a.ptr, n * 4)
        want = oracle.fill(n, xmpi.F32, xmpi.PAT_SIGNED, 900 + peer)
        for dst, src in ((b, a), (host, a), (b, raw.value)):
            if rank < peer:
                comm.send(src, n, xmpi.F32, peer, 11)
                comm.recv(dst, n, xmpi.F32, peer, 12)
            else:
                comm.recv(dst, n, xmpi.F32, peer, 11)
                comm.send(src, n, xmpi.F32, peer, 12)
            got = host if dst is host else b.download(np.float32, n)
            assert got.tobytes() == want.tobytes(), "Send / Receive echo differs"
            comm.memset(b, 0, n * 4)
            host[:] = 0
        if level & 4:  # what needs the mail slots says so instead of hanging
            try:
                comm.send_nowait(a, n, xmpi.F32, peer, 13)
                raise AssertionError("send_nowait worked in a job without windows")
            except xmpi.XmpiError as e:
                assert e.code == xmpi.ERR_UNSUPPORTED, e
        rt.hipDeviceSynchronize()
        rt.hipFree(raw)
        a.free()
        b.free()
    # the tuner in a degraded job: nothing to choose where ranks meet on the host; without windows every candidate is a
    # device-synchronised form and AUTO follows the table as anywhere
    comm.tune(16384)
    assert comm.get_param("tuned") == (0 if level & 2 else 1)
    for count in (1, 300, 5001):
        allreduce_case(comm, xmpi.I64, count, xmpi.ALGO_AUTO, exact=True)
        reduce_case(comm, xmpi.I64, count, size - 1, xmpi.ALGO_AUTO, pat=xmpi.PAT_UNIFORM)
        bcast_case(comm, xmpi.I64, count, 0, xmpi.ALGO_AUTO)
    comm.barrier()  # (every rank: one without a partner -- the last of an odd job -- must not run ahead into finalize)


def sc_peer_dies(comm, args):
    """The last rank leaves without a word (os._exit after a collective that worked) -- a process that crashed.  The reference's
    peers see their TCP connection fail at once (network.go:555,611,623: Send / Receive return the I/O error); here nobody closes
    a shared-memory block on a crash, so the library LOOKS: every rank's helper thread asks every 50 ms whether the processes that
    joined as the other ranks still exist and raises the job's abort flag, which every host wait loop and every waiting kernel
    polls.  `no_timeout`: with DEFAULT settings (XMPI_TIMEOUT_S unset = wait for ever) the survivors' next collective -- one kernel,
    meet / body / done, the ring kernel, LL lines launched or by the lingering agent -- or Receive (blocking or stream-ordered) from
    the dead rank comes back with XMPI_ERR_PEER within a second, naming the rank; the kernels that were waiting have ended (the
    device is usable afterwards), nothing hangs.  Without `no_timeout` (the test sets XMPI_TIMEOUT_S and XMPI_WATCHDOG_MS=0): the
    same through the no-progress limit alone."""
    import time
    rank, size = comm.rank(), comm.size()
    allreduce_case(comm, xmpi.I64, 4099, xmpi.ALGO_AUTO, exact=True)
    what = args.get("what", "allreduce")
    if what == "ll_agent":  # a blocking LL collective that worked: its agent lingers behind it on every rank
        comm.set_param("ll_agent_us", 200000)
        small = comm.alloc(512 * 8)
        comm.fill(small, 512, xmpi.I64, xmpi.PAT_INDEX, rank)
        comm.allreduce(small, small, 512, xmpi.I64, xmpi.SUM, xmpi.ALGO_LL)
    comm.barrier()
    if rank == size - 1:
        sys.stdout.flush()
        os._exit(0)
    limit = comm.get_param("timeout_s")
    if args.get("no_timeout"):
        assert limit == 0 and comm.get_param("watchdog_ms") > 0, "default settings: no no-progress limit, the watchdog on"
    else:
        assert 0 < limit <= 30 and comm.get_param("watchdog_ms") == 0, "the test sets XMPI_TIMEOUT_S and turns the watchdog off"
    buf, out = comm.alloc(4099 * 8), comm.alloc(4099 * 8)
    comm.fill(buf, 4099, xmpi.I64, xmpi.PAT_INDEX, rank)
    t0 = time.time()
    try:
        if what == "recv":
            comm.recv(out, 4099, xmpi.I64, size - 1, 3)
        elif what == "recv_on_stream":
            comm.recv_on_stream(out, 4099, xmpi.I64, size - 1, 3)
            comm.stream_sync()
        else:
            comm.set_param("dsync_split_bytes", 1 if what == "split" else 0)
            comm.set_param("ll_bytes", comm.get_param("ll_max_bytes") if what in ("ll", "ll_agent") else 0)
            if what == "ll":
                comm.set_param("agent_ll", 0)
            n = 512 if what in ("ll", "ll_agent") else 4099
            comm.allreduce(buf, out, n, xmpi.I64, xmpi.SUM, {"ring": xmpi.ALGO_RING, "ll": xmpi.ALGO_LL, "ll_agent": xmpi.ALGO_LL}.get(what, xmpi.ALGO_ZCOPY))
    except xmpi.XmpiError as e:
        took = time.time() - t0
        if args.get("no_timeout"):
            assert took < args.get("within", 1.0), f"{what}: the error took {took:.2f} s"
            assert e.code == xmpi.ERR_PEER, e
            assert comm.get_param("dead_rank") == size - 1, comm.get_param("dead_rank")
        else:
            assert took < 3 * limit + 10, f"{what}: the error took {took:.0f} s with a limit of {limit} s"
        # the device still works: a local kernel on the same buffers, checked
        comm.fill(out, 16, xmpi.I64, xmpi.PAT_INDEX, 7)
        assert out.download(np.int64, 16).tobytes() == oracle.fill(16, xmpi.I64, xmpi.PAT_INDEX, 7).tobytes()
        print(f"rank {rank}/{size} peer_dies[{what}]: ok (error after {took:.2f} s: {str(e)[:160]})")
        sys.stdout.flush()
        os._exit(0)  # (the job is broken: no finalize barrier to meet the dead rank in)
    raise AssertionError(f"{what} with a dead peer returned without an error")


def sc_mismatch(comm, args):
    """The ranks are NOT in the same call -- another length, another schedule, another operation, another root, another collective:
    the reference's peers would block for ever (a Receive nobody sends to, network.go:575-625) or decode garbage; here every kernel
    announces what call it is in with its buffers, and ranks that differ end with an ERROR -- all of them, at once, before anybody
    has touched another's memory."""
    import time
    rank, size = comm.rank(), comm.size()
    what = args["what"]
    n = 40000
    a, b = comm.alloc((n + 64) * 8), comm.alloc((n + 64) * 8 * (size if what == "collective" else 1))
    comm.fill(a, n + 64, xmpi.I64, xmpi.PAT_INDEX, rank)
    comm.memset(b, 0x5A, (n + 64) * 8)
    allreduce_case(comm, xmpi.I64, 4099, xmpi.ALGO_ZCOPY, exact=True)  # (everything mapped)
    comm.set_param("ll_bytes", 0)
    odd = rank == size - 1
    t0 = time.time()
    try:
        if what in ("length", "length_split"):  # (split: the meet kernel finds it, the body kernel moves nothing, the done kernel says so)
            comm.set_param("dsync_split_bytes", 1 if what == "length_split" else 0)
            comm.allreduce(a, b, n + (16 if odd else 0), xmpi.I64, xmpi.SUM, xmpi.ALGO_ZCOPY)
        elif what == "schedule":
            comm.allreduce(a, b, n, xmpi.I64, xmpi.SUM, xmpi.ALGO_RING if odd else xmpi.ALGO_ZCOPY)
        elif what == "form":
            comm.allreduce(a, b, n, xmpi.I64, xmpi.SUM, xmpi.ALGO_RING_PUSH if odd else xmpi.ALGO_RING)
        elif what == "operation":
            comm.allreduce(a, b, n, xmpi.I64, xmpi.MAX if odd else xmpi.SUM, xmpi.ALGO_ZCOPY if args.get("threads") else xmpi.ALGO_RHD)
        elif what == "root":
            comm.reduce(a, b, n, xmpi.I64, xmpi.SUM, 0 if odd else 1, xmpi.ALGO_ZCOPY)
        elif what == "shape":  # one rank was told another grid for the stepped kernels: worker w would wait for a worker that is not there
            if odd:
                comm.set_param("sched_grid", 3)
            comm.allreduce(a, b, n, xmpi.I64, xmpi.SUM, xmpi.ALGO_RING)
        else:  # "collective"
            if odd:
                comm.allgather(a, b, n, xmpi.I64, xmpi.ALGO_ZCOPY)
            else:
                comm.allreduce(a, b, n, xmpi.I64, xmpi.SUM, xmpi.ALGO_ZCOPY)
    except xmpi.XmpiError as e:
        took = time.time() - t0
        assert e.code == xmpi.ERR_ARG and "not in the same call" in str(e), e
        assert took < 5, f"{what}: the error took {took:.1f} s -- somebody waited for a clock"
        got = b.download(np.uint8, (n + 64) * 8)
        assert np.all(got == 0x5A), f"{what}: a peer wrote into this rank's buffer although the calls differed"
        print(f"rank {rank}/{size} mismatch[{what}]: ok (error after {took * 1e3:.0f} ms)")
        sys.stdout.flush()
        if args.get("threads"):
            return "aborted"  # (ranks as threads of one process: the harness skips their barrier and finalize)
        os._exit(0)  # (the job is aborted: no finalize barrier)
    raise AssertionError(f"{what}: ranks in different calls returned without an error")


CAND = {"fold": 0, "fold2": 1, "split": 2, "zpush": 3, "ring": 4, "rhd": 5, "ll": 6, "ring_push": 7, "rhd_push": 8, "tree": 9, "tree_push": 10}
CAND_ALGO = {0: xmpi.ALGO_ZCOPY, 2: xmpi.ALGO_ZCOPY, 3: xmpi.ALGO_ZPUSH, 4: xmpi.ALGO_RING, 5: xmpi.ALGO_RHD, 6: xmpi.ALGO_LL, 7: xmpi.ALGO_RING_PUSH,
             8: xmpi.ALGO_RHD_PUSH, 9: xmpi.ALGO_TREE, 10: xmpi.ALGO_TREE_PUSH}


def sc_corrupt(comm, args):
    """A machine on which ONE schedule gives wrong answers (tests/devsim DEVSIM_CORRUPT_FORM: one device flips a bit in one kind of
    data access of one kernel -- what a link, a cache policy or a mapping could do on a node this code has never met).  The
    library finds it by itself -- xmpi_tune checks every candidate's ANSWER before it times it, xmpi_init's self-check what untuned
    AUTO can reach --, every rank drops exactly the schedules that run that kernel, says which (tune_rejected_<collective>,
    xmpi_degraded(), degraded bit 8), refuses them by name on every rank alike, and AUTO stays oracle-exact.
    (The reference's own benchmark verifies every echo before it reports a time: examples/bounce/bounce.go:103-112,131-136.)"""
    rank, size = comm.rank(), comm.size()
    expect = {int(k): sorted(CAND[n] for n in v) for k, v in args.get("rejected", {}).items()}
    if args.get("tune", 1):
        comm.tune(args.get("max_bytes", 65536))
    got = {c: [k for k in range(11) if comm.get_param(f"tune_rejected_{c}") >> k & 1] for c in range(4)}
    want = {c: expect.get(c, []) for c in range(4)}
    assert got == want, f"rank {rank}: rejected {got}, expected {want} ({comm.degraded()!r})"
    assert comm.get_param("tune_rejected") == sum(len(v) for v in want.values())
    level = comm.get_param("degraded")
    assert level == args.get("level", 8 if any(want.values()) else 0), f"degraded = {level}: {comm.degraded()!r}"
    for needle in args.get("why", []):
        assert needle in comm.degraded(), (needle, comm.degraded())
    for k, v in args.get("params_after", {}).items():
        assert comm.get_param(k) == v, f"{k} = {comm.get_param(k)}, expected {v}"
    if args.get("report_selfcheck"):
        assert comm.get_param("init_selfcheck_us") > 0 and comm.get_param("init_selfcheck_ms") >= 1
        if rank == 0:
            print(f"init_selfcheck_us {comm.get_param('init_selfcheck_us')} ({size} ranks)", flush=True)
    if args.get("tune", 1) and not level & 2:
        assert comm.get_param("tuned") == 1
        for c in range(4):  # nothing rejected is in the table
            bad_algos = {CAND_ALGO[k] for k in want[c] if k not in (0, 1, 2)}
            table = [comm.get_param(f"tune_algo_{c}_{cls}") for cls in range(24)]
            assert not bad_algos & set(table), (c, table, bad_algos)
            if 2 in want[c]:
                assert all(comm.get_param(f"tune_split_{c}_{cls}") != 1 for cls in range(24))
            if 0 in want[c] and c != 2:  # the one-kernel fold is out: where the table says "the fold" it says "split"
                assert all(comm.get_param(f"tune_split_{c}_{cls}") == 1 for cls in range(24) if table[cls] == xmpi.ALGO_ZCOPY), (c, table)
    # AUTO: whatever it takes now, the right result
    for count in (1, 300, 4099, 100003):
        allreduce_case(comm, xmpi.I64, count, xmpi.ALGO_AUTO, exact=True)
        allreduce_case(comm, xmpi.F32, count, xmpi.ALGO_AUTO, exact=False)
        allgather_case(comm, xmpi.I64, count, xmpi.ALGO_AUTO)
        for root in sorted({0, size - 1}):
            bcast_case(comm, xmpi.I64, count, root, xmpi.ALGO_AUTO, what="bcast on the corrupt machine")
            reduce_case(comm, xmpi.I64, count, root, xmpi.ALGO_AUTO, what="reduce on the corrupt machine")
    # Send / Receive out of registered HBM into HBM, short (the lingering agent copies) and long (the pull kernel): every rank sends its
    # pattern to its right neighbour -- whatever way the library now moves a message (p2p_rejected), what arrives is what was sent
    right, left = (rank + 1) % size, (rank + size - 1) % size
    for n in (70001, 300007):
        a, b = comm.alloc(n * 4), comm.alloc(n * 4)
        comm.fill(a, n, xmpi.F32, xmpi.PAT_SIGNED, 900 + rank)
        comm.memset(b, 0xA5, n * 4)
        if rank % 2 == 0:
            comm.send(a, n, xmpi.F32, right, 21)
            comm.recv(b, n, xmpi.F32, left, 21)
        else:
            comm.recv(b, n, xmpi.F32, left, 21)
            comm.send(a, n, xmpi.F32, right, 21)
        assert b.download(np.float32, n).tobytes() == oracle.fill(n, xmpi.F32, xmpi.PAT_SIGNED, 900 + left).tobytes(), f"Send / Receive of {n} floats differs"
        a.free()
        b.free()
    # by name: the rejected ones are REFUSED on every rank alike (nobody hangs waiting for a rank that refused); every other one
    # still runs and is right
    calls = {0: lambda a, n: allreduce_case(comm, xmpi.F32, n, a, exact=False), 1: lambda a, n: allgather_case(comm, xmpi.I64, n, a),
             2: lambda a, n: bcast_case(comm, xmpi.I64, n, 0, a), 3: lambda a, n: reduce_case(comm, xmpi.I64, n, 0, a)}
    offered = {0: (0, 3, 4, 5, 6, 7, 8), 1: (0, 4, 6, 7), 2: (0, 6, 9, 10), 3: (0, 3, 6, 9, 10)}
    if not level & 2:
        for c in range(4):
            for k in offered[c]:
                n = 500 if k == 6 else 20011
                if k in want[c] and not (k == 0 and c != 2 and 2 not in want[c]):  # (the fold by name: a rank keeps to the form of it that is right here)
                    try:
                        calls[c](CAND_ALGO[k], n)
                        raise AssertionError(f"collective {c} by rejected candidate {k} was not refused")
                    except xmpi.XmpiError as e:
                        assert e.code == xmpi.ERR_UNSUPPORTED and "wrong answers" in str(e), e
                elif not (args.get("skip_named_ll") and k == 6):
                    calls[c](CAND_ALGO[k], n)
    comm.barrier()


def sc_guard(comm, args):
    """Virtual devices only (tests/devsim/runtime.cpp devsim_guarded_alloc): every buffer ends on the last byte of a page with an
    inaccessible page behind it, so a kernel that touches ONE byte past a buffer's end faults -- a tail a packet too wide, a tile
    rounded up: on a GPU such an access is served out of the allocation's 2 MiB granule and nobody ever knows.  Ranks are threads of
    this process (the guard is this mapping's).  The local kernels at ragged counts of every width, then every collective by every name
    a caller can give, out of REGISTERED user memory (xmpi_register: the buffers themselves, no stand-ins) -- results against the oracle."""
    import ctypes
    L = xmpi.lib()
    assert hasattr(L, "devsim_guarded_alloc"), "sc_guard runs on tests/devsim only"
    L.devsim_guarded_alloc.restype, L.devsim_guarded_alloc.argtypes = ctypes.c_void_p, [ctypes.c_size_t]
    L.devsim_guarded_free.restype, L.devsim_guarded_free.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t]
    rank, size = comm.rank(), comm.size()
    comm.sync()  # (this thread's device)

    def galloc(nbytes):
        p = L.devsim_guarded_alloc(nbytes)
        assert p, "devsim_guarded_alloc"
        return p

    def gfree(p, nbytes):
        rc = L.devsim_guarded_free(p, nbytes)
        assert rc == 0, "somebody stored BELOW a buffer's start" if rc == -2 else "devsim_guarded_free"

    def down(p, dtype, count):
        out = np.empty(count, dtype=xmpi.NUMPY_DTYPE[dtype])
        comm.memcpy(out, p, out.nbytes)
        return out

    counts = args.get("counts", [1, 3, 17, 255, 1000, 4099, 65536 + 5])
    if args.get("overrun"):  # the seeded fault: a kernel asked to read ONE byte more than the buffer has must take the process down
        a = galloc(4099)
        comm.fill(a, 4099, xmpi.U8, xmpi.PAT_UNIFORM, 1)
        comm.checksum(a, 4099)
        print("guard: in bounds ok, now one byte past the end", flush=True)
        comm.checksum(a, 4100)
        raise AssertionError("a read past the guarded end went unnoticed")
    # ---- the local kernels: two- and N-operand reduce, copy, fill, verify -----------------------------------------------------
    for dtype in (xmpi.U8, xmpi.F16, xmpi.BF16, xmpi.I32, xmpi.F32, xmpi.I64, xmpi.F64):
        es = xmpi.DTYPE_SIZE[dtype]
        for count in counts:
            nb = count * es
            a, b, c3, d = galloc(nb), galloc(nb), galloc(nb), galloc(nb)
            for p, seed in ((a, 11), (b, 12), (c3, 13)):
                comm.fill(p, count, dtype, xmpi.PAT_UNIFORM, seed + rank)
            ins = [oracle.fill(count, dtype, xmpi.PAT_UNIFORM, sd + rank) for sd in (11, 12, 13)]
            assert down(a, dtype, count).tobytes() == ins[0].tobytes(), "fill"
            comm.reduce_local(d, a, b, count, dtype, xmpi.SUM)
            check_reduced(down(d, dtype, count), ins[:2], dtype, xmpi.SUM, True, f"guard: reduce2 {xmpi.DTYPE_NAME[dtype]} n={count}")
            comm.reduce_local_n(d, [a, b, c3], count, dtype, xmpi.SUM)
            check_reduced(down(d, dtype, count), ins, dtype, xmpi.SUM, True, f"guard: reduce_n {xmpi.DTYPE_NAME[dtype]} n={count}")
            comm.reduce_local_multi([d, c3], [a, b], count, dtype, xmpi.SUM)
            check_reduced(down(c3, dtype, count), ins[:2], dtype, xmpi.SUM, True, f"guard: reduce_multi {xmpi.DTYPE_NAME[dtype]} n={count}")
            comm.copy_local(d, a, nb)
            assert down(d, dtype, count).tobytes() == ins[0].tobytes(), "copy"
            comm.copy_local_multi([d, c3], b, nb)
            assert down(c3, dtype, count).tobytes() == ins[1].tobytes(), "copy_multi"
            if nb % 16 == 0:  # (bench.py's box copy: 16-byte aligned buffers only, anything else is refused)
                comm.copy_local_pairs([d, c3], [a, b], nb)
                assert down(d, dtype, count).tobytes() == ins[0].tobytes() and down(c3, dtype, count).tobytes() == ins[1].tobytes(), "copy_pairs"
            comm.copy_local(c3, b, nb)
            comm.copy_local(d, b, nb)
            assert comm.count_mismatch(d, c3, nb) == 0 and comm.count_mismatch(a, b, nb) == oracle.count_mismatch(ins[0], ins[1])
            comm.checksum(a, nb)
            if dtype in (xmpi.F32, xmpi.F16):  # (what bench.py's parity check runs over the timed buffers)
                comm.diff_stats(a, b, count, dtype)
            for p in (a, b, c3, d):
                gfree(p, nb)
    comm.barrier()
    # ---- collectives out of registered user memory that ends at a guard page -------------------------------------------------------
    dsync = comm.get_param("dsync") == 1
    allreduce_algos = [xmpi.ALGO_AUTO, xmpi.ALGO_ZCOPY, xmpi.ALGO_ZPUSH, xmpi.ALGO_RING, xmpi.ALGO_RHD, xmpi.ALGO_DIRECT]
    if dsync:
        allreduce_algos += [xmpi.ALGO_RING_PUSH, xmpi.ALGO_RHD_PUSH, xmpi.ALGO_LL]
    for dtype, count in ((xmpi.F32, 1), (xmpi.U8, 7), (xmpi.I64, 333), (xmpi.F16, 1001), (xmpi.F32, 4099), (xmpi.BF16, 2049), (xmpi.F32, 65536 + 5), (xmpi.F64, 8191)):
        es = xmpi.DTYPE_SIZE[dtype]
        nb = count * es
        send, recv = galloc(nb), galloc(nb)
        comm.register(send, nb)
        comm.register(recv, nb)
        comm.fill(send, count, dtype, xmpi.PAT_UNIFORM, 300 + rank)
        ins = [oracle.fill(count, dtype, xmpi.PAT_UNIFORM, 300 + r) for r in range(size)]
        for split in ((0, 1) if dsync else (0,)):
            if dsync:
                comm.set_param("dsync_split_bytes", 1 if split else 0)
            for algo in allreduce_algos:
                if algo == xmpi.ALGO_LL and nb > 8192:
                    continue
                comm.memset(recv, 0xA5, nb)
                comm.allreduce(send, recv, count, dtype, xmpi.SUM, algo)
                # (rank-order schedules are bit-identical to the oracle; fp16 k/64 inputs are exactly summable in any order)
                exact = algo in (xmpi.ALGO_DIRECT, xmpi.ALGO_ZCOPY, xmpi.ALGO_AUTO, xmpi.ALGO_ZPUSH, xmpi.ALGO_LL) or size <= 2 or dtype == xmpi.F16
                check_reduced(down(recv, dtype, count), ins, dtype, xmpi.SUM, exact, f"guard: allreduce {xmpi.DTYPE_NAME[dtype]} n={count} algo={algo} split={split}")
        # in place
        comm.memcpy(recv, send, nb)
        comm.allreduce(recv, recv, count, dtype, xmpi.SUM, xmpi.ALGO_AUTO)
        check_reduced(down(recv, dtype, count), ins, dtype, xmpi.SUM, True, f"guard: allreduce in place {xmpi.DTYPE_NAME[dtype]} n={count}")
        # reduce to a root, broadcast from it
        root = size // 2
        for algo in (xmpi.ALGO_AUTO, xmpi.ALGO_TREE) + ((xmpi.ALGO_TREE_PUSH,) if dsync else ()):
            comm.memset(recv, 0xA5, nb)
            comm.reduce(send, recv, count, dtype, xmpi.SUM, root, algo)
            if rank == root:
                check_reduced(down(recv, dtype, count), ins, dtype, xmpi.SUM, algo == xmpi.ALGO_AUTO or size <= 2 or dtype == xmpi.F16,
                              f"guard: reduce {xmpi.DTYPE_NAME[dtype]} n={count} algo={algo}")
            comm.memcpy(recv, send, nb)
            comm.bcast(recv, count, dtype, root, algo)
            assert down(recv, dtype, count).tobytes() == ins[root].tobytes(), f"guard: bcast {xmpi.DTYPE_NAME[dtype]} n={count} algo={algo}"
        comm.deregister(send)
        comm.deregister(recv)
        gfree(recv, nb)
        # allgather: the receive buffer ends at the guard, every rank's block inside it
        big = galloc(nb * size)
        comm.register(big, nb * size)
        for algo in (xmpi.ALGO_AUTO, xmpi.ALGO_RING) + ((xmpi.ALGO_RING_PUSH,) if dsync else ()):
            comm.memset(big, 0x5A, nb * size)
            comm.allgather(send, big, count, dtype, algo)
            got = down(big, dtype, count * size)
            assert got.tobytes() == oracle.allgather(ins, dtype).tobytes(), f"guard: allgather {xmpi.DTYPE_NAME[dtype]} n={count} algo={algo}"
        comm.deregister(big)
        gfree(big, nb * size)
        gfree(send, nb)
    if dsync:
        comm.set_param("dsync_split_bytes", 4 << 20)
    # ... and out of guarded memory nobody registered: arena blocks stand in (ranks that meet on the device) or the staged schedules
    # take it through the windows (ranks that meet on the host) -- the copies in and out at ragged lengths
    for dtype, count in ((xmpi.U8, 3), (xmpi.F32, 4099), (xmpi.U8, 65536 * 3 + 1)):
        es = xmpi.DTYPE_SIZE[dtype]
        nb = count * es
        send, recv, big = galloc(nb), galloc(nb), galloc(nb * size)
        comm.fill(send, count, dtype, xmpi.PAT_UNIFORM, 500 + rank)
        ins = [oracle.fill(count, dtype, xmpi.PAT_UNIFORM, 500 + r) for r in range(size)]
        for algo in (xmpi.ALGO_AUTO, xmpi.ALGO_RING, xmpi.ALGO_DIRECT):
            comm.memset(recv, 0xA5, nb)
            comm.allreduce(send, recv, count, dtype, xmpi.SUM, algo)
            check_reduced(down(recv, dtype, count), ins, dtype, xmpi.SUM, algo != xmpi.ALGO_RING or size <= 2, f"guard: unregistered allreduce n={count} algo={algo}")
        comm.allgather(send, big, count, dtype, xmpi.ALGO_AUTO)
        assert down(big, dtype, count * size).tobytes() == oracle.allgather(ins, dtype).tobytes(), f"guard: unregistered allgather n={count}"
        comm.memcpy(recv, send, nb)
        comm.bcast(recv, count, dtype, size - 1, xmpi.ALGO_AUTO)
        assert down(recv, dtype, count).tobytes() == ins[size - 1].tobytes(), f"guard: unregistered bcast n={count}"
        if size >= 2 and (rank ^ 1) < size:
            if rank & 1 == 0:
                comm.send(send, count, dtype, rank ^ 1, 23)
            else:
                comm.recv(recv, count, dtype, rank ^ 1, 23)
                assert down(recv, dtype, count).tobytes() == ins[rank ^ 1].tobytes(), f"guard: unregistered p2p n={count}"
        comm.barrier()
        gfree(send, nb)
        gfree(recv, nb)
        gfree(big, nb * size)
    # Send / Receive out of and into guarded memory
    if size >= 2:
        for dtype, count in ((xmpi.U8, 1), (xmpi.F32, 4099), (xmpi.U8, 65536 * 3 + 1), (xmpi.F64, 100001)):
            es = xmpi.DTYPE_SIZE[dtype]
            nb = count * es
            buf = galloc(nb)
            comm.register(buf, nb)
            peer = rank ^ 1
            if peer < size:
                if rank & 1 == 0:
                    comm.fill(buf, count, dtype, xmpi.PAT_SIGNED, 900 + rank)
                    comm.send(buf, count, dtype, peer, 21)
                else:
                    comm.memset(buf, 0, nb)
                    comm.recv(buf, count, dtype, peer, 21)
                    assert down(buf, dtype, count).tobytes() == oracle.fill(count, dtype, xmpi.PAT_SIGNED, 900 + peer).tobytes(), f"guard: p2p n={count}"
            comm.barrier()
            # ... and the stream-ordered pair (one kernel on each side: sched.hip), with an allreduce enqueued behind it on the same stream
            if dsync and peer < size and size % 2 == 0:
                st = comm.stream_create()
                out = galloc(nb)
                comm.register(out, nb)
                if rank & 1 == 0:
                    comm.fill(buf, count, dtype, xmpi.PAT_SIGNED, 950 + rank)
                    comm.send_on_stream(buf, count, dtype, peer, 22, st)
                else:
                    comm.memset(buf, 0, nb)
                    comm.recv_on_stream(buf, count, dtype, peer, 22, st)
                comm.allreduce_on_stream(buf, out, count, dtype, xmpi.SUM, st)
                comm.stream_sync(st)
                if rank & 1:
                    assert down(buf, dtype, count).tobytes() == oracle.fill(count, dtype, xmpi.PAT_SIGNED, 950 + peer).tobytes(), f"guard: stream p2p n={count}"
                ins2 = [oracle.fill(count, dtype, xmpi.PAT_SIGNED, 950 + (r & ~1)) for r in range(size)]
                check_reduced(down(out, dtype, count), ins2, dtype, xmpi.SUM, True, f"guard: allreduce_on_stream behind a receive n={count}")
                comm.stream_destroy(st)
                comm.barrier()
                comm.deregister(out)
                gfree(out, nb)
            comm.deregister(buf)
            gfree(buf, nb)


SCENARIOS = {
    "guard": sc_guard,
    "corrupt": sc_corrupt,
    "mismatch": sc_mismatch,
    "linkprobe": sc_linkprobe,
    "rooted_bench": sc_rooted_bench,
    "degraded": sc_degraded,
    "peer_dies": sc_peer_dies,
    "traffic": sc_traffic,
    "xcd_flaky": sc_xcd_flaky,
    "devices": sc_devices,
    "ll": sc_ll,
    "sched": sc_sched,
    "soak": sc_soak,
    "split": sc_split,
    "multistream": sc_multistream,
    "p2p_stream": sc_p2p_stream,
    "tune": sc_tune,
    "allreduce_small": sc_allreduce_small,
    "allreduce_medium": sc_allreduce_medium,
    "allgather": sc_allgather,
    "bcast_reduce": sc_bcast_reduce,
    "bounce": sc_bounce,
    "heap_colours": sc_heap_colours,
    "host_payloads": sc_host_payloads,
    "helloworld": sc_helloworld,
    "p2p_semantics": sc_p2p_semantics,
    "fullsize": sc_fullsize,
    "zero_copy": sc_zero_copy,
    "nonblocking": sc_nonblocking,
    "stream_ordered": sc_stream_ordered,
    "lifecycle_stress": sc_lifecycle_stress,
}
