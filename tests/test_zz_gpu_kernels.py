"""(Named to run LAST: this module opens the GPU in the test runner's own process, and a GPU schedules the queues of
only 8 processes at once -- the 8-process tests of the other modules should not find a ninth already there.)

Parity of the HIP kernels with the CPU oracle, through the C ABI, on one MI355X.
Bit-exact for every dtype and operator (a two-operand combine has no ordering freedom; the
N-way fold is strictly left to right like the oracle's)."""
import numpy as np
import pytest

from mpi_amd import xmpi
from oracle import oracle

pytestmark = pytest.mark.gpu

ALL_DTYPES = [xmpi.U8, xmpi.I32, xmpi.I64, xmpi.F16, xmpi.F32, xmpi.F64, xmpi.BF16]
OPS = [xmpi.SUM, xmpi.PROD, xmpi.MIN, xmpi.MAX]


@pytest.fixture(scope="module")
def comm():
    import os
    import uuid
    c = xmpi.Comm(0, 1, 0, f"k{os.getpid()}-{uuid.uuid4().hex[:6]}")
    yield c
    c.finalize()


@pytest.mark.parametrize("dtype", ALL_DTYPES)
@pytest.mark.parametrize("pattern", [xmpi.PAT_UNIFORM, xmpi.PAT_INDEX, xmpi.PAT_CONST, xmpi.PAT_SIGNED])
def test_fill_matches_oracle(comm, dtype, pattern):
    n = 70001
    buf = comm.alloc(n * xmpi.DTYPE_SIZE[dtype])
    comm.fill(buf, n, dtype, pattern, 12345)
    got = buf.download(xmpi.NUMPY_DTYPE[dtype], n)
    want = oracle.fill(n, dtype, pattern, 12345)
    assert got.tobytes() == want.tobytes()
    buf.free()


@pytest.mark.parametrize("dtype", ALL_DTYPES)
@pytest.mark.parametrize("op", OPS)
def test_reduce2_bit_exact(comm, dtype, op):
    es = xmpi.DTYPE_SIZE[dtype]
    for count in (0, 1, 2, 15, 16, 17, 1023, 4096, 65536 + 3, (1 << 20) + 11):
        for pattern in (xmpi.PAT_UNIFORM, xmpi.PAT_SIGNED):
            a = oracle.fill(count, dtype, pattern, 7)
            b = oracle.fill(count, dtype, pattern, 8)
            da, db, dd = comm.alloc(count * es + 16), comm.alloc(count * es + 16), comm.alloc(count * es + 16)
            da.upload(a)
            db.upload(b)
            comm.memset(dd, 0xEE, count * es + 16)
            comm.reduce_local(dd, da, db, count, dtype, op)
            got = dd.download(xmpi.NUMPY_DTYPE[dtype], count)
            want = oracle.reduce2(a, b, dtype, op)
            assert got.tobytes() == want.tobytes(), (xmpi.DTYPE_NAME[dtype], op, count, pattern)
            guard = dd.download(np.uint8, 16, byte_offset=count * es)
            assert np.all(guard == 0xEE), "kernel wrote past the end"
            for x in (da, db, dd):
                x.free()


@pytest.mark.parametrize("dtype", [xmpi.F32, xmpi.F16, xmpi.U8, xmpi.I64])
def test_reduce2_unaligned_and_inplace(comm, dtype):
    """chunk boundaries can fall on any element: odd byte offsets take the element path"""
    es = xmpi.DTYPE_SIZE[dtype]
    count = 10007
    a = oracle.fill(count + 8, dtype, xmpi.PAT_SIGNED, 1)
    b = oracle.fill(count + 8, dtype, xmpi.PAT_SIGNED, 2)
    da, db = comm.alloc((count + 8) * es), comm.alloc((count + 8) * es)
    for off in (1, 3):
        da.upload(a)
        db.upload(b)
        comm.reduce_local(da.at(off * es), da.at(off * es), db.at(off * es), count, dtype, xmpi.SUM)  # dst == a
        got = da.download(xmpi.NUMPY_DTYPE[dtype], count + 8)
        want = a.copy()
        want[off:off + count] = oracle.reduce2(a[off:off + count], b[off:off + count], dtype, xmpi.SUM)
        assert got.tobytes() == want.tobytes()
    da.free()
    db.free()


@pytest.mark.parametrize("dtype", [xmpi.F32, xmpi.F16, xmpi.F64, xmpi.I32, xmpi.BF16])
@pytest.mark.parametrize("nsrc", [1, 2, 3, 5, 8, 12])
def test_reduce_n_is_left_to_right(comm, dtype, nsrc):
    es = xmpi.DTYPE_SIZE[dtype]
    for count in (1, 1000, 65536 + 7):
        ins = [oracle.fill(count, dtype, xmpi.PAT_SIGNED, 100 + r) for r in range(nsrc)]
        bufs = [comm.alloc(count * es).upload(x) for x in ins]
        dst = comm.alloc(count * es)
        for op in (xmpi.SUM, xmpi.MAX):
            comm.reduce_local_n(dst, bufs, count, dtype, op)
            got = dst.download(xmpi.NUMPY_DTYPE[dtype], count)
            want = oracle.reduce_ranks(ins, dtype, op)
            assert got.tobytes() == want.tobytes(), (xmpi.DTYPE_NAME[dtype], nsrc, count, op)
        for x in bufs + [dst]:
            x.free()


@pytest.mark.parametrize("dtype", [xmpi.F32, xmpi.F16, xmpi.I64, xmpi.BF16, xmpi.U8])
@pytest.mark.parametrize("nsrc,ndst", [(1, 1), (2, 2), (3, 1), (4, 4), (8, 8), (8, 1), (5, 7), (12, 3)])
def test_reduce_n_multi_fold_once_store_many(comm, dtype, nsrc, ndst):
    """the zero-copy allreduce kernel: every destination gets the left-to-right fold; a destination may be
    one of the sources (in place); odd offsets take the element path"""
    es = xmpi.DTYPE_SIZE[dtype]
    for count, shift in ((1, 0), (1000, 0), (65536 + 7, 0), (10007, 1)):
        ins = [oracle.fill(count, dtype, xmpi.PAT_SIGNED, 100 + r) for r in range(nsrc)]
        bufs = [comm.alloc((count + 2) * es) for _ in ins]
        for b, x in zip(bufs, ins):
            b.upload(x, byte_offset=shift * es)
        outs = [comm.alloc((count + 2) * es + 16) for _ in range(ndst - 1)]
        for o in outs:
            comm.memset(o, 0xEE, (count + 2) * es + 16)
        inplace = bufs[min(1, nsrc - 1)]  # the last destination aliases a source
        for op in (xmpi.SUM, xmpi.MIN):
            bufs[min(1, nsrc - 1)].upload(ins[min(1, nsrc - 1)], byte_offset=shift * es)
            dsts = [o.at(shift * es) for o in outs] + [inplace.at(shift * es)]
            comm.reduce_local_multi(dsts, [b.at(shift * es) for b in bufs], count, dtype, op)
            want = oracle.reduce_ranks(ins, dtype, op)
            for o in outs + [inplace]:
                got = o.download(xmpi.NUMPY_DTYPE[dtype], count, byte_offset=shift * es)
                assert got.tobytes() == want.tobytes(), (xmpi.DTYPE_NAME[dtype], nsrc, ndst, count, op)
            for o in outs:
                guard = o.download(np.uint8, 16, byte_offset=(shift + count) * es)
                assert np.all(guard == 0xEE), "kernel wrote past the end"
        for x in bufs + outs:
            x.free()


@pytest.mark.parametrize("ndst", [1, 2, 7, 15])
def test_copy_multi(comm, ndst):
    for n, shift in ((1, 0), (4099, 0), ((3 << 20) + 13, 0), (100001, 3)):
        a = oracle.fill(n + 8, xmpi.U8, xmpi.PAT_UNIFORM, 5)
        src = comm.alloc(n + 8).upload(a)
        outs = [comm.alloc(n + 24) for _ in range(ndst)]
        for o in outs:
            comm.memset(o, 0xEE, n + 24)
        # the source itself may be listed (in-place allgather block): it is skipped
        comm.copy_local_multi([o.at(shift) for o in outs] + [src.at(shift)], src.at(shift), n)
        for o in outs:
            got = o.download(np.uint8, n + 24)
            assert got[shift:shift + n].tobytes() == a[shift:shift + n].tobytes()
            assert np.all(got[:shift] == 0xEE) and np.all(got[shift + n:] == 0xEE)
        assert src.download(np.uint8, n + 8).tobytes() == a.tobytes()
        for x in outs + [src]:
            x.free()


def test_malloc_carves_blocks_out_of_arenas(comm):
    """xmpi_malloc: 256-byte aligned blocks of a few long-lived arenas; freed blocks coalesce and are reused"""
    a0, r0, u0 = (comm.get_param(k) for k in ("heap_arenas", "heap_reserved", "heap_in_use"))
    bufs = [comm.alloc(n) for n in (1, 255, 256, 257, 4096, 1 << 20, 3 << 20)]
    assert all(b.ptr % 256 == 0 for b in bufs)
    spans = sorted((b.ptr, b.ptr + max(1, b.nbytes)) for b in bufs)
    assert all(spans[i][1] <= spans[i + 1][0] for i in range(len(spans) - 1)), "blocks overlap"
    assert comm.get_param("heap_in_use") >= u0 + sum(b.nbytes for b in bufs)
    for i, b in enumerate(bufs):  # blocks are usable device memory
        comm.memset(b, i + 1, b.nbytes)
    for i, b in enumerate(bufs):
        assert np.all(b.download(np.uint8, b.nbytes) == i + 1)
    first = bufs[0].ptr
    r1 = comm.get_param("heap_reserved")
    assert r1 >= r0
    for b in bufs[::2] + bufs[1::2]:  # free out of order: neighbours merge
        b.free()
    assert comm.get_param("heap_in_use") == u0
    big = comm.alloc(4 << 20)  # fits where the seven blocks were, once they have coalesced
    assert comm.get_param("heap_reserved") == r1, "a new arena was reserved although the freed blocks had room"
    again = comm.alloc(1)
    assert again.ptr != big.ptr
    big.free()
    again.free()
    assert comm.get_param("heap_arenas") >= max(1, a0) and first % 256 == 0
    # a buffer larger than any arena gets one of its own
    huge = comm.alloc((1 << 30) + 4096)
    comm.memset(huge.at(1 << 30), 7, 4096)
    assert np.all(huge.download(np.uint8, 4096, byte_offset=1 << 30) == 7)
    huge.free()
    # not the start of a live block
    assert xmpi.lib().xmpi_free(comm.handle, first + 17) == xmpi.ERR_ARG


def test_zc_chunk_partition():
    for count, es, size in ((0, 4, 8), (1, 4, 8), (67108864, 4, 8), (4099, 8, 3), (1001, 1, 5), (17, 2, 16)):
        pos = 0
        for j in range(size):
            off, cnt = xmpi.zc_chunk(count, es, size, j)
            assert off == pos and (off * es) % 16 == 0 or cnt == 0
            pos = off + cnt
        assert pos == count


def test_copy_and_verify_kernels(comm):
    n = (3 << 20) + 13
    a = oracle.fill(n, xmpi.U8, xmpi.PAT_UNIFORM, 5)
    da, db = comm.alloc(n), comm.alloc(n)
    da.upload(a)
    comm.memset(db, 0, n)
    comm.copy_local(db, da, n)
    assert db.download(np.uint8, n).tobytes() == a.tobytes()
    assert comm.count_mismatch(da, db, n) == 0
    assert comm.checksum(da, n) == oracle.checksum(a)
    # flip a few bytes: the LDS/shuffle reduction must count exactly those
    b = a.copy()
    idx = [0, 1, 17, 4096, n // 2, n - 1]
    for i in idx:
        b[i] ^= 0x40
    db.upload(b)
    assert comm.count_mismatch(da, db, n) == len(idx) == oracle.count_mismatch(a, b)
    # unaligned views
    assert comm.count_mismatch(da.at(1), db.at(1), n - 1) == len(idx) - 1
    assert comm.checksum(da.at(3), n - 3) == oracle.checksum(a[3:])
    comm.copy_local(db.at(5), da.at(2), 1001)
    assert db.download(np.uint8, 1001, byte_offset=5).tobytes() == a[2:1003].tobytes()
    da.free()
    db.free()


@pytest.mark.parametrize("dtype", [xmpi.F32, xmpi.F16, xmpi.F64, xmpi.BF16])
def test_diff_stats(comm, dtype):
    n = 200003
    es = xmpi.DTYPE_SIZE[dtype]
    a = oracle.fill(n, dtype, xmpi.PAT_SIGNED, 1)
    b = oracle.fill(n, dtype, xmpi.PAT_SIGNED, 2)
    da, db = comm.alloc(n * es).upload(a), comm.alloc(n * es).upload(b)
    mx, sb, nn = comm.diff_stats(da, db, n, dtype)
    wmx, wsb, wnn = oracle.diff_stats(a, b, dtype)
    assert mx == wmx and nn == wnn == 0
    assert abs(sb - wsb) <= 1e-9 * wsb  # the sum's association differs
    da.free()
    db.free()


@pytest.mark.parametrize("dtype", [xmpi.F32, xmpi.F64, xmpi.F16])
def test_diff_rel_is_the_per_element_bound(comm, dtype):
    """max_i |a_i - b_i| / |b_i| over the whole buffer (what the full-size float checks use), against numpy"""
    n = 300007
    b = oracle.fill(n, dtype, xmpi.PAT_UNIFORM, 3)
    a = b.copy()
    rng = np.random.default_rng(11)
    idx = rng.choice(n, size=50, replace=False)
    a[idx] = (oracle.as_float64(a[idx], dtype) * (1 + 1e-3) + 1e-3).astype(a.dtype)
    da, db = comm.alloc(a.nbytes).upload(a), comm.alloc(b.nbytes).upload(b)
    got = comm.diff_rel(da, db, n, dtype)
    a64, b64 = oracle.as_float64(a, dtype), oracle.as_float64(b, dtype)
    d = np.abs(a64 - b64)
    with np.errstate(divide="ignore", invalid="ignore"):
        want = np.max(np.where(d == 0, 0.0, d / np.abs(b64)))
    assert got == want, (got, want)
    assert comm.diff_rel(db, db, n, dtype) == 0.0
    da.free()
    db.free()


def test_send_to_self_needs_no_peer(comm):
    """size-1 communicator: collectives degenerate to the local copy kernel"""
    n = 100001
    a = oracle.fill(n, xmpi.F32, 0, 9)
    da, db = comm.alloc(n * 4).upload(a), comm.alloc(n * 4)
    comm.allreduce(da, db, n, xmpi.F32, xmpi.SUM, xmpi.ALGO_RING)
    assert db.download(np.float32, n).tobytes() == a.tobytes()
    comm.allgather(da, db, n, xmpi.F32)
    assert db.download(np.float32, n).tobytes() == a.tobytes()
    comm.bcast(da, n, xmpi.F32, 0)
    comm.reduce(da, db, n, xmpi.F32, xmpi.SUM, 0)
    assert db.download(np.float32, n).tobytes() == a.tobytes()
    assert comm.rank() == 0 and comm.size() == 1
    da.free()
    db.free()
