"""Host-logic tests of the collective schedules (no GPU): every rank's step table, as emitted by
libxmpi.so (xmpi_plan_dump), is executed by tests/plan_sim.py over bounded FIFOs under a random
schedule and compared with the rank-order oracle."""
import numpy as np
import pytest

from mpi_amd import xmpi
from oracle import oracle
from tests import plan_sim

SIZES = [1, 2, 3, 4, 5, 8]


def _inputs(n, count, dtype_code, seed0=1000, pattern=oracle.lib and 0):
    return [oracle.fill(count, dtype_code, 0, seed0 + r) for r in range(n)]


@pytest.mark.parametrize("algo", [xmpi.ALGO_RING, xmpi.ALGO_RHD, xmpi.ALGO_DIRECT])
@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("count,piece", [(1, 4), (7, 4), (64, 4), (1000, 16), (4099, 64), (65536, 1024)])
def test_allreduce_i64_exact(algo, n, count, piece):
    ins = _inputs(n, count, oracle.I64)
    want = oracle.reduce_ranks(ins, oracle.I64, oracle.SUM)
    for channels in (1, 4):
        plans = plan_sim.get_plans(xmpi.COLL_ALLREDUCE, algo, n, 0, count, 8, channels, piece)
        for depth in (2, 8):
            got = plan_sim.simulate(plans, ins, count, np.int64, xmpi.SUM, depth, seed=count + n)
            for r in range(n):
                assert np.array_equal(got[r], want), f"rank {r}"


@pytest.mark.parametrize("algo", [xmpi.ALGO_RING, xmpi.ALGO_RHD, xmpi.ALGO_DIRECT])
@pytest.mark.parametrize("n", [2, 4, 8])
def test_allreduce_inplace(algo, n):
    count = 3001
    ins = _inputs(n, count, oracle.I64)
    want = oracle.reduce_ranks(ins, oracle.I64, oracle.SUM)
    plans = plan_sim.get_plans(xmpi.COLL_ALLREDUCE, algo, n, 0, count, 8, 4, 128)
    got = plan_sim.simulate(plans, ins, count, np.int64, xmpi.SUM, 2, seed=7, inplace=True)
    for r in range(n):
        assert np.array_equal(got[r], want)


@pytest.mark.parametrize("n", [2, 3, 4, 8])
def test_allreduce_direct_is_rank_order_f32(n):
    """DIRECT folds in rank order: bit-identical to the oracle for floats too."""
    count = 2053
    ins = [oracle.fill(count, oracle.F32, 3, 50 + r) for r in range(n)]  # signed, mixed binades
    want = oracle.reduce_ranks(ins, oracle.F32, oracle.SUM)
    plans = plan_sim.get_plans(xmpi.COLL_ALLREDUCE, xmpi.ALGO_DIRECT, n, 0, count, 4, 1, 256)
    got = plan_sim.simulate(plans, ins, count, np.float32, xmpi.SUM, 4, seed=3)
    for r in range(n):
        assert got[r].tobytes() == want.tobytes()


@pytest.mark.parametrize("n", [1, 2, 3, 5, 8])
@pytest.mark.parametrize("count,piece", [(1, 4), (7, 4), (1000, 16), (4099, 64), (65536, 65536)])
@pytest.mark.parametrize("inplace", [False, True])
def test_allreduce_oneshot(n, count, piece, inplace):
    """Small-message DIRECT: every rank pushes its whole buffer to every peer and folds all N in
    rank order itself -- one exchange instead of two, still bit-identical to the oracle."""
    ins = [oracle.fill(count, oracle.F32, 3, 70 + r) for r in range(n)]
    want = oracle.reduce_ranks(ins, oracle.F32, oracle.SUM)
    plans = plan_sim.get_plans(xmpi.COLL_ALLREDUCE, xmpi.ALGO_DIRECT, n, 0, count, 4, 1, piece,
                               oneshot_bytes=1 << 20)
    if n > 1:
        kinds = {s.kind for s in plans[0].steps}
        assert kinds == {0, 3, 4}, kinds  # SEND, RECV_HOLD, REDUCE_N only
        assert sum(s.kind == 0 for s in plans[0].steps) == (n - 1) * -(-count // max(piece, 4))
    for depth in (2, 8):
        got = plan_sim.simulate(plans, ins, count, np.float32, xmpi.SUM, depth, seed=count + n, inplace=inplace)
        for r in range(n):
            assert got[r].tobytes() == want.tobytes(), f"rank {r}"


@pytest.mark.parametrize("size", [1, 2, 3, 5, 8])
@pytest.mark.parametrize("count,es", [(0, 4), (1, 4), (67108864, 4), (4099, 8), (1001, 1), (17, 2), (536870912, 2)])
def test_zero_copy_chunks_partition_the_buffer(size, count, es):
    """zero-copy collectives: rank j folds / forwards chunk j; the chunks tile [0, count) in rank order,
    every boundary 16-byte aligned, and a numpy fold over them reproduces the oracle"""
    pos = 0
    for j in range(size):
        off, cnt = xmpi.zc_chunk(count, es, size, j)
        assert off == pos
        assert cnt == 0 or (off * es) % 16 == 0
        pos = off + cnt
    assert pos == count
    if count and count <= 5000 and es in (4, 8):
        code = oracle.F32 if es == 4 else oracle.F64
        ins = [oracle.fill(count, code, 3, 11 + r) for r in range(size)]
        out = np.zeros_like(ins[0])
        for j in range(size):  # what rank j's kernel computes
            off, cnt = xmpi.zc_chunk(count, es, size, j)
            out[off:off + cnt] = oracle.reduce_ranks([x[off:off + cnt] for x in ins], code, oracle.SUM)
        assert out.tobytes() == oracle.reduce_ranks(ins, code, oracle.SUM).tobytes()


@pytest.mark.parametrize("seed", [1, 2, 3, 12345])
def test_malloc_block_bookkeeping(seed):
    """xmpi_malloc's arena allocator (heap.cpp), host logic only: random allocate / free traffic never yields
    overlapping or misaligned blocks, refuses double frees and coalesces back into one free block"""
    assert xmpi.lib().xmpi_heap_selftest(seed, 20000) == 0


def test_allreduce_oneshot_threshold():
    """Above oneshot_bytes the two-phase form (reduce-scatter, then allgather) is used."""
    big = plan_sim.get_plans(xmpi.COLL_ALLREDUCE, xmpi.ALGO_DIRECT, 4, 0, 4096, 4, 1, 4096, oneshot_bytes=4096)
    small = plan_sim.get_plans(xmpi.COLL_ALLREDUCE, xmpi.ALGO_DIRECT, 4, 0, 1024, 4, 1, 4096, oneshot_bytes=4096)
    assert 2 in {s.kind for s in big[0].steps}       # RECV_COPY of the allgather phase
    assert 2 not in {s.kind for s in small[0].steps}


@pytest.mark.parametrize("n", [2, 4, 8])
@pytest.mark.parametrize("algo", [xmpi.ALGO_RING, xmpi.ALGO_RHD])
def test_allreduce_f32_tolerance(n, algo):
    """Ring / halving change the summation order: |delta| <= 1e-6 * sum_i |x_i| (BASELINE.md cfg 4)."""
    count = 4096
    ins = [oracle.fill(count, oracle.F32, 0, 1000 + r) for r in range(n)]
    want = oracle.reduce_ranks(ins, oracle.F32, oracle.SUM).astype(np.float64)
    bound = 1e-6 * np.sum([np.abs(x.astype(np.float64)) for x in ins], axis=0)
    plans = plan_sim.get_plans(xmpi.COLL_ALLREDUCE, algo, n, 0, count, 4, 4, 128)
    got = plan_sim.simulate(plans, ins, count, np.float32, xmpi.SUM, 4, seed=11)
    for r in range(n):
        assert np.all(np.abs(got[r].astype(np.float64) - want) <= bound)


@pytest.mark.parametrize("algo", [xmpi.ALGO_RING, xmpi.ALGO_DIRECT])
@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("count,piece", [(1, 2), (5, 2), (1000, 64), (4099, 256)])
def test_allgather_exact(algo, n, count, piece):
    ins = [oracle.fill(count, oracle.I64, 1, r) for r in range(n)]  # (r << 40) | i  (BASELINE cfg 3)
    want = oracle.allgather(ins, oracle.I64)
    for channels in (1, 4):
        plans = plan_sim.get_plans(xmpi.COLL_ALLGATHER, algo, n, 0, count, 8, channels, piece)
        got = plan_sim.simulate(plans, ins, count * n, np.int64, xmpi.SUM, 2, seed=n)
        for r in range(n):
            assert np.array_equal(got[r], want)
        got = plan_sim.simulate(plans, ins, count * n, np.int64, xmpi.SUM, 3, seed=n, inplace=True,
                                send_shift_bytes=count * 8)
        for r in range(n):
            assert np.array_equal(got[r], want)


@pytest.mark.parametrize("algo", [xmpi.ALGO_TREE, xmpi.ALGO_DIRECT, xmpi.ALGO_AUTO])
@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("root", [0, 1, -1])
def test_bcast(algo, n, root):
    """binary tree, and the full-mesh form (root scatters chunk j to rank j, every rank forwards its chunk)"""
    root = root % n
    for count, piece in ((777, 64), (5, 64), (2053, 32)):
        ins = [oracle.fill(count, oracle.I32, 0, 10 + r) for r in range(n)]
        plans = plan_sim.get_plans(xmpi.COLL_BCAST, algo, n, root, count, 4, 1, piece)
        for depth in (1, 4):
            got = plan_sim.simulate(plans, ins, count, np.int32, xmpi.SUM, depth, seed=5 + depth, inplace=True)
            for r in range(n):
                assert np.array_equal(got[r], ins[root])
    if algo == xmpi.ALGO_DIRECT and n > 2:  # two hops: nobody receives anything that travelled further
        p = plans[(root + 1) % n]
        assert {s.kind for s in p.steps} <= {0, 2}


@pytest.mark.parametrize("algo", [xmpi.ALGO_TREE, xmpi.ALGO_DIRECT, xmpi.ALGO_AUTO])
@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("root", [0, -1])
@pytest.mark.parametrize("oneshot", [0, 1 << 20])  # DIRECT: reduce-scatter + gather above, everything-to-root below
def test_reduce_to_root(algo, n, root, oneshot):
    root = root % n
    for count, piece in ((1500, 128), (3, 128), (2053, 32)):
        ins = [oracle.fill(count, oracle.I64, 0, 77 + r) for r in range(n)]
        want = oracle.reduce_ranks(ins, oracle.I64, oracle.SUM)
        plans = plan_sim.get_plans(xmpi.COLL_REDUCE, algo, n, root, count, 8, 1, piece, oneshot_bytes=oneshot)
        for depth in (1, 4):
            got = plan_sim.simulate(plans, ins, count, np.int64, xmpi.SUM, depth, seed=9 + depth)
            assert np.array_equal(got[root], want)
        # the root's buffers may be one and the same
        got = plan_sim.simulate(plans, ins, count, np.int64, xmpi.SUM, 2, seed=4, inplace=True)
        assert np.array_equal(got[root], want)


@pytest.mark.parametrize("oneshot", [0, 1 << 20])
def test_reduce_direct_is_rank_order_f32(oneshot):
    n, count = 8, 1025
    ins = [oracle.fill(count, oracle.F32, 3, 5 + r) for r in range(n)]
    want = oracle.reduce_ranks(ins, oracle.F32, oracle.SUM)
    plans = plan_sim.get_plans(xmpi.COLL_REDUCE, xmpi.ALGO_DIRECT, n, 3, count, 4, 1, 64, oneshot_bytes=oneshot)
    got = plan_sim.simulate(plans, ins, count, np.float32, xmpi.SUM, 2, seed=1)
    assert got[3].tobytes() == want.tobytes()
    if oneshot == 0:  # the fold is spread over the ranks: the root folds one chunk only
        folds = [s for s in plans[3].steps if s.kind == 4]
        assert sum(s.nbytes for s in folds) <= (count // n + 16) * 4


def test_min_max_prod_through_ring():
    n, count = 4, 513
    ins = [oracle.fill(count, oracle.I32, 0, 900 + r) for r in range(n)]
    for op in (oracle.MIN, oracle.MAX, oracle.PROD):
        want = oracle.reduce_ranks(ins, oracle.I32, op)
        plans = plan_sim.get_plans(xmpi.COLL_ALLREDUCE, xmpi.ALGO_RING, n, 0, count, 4, 2, 32)
        got = plan_sim.simulate(plans, ins, count, np.int32, op, 2, seed=op)
        for r in range(n):
            assert np.array_equal(got[r], want)


def test_headline_plan_shape():
    """BASELINE cfg 4: 256 MiB f32, 8 ranks, ring, 4 channels: 14 steps per channel, every rank sends and
    receives 2*(N-1)/N*S bytes, over 4 distinct neighbours in each direction."""
    n, count = 8, 64 << 20
    text = xmpi.plan_text(xmpi.COLL_ALLREDUCE, xmpi.ALGO_RING, n, 3, 0, count, 4, 4, (2 << 20) // 4)
    plan = plan_sim.parse_plan(text)
    # kinds: 0 SEND, 1 RECV_REDUCE, 2 RECV_COPY, 6 RECV_REDUCE_SEND, 7 RECV_COPY_SEND (fused pop + push)
    sent = sum(s.nbytes for s in plan.steps if s.kind in (0, 6, 7))
    recvd = sum(s.nbytes for s in plan.steps if s.kind in (1, 2, 6, 7))
    assert sent == recvd == 2 * (n - 1) * (count * 4) // n
    assert len({s.peer if s.kind == 0 else s.peer2 for s in plan.steps if s.kind in (0, 6, 7)}) == 4
    assert len({s.peer for s in plan.steps if s.kind in (1, 2, 6, 7)}) == 4
    assert max(s.nbytes for s in plan.steps) <= 2 << 20
    # fused ring: 2N-1 launches per piece instead of 4(N-1)
    pieces = len([s for s in plan.steps if s.kind == 0])
    assert len(plan.steps) == pieces * (2 * n - 1)


@pytest.mark.parametrize("n", [4, 6, 8])
def test_ring_channels_use_distinct_links(n):
    """Even N: the N-2 ring channels are the two directions of N/2-1 edge-disjoint Hamiltonian cycles
    (Walecki): every rank sends to N-2 different peers and no directed link carries two channels."""
    links = {}
    for r in range(n):
        text = xmpi.plan_text(xmpi.COLL_ALLREDUCE, xmpi.ALGO_RING, n, r, 0, n << 20, 4, n - 2, 1 << 16)
        plan = plan_sim.parse_plan(text)
        assert plan.channels == n - 2
        nxt = {s.peer for s in plan.steps if s.kind == 0}
        prv = {s.peer for s in plan.steps if s.kind in (1, 2)}
        assert len(nxt) == n - 2 and len(prv) == n - 2 and r not in nxt
        for peer in nxt:
            links[(r, peer)] = links.get((r, peer), 0) + 1
    assert len(links) == n * (n - 2) and set(links.values()) == {1}
    # both directions of every used link are used (bidirectional xGMI links fully loaded)
    assert all((b, a) in links for (a, b) in links)


def test_ring_channels_odd_world():
    for r in range(5):
        text = xmpi.plan_text(xmpi.COLL_ALLREDUCE, xmpi.ALGO_RING, 5, r, 0, 5 << 20, 4, 4, 1 << 16)
        nxt = {s.peer for s in plan_sim.parse_plan(text).steps if s.kind == 0}
        assert nxt == {(r + d) % 5 for d in (1, 2, 3, 4)}
