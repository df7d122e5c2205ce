"""The stepped kernels' step programs (ring / halving / allgather / tree), executed on the CPU under random
interleavings -- see tests/sched_sim.py.  No GPU needed: the programs come from libxmpi's host-callable step function."""
import pytest

from mpi_amd import xmpi
from tests import sched_sim as sim


def check(sched, size, count, **kw):
    root = kw.get("root", 0)
    recv, orig = sim.run(sched, size, count, **kw)
    want = sim.expected(sched, size, count, orig, root)
    for r in range(size):
        if want[r] is None:  # (a rank whose receive buffer means nothing: the non-roots of a reduce)
            continue
        got = list(recv[r])
        assert got == want[r], f"rank {r}: sched {sched} N={size} count={count} {kw}: first diff at " \
                               f"{next(i for i in range(len(got)) if got[i] != want[r][i])}"


@pytest.mark.parametrize("size", [2, 3, 4, 5, 6, 7, 8, 9])
@pytest.mark.parametrize("inplace", [False, True])
def test_ring_allreduce(size, inplace):
    for seed, (count, nchan, gx) in enumerate([(1, 1, 1), (size * 4, 1, 2), (37, 2, 2), (101, min(3, max(1, size - 2)), 3)]):
        nchan = min(nchan, max(1, size - 2)) if size >= 4 and size % 2 == 0 else 1
        check(xmpi.SCHED_RING_ALLREDUCE, size, count, nchan=nchan, gx=gx, inplace=inplace, seed=seed)
        check(xmpi.SCHED_RING_ALLREDUCE, size, count, nchan=nchan, gx=gx, inplace=inplace, seed=100 + seed, bias=seed % size)


# (in place beyond 9 ranks only where the shape changes: the CPU suite's time; the device runs these kernels with N <= 8)
@pytest.mark.parametrize("size,inplace", [(n, False) for n in range(2, 17)] + [(n, True) for n in (2, 3, 4, 5, 6, 7, 8, 9, 12, 15, 16)])
def test_recursive_halving_doubling(size, inplace):
    """any number of ranks: with no power of two the first 2 (N - 2^l) ranks pair up in a fold-in step, the even ones sit
    out the halving and doubling and fetch the result in a fold-out step"""
    for seed, (count, gx) in enumerate([(1, 1), (size, 2), (53, 3), (64, 4), (131, 5)]):
        check(xmpi.SCHED_RHD_ALLREDUCE, size, count, gx=gx, inplace=inplace, seed=seed)
        check(xmpi.SCHED_RHD_ALLREDUCE, size, count, gx=gx, inplace=inplace, seed=50 + seed, bias=(seed * 3) % size)


def test_halving_step_counts():
    for n, steps in ((2, 2), (3, 4), (4, 4), (5, 6), (6, 6), (7, 6), (8, 6), (9, 8), (16, 8)):
        for rank in range(n):
            text = xmpi.sched_text(xmpi.SCHED_RHD_ALLREDUCE, n, rank, 0, 1, 1000, 4, 1, 0)
            assert len(text.strip().split("\n")) == steps, (n, rank)


@pytest.mark.parametrize("size,inplace", [(n, False) for n in (2, 3, 4, 5, 6, 7, 8, 9, 12, 13, 16)] + [(n, True) for n in (2, 3, 4, 5, 7, 8, 13)])
def test_tree_reduce(size, inplace):
    """every node folds its children's partial results into its own, piece by piece; only the root's receive buffer counts"""
    for seed, (count, pieces, gx) in enumerate([(1, 1, 1), (40, 1, 2), (100, 4, 2), (257, 8, 3)]):
        for root in {0, size - 1, size // 2}:
            check(xmpi.SCHED_TREE_REDUCE, size, count, pieces=pieces, gx=gx, root=root, inplace=inplace, seed=seed)
            check(xmpi.SCHED_TREE_REDUCE, size, count, pieces=pieces, gx=gx, root=root, inplace=inplace, seed=seed + 20,
                  bias=(root + 1) % size)


@pytest.mark.parametrize("size", [2, 3, 4, 7, 8])
def test_ring_allgather(size):
    for seed, (count, es, nchan, gx) in enumerate([(1, 4, 1, 1), (5, 8, 1, 2), (33, 4, 2, 2), (64, 1, 1, 3)]):
        nchan = nchan if size >= 4 and size % 2 == 0 else 1
        check(xmpi.SCHED_RING_ALLGATHER, size, count, es=es, nchan=nchan, gx=gx, seed=seed)
        check(xmpi.SCHED_RING_ALLGATHER, size, count, es=es, nchan=nchan, gx=gx, seed=seed + 9, bias=size - 1)


@pytest.mark.parametrize("size", [2, 3, 5, 8, 13])
def test_tree_bcast(size):
    for seed, (count, pieces, gx) in enumerate([(1, 1, 1), (40, 1, 2), (100, 4, 2), (257, 8, 3)]):
        for root in {0, size - 1, size // 2}:
            check(xmpi.SCHED_TREE_BCAST, size, count, pieces=pieces, gx=gx, root=root, seed=seed)
            check(xmpi.SCHED_TREE_BCAST, size, count, pieces=pieces, gx=gx, root=root, seed=seed + 20, bias=(root + 1) % size)


def test_the_checker_notices_a_missing_wait(monkeypatch):
    """the same programs with every wait removed: some interleaving must read a region before it is complete"""
    real = sim.program

    def no_waits(*a):
        steps = real(*a)
        for s in steps:
            s["wait"] = (-1, 0)
        return steps

    monkeypatch.setattr(sim, "program", no_waits)
    bad = 0
    for seed in range(6):
        recv, orig = sim.run(xmpi.SCHED_RING_ALLREDUCE, 4, 40, gx=2, seed=seed, bias=seed % 4)
        want = sim.expected(xmpi.SCHED_RING_ALLREDUCE, 4, 40, orig)
        bad += any(list(recv[r]) != want[r] for r in range(4))
    assert bad > 0


def test_tuner_decision_function():
    """xmpi_tune_decide: the fastest candidate, but the default (index 0) stays unless beaten by more than the margin;
    candidates that did not run (<= 0) never win; nothing ran: -1"""
    d = xmpi.tune_decide
    assert d([100.0, 90.0, 120.0], 0.03) == 1
    assert d([100.0, 98.0, 120.0], 0.03) == 0          # 2 % is inside the noise margin: the default stays
    assert d([100.0, 97.0 - 1e-9, 120.0], 0.03) == 1   # just beyond it
    assert d([100.0, 0.0, -1.0, 50.0], 0.03) == 3      # not-run candidates are ignored
    assert d([0.0, 80.0, 70.0], 0.03) == 2             # the default itself did not run
    assert d([0.0, 0.0], 0.03) == -1
    assert d([5.0], 0.5) == 0
    assert d([10.0, 9.0], 0.0) == 1                    # no margin: plain argmin
    assert d([10.0, 10.0], 0.0) == 0                   # a tie keeps the default


def test_sched_dump_rejects_bad_arguments():
    L = xmpi.lib()
    assert L.xmpi_sched_dump(9, 4, 0, 0, 1, 100, 4, 1, 0, None, 0) == xmpi.ERR_ARG      # no such schedule
    assert L.xmpi_sched_dump(5, 4, 0, 0, 128, 100, 4, 1, 0, None, 0) == xmpi.ERR_ARG    # more pieces than step numbers
    assert L.xmpi_sched_dump(1, 4, 4, 0, 1, 100, 4, 1, 0, None, 0) == xmpi.ERR_ARG      # rank out of range
    assert L.xmpi_sched_dump(1, 4, 0, 0, 1, 100, 4, 9, 0, None, 0) == xmpi.ERR_ARG      # more channels than the kernel has
    assert L.xmpi_sched_dump(4, 5, 0, 7, 1, 100, 1, 1, 0, None, 0) == xmpi.ERR_ARG      # root out of range
    text = xmpi.sched_text(xmpi.SCHED_RING_ALLREDUCE, 8, 3, 0, 1, 1 << 20, 4, 2, 1)
    assert len(text.strip().split("\n")) == 14  # 2 (N - 1) steps


def test_ring_channels_are_different_cycles():
    """every ring channel of the stepped kernel is another cyclic order of the ranks: on an even mesh no two channels send
    over the same directed link (plan.cpp ring_order, Walecki's decomposition)"""
    for n in (4, 6, 8):
        nchan = min(n - 2, 8)
        links = set()
        for ch in range(nchan):
            for rank in range(n):
                first = sim.program(xmpi.SCHED_RING_ALLREDUCE, n, rank, 0, 1, n * 64, 4, nchan, ch)[1]  # step 2 waits for the previous rank
                prev = first["wait"][0]
                assert (prev, rank, ch) not in links
                assert not any((prev, rank, c2) in links for c2 in range(nchan) if c2 != ch), f"N={n}: link {prev}->{rank} used by two channels"
                links.add((prev, rank, ch))
