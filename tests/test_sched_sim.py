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


PUSH = pytest.mark.parametrize("push", [False, True], ids=["pull", "push"])


@pytest.mark.parametrize("size", [2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("inplace", [False, True])
@PUSH
def test_ring_allreduce(size, inplace, push):
    """push, out of place: partial results land in the next rank's receive buffer; in place: in the landing block it lends"""
    for seed, (count, nchan, gx) in enumerate([(1, 1, 1), (size * 4, 1, 2), (37, 2, 2), (101, min(3, max(1, size - 2)), 3)]):
        nchan = min(nchan, max(1, size - 2)) if size >= 4 and size % 2 == 0 else 1
        check(xmpi.SCHED_RING_ALLREDUCE, size, count, nchan=nchan, gx=gx, inplace=inplace, seed=seed, push=push)
        check(xmpi.SCHED_RING_ALLREDUCE, size, count, nchan=nchan, gx=gx, inplace=inplace, seed=100 + seed, bias=seed % size, push=push)


@pytest.mark.parametrize("size", [2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("inplace", [False, True])
@PUSH
def test_recursive_halving_doubling(size, inplace, push):
    """any number of ranks: with no power of two the first 2 (N - 2^l) ranks pair up in a fold-in step, the even ones sit
    out the halving and doubling and get the result in a fold-out step.  push: every level's half goes into a landing region of
    its own at the partner, which folds it while it cuts its accumulator for the next level"""
    for seed, (count, gx) in enumerate([(1, 1), (size, 2), (53, 3), (64, 4), (131, 5)]):
        check(xmpi.SCHED_RHD_ALLREDUCE, size, count, gx=gx, inplace=inplace, seed=seed, push=push)
        check(xmpi.SCHED_RHD_ALLREDUCE, size, count, gx=gx, inplace=inplace, seed=50 + seed, bias=(seed * 3) % size, push=push)


def test_halving_step_counts():
    for n, steps in ((2, 2), (3, 4), (4, 4), (5, 6), (6, 6), (7, 6), (8, 6)):
        for rank in range(n):
            text = xmpi.sched_text(xmpi.SCHED_RHD_ALLREDUCE, n, rank, 0, 1, 1000, 4, 1, 0)
            assert len(text.strip().split("\n")) == steps, (n, rank)
            text = xmpi.sched_text(xmpi.SCHED_RHD_ALLREDUCE, n, rank, 0, 1, 1000, 4, 1, 0, push=True)
            assert len(text.strip().split("\n")) == steps + 1, (n, rank)  # (the push form's last step waits for the last range)


@pytest.mark.parametrize("size", [2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("inplace", [False, True])
@PUSH
def test_tree_reduce(size, inplace, push):
    """every node folds its children's partial results into its own, piece by piece; only the root's receive buffer counts.
    push: the children store theirs into the parent's landing slots, the parent folds both in one pass and stores upwards"""
    for seed, (count, pieces, gx) in enumerate([(1, 1, 1), (40, 1, 2), (100, 4, 2), (257, 8, 3)]):
        for root in {0, size - 1, size // 2}:
            check(xmpi.SCHED_TREE_REDUCE, size, count, pieces=pieces, gx=gx, root=root, inplace=inplace, seed=seed, push=push)
            check(xmpi.SCHED_TREE_REDUCE, size, count, pieces=pieces, gx=gx, root=root, inplace=inplace, seed=seed + 20,
                  bias=(root + 1) % size, push=push)


@pytest.mark.parametrize("size", [2, 3, 4, 7, 8])
@pytest.mark.parametrize("inplace", [False, True])
@PUSH
def test_ring_allgather(size, inplace, push):
    for seed, (count, es, nchan, gx) in enumerate([(1, 4, 1, 1), (5, 8, 1, 2), (33, 4, 2, 2), (64, 1, 1, 3)]):
        nchan = nchan if size >= 4 and size % 2 == 0 else 1
        check(xmpi.SCHED_RING_ALLGATHER, size, count, es=es, nchan=nchan, gx=gx, seed=seed, push=push, inplace=inplace)
        check(xmpi.SCHED_RING_ALLGATHER, size, count, es=es, nchan=nchan, gx=gx, seed=seed + 9, bias=size - 1, push=push, inplace=inplace)


@pytest.mark.parametrize("size", [2, 3, 5, 8])
@PUSH
def test_tree_bcast(size, push):
    for seed, (count, pieces, gx) in enumerate([(1, 1, 1), (40, 1, 2), (100, 4, 2), (257, 8, 3)]):
        for root in {0, size - 1, size // 2}:
            check(xmpi.SCHED_TREE_BCAST, size, count, pieces=pieces, gx=gx, root=root, seed=seed, push=push)
            check(xmpi.SCHED_TREE_BCAST, size, count, pieces=pieces, gx=gx, root=root, seed=seed + 20, bias=(root + 1) % size, push=push)


@pytest.mark.parametrize("size", [2, 3, 4, 5, 6, 7, 8])
def test_push_and_pull_build_the_same_terms(size):
    """the elements as TERMS instead of numbers: both forms of a schedule combine the same operands in the same order and
    association -- the same floating-point bits, whatever the data (what the GPU suite then checks on the data)"""
    for sched, kw in ((xmpi.SCHED_RING_ALLREDUCE, dict(nchan=2 if size in (4, 6, 8) else 1)), (xmpi.SCHED_RHD_ALLREDUCE, {}),
                      (xmpi.SCHED_TREE_REDUCE, dict(pieces=3, root=size // 2)), (xmpi.SCHED_TREE_REDUCE, dict(pieces=1, root=0))):
        for inplace in (False, True):
            root = kw.get("root", 0)
            pull, _ = sim.run(sched, size, 67, gx=2, seed=3, symbolic=True, inplace=inplace, **kw)
            push, _ = sim.run(sched, size, 67, gx=2, seed=4, symbolic=True, inplace=inplace, push=True, **kw)
            for r in range(size):
                if sched == xmpi.SCHED_TREE_REDUCE and r != root:
                    continue
                assert list(pull[r]) == list(push[r]), (sched, size, r, inplace)
                if sched != xmpi.SCHED_TREE_REDUCE:
                    assert list(pull[r]) == list(pull[0]), "every rank holds the same term"


def test_no_push_step_leaves_the_block_it_stores_into():
    """at the sizes the kernels really run (up to 1 GiB, odd counts, every dtype width, every N): every landing-block address of every
    step of every rank's push program lies inside the block that rank lends, every receive-buffer address inside the buffer -- from
    the step function's own output, no simulation (the simulator checks the same at its small sizes)"""
    import random
    rng = random.Random(7)
    cases = [(xmpi.SCHED_RHD_ALLREDUCE, n, c, es, ip) for n in range(2, 9) for c, es, ip in ((1, 4, False), (268435456, 4, True), (536870912, 2, False), (67108864 + 3, 4, True))]
    for _ in range(60):
        sched = rng.choice([xmpi.SCHED_RING_ALLREDUCE, xmpi.SCHED_RHD_ALLREDUCE, xmpi.SCHED_TREE_REDUCE, xmpi.SCHED_RING_ALLGATHER, xmpi.SCHED_TREE_BCAST])
        cases.append((sched, rng.randint(2, 8), rng.choice([1, 3, 17, 4099, rng.randint(1, 1 << 28)]), rng.choice([1, 2, 4, 8]), rng.random() < 0.5))
    for sched, n, count, es, inplace in cases:
        root = rng.randrange(n)
        pieces = rng.choice([1, 3, 32]) if sched in (xmpi.SCHED_TREE_REDUCE, xmpi.SCHED_TREE_BCAST) else 1
        whole = count * es * (n if sched == xmpi.SCHED_RING_ALLGATHER else 1)
        lent = [xmpi.sched_land_bytes(sched, n, r, root, count, es, inplace) for r in range(n)]
        for r in range(n):
            for st in sim.program(sched, n, r, root, pieces, count, es, 1, 0, True, inplace):
                for m in st["moves"]:
                    assert 0 <= m["lo"] < m["hi"] <= whole, (sched, n, count, es, st)
                    for name in ("D", "D2", "A", "B", "C"):
                        ref = m[name]
                        if ref is None:
                            continue
                        owner, kind, off = ref
                        lo, hi = m["lo"] + off, m["hi"] + off
                        if kind == "l":
                            assert 0 <= lo and hi <= lent[owner], (sched, n, count, es, inplace, r, st["g"], name, ref, lo, hi, lent[owner])
                        elif kind == "r":
                            assert 0 <= lo and hi <= whole, (sched, n, count, es, r, st["g"], name, ref)
                        else:
                            assert 0 <= lo and hi <= count * es, (sched, n, count, es, r, st["g"], name, ref)


def test_landing_blocks_are_as_small_as_the_plan_says():
    """ring allreduce: none out of place, one buffer in place; halving: less than a buffer (+ one for the fold-in of an odd
    pair), nothing for a rank that sits out; tree reduce: a buffer per child; allgather and bcast: none"""
    S, es = 4000, 4
    for n in range(2, 9):
        for r in range(n):
            assert xmpi.sched_land_bytes(xmpi.SCHED_RING_ALLREDUCE, n, r, 0, S // es, es) == 0
            assert S <= xmpi.sched_land_bytes(xmpi.SCHED_RING_ALLREDUCE, n, r, 0, S // es, es, inplace=True) <= S + 32
            assert xmpi.sched_land_bytes(xmpi.SCHED_RING_ALLGATHER, n, r, 0, S // es, es) == 0
            assert xmpi.sched_land_bytes(xmpi.SCHED_TREE_BCAST, n, r, 0, S // es, es) == 0
            kids = (2 * r + 1 < n) + (2 * r + 2 < n)
            assert kids * S <= xmpi.sched_land_bytes(xmpi.SCHED_TREE_REDUCE, n, r, 0, S // es, es) <= kids * (S + 32)
            pow2 = n & (n - 1) == 0
            P = 1 << (n.bit_length() - 1)
            sits_out = r < 2 * (n - P) and r % 2 == 0
            got = xmpi.sched_land_bytes(xmpi.SCHED_RHD_ALLREDUCE, n, r, 0, S // es, es)
            if sits_out:
                assert got == 0
            else:
                assert S - S // P <= got - (0 if pow2 else S + 16) <= S + 80 * 3, (n, r, got)


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


def test_the_checker_notices_a_push_form_without_its_waits_or_its_landing_block(monkeypatch):
    """the push ring with every wait removed: somebody folds a chunk that has not landed; and the in-place push ring run with the
    out-of-place program (partial results stored into the next rank's receive buffer, which is its input): an input is clobbered
    before its owner has combined it -- the landing block is what prevents that"""
    real = sim.program

    def no_waits(*a):
        steps = real(*a)
        for s in steps:
            s["wait"] = (-1, 0)
        return steps

    def no_landing(sched, size, rank, root, pieces, count, es, nchan, ch, push=False, inplace=False):
        return real(sched, size, rank, root, pieces, count, es, nchan, ch, push, False)  # (its 'r' is the send buffer in place)

    for patch in (no_waits, no_landing):
        monkeypatch.setattr(sim, "program", patch)
        bad = 0
        for seed in range(6):
            recv, orig = sim.run(xmpi.SCHED_RING_ALLREDUCE, 4, 40, gx=2, seed=seed, bias=seed % 4, push=True, inplace=patch is no_landing)
            want = sim.expected(xmpi.SCHED_RING_ALLREDUCE, 4, 40, orig)
            bad += any(list(recv[r]) != want[r] for r in range(4))
        assert bad > 0, patch.__name__


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
    assert L.xmpi_sched_dump(9, 0, 0, 4, 0, 0, 1, 100, 4, 1, 0, None, 0) == xmpi.ERR_ARG      # no such schedule
    assert L.xmpi_sched_dump(5, 0, 0, 4, 0, 0, 128, 100, 4, 1, 0, None, 0) == xmpi.ERR_ARG    # more pieces than step numbers
    assert L.xmpi_sched_dump(1, 0, 0, 4, 4, 0, 1, 100, 4, 1, 0, None, 0) == xmpi.ERR_ARG      # rank out of range
    assert L.xmpi_sched_dump(1, 0, 0, 4, 0, 0, 1, 100, 4, 9, 0, None, 0) == xmpi.ERR_ARG      # more channels than the kernel has
    assert L.xmpi_sched_dump(4, 0, 0, 5, 0, 7, 1, 100, 1, 1, 0, None, 0) == xmpi.ERR_ARG      # root out of range
    assert L.xmpi_sched_dump(1, 2, 0, 4, 0, 0, 1, 100, 4, 1, 0, None, 0) == xmpi.ERR_ARG      # no such form
    text = xmpi.sched_text(xmpi.SCHED_RING_ALLREDUCE, 8, 3, 0, 1, 1 << 20, 4, 2, 1)
    assert len(text.strip().split("\n")) == 14  # 2 (N - 1) steps
    text = xmpi.sched_text(xmpi.SCHED_RING_ALLREDUCE, 8, 3, 0, 1, 1 << 20, 4, 2, 1, push=True)
    assert len(text.strip().split("\n")) == 15  # ... and one that waits for the last chunk


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
