"""The stream-ordered Send / Receive protocol under random interleavings -- see tests/p2p_sim.py (CPU only)."""
import pytest

from tests import p2p_sim as sim


def test_ping_pong_and_more_messages_than_boxes():
    for seed in range(20):
        # rank 0 sends 30 messages on one stream, rank 1 receives them in order; then the echo
        ops = {0: [[("send", 1, 100 + k, f"m{k}") for k in range(30)] + [("recv", 1, 7)]],
               1: [[("recv", 0, 100 + k) for k in range(30)] + [("send", 0, 7, "echo")]]}
        got = sim.run(ops, seed=seed)
        assert [got[(1, 0, k)] for k in range(30)] == [f"m{k}" for k in range(30)] and got[(0, 0, 30)] == "echo"


def test_concurrent_streams_tags_in_any_order():
    """several streams per rank: messages with distinct tags overtake each other, every receive still gets ITS message"""
    for seed in range(40):
        ops = {0: [[("send", 1, t, f"a{t}")] for t in (1, 2, 3, 4)] + [[("recv", 1, 9)]],
               1: [[("recv", 0, t)] for t in (4, 2, 3, 1)] + [[("send", 0, 9, "z")]]}
        got = sim.run(ops, seed=seed)
        for si, t in enumerate((4, 2, 3, 1)):
            assert got[(1, si, 0)] == f"a{t}"
        assert got[(0, 4, 0)] == "z"


def test_the_same_tag_twice_is_first_come_first_served():
    for seed in range(20):
        ops = {0: [[("send", 1, 5, "first"), ("send", 1, 5, "second")]], 1: [[("recv", 0, 5), ("recv", 0, 5)]]}
        got = sim.run(ops, seed=seed)
        assert (got[(1, 0, 0)], got[(1, 0, 1)]) == ("first", "second")


def test_three_ranks_every_pair():
    for seed in range(20):
        ops = {r: [[("send", p, 10 * r + p, f"{r}->{p}") for p in range(3) if p != r], [("recv", p, 10 * p + r) for p in range(3) if p != r]]
               for r in range(3)}
        got = sim.run(ops, seed=seed)
        for r in range(3):
            assert [got[(r, 1, i)] for i in range(2)] == [f"{p}->{r}" for p in range(3) if p != r]


def test_a_stream_that_waits_for_work_behind_it_deadlocks():
    """the documented rule, shown: a send whose receive is queued BEHIND a receive that needs the peer's later send"""
    ops = {0: [[("send", 1, 1, "x"), ("send", 1, 2, "y")]], 1: [[("recv", 0, 2), ("recv", 0, 1)]]}
    with pytest.raises(sim.Violation, match="deadlock"):
        sim.run(ops, seed=1)


def test_records_of_an_earlier_communicator_are_not_matched():
    """pages are pooled and never cleared: a box may still hold a message of the communicator before, even one that was
    never consumed (its job was aborted) -- the communicator number in the message number keeps it out"""
    stale = {(0, 1): [((1 << 32) | (b + 1), 5) for b in range(sim.K)]}  # communicator 1 left 8 messages with tag 5
    for seed in range(10):
        ops = {0: [[("send", 1, 5, "fresh")]], 1: [[("recv", 0, 5)]]}
        got = sim.run(ops, seed=seed, comm_tag=2, stale=stale, bugs=("stale_unconsumed",))
        assert got[(1, 0, 0)] == "fresh"
    with pytest.raises(AssertionError):  # ... and the checker notices when the communicator number is not looked at
        for seed in range(10):
            got = sim.run({0: [[("send", 1, 5, "fresh")]], 1: [[("recv", 0, 5)]]}, seed=seed, comm_tag=2, stale=stale,
                          bugs=("stale_unconsumed", "no_comm_check"))
            assert got[(1, 0, 0)] == "fresh"


def test_the_checker_notices_a_missing_box_wait():
    """a sender that does not wait for the previous occupant's ack overwrites an unconsumed message once the boxes wrap"""
    ops = {0: [[("send", 1, k, k)] for k in range(12)], 1: [[("recv", 0, k) for k in reversed(range(12))]]}
    bad = 0
    for seed in range(30):
        try:
            sim.run(ops, seed=seed, bugs=("no_box_wait",))
        except sim.Violation:
            bad += 1
    assert bad > 0
