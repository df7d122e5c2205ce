"""The host lanes (engine.cpp p2p_send / p2p_recv, DIRECT_HOST) under random interleavings -- tests/lane_sim.py (CPU only)."""
import pytest

from tests import lane_sim as sim

LENGTHS = [1, 7, 8, 9, 31, 32, 33, 40, 64, 65, 100, 257]  # ring = 4 pieces of 8 units = 32


@pytest.mark.parametrize("device_dst", [False, True])
def test_every_length_arrives_whole(device_dst):
    for n in LENGTHS:
        for seed in range(40):
            s, r, got = sim.run(n, device_dst=device_dst, seed=seed, kernel_turns=2 + 5 * (seed % 12))
            assert (s, r) == (sim.OK, sim.OK) and got == sim.expected(n), (n, seed)


def test_a_late_receiver_finds_the_ring_full_and_drains_it():
    for n in (5, 32, 200):
        for seed in range(40):
            s, r, got = sim.run(n, receiver_delay=300, device_dst=seed % 2 == 1, seed=seed)
            assert (s, r) == (sim.OK, sim.OK) and got == sim.expected(n)


def test_truncation_reaches_both_sides_and_frees_the_entry():
    for n, cap in ((10, 3), (33, 32), (300, 100)):
        for seed in range(60):
            s, r, _ = sim.run(n, capacity=cap, seed=seed, receiver_delay=seed % 50)
            assert (s, r) == (sim.TRUNCATE, sim.TRUNCATE), (n, cap, seed)


def test_an_unanswered_send_withdraws_but_never_a_matched_one():
    for seed in range(80):
        # the receiver shows up around the moment the sender loses patience: either the message is withdrawn (and the
        # receiver finds nothing), or it was matched first and then goes through whole -- never half of each
        s, r, got = sim.run(100, sender_patience=40, receiver_delay=seed * 5, seed=seed)
        if s == sim.TIMEOUT:
            assert r == sim.TIMEOUT
        else:
            assert (s, r) == (sim.OK, sim.OK) and got == sim.expected(100)


def test_the_model_catches_a_sender_that_does_not_wait_for_room():
    caught = 0
    for seed in range(60):
        try:
            s, r, got = sim.run(200, receiver_delay=100, seed=seed, bugs=("no_room_check",))
            caught += got != sim.expected(200)
        except sim.Violation:
            caught += 1
    assert caught >= 50


def test_the_model_catches_a_kernel_run_across_the_wrap():
    caught = 0
    for seed in range(60):
        try:
            # (a slow kernel: the sender gets ahead, and a run that starts in the middle of the ring reaches past its end)
            sim.run(203, device_dst=True, seed=seed, bugs=("run_wraps",), kernel_turns=10 + 7 * (seed % 20))
        except sim.Violation:
            caught += 1
    assert caught >= 30
