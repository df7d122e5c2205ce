"""The LL small-collective protocol (mpi_amd/csrc/ll.hip) under random interleavings on the CPU (tests/ll_sim.py): the
parity argument of kernels.h -- no half-line is overwritten under a reader, every accepted line is this epoch's, every
kernel ends -- for mixed programs (LL allreduce / allgather / broadcast / reduce and zero-copy folds), several
communicators in a row on the same uncleared pages; and the known-bad variants are caught."""
import itertools
import random

import pytest

from tests import ll_sim

MIXED = [("ar", 0), ("bc", 0), ("bc", 0), ("bc", 0), ("ar", 0), ("fold", 0), ("bc", 1), ("rd", 1), ("rd", 1), ("ag", 0),
         ("fold", 0), ("fold", 0), ("bc", 0), ("ar", 0), ("rd", 0), ("bc", 1), ("ar", 0)]


@pytest.mark.parametrize("n", [2, 3, 4, 8])
def test_mixed_program_holds_under_random_interleavings(n):
    for seed in range(12):
        ll_sim.run(n, lines=2, program=MIXED, seed=seed, comms=2)


def test_random_programs():
    rng = random.Random(7)
    for trial in range(40):
        n = rng.choice([2, 3, 5])
        prog = [(rng.choice(["ar", "ag", "bc", "rd", "fold"]), rng.randrange(n)) for _ in range(rng.randrange(3, 14))]
        ll_sim.run(n, lines=rng.choice([1, 3]), program=prog, seed=trial, comms=rng.choice([1, 3]))


def test_runs_of_broadcasts_from_one_root():
    for n, seed in itertools.product([2, 4], range(10)):
        ll_sim.run(n, lines=2, program=[("bc", 0)] * 12, seed=seed)
        ll_sim.run(n, lines=2, program=[("rd", n - 1)] * 12, seed=seed)


@pytest.mark.parametrize("bug,program", [
    ("no_here", [("bc", 0)] * 10),                 # the root laps a slow reader
    ("no_here", [("ar", 0), ("bc", 0), ("ar", 0)] * 4),  # ... and so does a rank behind a broadcast
    ("one_slot", [("ar", 0)] * 8),                 # without the parity a fast rank overwrites the line a slow one still needs
    ("flag_on_first_half_only", [("ar", 0)] * 8),  # a torn 16-byte load
    ("no_epoch_base", [("ar", 0)] * 4),            # a later communicator takes stale lines for its own
])
def test_known_bad_variants_are_caught(bug, program):
    caught = 0
    for seed in range(60):
        try:
            ll_sim.run(3, lines=2, program=program, seed=seed, comms=2, bugs=(bug,))
        except ll_sim.Violation:
            caught += 1
    assert caught > 0, f"the checker does not notice '{bug}'"
