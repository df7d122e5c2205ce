"""The rendezvous / completion protocol of the device-synchronised collectives, model-checked on the CPU
(tests/dsync_sim.py): random interleavings of every block of every rank's kernels, several communicators in a row on
pooled, uncleared flag pages.  The real kernels run in the GPU suite; this is the N > 1 logic without a GPU."""
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st

from tests import dsync_sim


@pytest.mark.parametrize("n", [2, 3, 4, 8])
@pytest.mark.parametrize("blocks", [1, 2, 5])
def test_protocol_is_safe_and_live(n, blocks):
    for seed in range(6):
        dsync_sim.run(n, blocks, epochs_per_comm=4, seed=seed, comms=3)


@settings(max_examples=60, deadline=None)
@given(n=st.integers(2, 6), blocks=st.integers(1, 4), epochs=st.integers(1, 5), comms=st.integers(1, 3), seed=st.integers(0, 10**9))
def test_protocol_property(n, blocks, epochs, comms, seed):
    dsync_sim.run(n, blocks, epochs, seed, comms)


def _caught(bug, **kw):
    hits = 0
    for seed in range(40):
        try:
            dsync_sim.run(seed=seed, bugs=(bug,), **kw)
        except dsync_sim.Violation:
            hits += 1
    return hits


def test_the_checker_notices_a_completion_signalled_too_early():
    """a block that says "done" before every block of its kernel has finished lets a peer take its buffers back early"""
    assert _caught("early_done", n=3, blocks=4, epochs_per_comm=3, comms=1) > 0


def test_the_checker_notices_epochs_restarting_on_an_uncleared_page():
    """a second communicator whose epochs restart at 1 on pages that still hold the first one's flags sails through its
    waits and reads stale buffer references"""
    assert _caught("no_epoch_base", n=3, blocks=2, epochs_per_comm=3, comms=2) > 0


def test_the_checker_notices_a_kernel_that_leaves_without_waiting_for_the_peers():
    """a kernel that ends right after saying "done" hands its buffers back while slower peers still read and write them"""
    assert _caught("no_done_wait", n=3, blocks=2, epochs_per_comm=3, comms=1) > 0
