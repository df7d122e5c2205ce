"""The receive agent's protocol (sched.hip p2p_agent_kernel / engine.cpp agent_submit), model-checked on the CPU under
random interleavings -- tests/agent_sim.py.  The agent is the one kernel of the library that waits for the HOST; the
hang it had in round 3 (a stale word of the launch before) is kept as a known-bad variant the model must catch."""
import pytest

from tests import agent_sim


@pytest.mark.parametrize("blocks", [1, 2, 4, 8])
def test_every_message_once_and_acked_after_every_block(blocks):
    launches = 0
    for seed in range(150):
        r = agent_sim.run([1, 3, 9, 2, 17, 4, 5, 30], blocks=blocks, seed=seed)
        assert r["served"] == 8
        launches += r["launches"]
    assert launches >= 150  # at least one launch per run; more whenever the patience ran out between two Receives


def test_back_to_back_receives_need_one_launch():
    for seed in range(100):
        r = agent_sim.run([2, 9, 1, 12, 3], blocks=4, patience=50, seed=seed, gaps=[0, 0, 0, 0, 0])
        assert r["launches"] == 1 and r["served"] == 5


def test_a_receive_after_the_patience_ran_out_launches_again():
    for seed in range(100):
        r = agent_sim.run([2, 9, 1], blocks=4, patience=3, seed=seed, gaps=[0, 5000, 5000])
        assert r["launches"] == 3 and r["served"] == 3


def test_the_race_between_a_command_and_the_end_of_the_patience():
    # gaps around the patience: the command lands while the agent decides to go, in every order the scheduler finds
    for seed in range(400):
        for gap in (4, 8, 16, 32):
            r = agent_sim.run([1, 9, 2, 11], blocks=3, patience=2, seed=seed, gaps=[0, gap, gap, gap])
            assert r["served"] == 4


def test_short_after_wide_after_short_across_launches():
    # what the stale word needs: a launch that only served short messages, then a launch whose first message is wide
    for seed in range(200):
        r = agent_sim.run([1, 2, 12, 3, 14], blocks=4, patience=2, seed=seed, gaps=[0, 0, 3000, 3000, 3000])
        assert r["served"] == 5


def test_the_model_finds_the_stale_word_hang():
    caught = 0
    for seed in range(40):
        try:
            agent_sim.run([1, 2, 12, 3, 14], blocks=4, patience=2, seed=seed, gaps=[0, 0, 3000, 3000, 3000], bugs=("stale_key",),
                          max_steps=60000)
        except (agent_sim.Hang, agent_sim.Violation):
            caught += 1
    assert caught >= 30


def test_the_model_finds_an_ack_before_the_last_block():
    caught = 0
    for seed in range(200):
        try:
            agent_sim.run([20, 24, 28], blocks=4, seed=seed, gaps=[0, 0, 0], bugs=("no_wait_for_all",))
        except agent_sim.Violation:
            caught += 1
    assert caught >= 20


def test_no_stop_needed_the_agent_goes_by_itself():
    for seed in range(50):
        r = agent_sim.run([5, 6], blocks=2, seed=seed, stop_at_end=False)
        assert r["served"] == 2
