"""Property tests of the schedules (hypothesis): for random world sizes, element counts, piece sizes,
channel counts, FIFO depths and random interleavings, every rank's step table executed over bounded
FIFOs reproduces the rank-order oracle exactly (int64: any data-movement or ordering bug shows), never
deadlocks, and leaves every pipe empty."""
import numpy as np
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from mpi_amd import xmpi
from oracle import oracle
from tests import plan_sim

ALGOS = {xmpi.COLL_ALLREDUCE: [xmpi.ALGO_RING, xmpi.ALGO_RHD, xmpi.ALGO_DIRECT, xmpi.ALGO_AUTO],
         xmpi.COLL_ALLGATHER: [xmpi.ALGO_RING, xmpi.ALGO_DIRECT],
         xmpi.COLL_BCAST: [xmpi.ALGO_TREE, xmpi.ALGO_DIRECT, xmpi.ALGO_AUTO],
         xmpi.COLL_REDUCE: [xmpi.ALGO_TREE, xmpi.ALGO_DIRECT, xmpi.ALGO_AUTO]}


@settings(max_examples=150, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(coll=st.sampled_from(sorted(ALGOS)), n=st.integers(1, 9), count=st.integers(1, 3000), piece=st.integers(2, 700),
       channels=st.integers(1, 8), depth=st.integers(1, 5), seed=st.integers(0, 10 ** 6), data=st.data())
def test_any_schedule_matches_oracle(coll, n, count, piece, channels, depth, seed, data):
    algo = data.draw(st.sampled_from(ALGOS[coll]))
    root = data.draw(st.integers(0, n - 1))
    ins = [oracle.fill(count, oracle.I64, 0, seed + r) for r in range(n)]
    oneshot = data.draw(st.sampled_from([0, 4096, 1 << 20]))  # direct allreduce: two-phase / one-shot
    plans = plan_sim.get_plans(coll, algo, n, root, count, 8, channels, piece, fifo_depth=depth,
                               oneshot_bytes=oneshot)
    if coll == xmpi.COLL_ALLREDUCE:
        want = oracle.reduce_ranks(ins, oracle.I64, oracle.SUM)
        got = plan_sim.simulate(plans, ins, count, np.int64, xmpi.SUM, depth, seed=seed)
        assert all(np.array_equal(g, want) for g in got)
    elif coll == xmpi.COLL_ALLGATHER:
        want = oracle.allgather(ins, oracle.I64)
        got = plan_sim.simulate(plans, ins, count * n, np.int64, xmpi.SUM, depth, seed=seed)
        assert all(np.array_equal(g, want) for g in got)
    elif coll == xmpi.COLL_BCAST:
        got = plan_sim.simulate(plans, ins, count, np.int64, xmpi.SUM, depth, seed=seed, inplace=True)
        assert all(np.array_equal(g, ins[root]) for g in got)
    else:
        want = oracle.reduce_ranks(ins, oracle.I64, oracle.SUM)
        got = plan_sim.simulate(plans, ins, count, np.int64, xmpi.SUM, depth, seed=seed)
        assert np.array_equal(got[root], want)


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(n=st.integers(2, 8), count=st.integers(1, 2000), piece=st.integers(2, 500), seed=st.integers(0, 10 ** 6),
       oneshot=st.sampled_from([0, 2048, 1 << 20]))
def test_direct_fold_is_bitwise_rank_order_for_floats(n, count, piece, seed, oneshot):
    ins = [oracle.fill(count, oracle.F32, 3, seed + r) for r in range(n)]
    want = oracle.reduce_ranks(ins, oracle.F32, oracle.SUM)
    plans = plan_sim.get_plans(xmpi.COLL_ALLREDUCE, xmpi.ALGO_DIRECT, n, 0, count, 4, 1, piece,
                               oneshot_bytes=oneshot)
    got = plan_sim.simulate(plans, ins, count, np.float32, xmpi.SUM, 2, seed=seed)
    assert all(g.tobytes() == want.tobytes() for g in got)
