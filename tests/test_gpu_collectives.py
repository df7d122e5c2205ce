"""Multi-rank parity tests on real hardware.  One process per rank (the production path: hipIpc
windows + shared control block); on a 1-GPU box the ranks share device 0, on an N-GPU box rank i
uses GPU i % ndev.  Every rank checks its own result against the CPU oracle (tests/scenarios.py)."""
import pytest

from tests.gpu_harness import run_ranks, run_threads

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("size", [2, 4, 8])
def test_allreduce_small(size):
    run_ranks("allreduce_small", size, timeout=600)


@pytest.mark.parametrize("size", [3, 5])
def test_allreduce_odd_world(size):
    run_ranks("allreduce_small", size, {"counts": [1, 1000, 4099], "dtypes": [4, 2, 3]}, timeout=600)


@pytest.mark.parametrize("size", [2, 8])
def test_allreduce_pipelined(size):
    run_ranks("allreduce_medium", size, timeout=600)


def test_allreduce_tiny_slots():
    """4 KiB slots and a 2-deep FIFO force heavy slot reuse and back-pressure"""
    run_ranks("allreduce_medium", 4, timeout=600,
              env={"XMPI_SLOT_BYTES": "65536", "XMPI_FIFO_DEPTH": "2", "XMPI_P2P_SLOT_BYTES": "8192", "XMPI_DSYNC": "0"})


@pytest.mark.parametrize("size", [2, 8])
def test_copy_kernel_transport_and_batched_copies(size):
    """peer pushes by the copy kernel; all ready pushes / slot drains of a rank go out in one launch"""
    # (XMPI_DSYNC=0: the host-driven step tables, which is what these transports belong to -- with ranks that meet on the
    # device RING / RHD name the stepped kernels instead, tests below)
    env = {"XMPI_COPY_ENGINE": "1", "XMPI_BATCH_COPIES": "1", "XMPI_DSYNC": "0"}
    run_ranks("allreduce_small", size, {"counts": [1, 4099, 65536 + 5], "dtypes": [4, 2, 3]}, timeout=600, env=env)
    run_ranks("allgather", size, timeout=600, env=env)
    run_ranks("bcast_reduce", size, timeout=600, env=env)
    run_ranks("bounce", 2, timeout=600, env=env)
    run_ranks("allreduce_medium", size, timeout=600, env={**env, "XMPI_SLOT_BYTES": "262144", "XMPI_FIFO_DEPTH": "3"})


def test_copy_kernel_transport_threads():
    run_threads("allreduce_small", 4, {"counts": [1, 4099, 300001], "dtypes": [4, 2], "params": {"copy_engine": 1}})
    run_threads("allgather", 3, {"params": {"copy_engine": 1}})


@pytest.mark.parametrize("size", [2, 4, 8])
def test_allgather(size):
    run_ranks("allgather", size, timeout=600)


@pytest.mark.parametrize("size", [2, 4, 7])
def test_bcast_reduce(size):
    run_ranks("bcast_reduce", size, timeout=600)


@pytest.mark.parametrize("size", [2, 3, 8])
def test_zero_copy_processes(size):
    """zero-copy collectives between processes: the peers' user buffers are mapped through hipIpc and
    read / written in place; every result bit-identical to the rank-order oracle"""
    run_ranks("zero_copy", size, timeout=600)


def test_zero_copy_threads():
    """ranks hosted by threads of one process use each other's pointers directly"""
    run_threads("zero_copy", 4, {"counts": [1, 17, 4099, 65536 + 5]})
    # ... and by default the lowest of them folds everybody's chunks in one launch; without that:
    run_threads("zero_copy", 3, {"counts": [1, 4099], "params": {"zc_group_launch": 0}})


def test_zero_copy_disabled_by_param():
    """XMPI_ZERO_COPY=0: AUTO keeps to the staged schedules"""
    run_ranks("allreduce_small", 2, {"counts": [1, 4099], "dtypes": [4]}, timeout=300, env={"XMPI_ZERO_COPY": "0"})


@pytest.mark.parametrize("size", [2, 4, 8])
def test_stream_ordered_collectives(size):
    """xmpi_*_on_stream between processes: one kernel per rank is the collective, the ranks meet through flag words in HBM"""
    run_ranks("stream_ordered", size, timeout=600)


def test_stream_ordered_collectives_threads():
    """ranks hosted by threads of one process share a stream: the same API, met on the host"""
    run_threads("stream_ordered", 3, {"counts": [1, 4099]})


def test_device_sync_disabled_by_env():
    """XMPI_DSYNC=0: processes meet through the control block again (the round-1 path stays alive)"""
    run_ranks("zero_copy", 2, {"counts": [1, 4099]}, timeout=600, env={"XMPI_DSYNC": "0"})
    run_ranks("stream_ordered", 2, {"counts": [1, 4099]}, timeout=600, env={"XMPI_DSYNC": "0"})


@pytest.mark.parametrize("size", [2, 8])
def test_lifecycle_stress(size):
    """200 / 80 communicator lifetimes with the copy kernel as transport: init -> collectives -> finalize at each
    rank's own pace (the one unexplained GPU fault of round 1 was in this configuration)"""
    run_ranks("lifecycle_stress", size, {"iters": 200 if size == 2 else 80}, timeout=300,
              env={"XMPI_COPY_ENGINE": "1", "XMPI_BATCH_COPIES": "1", "XMPI_TEST_DUMP_AFTER": "200", "XMPI_TRACE": "1"})


@pytest.mark.parametrize("size", [2, 5])
def test_nonblocking_collectives(size):
    run_ranks("nonblocking", size, timeout=300)


def test_nonblocking_collectives_threads():
    run_threads("nonblocking", 3, {"count": 50021})


@pytest.mark.parametrize("size", [2, 4])
def test_bounce(size):
    """examples/bounce/bounce.go at its own message lengths + BASELINE cfg 2 (1 MiB f32)"""
    run_ranks("bounce", size, timeout=600)


def test_bounce_small_p2p_slots():
    """everything through 8 KiB mail slots (the path of unregistered / host buffers): heavy slot reuse"""
    run_ranks("bounce", 2, timeout=600, env={"XMPI_P2P_SLOT_BYTES": "8192", "XMPI_P2P_DIRECT_BYTES": "-1"})


def test_bounce_without_the_receive_agent():
    """XMPI_P2P_AGENT_US=0: one launch of the copy-and-ack kernel per message; XMPI_P2P_KERNEL_ACK=0: round 2's path
    (hipMemcpyAsync + event + host ack)"""
    run_ranks("bounce", 2, timeout=600, env={"XMPI_P2P_AGENT_US": "0"})
    run_ranks("bounce", 2, timeout=600, env={"XMPI_P2P_KERNEL_ACK": "0"})
    run_ranks("p2p_semantics", 2, timeout=300, env={"XMPI_P2P_SLOT_BYTES": "65536", "XMPI_P2P_AGENT_US": "0"})


def test_host_payloads():
    """Go slices on either side of Send / Receive: the host lanes of the shared segment, DMA out of a lane, one copy out of
    the sender's HBM -- every mix, around the lane's piece and ring sizes; and the same with the lanes switched off (staged
    through the HBM slots)"""
    for env in ({}, {"XMPI_HOST_LANES": "0"}):
        outs = run_ranks("host_payloads", 2, timeout=600, env=env)
        for line in (outs[0] or "").splitlines():  # what a round trip / an allreduce of slices costs (pytest -s shows it)
            if "host slices" in line:
                print(line)


def test_large_blocks_are_coloured_where_ranks_share_a_heap():
    """xmpi_malloc, heap.cpp: 16 consecutive large blocks of a heap serving several ranks lie in 16 different 4 KiB slots of a
    64 KiB frame (the fold's 16 streams then use different HBM banks); with one rank per process nothing moves"""
    run_threads("heap_colours", 2, {"threads": 1})
    run_threads("heap_colours", 8, {"threads": 1})
    run_ranks("heap_colours", 2, timeout=300)


def test_bounce_threads():
    """ranks as threads of one process: the receiver reads the sender's buffer through its own pointer"""
    run_threads("bounce", 2)


@pytest.mark.parametrize("size", [1, 2, 4])
def test_helloworld(size):
    """examples/helloworld/helloworld.go: BASELINE cfg 1 (all-to-all strings incl. self-send)"""
    run_ranks("helloworld", size, timeout=300)


@pytest.mark.parametrize("direct", ["1", "4096", "-1"])
def test_p2p_semantics(direct):
    run_ranks("p2p_semantics", 2, timeout=300, env={"XMPI_P2P_SLOT_BYTES": "65536", "XMPI_P2P_DIRECT_BYTES": direct})


def test_ranks_as_threads():
    """ranks hosted by threads of one process share window pointers instead of hipIpc handles"""
    run_threads("allreduce_small", 4, {"counts": [1, 4099], "dtypes": [4, 2]})
    run_threads("helloworld", 3)


def test_cfg3_allgather_int64_full():
    """BASELINE cfg 3: allgather int64, 16 MiB per rank, 4 ranks, bit-exact"""
    run_ranks("fullsize", 4, {"which": "cfg3"}, timeout=600)


def test_cfg4_allreduce_f32_full():
    """BASELINE cfg 4 (headline): allreduce-sum f32 256 MiB, 8 ranks"""
    run_ranks("fullsize", 8, {"which": "cfg4"}, timeout=900)


def test_cfg5_allreduce_f16_large():
    """BASELINE cfg 5: fp16 allreduce, ring vs recursive halving, exactly-summable inputs, at the FULL 1 GiB per rank
    whenever the GPU(s) have room for every rank's three buffers (one MI355X has: 8 ranks x 3 GiB of its 288),
    otherwise 256 MiB per rank."""
    run_ranks("fullsize", 8, {"which": "cfg5", "count": "auto"}, timeout=900)


@pytest.mark.parametrize("size", [2, 3, 4, 5, 6, 8])
def test_stepped_kernels(size):
    """ring allreduce / allgather, recursive halving + doubling (any number of ranks: a fold-in and a fold-out step when it
    is no power of two), binary-tree broadcast and reduce: every step inside ONE kernel per rank, released by flag words
    between the peers' kernels"""
    args = {} if size in (2, 8) else {"shapes": [(0, 0), (2, 3)], "counts": [1, 4099, 65536 + 5]}
    run_ranks("sched", size, args, timeout=900)


@pytest.mark.parametrize("size", [2, 3, 4, 5, 8])
def test_ll_small_collectives(size):
    """LL lines (ll.hip): allreduce / reduce / bcast / allgather in one one-way hop, bit-identical to the rank-order oracle"""
    args = {} if size in (2, 8) else {"counts": [1, 3, 17, 1000, 4099]}
    run_ranks("ll", size, args, timeout=900)


def test_ll_named_with_ranks_as_threads():
    """ranks that meet on the host: XMPI_ALGO_LL means the library's own choice"""
    run_threads("ll", 3)


def test_staged_schedules_between_processes():
    """XMPI_DSYNC=0: RING / RHD / TREE between processes are the host-driven step tables again"""
    run_ranks("allreduce_small", 4, {"counts": [1, 4099, 65536 + 5], "dtypes": [4, 2]}, timeout=600, env={"XMPI_DSYNC": "0"})
    run_ranks("allgather", 3, timeout=600, env={"XMPI_DSYNC": "0"})


@pytest.mark.parametrize("size", [2, 5, 8])
def test_split_form(size):
    """meet / body / done: the zero-copy collectives as three launches, only two blocks of which ever wait"""
    run_ranks("split", size, timeout=600)


@pytest.mark.parametrize("size,seed", [(2, 1), (3, 2), (5, 3), (8, 4)])
def test_soak_every_form_mixed(size, seed):
    """a seeded random walk over every collective in every form, blocking and stream-ordered, with Send / Receive rings between
    them and parameters flipped between calls: what one form leaves on the shared flag page for the next"""
    run_ranks("soak", size, {"seed": seed, "steps": 600 if size < 8 else 300}, timeout=900)


def test_soak_threads_layout():
    run_threads("soak", 4, {"seed": 9, "steps": 200})


def test_roctx_ranges_on():
    """XMPI_ROCTX=1: the marker library is found at run time (nothing links it) and every collective / launch / message pushes and
    pops its range; the results are what they are without"""
    run_ranks("allreduce_small", 2, {"counts": [1, 4099, 65536 + 5], "dtypes": [4, 3], "expect_params": {"roctx": 1}}, timeout=300,
              env={"XMPI_ROCTX": "1"})
    run_ranks("bounce", 2, timeout=300, env={"XMPI_ROCTX": "1"})
    run_ranks("allreduce_small", 2, {"counts": [17], "dtypes": [4], "expect_params": {"roctx": 0}}, timeout=300, env={"XMPI_ROCTX": "0"})


def test_split_form_xcd_guard_trips():
    """the done kernel refuses a collective whose meet / done blocks did not reach as many XCDs as it was told the GPU has"""
    run_ranks("split", 2, {"counts": [4099], "trip": 1}, timeout=300)


@pytest.mark.parametrize("size,env,level", [(2, {}, 0), (3, {}, 0), (4, {"XMPI_DSYNC": "1"}, 0)])
def test_init_vote_on_a_healthy_machine(size, env, level):
    """xmpi_init's vote (windows, flag pages tried with the self-test kernel, the LL limit) on real hardware: nothing degraded, and
    what the degraded levels rely on -- every collective by every name, helloworld, Send / Receive out of registered, host and never
    registered device memory -- works here (tests/scenarios.py sc_degraded; the fault injections themselves run on virtual devices
    in the CPU suite)"""
    run_ranks("degraded", size, {"expect": level, "why": ""}, timeout=300, env=env)


@pytest.mark.parametrize("size", [2, 4])
def test_collectives_on_several_streams(size):
    run_ranks("multistream", size, timeout=300)


@pytest.mark.parametrize("size", [2, 4])
def test_stream_ordered_send_recv(size):
    run_ranks("p2p_stream", size, timeout=300)


@pytest.mark.parametrize("size", [2, 8])
def test_library_tuner(size):
    run_ranks("tune", size, timeout=600)


def test_library_tuner_from_the_environment():
    """XMPI_AUTOTUNE_BYTES: a program that never heard of tuning gets the tuned table from xmpi_init"""
    run_ranks("tune", 4, {"tuned_by_init": True}, timeout=600, env={"XMPI_AUTOTUNE_BYTES": str(2 << 20)})


@pytest.mark.parametrize("size", [2, 8])
def test_init_selfcheck_on_the_gpu(size):
    """XMPI_SELFCHECK=1 (the default only where ranks sit on different GPUs): xmpi_init runs what untuned AUTO can reach -- LL lines,
    the one-kernel fold, meet / body / done, the other collectives' folds -- on patterned inputs and compares with the locally
    computed result before the first caller's data goes through; on this machine nothing is rejected, nothing degraded, the cost is
    readable, and every collective is exact afterwards (tests/scenarios.py sc_corrupt with nothing corrupt; the fault injections run
    on virtual devices in the CPU suite)"""
    outs = run_ranks("corrupt", size, {"rejected": {}, "tune": 0, "params_after": {"selfcheck": 1}, "report_selfcheck": 1}, timeout=300,
                     env={"XMPI_SELFCHECK": "1"})
    assert any("init_selfcheck_us" in o for o in outs), outs[0][-500:]


def test_library_tuner_threads():
    """ranks that meet on the host have one schedule: nothing to tune, AUTO unchanged"""
    run_threads("tune", 3)


@pytest.mark.parametrize("what", ["allreduce", "recv"])
def test_a_peer_that_dies_is_an_error_not_a_hang(what):
    """rank 1 of 2 exits without a word after a collective that worked: rank 0's next allreduce (its kernel waits for a flag word
    that will never come) / Receive returns an error within the no-progress limit, the kernel has ended, the GPU still works
    (the reference's peers get a TCP error: network.go:518-571)"""
    outs = run_ranks("peer_dies", 2, {"what": what}, timeout=120, env={"XMPI_TIMEOUT_S": "5", "XMPI_WATCHDOG_MS": "0"})
    assert sum("ok (error after" in o for o in outs) == 1, "\n".join(outs)


@pytest.mark.gpu
@pytest.mark.parametrize("what", ["allreduce", "split", "ring", "ll", "ll_agent", "recv", "recv_on_stream"])
def test_a_peer_that_dies_is_an_error_at_once_with_default_settings(what):
    """XMPI_TIMEOUT_S unset = wait for ever (the default, as the reference's blocking calls): rank 1 of 2 exits without a word and
    rank 0's next collective -- one kernel, meet / body / done, the ring kernel, LL lines launched or by the lingering agent -- or
    Receive (blocking / stream-ordered) comes back with XMPI_ERR_PEER WITHIN A SECOND, naming the rank: the helper thread asks every
    50 ms whether the peers' processes still exist (pid + start time) and raises the job's abort flag, which the waiting kernel
    polls; the kernel has ended, the GPU still works.  (The reference's peers get the error from their sockets at once:
    network.go:555,611,623.)"""
    outs = run_ranks("peer_dies", 2, {"what": what, "no_timeout": 1, "within": 1.0}, timeout=120, env={"XMPI_TIMEOUT_S": "0"})
    assert sum("ok (error after" in o for o in outs) == 1, "\n".join(outs)
    assert "the process of rank 1" in outs[0], outs[0]


@pytest.mark.parametrize("what", ["length", "length_split", "schedule", "form", "root", "shape", "collective"])
def test_ranks_in_different_calls_get_an_error_not_a_hang(what):
    """every kernel announces the call it is in (collective, schedule, bytes, dtype, operation, root, the stepped kernels' shape) in
    the upper half of a word it stores anyway; ranks that differ all end with an error at once -- no hang, no fold over buffers of
    different lengths -- and nobody's receive buffer was written (the reference: a Receive nobody sends to blocks for ever)"""
    outs = run_ranks("mismatch", 3, {"what": what}, timeout=120, env={"XMPI_TIMEOUT_S": "20"})
    assert sum("ok (error after" in o for o in outs) == 3, "\n".join(outs)


@pytest.mark.parametrize("what", ["length", "collective"])
def test_rank_threads_in_different_calls_get_an_error_not_a_hang(what):
    """... and where ranks meet on the host (threads of one process): the descriptors they publish carry the same signature"""
    out = run_threads("mismatch", 3, {"what": what, "threads": 1}, timeout=120)
    assert out.count("ok (error after") == 3, out
