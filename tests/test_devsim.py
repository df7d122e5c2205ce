"""The library WHOLE on the CPU: its host sources and its gfx950 kernel sources (kernels.hip, sched.hip, ll.hip, kdev.h -- the
files that ship, unchanged) compiled by clang++ as C++ over tests/devsim, a HIP runtime with N virtual devices whose kernels run
as threads (blocks) and fibers (lanes) and whose device memory is shared memory that another process maps at another address.

Two things no 1-GPU box can show ("nothing with hipGetDeviceCount() > 1 has ever executed", "nothing races the actual
atomics"):
  * one process per DEVICE -- rank i on device i, peer access, cross-device IPC mappings, the cross-device branches of the host
    code -- under the scenarios of the GPU suite (tests/scenarios.py, every result against the oracle);
  * the device-side protocols (dsync_begin / dsync_end, meet / body / done, LL lines, the stepped ring / halving / tree kernels,
    the Send / Receive kernels and the receive agent) under ThreadSanitizer, with the kernels' data stores as PLAIN stores: a
    reader that no chain of flag words has ordered behind them is reported.

What it cannot show stays with the GPU suite: cache maintenance, s_waitcnt, write-through -- everything about WHEN a store
becomes visible rather than in which order the protocol allows it to be looked at -- and every rate.  No GPU involved; nothing
here is product code (the product has no switch that leads here: tests/rank_worker.py points the binding at the stand-in)."""
import os
import subprocess

import pytest

from tests.gpu_harness import run_ranks, run_threads

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _everything_built():
    """the four artefacts side by side (a cold build of each is ~1 min of clang++ on the kernel templates; nothing when current)"""
    import sys
    jobs = [subprocess.Popen([sys.executable, "-m", "tests.devsim.build", *flag], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            for flag in ([], ["--tsan"], ["--traffic"], ["--ubsan"])]
    out, _ = jobs[1].communicate()  # the mutants swap single objects of the sanitizer build: they start when it is there
    assert jobs[1].returncode == 0, out[-4000:]
    jobs[1] = subprocess.Popen([sys.executable, "-m", "tests.devsim.build", "--mutant"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    for j in jobs:
        out, _ = j.communicate()
        assert j.returncode == 0, out[-4000:]
    from tests.devsim import build
    build.build_driver(False)  # (shares the library's objects)


@pytest.fixture(scope="module")
def devsim_lib():
    from tests.devsim import build
    path = build.build_lib()
    old = {k: os.environ.get(k) for k in ("XMPI_DEVSIM_LIB", "XMPI_TUNE_ITERS")}
    os.environ["XMPI_DEVSIM_LIB"] = path  # (run_ranks hands the environment on to the rank processes)
    # a "kernel" here is a host thread: the tuner's TIMES mean nothing, its checked run of every candidate at every size is what these
    # tests are about -- two timed runs per candidate instead of twenty (the lone-process scale command: 66 -> 18 s, nearly all of it tuning)
    os.environ["XMPI_TUNE_ITERS"] = "2"
    yield path
    for k, v in old.items():
        if v is None:
            del os.environ[k]
        else:
            os.environ[k] = v


@pytest.fixture(scope="module")
def plain_bin():
    from tests.devsim import build
    return build.build_driver(False)


@pytest.fixture(scope="module")
def tsan_bin():
    from tests.devsim import build
    return build.build_driver(True)


def run(binp, *args, **env):
    e = dict(os.environ, TSAN_OPTIONS="exitcode=66 halt_on_error=0 report_signal_unsafe=0", **{k: str(v) for k, v in env.items()})
    for k in ("XMPI_DEVSIM_LIB", "XMPI_TUNE_ITERS"):
        e.pop(k, None)  # (what the devsim_lib fixture sets for the tests further down is not for the drivers' runs)
    return subprocess.run([binp, *args], capture_output=True, text=True, timeout=900, env=e, cwd="/tmp")


# ---- under the sanitizer: ranks as threads, every rank on a device of its own ----------------------------------------------------
def test_the_sanitizer_sees_the_kernels_stores(tsan_bin):
    """a rank that reads its receive buffer while the collective is still in flight: reported, with the kernel's store named"""
    r = run(tsan_bin, "--seed-race", "2")
    assert r.returncode == 66 and "ThreadSanitizer: data race" in r.stderr, r.stderr[-3000:]
    assert "dsync_fold_kernel" in r.stderr, r.stderr[-3000:]


@pytest.mark.parametrize("mutant,scenario,where", [("step", "sched", ("st_sys128", "ld_sys128_issue")), ("done", "fold", ("hipMemcpyAsync",)),
                                                   ("agent", "ll", ("ll_agent_collective", "hipMemcpyAsync")),
                                                   ("land", "sched", ("st_sys128", "ld_sys128_issue"))])
def test_the_sanitizer_finds_a_wait_taken_out_of_a_kernel(tsan_bin, mutant, scenario, where):
    """mutation: COPIES of the kernel sources with one wait removed (tests/devsim/build.py MUTATIONS; the product source is
    untouched) -- `step`: the stepped kernels no longer wait for the peer's step flag (races between a step's loads and the
    peer's written-through stores); `done`: the closing block no longer waits for the peers' "done" (the caller reads / refills
    buffers the peers' kernels still store into); `agent`: the LL agent's lane 0 answers its caller without waiting for the block's
    other lanes (the caller downloads a receive buffer they still store into); `land`: the push form of the halving kernel lands
    every level's half in ONE region of the partner's landing block instead of a region per level (the next level's partner stores
    over what the owner may still be folding).  The harness must say so, in those places"""
    from tests.devsim import build
    # (XMPI_SELFCHECK=0: with every rank on a device of its own xmpi_init checks the schedules' answers itself, and a mutant whose
    # race happens to spoil an answer THERE is dropped -- "refused: it gave wrong answers on this machine" -- before the scenario gets
    # to it; this test is about what the sanitizer harness sees)
    r = run(build.build_mutant(mutant), "4", "1", scenario, XMPI_SELFCHECK=0)
    assert r.returncode != 0 and "ThreadSanitizer: data race" in r.stderr, r.stderr[-3000:]
    assert all(w in r.stderr for w in where), r.stderr[-3000:]


@pytest.mark.parametrize("ranks,fuzz", [(2, 0), (3, 5), (5, 0), (8, 9)])
def test_device_protocols_are_race_free(tsan_bin, ranks, fuzz):
    """every form of every collective, stream-ordered and blocking Send / Receive, the agent, graphs: results right, nothing
    reported (fuzz: lanes and threads give way at random before system-scope accesses)"""
    r = run(tsan_bin, str(ranks), "1", DEVSIM_FUZZ=fuzz)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-6000:]
    assert r.returncode == 0 and "devsim driver ok" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]


@pytest.mark.parametrize("ranks,fuzz", [(4, 3), (8, 5)])
def test_the_undefined_behaviour_sanitizer_finds_nothing(ranks, fuzz):
    """-fsanitize=undefined,bounds over the library whole -- host sources AND the kernel sources, compiled for the CPU -- while the driver
    walks every form of every collective, Send / Receive, the agents, graphs (the GPU pool offers no sanitizer run: this is where one
    can).  What it found when first run: pointer arithmetic on the null bases of a stepped kernel's unused operands (sched.hip
    tile_apply: `C + lo`, `D2 + lo` with C / D2 null -- never dereferenced, undefined all the same)"""
    from tests.devsim import build
    r = run(build.build_driver_ubsan(), str(ranks), str(fuzz), UBSAN_OPTIONS="print_stacktrace=1")
    assert r.returncode == 0 and "devsim driver ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    assert "runtime error" not in r.stderr, r.stderr[-4000:]


@pytest.mark.parametrize("ranks,fuzz", [(4, 3), (8, 5)])
def test_ranks_that_share_a_gpu_are_race_free(tsan_bin, ranks, fuzz):
    """--shared: every rank a thread on device 0 -- bench.py's N = 1 layout.  They meet on the HOST (zcopy.cpp's rendezvous and
    group launch: reduce_n_multi_kernel folds everybody's chunks, one launch), ring / halving / direct run as host-driven step
    tables through the windows, Send / Receive through the agent: under the sanitizer, nothing reported"""
    r = run(tsan_bin, "--shared", str(ranks), "1", DEVSIM_FUZZ=fuzz)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-6000:]
    assert r.returncode == 0 and "devsim driver ok" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]


@pytest.mark.parametrize("xcd_map,fuzz", [("rr", 1), ("continue", 2), ("pairs", 3)])
def test_driver_under_perturbed_schedules(plain_bin, xcd_map, fuzz):
    """the same walk without the sanitizer (its 5-10x), 8 ranks, two rounds, other dispatch orders round the XCDs"""
    r = run(plain_bin, "8", "2", DEVSIM_FUZZ=fuzz, DEVSIM_XCD_MAP=xcd_map)
    assert r.returncode == 0 and "devsim driver ok" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]


def test_no_collective_needs_two_of_its_blocks_resident(plain_bin):
    """DEVSIM_RESIDENT=1: the blocks of a launch run ONE AFTER THE OTHER.  Every collective kernel still completes -- its blocks wait
    for peers' blocks and for blocks before them, never for a block behind them -- so no grid cap is load-bearing for progress
    (what a GPU shared by eight processes, or one with fewer free wave slots than the grid, relies on).  The one exception is
    by design and named: the lingering receive agent's 8 blocks wait for block 0's word (p2p_block, left out here)."""
    r = run(plain_bin, "3", "1", "fold", "split", "ll", "sched", "bcast", "reduce", "allgather", "stream", "graph", "p2p_stream",
            DEVSIM_RESIDENT=1)
    assert r.returncode == 0 and "devsim driver ok" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]


# ---- one PROCESS per device: the GPU suite's scenarios ---------------------------------------------------------------------------
@pytest.mark.parametrize("size", [2, 3, 5, 8])
def test_one_process_per_device(devsim_lib, size):
    """rank i <-> device i: peer access, cross-device mappings, link probe, every form of the collectives, Send / Receive"""
    run_ranks("devices", size, timeout=600)


SCENARIOS = [
    ("allreduce_small", 8, None),
    ("allreduce_small", 3, {"counts": [1, 1000, 4099], "dtypes": [4, 2, 3]}),
    ("allgather", 4, None),
    ("bcast_reduce", 7, None),
    ("nonblocking", 4, None),
    ("bounce", 2, None),
    ("helloworld", 4, None),
    ("p2p_semantics", 2, None),
    ("stream_ordered", 4, None),
    ("ll", 5, None),
    # the stepped kernels, pull and push form: every dtype and operator, in place, odd alignments, channels and workers; the push
    # form's bits against the pull form's
    ("sched", 4, {"shapes": [(0, 0), (2, 3)], "counts": [1, 17, 4099], "quick": 1}),
    ("sched", 3, {"shapes": [(1, 2)], "counts": [1, 4099], "quick": 1}),
    ("sched", 8, {"shapes": [(0, 0)], "counts": [17], "quick": 1}),
    ("split", 4, {"counts": [1, 17, 4099]}),
    ("multistream", 4, None),
    ("p2p_stream", 4, None),
    ("soak", 3, None),
    ("lifecycle_stress", 2, None),
    # the library's tuner over all four collectives (times mean nothing here: that every candidate of every collective runs, the
    # ranks agree on the table, and AUTO follows it to the right result)
    ("tune", 4, {"max_bytes": 65536}),
    ("tune", 7, {"max_bytes": 4096}),
]


# More ranks than the device side is sized for (kernels.h kDsyncRanks = 8: one node, one rank per GPU; the job limit is ctl.h
# kMaxRanks = 16 -- the reference takes len(addrs), network.go:94-109): 9 .. 16 ranks, two per virtual GPU, run on the generic path
# -- the ranks meet on the host (zcopy.cpp's rendezvous with up to 16 sources per fold, the step tables through the windows) --
# and every result still matches the oracle.  Not a degradation (get_param("degraded") stays 0) and no device-synchronised form.
SCENARIOS += [
    ("allreduce_small", 9, {"counts": [1, 1000, 4099], "dtypes": [4, 2], "expect_params": {"dsync": 0, "degraded": 0}}),
    ("bcast_reduce", 11, {"expect_params": {"dsync": 0}}),
    ("allgather", 12, {"counts": [0, 1, 1000, 4099, 65536 + 3], "expect_params": {"dsync": 0}}),
    ("helloworld", 16, {"expect_params": {"dsync": 0, "degraded": 0}}),
]


@pytest.mark.parametrize("scenario,size,args", SCENARIOS, ids=[f"{s}-{n}" for s, n, _ in SCENARIOS])
def test_gpu_scenarios_on_virtual_devices(devsim_lib, scenario, size, args):
    run_ranks(scenario, size, args, timeout=600, env={"DEVSIM_DEVICES": "8"} if size > 8 else None)


@pytest.mark.parametrize("devices,ranks", [(1, 4), (4, 4), (3, 3), (8, 8)])
def test_no_kernel_touches_a_byte_past_its_buffers(devsim_lib, devices, ranks):
    """every buffer ends on the last byte of a page with an inaccessible page behind it (runtime.cpp devsim_guarded_alloc): the local
    kernels at ragged counts of every element width, then -- out of registered user memory -- every collective by every name, in place
    and out, Send / Receive; rank threads sharing ONE device (the host rendezvous, `reduce_n_multi_kernel`: the headline's kernel)
    and on a device each (the device-synchronised kernels: fold, split, stepped pull / push, LL; 3, 4 and 8 ranks).  An access one byte past a buffer's end -- which
    a GPU serves out of the allocation's granule without a word -- kills the process here; the results are held to the oracle"""
    import sys
    e = dict(os.environ, DEVSIM_DEVICES=str(devices), XMPI_TIMEOUT_S="60")
    cmd = [sys.executable, os.path.join(ROOT, "tests", "rank_worker.py"), "--threads", "guard", str(ranks)]
    p = subprocess.run(cmd + ['{"counts": [1, 17, 4099]}' if ranks == 8 else "{}"], cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert p.returncode == 0 and f"{ranks} rank threads guard: ok" in p.stdout, p.stdout[-4000:]
    if devices == 1:  # ... and the guard is real: a checksum asked for one byte more than the buffer has dies of it
        p = subprocess.Popen(cmd[:-1] + ["1", '{"overrun": 1}'], cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        out, _ = p.communicate(timeout=300)
        import glob
        for f in glob.glob(f"/dev/shm/devsim.{p.pid}.*") + glob.glob(f"/dev/shm/xmpi-*-th{p.pid}-*"):  # (it died with its allocations mapped)
            os.unlink(f)
        assert p.returncode < 0 and "one byte past the end" in out and "went unnoticed" not in out, (p.returncode, out[-2000:])


@pytest.mark.parametrize("layout", ["threads", "processes", "processes_on_the_host"])
def test_host_slices_run_the_schedules_registered_buffers_run(devsim_lib, layout):
    """what a caller of the reference passes -- host slices -- is stood in for by arena blocks and takes the zero-copy path in every
    layout: rank threads of one process and processes that meet through the control block (api.cpp collective: no fallback to the
    staged schedule, `zc_fallbacks_unregistered` stays), processes that meet on the device (dsync.cpp: `dsync_bounced`); out of place,
    in place, host operand with the result in HBM: bit-exact against the oracle (tests/scenarios.py sc_zero_copy)"""
    if layout == "threads":
        run_threads("zero_copy", 4, {"counts": [1, 4099]})
    else:
        run_ranks("zero_copy", 3, {"counts": [1, 4099]}, timeout=600, env={"XMPI_DSYNC": "0"} if layout.endswith("host") else None)


DEGRADED = [
    # every open of another device's UNCACHED allocation fails (the flag pages: "has never executed anywhere" before an 8-GPU
    # node): the ranks meet on the host, zero-copy collectives with a host rendezvous, everything else as before
    ("flag pages", {"DEVSIM_FAIL_IPC_KIND": "uncached"}, 2, "flag page", 4),
    # ... on ONE rank only: the same level for everybody (the vote), nobody waits for a rank that went another way
    ("flag pages, one rank", {"DEVSIM_FAIL_IPC_KIND": "uncached@1"}, 2, "rank 1: hipIpcOpenMemHandle(flag page", 3),
    # the pages OPEN on rank 0's device and carry nothing (its stores into them never reach their owners): found by trying the flag
    # words inside xmpi_init, with a clock -- not by a first collective that never ends
    ("flag pages that carry nothing", {"DEVSIM_PRIVATE_UNCACHED": "1@0"}, 2, "never arrived", 3),
    # no uncached device memory at all (one rank's runtime refuses it): nobody has a use for the others' pages
    ("no uncached memory", {"DEVSIM_FAIL_UNCACHED_ALLOC": "1@2"}, 2, "rank 2: hipExtMallocWithFlags", 4),
    # rank 1's first open -- rank 0's window -- fails: no windows for the job (no staged step tables, no mail slots), the
    # device-synchronised collectives serve every call, Send / Receive go out of registered blocks and through the host lanes
    ("a window, one rank", {"DEVSIM_FAIL_IPC_OPEN": "1@1"}, 4, "rank 1: hipIpcOpenMemHandle(window of rank 0)", 3),
]


@pytest.mark.parametrize("what,env,level,why,size", DEGRADED, ids=[d[0].replace(" ", "_").replace(",", "") for d in DEGRADED])
def test_a_mapping_the_driver_refuses_degrades_the_job_not_kills_it(devsim_lib, what, env, level, why, size):
    """xmpi_init's vote: every rank publishes what it could map, all take the best level everybody reached, and the communicator
    WORKS -- level and reason readable (xmpi_get_param("degraded"), xmpi_degraded()), every collective by every name against the
    oracle, the reference's helloworld and a ping-pong out of HBM (tests/scenarios.py sc_degraded)"""
    run_ranks("degraded", size, {"expect": level, "why": why}, timeout=600, env=dict(env, XMPI_INIT_TIMEOUT_S="20", XMPI_TIMEOUT_S="30"))


# ---- a node on which one schedule gives WRONG ANSWERS --------------------------------------------------------------------------------
CORRUPT = [
    # (what the node gets wrong, rejected candidates per collective 0 allreduce / 1 allgather / 2 bcast / 3 reduce, extra scenario arguments, extra environment)
    # the one-kernel fold's stores into peer memory: the fold in both unrolls, push-only (its two kernels ARE that kernel) -- the table
    # says "split" wherever it said "the fold"; bcast takes the tree kernel
    ("fold", {0: ["fold", "fold2", "zpush"], 1: ["fold"], 2: ["fold"], 3: ["fold", "zpush"]}, {"why": ["allreduce: fold (one kernel) gives wrong answers", "bcast: fold"], "max_bytes": 1 << 20}, {}),  # (1 MiB: bcast's fold beyond zc_bcast_push_bytes, where every rank forwards)
    # the split form's data kernel with ordinary (cached, non-temporal) stores: the ladder's FIRST rung -- its system-scope form is
    # right, takes over (degraded bit 1), and nothing is rejected
    ("split", {}, {"level": 8 | 1, "why": ["its system-scope data kernel is right and takes over"], "params_after": {"body_sys": 1}}, {"XMPI_KERNEL_MODE": "1"}),
    # ... in both forms: split leaves every table and the untuned rule (dsync_split_bytes 0)
    ("split_sys", {0: ["split"], 1: ["split"], 3: ["split"]}, {"why": ["its system-scope data kernel does too"], "params_after": {"body_sys": 0, "dsync_split_bytes": 0}},
     {"XMPI_KERNEL_MODE": "1"}),
    # NOT a flipped bit: the data kernel's loads from a peer's memory return what the FIRST load of that address returned -- an L2 the
    # once-per-XCD acquire never reached.  Every buffer read once: nothing to see; the check's SECOND pass (every rank's input changed in
    # place between two runs) catches it -- and the system-scope data kernel, which no cache serves, takes over
    ("split_stale", {}, {"level": 8 | 1, "why": ["its system-scope data kernel is right and takes over"], "params_after": {"body_sys": 1}}, {"XMPI_KERNEL_MODE": "1"}),
    # ... the same in the one-kernel fold (its per-block acquire is what prevents this on a GPU): out where it LOADS from peers -- allreduce
    # and reduce; push-only, allgather and bcast read local memory only and stay
    ("fold_stale", {0: ["fold", "fold2"], 3: ["fold"]}, {}, {}),
    ("ring", {0: ["ring"], 1: ["ring"]}, {}, {}),            # what the ring kernel LOADS over a link (pull form)
    ("ring_push", {0: ["ring_push"], 1: ["ring_push"]}, {}, {}),  # ... STORES over a link (push form): the pull form stays
    ("rhd_push", {0: ["rhd_push"]}, {}, {}),
    ("tree", {2: ["tree"], 3: ["tree"]}, {}, {}),
    # LL lines into another device's flag allocation (bcast's lines come from the root, device 0 -- right here, but it is ONE mechanism:
    # wrong for one collective, trusted for none): no LL for untuned AUTO either
    ("ll", {0: ["ll"], 1: ["ll"], 2: ["ll"], 3: ["ll"]}, {"params_after": {"ll_bytes": 0, "agent_ll": 0}}, {}),
]


@pytest.mark.parametrize("form,rejected,extra,env", CORRUPT, ids=[c[0] for c in CORRUPT])
def test_the_tuner_drops_a_schedule_that_gives_wrong_answers_here(devsim_lib, form, rejected, extra, env):
    """DEVSIM_CORRUPT_FORM: device 1 flips a bit in that kernel's data accesses to other devices' memory.  xmpi_tune checks every
    candidate's answer (patterned inputs, the expected result computed locally, a job-wide vote) BEFORE it believes its time: exactly
    the schedules that run that kernel are rejected -- on every rank, by name, with the reason readable --, AUTO stays oracle-exact,
    a rejected schedule named by a caller is refused on every rank alike, everything else still runs (tests/scenarios.py sc_corrupt)"""
    run_ranks("corrupt", 4, dict({"rejected": {str(k): v for k, v in rejected.items()}}, **extra), timeout=600,
              env=dict({"DEVSIM_CORRUPT_FORM": form, "XMPI_SELFCHECK": "0"}, **env))


SELFCHECK = [
    # an UNTUNED job (nobody calls xmpi_tune): xmpi_init's self-check runs what untuned AUTO can reach on patterned inputs.  LL lines
    # wrong: no LL for AUTO (the fold takes the small messages)
    ("ll", {0: ["ll"], 1: ["ll"], 2: ["ll"], 3: ["ll"]}, {"level": 8, "params_after": {"ll_bytes": 0, "agent_ll": 0}}, {}),
    # split wrong with ordinary stores: the system-scope data kernel takes over
    ("split", {}, {"level": 8 | 1, "params_after": {"body_sys": 1}, "why": ["xmpi_init self-check", "takes over"]}, {"XMPI_KERNEL_MODE": "1"}),
    ("split_sys", {0: ["split"]}, {"level": 8, "params_after": {"dsync_split_bytes": 0}}, {"XMPI_KERNEL_MODE": "1"}),
    # the one-kernel fold wrong: an untuned job has no table to route round it -- the ranks meet on the host (level 2), every collective right
    ("fold", {0: ["fold"], 1: ["fold"], 2: ["fold"], 3: ["fold"]}, {"level": 8 | 2, "why": ["the ranks meet on the host"]}, {}),
    # what the blocking Receive's copy kernels LOAD out of the sender's memory (agent and pull kernel): messages go through the mail
    # slots instead (the sender's engine pushes, the receiver drains locally), checked in turn -- and every echo is right
    ("p2p", {}, {"level": 8, "why": ["Send / Receive: the receiver's direct pull", "messages travel through the mail slots"],
                 "params_after": {"p2p_rejected": 1, "p2p_direct_bytes": -1}}, {}),
]


@pytest.mark.parametrize("form,rejected,extra,env", SELFCHECK, ids=[c[0] for c in SELFCHECK])
def test_init_checks_what_untuned_auto_can_reach(devsim_lib, form, rejected, extra, env):
    """the same node, a job that never tunes: ranks on different GPUs => xmpi_init runs the self-check by default (XMPI_SELFCHECK)"""
    run_ranks("corrupt", 4, dict({"rejected": {str(k): v for k, v in rejected.items()}, "tune": 0}, **extra), timeout=600,
              env=dict({"DEVSIM_CORRUPT_FORM": form}, **env))


def test_the_selfcheck_runs_where_ranks_sit_on_different_gpus_and_finds_nothing_on_a_healthy_node(devsim_lib):
    run_ranks("corrupt", 3, {"rejected": {}, "tune": 0, "params_after": {"selfcheck": 1}, "expect_selfcheck": 1}, timeout=600)


def test_nothing_degraded_on_a_healthy_node(devsim_lib):
    run_ranks("degraded", 3, {"expect": 0, "why": ""}, timeout=600)


def test_no_transport_at_all_fails_init_on_every_rank_at_once(devsim_lib):
    """neither the windows nor the flag pages can be mapped (IPC is broken outright): xmpi_init returns an error that names the
    calls -- on EVERY rank, promptly (the vote is collective: nobody waits for the bootstrap's clock to run out, nobody carries on
    alone)"""
    import time
    t0 = time.time()
    with pytest.raises(AssertionError) as e:
        run_ranks("helloworld", 3, timeout=120, env={"DEVSIM_FAIL_IPC_OPEN": "1@1", "DEVSIM_FAIL_IPC_KIND": "uncached", "XMPI_INIT_TIMEOUT_S": "60",
                                                     "XMPI_TIMEOUT_S": "10"}, expect_failure=True)
    text = str(e.value)
    assert "ranks [0, 1, 2] failed" in text and text.count("no transport left") == 3, text[-3000:]
    assert "hipIpcOpenMemHandle(window of rank 0)" in text and "flag page" in text and "killed after timeout" not in text, text[-3000:]
    assert time.time() - t0 < 40, "the ranks waited for a clock instead of each other"


@pytest.mark.parametrize("what", ["allreduce", "split", "ring", "ll", "recv"])
def test_a_peer_that_dies_is_an_error_not_a_hang(devsim_lib, what):
    """the last of 3 ranks exits without a word; the survivors' next collective (one kernel / meet-body-done / ring kernel / LL
    lines) or Receive from it returns an error within the no-progress limit, the waiting kernels have ended, the device works"""
    outs = run_ranks("peer_dies", 3, {"what": what}, timeout=120, env={"XMPI_TIMEOUT_S": "4", "XMPI_WATCHDOG_MS": "0"})
    assert sum("ok (error after" in o for o in outs) == 2, "\n".join(outs)


@pytest.mark.parametrize("what", ["allreduce", "split", "ring", "ll", "ll_agent", "recv", "recv_on_stream"])
def test_a_peer_that_dies_is_an_error_at_once_with_default_settings(devsim_lib, what):
    """XMPI_TIMEOUT_S unset (wait for ever, the default): the watchdog -- every rank's helper thread asks every 50 ms whether the
    other ranks' processes still exist -- raises the job's abort flag; XMPI_ERR_PEER within a second, the dead rank named"""
    outs = run_ranks("peer_dies", 3, {"what": what, "no_timeout": 1, "within": 1.5}, timeout=120, env={"XMPI_TIMEOUT_S": "0"})
    assert sum("ok (error after" in o for o in outs) == 2, "\n".join(outs)
    assert all("the process of rank 2" in o for o in outs[:2]), "\n".join(outs)


@pytest.mark.parametrize("what", ["length", "length_split", "schedule", "form", "operation", "root", "shape", "collective"])
def test_ranks_in_different_calls_get_an_error_not_a_hang(devsim_lib, what):
    """every kernel announces the call it is in; ranks that differ all fail at once, and nobody's buffer was written"""
    outs = run_ranks("mismatch", 3, {"what": what}, timeout=120, env={"XMPI_TIMEOUT_S": "20"})
    assert sum("ok (error after" in o for o in outs) == 3, "\n".join(outs)


@pytest.mark.parametrize("what", ["length", "operation", "root", "collective"])
def test_rank_threads_in_different_calls_get_an_error_not_a_hang(devsim_lib, what):
    """the same where ranks meet on the host: the descriptors they publish carry the call's signature"""
    out = run_threads("mismatch", 3, {"what": what, "threads": 1}, timeout=120)
    assert out.count("ok (error after") == 3, out


def test_the_stepped_kernels_shape_follows_the_most_crowded_gpu(devsim_lib):
    """5 ranks on 2 GPUs sit 3 + 2: what shapes a protocol (channels, workers of the stepped kernels) is read off the job's most
    crowded GPU -- the same figure on every rank -- not off the rank's own; every form still right in that layout"""
    run_ranks("sched", 5, {"shapes": [(0, 0)], "counts": [17, 4099], "quick": 1, "expect_params": {"dsync_sharers_job": 3}}, timeout=600,
              env={"DEVSIM_DEVICES": "2", "XMPI_NGPUS": "2"})


def test_ranks_that_share_a_device(devsim_lib):
    """the threads layout (one process, one device, pid-equal peers) through the same stand-in"""
    run_threads("allreduce_small", 4, {"counts": [1, 4099], "dtypes": [4, 2]})


# ---- the product's own executables: launcher + C++ front end + example programs, one process per virtual GPU -----------------------------
@pytest.fixture()
def stage(devsim_lib, tmp_path):
    """the binaries under mpi_amd/bin resolve libxmpi.so through their RUNPATH; LD_LIBRARY_PATH comes first: a directory whose
    libxmpi.so IS the stand-in puts the executables that ship on virtual devices, unchanged"""
    os.symlink(devsim_lib, tmp_path / "libxmpi.so")
    return dict(os.environ, LD_LIBRARY_PATH=str(tmp_path), XMPI_TIMEOUT_S="60")


def launch(env, ranks, prog, *args, port):
    binp = os.path.join(ROOT, "mpi_amd", "bin")
    e = dict(env, XMPI_NGPUS=str(ranks), DEVSIM_DEVICES=str(ranks), XMPI_BASEPORT=str(port))
    e.pop("XMPI_DEVSIM_LIB", None)
    r = subprocess.run([os.path.join(binp, "xmpirun"), str(ranks), os.path.join(binp, prog), *args], capture_output=True, text=True,
                       timeout=600, cwd=ROOT, env=e)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_reference_examples_one_process_per_gpu(stage):
    """the reference's two programs (examples/helloworld/helloworld.go:33-82, examples/bounce/bounce.go:37-153) as the C++ front end
    runs them, under the launcher that replaces gompirun.go:46-93, rank i on GPU i"""
    out = launch(stage, 4, "helloworld", port=7600)
    for me in range(4):
        for other in range(4):
            want = (f"I'm just node {me} talking to myself" if other == me else f"Hello node {me}, I'm node {other}")
            assert f'I, node {me}, received a message: "{want}"' in out, out
    out = launch(stage, 2, "bounce", port=7620)
    assert "Number of nodes =  2" in out and "Average float64 trip time" in out, out


def test_collective_programs_one_process_per_gpu(stage):
    """every schedule by name at 8 ranks on 8 GPUs (the tuner included), cfg 3 and cfg 5 at small sizes: exact, and nobody
    shares a device"""
    import json
    d = json.loads(launch(stage, 8, "allreduce_bench", "65536", "2", "1", "auto", "fused", "split", "ring", "rhd", "ring_push", "rhd_push", port=7640).strip().split("\n")[-1])
    assert d["ranks"] == 8 and d["sharers"] == 1 and d["exact"] is True and d["xcd_short"] == 0, d
    assert [r["mode"] for r in d["rows"]] == ["auto", "fused", "split", "ring", "rhd", "ring_push", "rhd_push"] and all(r["us_per_step"] > 0 for r in d["rows"])
    d = json.loads(launch(stage, 4, "cfg3_allgather", "32768", "2", port=7660).strip().split("\n")[-1])
    assert d["exact"] is True and d["ranks"] == 4 and d["ring"]["bit_exact_and_in_place"] and d["ring_push"]["bit_exact_and_in_place"] and d["auto"]["bit_exact_and_in_place"], d
    d = json.loads(launch(stage, 8, "cfg5_sweep", str(1 << 18), "2", port=7680).strip().split("\n")[-1])
    assert d["all_bit_identical"] is True, d
    out = launch(stage, 4, "allreduce", port=7700)
    assert "every result exact" in out, out


# ---- the driver's own multi-GPU command --------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("gpus", [8, 4, 2])
def test_the_scale_command_on_virtual_gpus(devsim_lib, gpus, tmp_path):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N
    --steps K --warmup W` -- the command the driver runs on an 8-GPU node and no 1-GPU box can (bench.py binds through ctypes:
    tests/devsim/site/usercustomize.py on PYTHONPATH puts every process of it, the zero-copy probe's children included, on the
    stand-in).  A quarter MiB per rank instead of 256: the line's shape and the path, not a rate.  Found this way: every process
    took all ranks to be on ITS device, so an 8-GPU line would have said "intra-HBM" and left the xgmi block out."""
    import json
    import sys
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "tests", "devsim", "site"), XMPI_DEVSIM_LIB=devsim_lib, DEVSIM_DEVICES=str(gpus),
               XMPI_TIMEOUT_S="60", XMPI_BENCH_EXTRAS_DIR=str(tmp_path))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
                        "--master-port", str(29600 + gpus), os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1",
                        "--size-mib", "0.25", "--no-cpu"], capture_output=True, text=True, timeout=900, cwd=str(tmp_path), env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 alone prints, one line
    d = json.loads(lines[0])
    assert d["n_gpus"] == gpus and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "strong" and d["higher_is_better"] is True
    assert d["config"]["ranks"] == 8 and d["config"]["ranks_per_gpu"] == 8 // gpus
    assert d["parity"]["ok"] is True and d["parity_failures"] == 0 and d["value"] > 0
    assert d["xgmi"]["link_probe"] and d["xgmi"]["meaningful"] is (gpus == 8)
    assert d["zero_copy_probe"].startswith("ok")
    if gpus == 8:  # north_star's layout: one rank per GPU, ranks meet on the device, the library's tuner chose the schedule
        assert d["config"]["transport"] == "xGMI (one rank per GPU)" and d["ranks_meet"] == "on the device (dsync)"
        assert d["roofline_hbm"]["kernel"].startswith("dsync_") and d["config"]["tuned"]
        # the links bound it: `roofline` is the link roofline of the schedule that was timed -- what its busiest link direction carries
        #  over the step time, against one direction of one link -- recomputable from the line's own fields
        rf, x = d["roofline"], d["xgmi"]
        assert rf["bound"] == "xgmi" and rf["peak"] == 76.8 and rf["unit"] == "GB/s" and rf["kernel"] == d["roofline_hbm"]["kernel"]
        share = {"ring": 1.75 / 6, "ring_push": 1.75 / 6, "rhd": 1.0, "rhd_push": 1.0}.get(rf["schedule"], 0.25)
        assert rf["busiest_link_direction_bytes_over_S"] == share == x["busiest_link_direction_bytes_over_S"] and rf["schedule"] == x["schedule"], rf
        assert rf["algorithmic_bytes_per_link_direction"] == share * d["config"]["bytes_per_rank"]
        assert abs(rf["achieved"] - rf["algorithmic_bytes_per_link_direction"] / (d["ms_per_step"] * 1e-3) / 1e9) < 1e-9
        assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and abs(x["frac_of_link_peak"] - rf["frac"]) < 1e-12
        assert rf["peak_measured"] == x["link_probe"]["copy_kernel_write_GBps"] > 0 and abs(rf["frac_of_measured"] - rf["achieved"] / rf["peak_measured"]) < 1e-12
        # north_star's target is quoted on ring: both forms of the ring kernel timed by name beside the library's choice
        ring = d["ring"]
        assert ring["best"] in ("pull", "push") and ring["channels"] == 6
        for form in ("pull", "push"):
            r = ring[form]
            assert r["parity_ok"] is True and r["busbw_GBps"] > 0
            assert abs(r["frac_of_link_peak"] - 1.75 / 6 * d["config"]["bytes_per_rank"] / (r["ms_per_step"] * 1e-3) / 76.8e9) < 1e-3 * r["frac_of_link_peak"]
        assert d["cpu_baseline"]["see"].startswith("the N = 1 line")
    else:
        assert d["config"]["transport"].startswith(f"mixed: 8 ranks on {gpus} GPUs")
        # several ranks per GPU: the one link between a pair of GPUs carries 2 R / gpus^2 x S per direction (at 2 GPUs: 4 S) -- that, not
        # an HBM, bounds the step; the HBM figure stays beside it
        rf = d["roofline"]
        assert rf["bound"] == "xgmi" and rf["busiest_link_direction_bytes_over_S"] == 2.0 * 8 / gpus ** 2 and rf["ranks_per_gpu"] == 8 // gpus, rf
        assert abs(rf["achieved"] - rf["algorithmic_bytes_per_link_direction"] / (d["ms_per_step"] * 1e-3) / 1e9) < 1e-9
        assert d["roofline_hbm"]["bound"] == "hbm" and "ring" not in d


@pytest.mark.parametrize("gpus", [8, 4, 2])
def test_the_scale_command_as_a_lone_process(devsim_lib, gpus, tmp_path):
    """`python bench.py --gpus N --steps K --warmup W` WITHOUT torch.distributed.run -- the `--gpus 1` command's shape carried to
    N GPUs: ONE process hosts all 8 ranks as threads and drives every GPU (bench.py Job.device_of: ranks spread over the visible
    devices; same-pid peers address each other's memory by pointer instead of through hipIpc, peer access enabled by xmpi_init --
    dsync.cpp dsync_connect).  At N = 8 every rank thread has a GPU to itself: the ranks meet on the device, the library tunes
    itself, the line's roofline is the link roofline -- the same line the torchrun form gives."""
    import json
    import sys
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "tests", "devsim", "site"), XMPI_DEVSIM_LIB=devsim_lib, DEVSIM_DEVICES=str(gpus),
               XMPI_TIMEOUT_S="60", XMPI_BENCH_EXTRAS_DIR=str(tmp_path))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    # (--no-probe: the child-process probe of every schedule is the torchrun form's test above; here: the lone process' own path)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1", "--size-mib", "0.25", "--no-cpu",
                        "--no-production", "--no-probe", "--no-extras"], capture_output=True, text=True, timeout=900, cwd=str(tmp_path), env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == gpus and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "strong"
    assert d["config"]["ranks"] == 8 and d["config"]["ranks_per_gpu"] == 8 // gpus
    assert d["parity"]["ok"] is True and d["parity_failures"] == 0 and d["value"] > 0
    assert d["roofline"]["bound"] == "xgmi"
    if gpus == 8:
        assert d["xgmi"]["meaningful"] is True
        assert d["config"]["transport"] == "xGMI (one rank per GPU)" and d["ranks_meet"] == "on the device (dsync)" and d["config"]["tuned"]
        assert d["roofline_hbm"]["kernel"].startswith("dsync_") and d["ring"]["best"] in ("pull", "push")
        assert "rejected" not in d["config"]["tuned"]  # (the tuner checked every candidate's answer: nothing wrong on these devices)
    else:
        assert d["config"]["transport"].startswith(f"mixed: 8 ranks on {gpus} GPUs") and d["roofline"]["ranks_per_gpu"] == 8 // gpus


def test_the_8gpu_script_rehearsal(stage, devsim_lib, tmp_path):
    """scripts/profile_8gpu.sh -- the one command for the day an 8-GPU node exists -- with XMPI_8GPU_REHEARSAL=1 (small sizes, no
    rocprofv3, no GPU suite): every other command line of it as written, on 8 virtual GPUs; every file it leaves parses, every
    result in them is exact, every mode it names ran, nobody shares a device"""
    import glob
    import json
    import sys
    env = dict(stage, PYTHONPATH=os.path.join(ROOT, "tests", "devsim", "site"), XMPI_DEVSIM_LIB=devsim_lib, DEVSIM_DEVICES="8", XMPI_NGPUS="8",
               XMPI_8GPU_REHEARSAL="1", XMPI_8GPU_OUT=str(tmp_path / "out"))
    r = subprocess.run(["bash", os.path.join(ROOT, "scripts", "profile_8gpu.sh"), "8"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = str(tmp_path / "out")
    line = json.loads(open(os.path.join(out, "bench_n8.json")).read().strip().split("\n")[-1])
    assert line["n_gpus"] == 8 and line["xgmi"]["meaningful"] is True and line["parity"]["ok"] is True
    json.load(open(os.path.join(out, "bench_n8_extras.json")))
    seen = set()
    for f in glob.glob(os.path.join(out, "*.json")):
        name = os.path.basename(f)
        if name.startswith("bench_n8"):
            continue
        d = json.loads(open(f).read().strip().split("\n")[-1])
        assert d.get("exact", d.get("all_bit_identical")) is True, (name, d)
        if "sharers" in d:
            assert d["sharers"] == 1 and d["xcd_short"] == 0, (name, d)
            seen |= {row["mode"] for row in d["rows"]}
        if name.startswith("split_body_sys"):
            assert d["body_sys"] == int(name[len("split_body_sys")])
        if name == "cfg5_n8.json":
            assert d["rows"], d
    assert seen == {"auto", "fused", "fused2", "split", "zpush", "ring", "ring_push", "rhd", "rhd_push"}, seen
    # (nothing but the launcher's one line per job: `xmpirun: 8 ranks on 8 GPUs`)
    assert [ln for ln in open(os.path.join(out, "prod.err")).read().splitlines() if ln.strip() and not ln.startswith("xmpirun: ")] == []
    # ... and the one-screen reading of it names what the first hour on a node has to look at
    rep = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "first_hour_report.py"), out], capture_output=True, text=True, timeout=120)
    assert rep.returncode == 0, rep.stderr[-2000:]
    for needle in ("degraded: no", "link roofline: schedule", "link probe sys_kernel", "the tuner chose", "found WRONG on this node: none", "self-check at init", "schedules rejected 0", "bcast     1 KiB", "reduce    1 KiB", "ring by name, pull", "ring by name, push",
                   "ring_push", "rhd_push", "XCD masks meet / done 0xff / 0xff", "link bound", "cfg 5"):
        assert needle in rep.stdout, (needle, rep.stdout[-3000:])


# ---- what the links would carry -----------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def traffic_lib():
    from tests.devsim import build
    return build.build_traffic_lib()


@pytest.mark.parametrize("size,count", [(2, 0), (3, 0), (4, 0), (5, 0), (8, 262144)])
def test_link_traffic_follows_the_plan(traffic_lib, size, count):
    """every load and store of the kernels traced, one process per virtual device: the payload bytes device i moves in device
    j's memory are the schedule's plan to the byte -- fold: S / N each way over every link; push-only 2 S / N stores; ring
    2 (N - 1) / N x S per rank over the Walecki cycles; halving S x d / N per partner; allgather, bcast (both forms), reduce,
    tree kernels, Send / Receive = one pull; LL = 2 S of lines per peer and no remote load -- and each device's memory serves
    2 S per allreduce by the fold (N reads + N writes per element of its chunk)"""
    outs = run_ranks("traffic", size, {"count": count} if count else None, timeout=600,
                     env={"XMPI_DEVSIM_LIB": traffic_lib, "DEVSIM_TRAFFIC": "1"})
    assert any(line.startswith("TRAFFIC ") for o in outs for line in o.splitlines())


# ---- dispatchers that do not deal small grids round every XCD ----------------------------------------------------------------------
def test_a_dispatcher_that_never_reaches_one_xcd(devsim_lib):
    """the 16-block probe grid of xmpi_init misses an XCD: the split form takes the system-scope data kernel by itself"""
    run_ranks("split", 2, {"counts": [4099]}, timeout=600, env={"DEVSIM_XCD_MAP": "small_miss"})
    run_ranks("devices", 3, {"counts": [4099], "expect_params": {"body_sys": 1}}, timeout=600, env={"DEVSIM_XCD_MAP": "small_miss"})


def test_the_guard_trips_on_virtual_devices(devsim_lib):
    run_ranks("split", 2, {"counts": [4099], "trip": 1}, timeout=600)


@pytest.mark.parametrize("seed", [5, 6])
def test_a_dispatcher_that_misses_an_xcd_now_and_then(devsim_lib, seed):
    """the probe passes, a later launch falls short: that collective is refused and the job aborted, no rank returns a result"""
    outs = run_ranks("xcd_flaky", 3, timeout=600, env={"DEVSIM_XCD_MAP": "flaky", "DEVSIM_FUZZ": str(seed)})
    assert any("the guard refused the launch" in o for o in outs), "\n".join(outs)
