"""Launch N rank processes (one per GPU when the box has N, otherwise sharing device 0) and wait."""
import json
import os
import subprocess
import sys
import threading
import uuid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_ranks(scenario: str, size: int, args: dict | None = None, timeout: float = 300.0, env: dict | None = None,
              expect_failure: bool = False):
    """expect_failure: the test WANTS ranks to fail (fault injection) -- the AssertionError is raised all the same, but no
    post-mortem is left in gpurun_out/ (a fail_*.log there reads as an unexplained failure)"""
    key = f"t{os.getpid()}-{uuid.uuid4().hex[:8]}"
    e = dict(os.environ)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e.setdefault("XMPI_TIMEOUT_S", "60")
    e.setdefault("GPU_MAX_HW_QUEUES", "2")  # the ranks of a test share one GPU and its hardware queues (see launcher/xmpirun.cpp)
    e.setdefault("XMPI_TEST_DUMP_AFTER", str(max(30.0, timeout - 30.0)))  # a rank that hangs says where before it is killed
    e.update(env or {})
    if e.get("XMPI_DEVSIM_LIB"):  # the CPU suite's tests/devsim runs: every rank on a virtual device of its own
        e.setdefault("DEVSIM_DEVICES", str(size))
    procs = []
    for r in range(size):
        cmd = [sys.executable, os.path.join(ROOT, "tests", "rank_worker.py"), scenario, str(r), str(size), key,
               json.dumps(args or {})]
        procs.append(subprocess.Popen(cmd, cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [None] * size

    def wait(i):
        try:
            outs[i], _ = procs[i].communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            procs[i].kill()
            outs[i] = (procs[i].communicate()[0] or "") + "\n[harness] killed after timeout"

    ts = [threading.Thread(target=wait, args=(i,)) for i in range(size)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if e.get("XMPI_DEVSIM_LIB"):
        # a rank that was killed, aborted or left through os._exit (the fault-injection tests) never unlinked the shared-memory files
        # behind its virtual device's allocations, nor the job's control block: they are this container's MEMORY until somebody does
        import glob
        for f in [x for p in procs for x in glob.glob(f"/dev/shm/devsim.{p.pid}.*")] + glob.glob(f"/dev/shm/xmpi-*-{key}*"):
            try:
                os.unlink(f)
            except OSError:
                pass
    failed = [i for i, p in enumerate(procs) if p.returncode != 0]
    if failed:
        msg = "\n".join(f"--- rank {i} (exit {procs[i].returncode}) ---\n{outs[i]}" for i in range(size))
        dump = os.path.join(ROOT, "gpurun_out")  # survives the GPU box: post-mortem of intermittent failures
        if os.path.isdir(dump) and not expect_failure:
            with open(os.path.join(dump, f"fail_{scenario}_{size}_{key}.log"), "w") as f:
                f.write(msg)
        raise AssertionError(f"scenario {scenario} size {size}: ranks {failed} failed\n{msg}")
    return outs


def run_threads(scenario: str, size: int, args: dict | None = None, timeout: float = 600.0):
    """All ranks as threads of ONE process (pid-equal peers share pointers instead of hipIpc) -- a child process, not
    the test runner: a GPU schedules the queues of at most 8 processes at once (8 compute VMIDs), so a runner that held
    a HIP context of its own would make every 8-process test the 9th process' problem (time-sliced: ~25 s per
    communicator lifetime instead of 0.15 s, measured with tests/test_gpu_collectives.py::test_lifecycle_stress)."""
    e = dict(os.environ)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e.setdefault("XMPI_TIMEOUT_S", "60")
    if e.get("XMPI_DEVSIM_LIB"):
        e.setdefault("DEVSIM_DEVICES", "1")  # (the threads layout: ranks that share a device)
    cmd = [sys.executable, os.path.join(ROOT, "tests", "rank_worker.py"), "--threads", scenario, str(size), json.dumps(args or {})]
    p = subprocess.run(cmd, cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    assert p.returncode == 0, f"scenario {scenario} with {size} rank threads failed\n{p.stdout[-6000:]}"
    return p.stdout
