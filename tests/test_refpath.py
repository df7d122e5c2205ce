"""World-size 2 and 4 runs of the reference-path restatement (oracle/refpath.cpp) on the CPU:
N OS processes on localhost TCP ports, launched the way mpirun/gompirun/gompirun.go:46-93 does
(-mpi-addr :6000+i -mpi-alladdr csv).  Checks BASELINE config 1 (helloworld strings) and the
lossless bounce echo (bounce.go:105,133)."""
import json
import os
import random
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "refpath_bin")


def free_ports(n):
    """n TCP ports nobody listens on right now (asked of the kernel, below the ephemeral range's churn where possible)"""
    import socket
    ports = []
    for _ in range(200):
        p = random.randint(10000, 30000)
        if p in ports:
            continue
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            try:
                s.bind(("", p))
            except OSError:
                continue
        ports.append(p)
        if len(ports) == n:
            return ports
    raise RuntimeError("no free TCP ports")


def launch(mode, n, extra=(), timeout=120):
    for attempt in range(3):  # (a port can still be taken between the check and the rank's listen: another set, not a failure)
        ports = [f":{p}" for p in free_ports(n)]
        procs = [subprocess.Popen([BIN, mode, *extra, "-mpi-addr", p, "-mpi-alladdr", ",".join(ports)],
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for p in ports]
        outs = []
        for p in procs:
            try:
                out, _ = p.communicate(timeout=timeout)
            except subprocess.TimeoutExpired:
                p.kill()
                out, _ = p.communicate()
            outs.append(out)
        if attempt < 2 and any("listen failed" in o for o in outs):
            continue
        for p, out in zip(procs, outs):
            assert p.returncode == 0, out
        return outs


@pytest.mark.parametrize("n", [1, 2, 4])
def test_helloworld_matches_reference_text(n):
    """examples/helloworld/helloworld.go:51,59-62,78"""
    outs = launch("helloworld", n)
    seen = set()
    for out in outs:
        lines = out.strip().split("\n")
        rank = int(lines[0].split("node ")[1].split(" ")[0])
        seen.add(rank)
        assert lines[0] == f"Hello world, I'm node {rank} in a land with {n} nodes"
        got = sorted(lines[1:])
        want = sorted(
            f"I, node {rank}, received a message: " +
            (f"\"I'm just node {rank} talking to myself\"" if i == rank else f"\"Hello node {rank}, I'm node {i}\"")
            for i in range(n))
        assert got == want
    assert seen == set(range(n))


def test_bounce_echo_is_lossless():
    outs = launch("bounce", 2, extra=["100000", "2"])
    rows = [json.loads(o.strip().split("\n")[-1]) for o in outs]
    assert all(len(r["bytes_us"]) == 7 and len(r["float64_us"]) == 7 for r in rows)  # lengths 0, 1, 10 .. 1e5


@pytest.mark.parametrize("n", [2, 4])
def test_user_allreduce_rank_order(n):
    outs = launch("allreduce_f32", n, extra=["50000", "2"])
    for o in outs:
        r = json.loads(o.strip().split("\n")[-1])
        assert r["bad"] == 0 and r["ranks"] == n
