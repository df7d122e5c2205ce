"""The host side of the library -- the shared-memory protocols of ctl.cpp and engine.cpp's blocking Send / Receive of host slices --
built with -fsanitize=thread and raced with the ranks as threads of one process (tests/tsan_host_driver.cpp).  The *_sim.py tests
check models of the protocols; this checks the atomics of the code that ships.  No GPU involved."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tsan_bin():
    from mpi_amd import build
    return build.build_tsan()


def run(binp, *args):
    env = dict(os.environ, TSAN_OPTIONS="exitcode=66 halt_on_error=0 report_signal_unsafe=0")
    return subprocess.run([binp, *args], capture_output=True, text=True, timeout=600, env=env, cwd="/tmp")


def test_the_sanitizer_sees_the_control_block(tsan_bin):
    """a seeded race on a plain field of a mail entry, written by two ranks through the shared mapping, is reported"""
    r = run(tsan_bin, "--seed-race")
    assert r.returncode == 66 and "ThreadSanitizer: data race" in r.stderr, r.stderr[-2000:]


@pytest.mark.parametrize("ranks", [1, 2, 3, 4, 8])
def test_host_protocols_are_race_free(tsan_bin, ranks):
    r = run(tsan_bin, str(ranks), "4")
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0 and "tsan_host_driver ok" in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]
