"""Build the CPU stand-in of libxmpi.so that tests/devsim runs (TEST INFRASTRUCTURE -- mpi_amd.build knows nothing of it).

  tests/devsim/libxmpi_devsim.so   the library's host sources + its gfx950 kernel sources, compiled by clang++ AS C++ against
                                   tests/devsim/include (a HIP runtime with N virtual devices, kernels as threads / fibers)
  tests/devsim/libxmpi_devsim_traffic.so  the same, the kernel sources with every load / store traced (--traffic): which device's
                                   memory the kernels of which device touch, in bytes
  tests/devsim/devsim_tsan_bin     the same objects with -fsanitize=thread + tests/devsim/driver.cpp (ranks as threads,
                                   every rank on a device of its own)

`python -m tests.devsim.build [--tsan | --driver | --traffic] [--force]`.  Nothing here needs hipcc's device side or a GPU.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

from mpi_amd import build as b

ROOT = b.ROOT
HERE = os.path.join(ROOT, "tests", "devsim")
INC = os.path.join(HERE, "include")
LIB = os.path.join(HERE, "libxmpi_devsim.so")
TSAN_BIN = os.path.join(HERE, "devsim_tsan_bin")
PLAIN_BIN = os.path.join(HERE, "devsim_bin")
TRAFFIC_LIB = os.path.join(HERE, "libxmpi_devsim_traffic.so")


def _clang() -> str:
    for c in ("/opt/rocm/lib/llvm/bin/clang++", shutil.which("clang++") or "", shutil.which("amdclang++") or ""):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("tests/devsim needs clang++ (the kernel sources use clang vector extensions)")


def _sources() -> list[str]:
    return [os.path.join(b.CSRC, s) for s in b.LIB_SOURCES] + [os.path.join(HERE, "runtime.cpp")]


def _headers() -> list[str]:
    hs = [os.path.join(b.CSRC, h) for h in b.LIB_HEADERS]
    for d, _, fs in os.walk(INC):
        hs += [os.path.join(d, f) for f in fs]
    return sorted(hs)


def _flat(path: str) -> bool:
    """sources whose code runs on lane stacks (see runtime.cpp: no function entry / exit instrumentation under the sanitizer)"""
    return path.endswith(".hip") or os.path.basename(path) == "runtime.cpp"


def _objects(tag: str, flags: list[str], extra: list[str], force: bool) -> tuple[list[str], bool]:
    objdir = os.path.join(HERE, "obj_" + tag)
    os.makedirs(objdir, exist_ok=True)
    cxx = _clang()
    headers = _headers()
    objs, jobs = [], []
    for path in _sources() + extra:
        obj = os.path.join(objdir, os.path.splitext(os.path.basename(path))[0] + ".o")
        objs.append(obj)
        fl = flags + (["-mllvm", "-tsan-instrument-func-entry-exit=0"] if "-fsanitize=thread" in flags and _flat(path) else [])
        if tag == "traffic" and path.endswith(".hip"):  # every load / store of the kernels calls a hook of runtime.cpp
            fl = fl + ["-DDEVSIM_TRACED", "-fsanitize-coverage=func,trace-pc-guard,trace-loads,trace-stores"]
        d = b._digest([path] + headers, " ".join(fl))
        if force or b._stale(obj, d):
            jobs.append((subprocess.Popen([cxx, *fl, "-c", path, "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True),
                         obj, d, path))
    for proc, obj, d, path in jobs:
        out, _ = proc.communicate()
        if proc.returncode != 0:
            sys.stderr.write(f"clang++ {path}:\n{out}")
            raise RuntimeError("build failed: devsim " + os.path.basename(path))
        b._record(obj, d)
    return objs, bool(jobs)


def _flags(opt: str, *more: str) -> list[str]:
    return ["-x", "c++", "-std=c++17", opt, "-g", "-fPIC", "-DXMPI_DEVSIM", "-ffp-contract=off", "-mf16c", "-Wall", "-Wextra",
            "-I", INC, "-I", b.CSRC, *more]


def build_lib(force: bool = False) -> str:
    objs, rebuilt = _objects("plain", _flags("-O2"), [], force)
    link = b._digest(objs, "devsim link")
    if force or rebuilt or b._stale(LIB, link):
        b._run([_clang(), "-shared", "-fPIC", *objs, "-o", LIB, "-lpthread", "-lrt", "-ldl"])
        b._record(LIB, link)
    return LIB


def build_traffic_lib(force: bool = False) -> str:
    """the stand-in with the kernels' loads and stores counted per owning device (runtime.cpp, DEVSIM_TRAFFIC=1)"""
    objs, rebuilt = _objects("traffic", _flags("-O2"), [], force)
    link = b._digest(objs, "devsim traffic link")
    if force or rebuilt or b._stale(TRAFFIC_LIB, link):
        b._run([_clang(), "-shared", "-fPIC", *objs, "-o", TRAFFIC_LIB, "-lpthread", "-lrt", "-ldl"])
        b._record(TRAFFIC_LIB, link)
    return TRAFFIC_LIB


def build_driver(tsan: bool, force: bool = False) -> str:
    driver = os.path.join(HERE, "driver.cpp")
    if tsan:
        objs, rebuilt = _objects("tsan", _flags("-O1", "-fsanitize=thread"), [driver], force)
        out, extra = TSAN_BIN, ["-fsanitize=thread"]
    else:
        objs, rebuilt = _objects("plain", _flags("-O2"), [driver], force)
        out, extra = PLAIN_BIN, []
    link = b._digest(objs, "devsim driver link")
    if force or rebuilt or b._stale(out, link):
        b._run([_clang(), *extra, *objs, "-o", out, "-lpthread", "-lrt", "-ldl"])
        b._record(out, link)
    return out


MUTANT_BIN = os.path.join(HERE, "devsim_tsan_mutant_bin")
# (what is cut out, what goes in its place): the stepped kernels' wait for the previous step's flag word of the peer
MUTATION = ("const uint32_t why = dsync_spin(step_flags(mine) + (size_t)wait_rank * kStepSlots + w, (sh.epoch << 8) | st.wait_val, a.d);",
            "const uint32_t why = DSYNC_OK;  /* MUTANT: the step does not wait for its peer */")


def build_mutant(force: bool = False) -> str:
    """the sanitizer driver with ONE wait taken out of a copy of sched.hip (the ring / halving / tree kernels no longer wait for the
    peer's step): what the harness must find.  The copy lives under obj_mutant/ and is never anything but a test's input."""
    objs, _ = _objects("tsan", _flags("-O1", "-fsanitize=thread"), [os.path.join(HERE, "driver.cpp")], force)
    objdir = os.path.join(HERE, "obj_mutant")
    os.makedirs(objdir, exist_ok=True)
    src = os.path.join(b.CSRC, "sched.hip")
    text = open(src).read()
    assert text.count(MUTATION[0]) == 1, "sched.hip no longer holds the line the mutant removes: update tests/devsim/build.py MUTATION"
    mutated = os.path.join(objdir, "sched_mutant.hip")
    new_text = text.replace(MUTATION[0], MUTATION[1])
    if not os.path.exists(mutated) or open(mutated).read() != new_text:
        with open(mutated, "w") as f:
            f.write(new_text)
    obj = os.path.join(objdir, "sched.o")
    fl = _flags("-O1", "-fsanitize=thread") + ["-mllvm", "-tsan-instrument-func-entry-exit=0"]
    d = b._digest([mutated] + _headers(), " ".join(fl))
    rebuilt = False
    if force or b._stale(obj, d):
        b._run([_clang(), *fl, "-c", mutated, "-o", obj])
        b._record(obj, d)
        rebuilt = True
    link_objs = [obj if os.path.basename(o) == "sched.o" else o for o in objs]
    link = b._digest(link_objs, "devsim mutant link")
    if force or rebuilt or b._stale(MUTANT_BIN, link):
        b._run([_clang(), "-fsanitize=thread", *link_objs, "-o", MUTANT_BIN, "-lpthread", "-lrt", "-ldl"])
        b._record(MUTANT_BIN, link)
    return MUTANT_BIN


if __name__ == "__main__":
    force = "--force" in sys.argv
    if "--tsan" in sys.argv:
        print("built:", build_driver(True, force))
    elif "--mutant" in sys.argv:
        print("built:", build_mutant(force))
    elif "--traffic" in sys.argv:
        print("built:", build_traffic_lib(force))
    elif "--driver" in sys.argv:
        print("built:", build_driver(False, force))
    else:
        print("built:", build_lib(force))
