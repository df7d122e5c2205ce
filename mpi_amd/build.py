"""Build the native pieces in-tree (no JIT cache: the .so files travel with the repo snapshot).

  libxmpi.so        HIP kernels + engine + C ABI  (hipcc --offload-arch=gfx950)
  libxmpi_host.so   C++ host mirror of the reference's Go API (mpi.hpp) on top of the C ABI
  bin/xmpirun       one-process-per-GPU launcher (replaces mpirun/gompirun/gompirun.go)
  bin/helloworld, bin/bounce   the reference's two example programs against the mirror

`python -m mpi_amd.build` rebuilds whatever is out of date; `--force` rebuilds everything.
"""
from __future__ import annotations

import hashlib
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mpi_amd", "csrc")
LIB = os.path.join(ROOT, "mpi_amd", "libxmpi.so")
HOSTLIB = os.path.join(ROOT, "mpi_amd", "libxmpi_host.so")
BIN = os.path.join(ROOT, "mpi_amd", "bin")

HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
ARCH = "gfx950"
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wextra"]

LIB_SOURCES = ["kernels.hip", "sched.hip", "ll.hip", "engine.cpp", "api.cpp", "ctl.cpp", "plan.cpp", "zcopy.cpp", "heap.cpp", "dsync.cpp", "trace.cpp"]
LIB_HEADERS = ["kernels.h", "kdev.h", "sched_steps.h", "comm.h", "ctl.h", "plan.h", "trace.h", os.path.join("..", "..", "include", "xmpi.h"),
               os.path.join("..", "..", "include", "xmpi_test.h")]


MANIFEST = os.path.join(ROOT, "mpi_amd", ".build_manifest.json")
OBJDIR = os.path.join(ROOT, "mpi_amd", "csrc", "obj")


def _digest(paths: list[str], extra: str = "") -> str:
    """Content hash of the files (and the command line): what a target was built FROM.  Targets are rebuilt when
    this differs from what the manifest beside them recorded -- never by mtimes, which a fresh checkout, a copy to
    the GPU box or `touch` make meaningless (a stale binary must not be able to pass for a current one)."""
    h = hashlib.sha256(extra.encode())
    for q in paths:
        h.update(os.path.basename(q).encode())
        if os.path.exists(q):
            with open(q, "rb") as f:
                h.update(f.read())
    return h.hexdigest()


def _manifest() -> dict:
    try:
        with open(MANIFEST) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def _stale(target: str, digest: str) -> bool:
    return not os.path.exists(target) or _manifest().get(os.path.relpath(target, ROOT)) != digest


def _record(target: str, digest: str) -> None:
    m = _manifest()
    m[os.path.relpath(target, ROOT)] = digest
    tmp = MANIFEST + f".{os.getpid()}.tmp"
    with open(tmp, "w") as f:
        json.dump(m, f, indent=1, sort_keys=True)
    os.replace(tmp, MANIFEST)


def _newer(target: str, deps: list[str]) -> bool:
    """`target` is missing or was built from other contents than `deps` have now"""
    return _stale(target, _digest(deps))


def _run(cmd: list[str], target: str | None = None, deps: list[str] | None = None) -> None:
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise RuntimeError("build failed: " + cmd[0])
    if target is not None:
        _record(target, _digest(deps or []))


def build_lib(force: bool = False) -> str:
    """One object per source (kernels.hip alone takes over a minute), then the link."""
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in LIB_HEADERS]
    flags = [f"--offload-arch={ARCH}", *CXXFLAGS]
    objs, jobs = [], []
    for src in LIB_SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        d = _digest([path] + headers, " ".join(flags))
        if force or _stale(obj, d):
            jobs.append((subprocess.Popen([HIPCC, *flags, "-c", path, "-o", obj], stdout=subprocess.PIPE,
                                          stderr=subprocess.STDOUT, text=True), obj, d, src))
    for proc, obj, d, src in jobs:  # the compilations run side by side
        out, _ = proc.communicate()
        if proc.returncode != 0:
            sys.stderr.write(f"hipcc {src}:\n{out}")
            raise RuntimeError("build failed: hipcc " + src)
        _record(obj, d)
    link = _digest(objs, "link")
    if force or jobs or _stale(LIB, link):
        _run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB, "-lpthread", "-lrt", "-ldl"])
        _record(LIB, _digest(objs, "link"))
    return LIB


def build_host(force: bool = False) -> list[str]:
    """C++ host mirror + launcher + example programs (skipped silently if sources are absent)."""
    outs = []
    os.makedirs(BIN, exist_ok=True)
    inc = ["-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "mpi_amd", "host")]
    host_dir = os.path.join(ROOT, "mpi_amd", "host")
    mpi_cpp = os.path.join(host_dir, "mpi.cpp")
    net_cpp = os.path.join(host_dir, "network.cpp")
    if os.path.exists(mpi_cpp):
        deps = [mpi_cpp, net_cpp, os.path.join(host_dir, "mpi.hpp"), os.path.join(host_dir, "network.hpp"),
                os.path.join(host_dir, "gobwire.hpp"), os.path.join(ROOT, "include", "xmpi.h"),
                os.path.join(ROOT, "include", "xmpi_test.h")]
        if force or _newer(HOSTLIB, deps):
            _run(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wextra", "-shared", *inc, mpi_cpp, net_cpp, "-o", HOSTLIB,
                  "-L", os.path.dirname(LIB), "-lxmpi", "-Wl,-rpath,$ORIGIN", "-lpthread"], HOSTLIB, deps)
        outs.append(HOSTLIB)
        for name, src in (("helloworld", os.path.join(ROOT, "examples", "helloworld.cpp")),
                          ("bounce", os.path.join(ROOT, "examples", "bounce.cpp")),
                          ("allreduce", os.path.join(ROOT, "examples", "allreduce.cpp")),
                          ("coll_sweep", os.path.join(ROOT, "examples", "coll_sweep.cpp")),
                          ("allreduce_bench", os.path.join(ROOT, "examples", "allreduce_bench.cpp")),
                          ("cfg5_sweep", os.path.join(ROOT, "examples", "cfg5_sweep.cpp")),
                          ("cfg3_allgather", os.path.join(ROOT, "examples", "cfg3_allgather.cpp"))):
            if os.path.exists(src):
                out = os.path.join(BIN, name)
                if force or _newer(out, [src] + deps):
                    _run(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", *inc, src, "-o", out, "-L",
                          os.path.dirname(LIB), "-lxmpi_host", "-lxmpi", "-Wl,-rpath,$ORIGIN/..", "-lpthread"],
                         out, [src] + deps)
                outs.append(out)
    launcher = os.path.join(ROOT, "launcher", "xmpirun.cpp")
    if os.path.exists(launcher):
        out = os.path.join(BIN, "xmpirun")
        if force or _newer(out, [launcher]):
            _run(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", launcher, "-o", out], out, [launcher])
        outs.append(out)
    return outs


def build_oracle(force: bool = False) -> list[str]:
    """The CPU oracle is test infrastructure; building the checker is not using it."""
    odir = os.path.join(ROOT, "oracle")
    outs = []
    src = os.path.join(odir, "xmpi_oracle.c")
    out = os.path.join(odir, "liboracle.so")
    if force or _newer(out, [src, os.path.join(odir, "xmpi_oracle.h")]):
        _run(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wall",
              "-Wextra", src, "-o", out, "-lm"], out, [src, os.path.join(odir, "xmpi_oracle.h")])
    outs.append(out)
    ref = os.path.join(odir, "refpath.cpp")
    if os.path.exists(ref):
        out = os.path.join(odir, "librefpath.so")
        if force or _newer(out, [ref, os.path.join(odir, "gob_codec.h")]):
            _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", ref, "-o", out, "-lpthread"],
                 out, [ref, os.path.join(odir, "gob_codec.h")])
        outs.append(out)
        out = os.path.join(odir, "refpath_bin")
        if force or _newer(out, [ref, os.path.join(odir, "gob_codec.h")]):
            _run(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-DREFPATH_MAIN", ref, "-o", out, "-lpthread"],
                 out, [ref, os.path.join(odir, "gob_codec.h")])
        outs.append(out)
    return outs


TSAN_BIN = os.path.join(ROOT, "tests", "tsan_host_bin")
TSAN_HOST_SOURCES = [s for s in LIB_SOURCES if s.endswith(".cpp")]


def build_tsan(force: bool = False) -> str:
    """tests/tsan_host_bin: the library's host sources (ctl.cpp, engine.cpp, heap.cpp, api.cpp, ...) compiled with
    -fsanitize=thread and driven by tests/tsan_host_driver.cpp with the ranks as threads (test infrastructure: the shared-memory
    protocols of the code that ships, raced under the sanitizer on the CPU).  The kernel objects are the library's own."""
    build_lib(False)
    objdir = os.path.join(ROOT, "mpi_amd", "csrc", "obj_tsan")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in LIB_HEADERS]
    flags = [f"--offload-arch={ARCH}", "-O1", "-g", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fsanitize=thread", "-Wall", "-Wextra"]
    driver = os.path.join(ROOT, "tests", "tsan_host_driver.cpp")
    objs, jobs = [], []
    for path in [os.path.join(CSRC, s) for s in TSAN_HOST_SOURCES] + [driver]:
        obj = os.path.join(objdir, os.path.splitext(os.path.basename(path))[0] + ".o")
        objs.append(obj)
        d = _digest([path] + headers, " ".join(flags))
        if force or _stale(obj, d):
            jobs.append((subprocess.Popen([HIPCC, *flags, "-c", path, "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True),
                         obj, d, path))
    for proc, obj, d, path in jobs:
        out, _ = proc.communicate()
        if proc.returncode != 0:
            sys.stderr.write(f"hipcc -fsanitize=thread {path}:\n{out}")
            raise RuntimeError("build failed: tsan " + os.path.basename(path))
        _record(obj, d)
    kernel_objs = [os.path.join(OBJDIR, os.path.splitext(s)[0] + ".o") for s in LIB_SOURCES if s.endswith(".hip")]
    link = _digest(objs + kernel_objs, "tsan link")
    if force or jobs or _stale(TSAN_BIN, link):
        _run([HIPCC, f"--offload-arch={ARCH}", "-fsanitize=thread", *objs, *kernel_objs, "-o", TSAN_BIN, "-lpthread", "-lrt", "-ldl"])
        _record(TSAN_BIN, link)
    return TSAN_BIN


def build_all(force: bool = False) -> None:
    build_lib(force)
    build_host(force)
    build_oracle(force)


if __name__ == "__main__":
    if "--tsan" in sys.argv:
        print("built:", build_tsan("--force" in sys.argv))
    else:
        build_all("--force" in sys.argv)
        print("built:", LIB)
