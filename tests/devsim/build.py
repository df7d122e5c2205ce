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


UBSAN_BIN = os.path.join(HERE, "devsim_ubsan_bin")


def build_driver_ubsan(force: bool = False) -> str:
    """the driver with -fsanitize=undefined,bounds over the library whole, kernels included (no GPU pool offers a sanitizer run: on the CPU
    build only) -- found: pointer arithmetic on the null bases of a stepped kernel's unused operands (sched.hip tile_apply)"""
    objs, rebuilt = _objects("ubsan", _flags("-O1", "-fsanitize=undefined,bounds", "-fno-omit-frame-pointer"), [os.path.join(HERE, "driver.cpp")], force)
    link = b._digest(objs, "devsim ubsan driver link")
    if force or rebuilt or b._stale(UBSAN_BIN, link):
        b._run([_clang(), "-fsanitize=undefined", *objs, "-o", UBSAN_BIN, "-lpthread", "-lrt", "-ldl"])
        b._record(UBSAN_BIN, link)
    return UBSAN_BIN


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


# Mutants: the sanitizer driver with ONE wait taken out of a COPY of the kernel sources (never anything but a test's input; the
# product source is untouched).  name -> (file the line is in, what is cut out, what goes in its place)
MUTATIONS = {
    # the stepped kernels no longer wait for the previous step's flag word of the peer
    "step": ("sched.hip",
             "const uint32_t why = dsync_spin(step_flags(mine) + (size_t)wait_rank * kStepSlots + w, (sh.epoch << 8) | st.wait_val, a.d);",
             "const uint32_t why = DSYNC_OK;  /* MUTANT: the step does not wait for its peer */"),
    # the closing block no longer waits for the peers' "done": the caller is told its buffers are final while peers still store into them
    # the LL agent answers its caller before its lanes have met: the caller reads a receive buffer the agent's other lanes still store into
    "agent": ("ll.hip",
              "    __syncthreads();\n    if (t == 0) {  // what ll_end does for a launched kernel of one block",
              "    /* MUTANT: lane 0 does not wait for the block's other lanes */\n    if (t == 0) {  // what ll_end does for a launched kernel of one block"),
    # push form of the halving kernel: every level's half lands in ONE region of the partner's landing block instead of a region per
    # level -- the partner of level k + 1 stores over what the owner may still be folding from level k, and no flag orders the two
    "land": ("sched_steps.h",
             "  for (int j = 1; j < k; j++) off += rhd_level_bytes(whole, j);",
             "  (void)k;  /* MUTANT: every halving level lands in the same region */"),
    "done": ("kdev.h",
             "if (sh.fail == DSYNC_OK) why = dsync_spin(&mine->done[t][0], sh.epoch, a);",
             "/* MUTANT: nobody waits for the peers' done */"),
}


def mutant_bin(name: str) -> str:
    return os.path.join(HERE, f"devsim_tsan_mutant_{name}_bin")


def build_mutant(name: str = "step", force: bool = False) -> str:
    fname, cut, put = MUTATIONS[name]
    objs, _ = _objects("tsan", _flags("-O1", "-fsanitize=thread"), [os.path.join(HERE, "driver.cpp")], force)
    objdir = os.path.join(HERE, "obj_mutant", name)
    os.makedirs(objdir, exist_ok=True)
    text = open(os.path.join(b.CSRC, fname)).read()
    assert text.count(cut) == 1, f"{fname} no longer holds the line mutant '{name}' removes: update tests/devsim/build.py MUTATIONS"
    # a header's mutant needs every file that includes it beside it (quoted includes look there first)
    copies = {fname: text.replace(cut, put)}
    units = [fname] if fname.endswith(".hip") else [s for s in b.LIB_SOURCES if s.endswith(".hip")]
    for u in units:
        copies.setdefault(u, open(os.path.join(b.CSRC, u)).read())
    for f, t in copies.items():
        path = os.path.join(objdir, f)
        if not os.path.exists(path) or open(path).read() != t:
            with open(path, "w") as fh:
                fh.write(t)
    fl = _flags("-O1", "-fsanitize=thread") + ["-mllvm", "-tsan-instrument-func-entry-exit=0"]
    jobs, rebuilt, swapped = [], False, {}
    for u in units:
        src = os.path.join(objdir, u)
        obj = os.path.join(objdir, os.path.splitext(u)[0] + ".o")
        swapped[os.path.splitext(u)[0] + ".o"] = obj
        d = b._digest([os.path.join(objdir, f) for f in sorted(copies)] + _headers(), " ".join(fl) + u)
        if force or b._stale(obj, d):
            jobs.append((subprocess.Popen([_clang(), *fl, "-c", src, "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True), obj, d, u))
    for proc, obj, d, u in jobs:
        out, _ = proc.communicate()
        if proc.returncode != 0:
            sys.stderr.write(f"clang++ mutant {name} {u}:\n{out}")
            raise RuntimeError("build failed: devsim mutant " + u)
        b._record(obj, d)
        rebuilt = True
    link_objs = [swapped.get(os.path.basename(o), o) for o in objs]
    out = mutant_bin(name)
    link = b._digest(link_objs, "devsim mutant link " + name)
    if force or rebuilt or b._stale(out, link):
        b._run([_clang(), "-fsanitize=thread", *link_objs, "-o", out, "-lpthread", "-lrt", "-ldl"])
        b._record(out, link)
    return out


if __name__ == "__main__":
    force = "--force" in sys.argv
    if "--tsan" in sys.argv:
        print("built:", build_driver(True, force))
    elif "--mutant" in sys.argv:
        print("built:", [build_mutant(m, force) for m in MUTATIONS])
    elif "--traffic" in sys.argv:
        print("built:", build_traffic_lib(force))
    elif "--ubsan" in sys.argv:
        print("built:", build_driver_ubsan(force))
    elif "--driver" in sys.argv:
        print("built:", build_driver(False, force))
    else:
        print("built:", build_lib(force))
