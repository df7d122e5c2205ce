"""The C-ABI shared library loads and exports every symbol include/xmpi.h declares (no compute
calls: this runs without a GPU), and the product fails loudly -- no CPU fallback -- when no HIP
device is visible."""
import ctypes
import os
import re

import pytest

from mpi_amd import xmpi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "xmpi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(xmpi_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    L = xmpi.lib()
    bound = {name for name, _, _ in xmpi.SYMBOLS}
    decl = declared_functions()
    assert len(decl) >= 30
    for name in decl:
        assert hasattr(L, name), f"{name} is declared in include/xmpi.h but not exported by libxmpi.so"
        assert name in bound, f"{name} has no ctypes prototype in mpi_amd/xmpi.py"
    assert bound <= set(decl), f"bound but undeclared: {bound - set(decl)}"


def test_header_cites_the_reference_interface():
    text = open(os.path.join(ROOT, "include", "xmpi.h")).read()
    for cite in ("mpi.go:96-98", "network.go:53-65", "mpi.go:126-128", "network.go:518-572", "mpi.go:157-159",
                 "network.go:575-625", "mpi.go:102-104", "mpi.go:112-119", "mpi.go:130"):
        assert cite in text, cite


def test_uninitialised_answers_match_reference():
    """mpi.go:110-111 / network.go:41-50: Rank() == -1 and Size() == 0 before Init"""
    L = xmpi.lib()
    assert L.xmpi_rank(None) == -1
    assert L.xmpi_size(None) == 0
    assert L.xmpi_barrier(None) == xmpi.ERR_STATE
    assert L.xmpi_send(None, None, 0, xmpi.U8, 0, 0) == xmpi.ERR_STATE


def test_dtype_sizes_and_strerror():
    L = xmpi.lib()
    for dt, sz in xmpi.DTYPE_SIZE.items():
        assert L.xmpi_dtype_size(dt) == sz
    assert L.xmpi_dtype_size(99) == 0
    assert b"tag" in L.xmpi_strerror(xmpi.ERR_TAG_EXISTS)
    assert b"fallback" in L.xmpi_strerror(xmpi.ERR_NOGPU)
    assert L.xmpi_version().startswith(b"xmpi")


def _no_gpu():
    return not os.path.exists("/dev/kfd")


@pytest.mark.skipif(not _no_gpu(), reason="a GPU is present")
def test_init_fails_loudly_without_a_gpu():
    h = ctypes.c_void_p()
    rc = xmpi.lib().xmpi_init(0, 1, -1, b"nogpu", ctypes.byref(h))
    assert rc == xmpi.ERR_NOGPU and not h.value
    with pytest.raises(xmpi.XmpiError) as ei:
        xmpi.Comm(0, 1, -1, "nogpu")
    assert ei.value.code == xmpi.ERR_NOGPU


def test_product_does_not_touch_the_oracle():
    """nothing under mpi_amd/ or include/ may import, link or call anything under oracle/"""
    bad = []
    for base in ("mpi_amd", "include", "launcher", "examples", "go"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn == "build.py":  # compiles the checker (allowed); never loads it
                    continue
                if fn.endswith((".py", ".cpp", ".hip", ".h", ".hpp", ".go", ".c")):
                    src = open(os.path.join(dp, fn), errors="ignore").read()
                    for ln in src.splitlines():
                        s = ln.strip()
                        if s.startswith(("//", "#", "*", "/*")) and "include" not in s and "import" not in s:
                            continue
                        if re.search(r"(import|from)\s+oracle|#include\s+[\"<].*oracle|liboracle|librefpath", s):
                            bad.append((fn, s))
    assert not bad, bad
