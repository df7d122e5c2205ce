"""The C-ABI shared library loads and exports every symbol include/xmpi.h declares (no compute
calls: this runs without a GPU), and the product fails loudly -- no CPU fallback -- when no HIP
device is visible."""
import ctypes
import os
import re

import pytest

from mpi_amd import xmpi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header="xmpi.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
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
    scaffolding = declared_functions("xmpi_test.h")  # the same library's entry points for the test suites: not the boundary
    for name in scaffolding:
        assert hasattr(L, name), f"{name} is declared in include/xmpi_test.h but not exported by libxmpi.so"
    assert bound <= set(decl) | set(scaffolding), f"bound but undeclared: {bound - set(decl) - set(scaffolding)}"


def test_header_cites_the_reference_interface():
    text = open(os.path.join(ROOT, "include", "xmpi.h")).read()
    for cite in ("mpi.go:96-98", "network.go:53-65", "mpi.go:126-128", "network.go:518-572", "mpi.go:157-159",
                 "network.go:575-625", "mpi.go:102-104", "mpi.go:112-119", "mpi.go:130"):
        assert cite in text, cite


def test_every_declared_entry_of_the_boundary_header_cites_the_reference_or_says_it_has_no_counterpart():
    """include/xmpi.h is the drop-in boundary: every function it declares sits under a comment that names the reference interface
    it replaces (file:line) or says that the reference has none.  The test scaffolding lives in include/xmpi_test.h (same library,
    not bound by go/)"""
    import re
    text = open(os.path.join(ROOT, "include", "xmpi.h")).read()
    for name in ("xmpi_fill_pattern", "xmpi_ctl_selftest", "xmpi_plan_dump", "xmpi_sched_dump", "xmpi_heap_selftest", "xmpi_tune_decide", "xmpi_zc_chunk"):
        assert name + "(" not in text, f"{name} belongs in include/xmpi_test.h"
        assert name + "(" in open(os.path.join(ROOT, "include", "xmpi_test.h")).read()
    go = "".join(open(os.path.join(ROOT, "go", d, f)).read() for d, f in (("xgmi", "xgmi.go"), ("mpi_collectives", "collectives.go")))
    assert "xmpi_test.h" not in go
    # comment blocks and the declarations that follow each of them
    blocks = re.split(r"(/\*.*?\*/)", text, flags=re.S)
    comment = ""
    seen = 0
    for part in blocks:
        if part.startswith("/*"):
            comment = part if not re.fullmatch(r"/\* -+ .* -+ ?\*/", part.replace("\n", " ")) or len(part) > 200 else comment + part
            continue
        for m in re.finditer(r"\b(xmpi_[a-z0-9_]+)\s*\(", part):
            seen += 1
            cites = re.search(r"(mpi|network|flags|bounce|helloworld|gompirun)\.go:\d+", comment) or \
                re.search(r"reference has no|no counterpart|absent from the reference|reference has none|no reference counterpart", comment, re.I)
            assert cites, f"{m.group(1)}: the comment above it neither cites the reference nor says there is no counterpart:\n{comment[:300]}"
    assert seen >= 60


def test_every_knob_is_in_the_table():
    """INTEGRATION.md section 5 is THE table of environment variables and xmpi_set_param names: what the product code reads
    (getenv / env_long / env_size) or accepts (xmpi_set_param) must have a row, and a row must name something the code reads"""
    import glob
    import re
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = doc[doc.index("## 5. Every knob"):]
    rows = [ln for ln in sec.splitlines() if ln.startswith("| `XMPI_") or ln.startswith("| — |")]
    env_doc = set(re.findall(r"`(XMPI_[A-Z0-9_]+)`", "\n".join(ln.split("|")[1] for ln in rows)))
    par_doc = set(re.findall(r"`([a-z0-9_]+)`", "\n".join(ln.split("|")[2] for ln in rows)))
    src = ""
    for pat in ("mpi_amd/csrc/*.cpp", "mpi_amd/csrc/*.h", "mpi_amd/csrc/*.hip", "mpi_amd/host/*.cpp", "mpi_amd/host/*.hpp", "launcher/*.cpp"):
        for f in glob.glob(os.path.join(ROOT, pat)):
            src += open(f).read()
    env_code = set(re.findall(r"(?:getenv|env_long|env_size)\(\s*\"(XMPI_[A-Z0-9_]+)\"", src))
    assert env_code - env_doc == set(), f"read by the code, missing from INTEGRATION.md section 5: {sorted(env_code - env_doc)}"
    assert env_doc - env_code == set(), f"in the table, read by nothing: {sorted(env_doc - env_code)}"
    api = open(os.path.join(ROOT, "mpi_amd", "csrc", "api.cpp")).read()
    setter = api[api.index("int xmpi_set_param("):api.index("long xmpi_get_param(")]
    par_code = set(re.findall(r'n == "([a-z0-9_]+)"', setter))
    assert par_code - par_doc == set(), f"accepted by xmpi_set_param, missing from the table: {sorted(par_code - par_doc)}"
    assert par_doc - par_code == set(), f"in the table, not accepted by xmpi_set_param: {sorted(par_doc - par_code)}"
    getter = api[api.index("long xmpi_get_param("):api.index("int xmpi_tune_decide(") if "int xmpi_tune_decide(" in api else len(api)]
    getter = getter[:getter.index("\n}\n")]
    for name in set(re.findall(r'n == "([a-z0-9_]+)"', getter)):
        assert f"`{name}`" in sec, f"xmpi_get_param(\"{name}\") is not in INTEGRATION.md section 5"
    assert len(env_code) < 60, "knob sprawl"


def _split_args(text):
    """top-level comma split of an argument list (parentheses / brackets / braces nest)"""
    out, depth, cur = [], 0, ""
    for ch in text:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def test_the_go_shim_calls_the_boundary_as_declared():
    """go/xgmi/xgmi.go has never met a Go compiler (none in the image): what CAN be held to the header without one is -- every
    C.xmpi_* it calls is declared in include/xmpi.h (not in the test header), with as many arguments as the declaration has
    parameters; every C.XMPI_* constant exists; braces balance"""
    go = open(os.path.join(ROOT, "go", "xgmi", "xgmi.go")).read()
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "xmpi.h")).read(), flags=re.S)
    params = {}
    for m in re.finditer(r"\b(xmpi_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        args = m.group(2).strip()
        params[m.group(1)] = 0 if args in ("", "void") else len(_split_args(args))
    calls = 0
    for m in re.finditer(r"C\.(xmpi_[a-z0-9_]+)\(", go):
        name = m.group(1)
        if name in ("xmpi_dtype", "xmpi_op", "xmpi_algo"):  # (conversions to the header's enums)
            continue
        assert name in params, f"xgmi.go calls {name}, which include/xmpi.h does not declare"
        depth, i = 1, m.end()
        while depth:
            depth += {"(": 1, ")": -1}.get(go[i], 0)
            i += 1
        given = len(_split_args(go[m.end():i - 1]))
        assert given == params[name], f"xgmi.go calls {name} with {given} arguments, the header declares {params[name]}"
        calls += 1
    assert calls >= 40
    consts = set(re.findall(r"\bC\.(XMPI_[A-Z0-9_]+)", go))
    assert consts <= set(re.findall(r"\b(XMPI_[A-Z0-9_]+)\b", hdr)), consts - set(re.findall(r"\b(XMPI_[A-Z0-9_]+)\b", hdr))
    code = re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", go, flags=re.S))
    code = re.sub(r"\"(?:\\.|[^\"\\])*\"|`[^`]*`|'(?:\\.|[^'\\])'", "", code)
    for a, b in ("{}", "()", "[]"):
        assert code.count(a) == code.count(b), f"xgmi.go: {code.count(a)} '{a}' against {code.count(b)} '{b}'"


def test_design_describes_head_and_history_lives_in_the_changelog():
    """DESIGN.md is what the next hardware session reads first: what HEAD does, under 36 KB (30 before the answer checks, the
    self-check and the watchdog had to be described), no round-by-round history (that is CHANGELOG.md's), the 8-GPU checklist at the
    top naming the one script (rehearsed on virtual GPUs by tests/test_devsim.py)"""
    import re
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert len(text.encode()) < 36 * 1024, len(text.encode())
    assert not re.search(r"\bround\s+\d", text, re.I), re.search(r"\bround\s+\d", text, re.I)
    assert "## 0. The first hour on an 8-GPU node" in text and "scripts/profile_8gpu.sh" in text
    for form in ("Fold, one kernel", "Fold, split", "Push-only", "Stepped kernels", "pull", "push", "LL lines", "LL agent", "Host rendezvous",
                 "Host-driven step tables", "receive agent"):
        assert form in text, form
    assert os.path.exists(os.path.join(ROOT, "CHANGELOG.md"))


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
