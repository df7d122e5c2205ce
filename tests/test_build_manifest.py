"""mpi_amd/build.py decides what to rebuild from CONTENT hashes recorded beside the binaries, never from mtimes: a
stale binary must not be able to pass for a current one on a fresh checkout or after a copy to the GPU box."""
import os
import time

from mpi_amd import build


def test_digest_follows_content_not_mtime(tmp_path):
    src = tmp_path / "a.cpp"
    src.write_text("int f() { return 1; }\n")
    d1 = build._digest([str(src)], "flags")
    os.utime(src, (time.time() + 1000, time.time() + 1000))  # touch: same bytes
    assert build._digest([str(src)], "flags") == d1
    src.write_text("int f() { return 2; }\n")
    assert build._digest([str(src)], "flags") != d1
    assert build._digest([str(src)], "other flags") != build._digest([str(src)], "flags")


def test_a_target_without_a_matching_record_is_stale(tmp_path, monkeypatch):
    monkeypatch.setattr(build, "MANIFEST", str(tmp_path / "manifest.json"))
    target = os.path.join(build.ROOT, "mpi_amd", "libxmpi.so")
    assert build._stale(target, "no such digest")           # exists, but nothing says what it was built from
    build._record(target, "abc")
    assert not build._stale(target, "abc")                  # built from exactly this
    assert build._stale(target, "abd")                      # a source changed
    assert build._stale(str(tmp_path / "missing.so"), "abc")


def test_the_shipped_library_matches_its_sources():
    """after build_all() (the session fixture) every recorded digest equals the digest of the sources as they are now"""
    objs = [os.path.join(build.OBJDIR, os.path.splitext(s)[0] + ".o") for s in build.LIB_SOURCES]
    assert all(os.path.exists(o) for o in objs)
    assert not build._stale(build.LIB, build._digest(objs, "link"))
