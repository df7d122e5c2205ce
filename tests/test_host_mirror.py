"""The C++ mirror of the reference's Go package (mpi_amd/host/mpi.hpp): host logic on the CPU, and the
reference's two example programs on the GPU through the launcher."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "mpi_amd", "bin")


def test_host_api_semantics(tmp_path):
    exe = str(tmp_path / "host_api_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I",
                           os.path.join(ROOT, "mpi_amd", "host"), os.path.join(ROOT, "tests", "host_api_check.cpp"),
                           "-o", exe, "-L", os.path.join(ROOT, "mpi_amd"), "-lxmpi_host", "-lxmpi",
                           "-Wl,-rpath," + os.path.join(ROOT, "mpi_amd"), "-lpthread"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr


def test_launcher_contract(tmp_path):
    """xmpirun passes -mpi-addr / -mpi-alladdr exactly like gompirun.go:46-51,77 and propagates the
    worst exit status (the reference drops it, gompirun.go:89)."""
    script = tmp_path / "probe.sh"
    script.write_text("#!/bin/sh\necho \"$@ dev=$XMPI_DEVICE job=$XMPI_JOB\"\ncase \"$*\" in *'-mpi-addr :6001'*) exit 3;; esac\nexit 0\n")
    script.chmod(0o755)
    r = subprocess.run([os.path.join(BIN, "xmpirun"), "3", str(script), "extra"], capture_output=True, text=True,
                       timeout=60, env={**os.environ, "XMPI_NGPUS": "2"})
    assert r.returncode == 3
    lines = sorted(r.stdout.strip().split("\n"))
    assert len(lines) == 3
    for i, ln in enumerate(lines):
        assert ln.startswith(f"extra -mpi-addr :600{i} -mpi-alladdr :6000,:6001,:6002 dev={i % 2} job=x")
    assert len({ln.split("job=")[1] for ln in lines}) == 1


def _fake_kfd(root, gpus, cpus=1):
    """a KFD topology directory: `cpus` CPU nodes (simd_count 0) and `gpus` GPU nodes, as /sys/class/kfd/kfd/topology lays them out"""
    for i in range(cpus + gpus):
        d = root / "nodes" / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count {64 if i < cpus else 0}\nsimd_count {0 if i < cpus else 1024}\nmem_banks_count 1\ngfx_target_version {0 if i < cpus else 90500}\n")
    return str(root)


def test_launcher_counts_the_gpus_the_ranks_will_see(tmp_path):
    """G comes from the KFD topology (the nodes with SIMDs: a CPU node, or another vendor's render node, is no GPU) narrowed by
    ROCR_VISIBLE_DEVICES and HIP_VISIBLE_DEVICES: on a masked node -- 4 of 8 GPUs visible -- 8 ranks sit two per GPU on devices
    0 .. 3 (rank 5 on device 1, not on a device 5 that xmpi_init would refuse); the launcher says so once; an XMPI_DEVICE that no
    rank will see is refused with a sentence.  (gompirun.go:57-93: N copies on one machine.)"""
    script = tmp_path / "probe.sh"
    script.write_text("#!/bin/sh\necho \"rank=$XMPI_RANK dev=$XMPI_DEVICE queues=$GPU_MAX_HW_QUEUES\"\n")
    script.chmod(0o755)
    kfd = _fake_kfd(tmp_path / "kfd", 8)
    base = {k: v for k, v in os.environ.items() if k not in ("XMPI_NGPUS", "XMPI_DEVICE", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES",
                                                              "GPU_MAX_HW_QUEUES")}
    base["XMPI_KFD_TOPOLOGY"] = kfd

    def run(n, **env):
        r = subprocess.run([os.path.join(BIN, "xmpirun"), str(n), str(script)], capture_output=True, text=True, timeout=60, env={**base, **env})
        return r, sorted(r.stdout.strip().split("\n"), key=lambda ln: int(ln.split()[0].split("=")[1])) if r.stdout.strip() else []

    r, lines = run(8)
    assert r.returncode == 0 and "xmpirun: 8 ranks on 8 GPUs (KFD topology)" in r.stderr and r.stderr.count("xmpirun:") == 1, r.stderr
    assert [ln.split()[1] for ln in lines] == [f"dev={i}" for i in range(8)] and all(ln.endswith("queues=") for ln in lines)
    r, lines = run(8, HIP_VISIBLE_DEVICES="0,1,2,3")
    assert r.returncode == 0 and "8 ranks on 4 GPUs" in r.stderr and "4 of 8 visible" in r.stderr and "several ranks per GPU" in r.stderr, r.stderr
    assert [ln.split()[1] for ln in lines] == [f"dev={i % 4}" for i in range(8)] and all(ln.endswith("queues=2") for ln in lines)
    r, lines = run(4, ROCR_VISIBLE_DEVICES="4,5,6,7,9", HIP_VISIBLE_DEVICES="1,3")  # (9: no such GPU -- the list ends there; HIP's indices count within ROCr's four)
    assert "4 ranks on 2 GPUs" in r.stderr and [ln.split()[1] for ln in lines] == ["dev=0", "dev=1", "dev=0", "dev=1"], r.stderr + r.stdout
    r, lines = run(2, ROCR_VISIBLE_DEVICES="GPU-abcdef0123456789,GPU-0123456789abcdef,GPU-00000000deadbeef")  # UUIDs are taken at their word
    assert "2 ranks on 3 GPUs" in r.stderr, r.stderr
    r, lines = run(8, HIP_VISIBLE_DEVICES="0,1,2,3", XMPI_DEVICE="5")
    assert r.returncode == 2 and lines == [] and "XMPI_DEVICE=5" in r.stderr and "4 GPUs" in r.stderr, r.stderr
    r, lines = run(3, XMPI_NGPUS="2")  # an explicit count wins
    assert "3 ranks on 2 GPUs (XMPI_NGPUS)" in r.stderr and [ln.split()[1] for ln in lines] == ["dev=0", "dev=1", "dev=0"]


def test_pin_parity_script_rehearsal(tmp_path):
    """scripts/pin_parity.sh -- the one command for the day a Go toolchain exists (the image has none: `go: command not found`) --
    rehearsed with a stub `go` on PATH that records where and how it was called: the sequence a maintainer would otherwise type
     -- go mod init in a COPY of the reference, collectives.go dropped in with its build tag stripped,
    go mod edit -replace in a copy of go/, go vet / build, go run ./golden -out tests/golden, the reference's helloworld.go and
    bounce.go byte for byte beside the one-line Register file -- and that the reference checkout is only ever read
    (/root/reference is read-only for us; here: a read-only copy, hashed before and after).  (mpi.go:56-67.)"""
    import hashlib
    import shutil
    import stat
    if not os.path.isdir("/root/reference"):
        pytest.skip("the reference checkout is not on this machine")
    ref = tmp_path / "ref_ro"
    shutil.copytree("/root/reference", ref)

    def tree_hash(root):
        h = hashlib.sha256()
        for d, _, fs in sorted(os.walk(root)):
            for f in sorted(fs):
                h.update(os.path.relpath(os.path.join(d, f), root).encode())
                h.update(open(os.path.join(d, f), "rb").read())
        return h.hexdigest(), sorted(os.listdir(root))
    before = tree_hash(ref)
    for d, ds, fs in os.walk(ref):
        for x in fs:
            os.chmod(os.path.join(d, x), stat.S_IRUSR | stat.S_IRGRP | stat.S_IROTH)
        os.chmod(d, stat.S_IRUSR | stat.S_IXUSR)
    log = tmp_path / "go.log"
    stub = tmp_path / "bin"
    stub.mkdir()
    (stub / "go").write_text(f"""#!/bin/bash
echo "$PWD|$*" >> {log}
case "$1 $2" in
  "version "*) echo "go version go1.22.0 stub/amd64";;
  "mod init") echo "module $3" > go.mod;;
  "mod edit") echo "// $3 $4" >> go.mod;;
esac
prev=""
for a in "$@"; do
  if [ "$prev" = "-o" ]; then printf '#!/bin/sh\nexit 0\n' > "$a"; chmod +x "$a"; fi
  prev="$a"
done
exit 0
""")
    (stub / "go").chmod(0o755)
    env = dict(os.environ, PATH=f"{stub}:{os.environ['PATH']}", REF=str(ref), PIN_PARITY_REHEARSAL="1", PIN_PARITY_KEEP="1")
    try:
        r = subprocess.run(["bash", os.path.join(ROOT, "scripts", "pin_parity.sh")], capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    finally:
        for d, ds, fs in os.walk(ref):
            os.chmod(d, 0o755)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert tree_hash(ref) == before, "the reference checkout was written to"
    calls = [ln.split("|", 1) for ln in log.read_text().splitlines()]
    work = next(cwd for cwd, a in calls if a == "mod init github.com/btracey/mpi")
    T = os.path.dirname(work)
    assert work == os.path.join(T, "ref") and T.startswith("/tmp/pin_parity.")
    seq = [(os.path.relpath(cwd, T), a) for cwd, a in calls if not a.startswith("version")]
    want = [("ref", "mod init github.com/btracey/mpi"),
            ("go", f"mod edit -replace github.com/btracey/mpi={T}/ref"),
            ("ref", "vet ."), ("ref", "build ."),
            ("go", "vet ./..."), ("go", "build ./..."),
            ("go", f"run ./golden -out {ROOT}/tests/golden"),
            ("run", f"build -o {T}/run/helloworld.bin ./helloworld"),
            ("run", f"build -o {T}/run/bounce.bin ./bounce")]
    assert seq == want, seq
    # what the compiler would have been given: collectives.go as a file of package mpi (tag stripped), the shim's sources, the
    # reference's programs byte for byte, each beside the one-line Register
    col = open(os.path.join(T, "ref", "collectives.go")).read()
    assert not col.startswith("//go:build") and "\npackage mpi\n" in col and "type Collective interface" in col
    assert os.path.exists(os.path.join(T, "go", "xgmi", "xgmi.go")) and os.path.exists(os.path.join(T, "go", "golden", "gen_golden.go"))
    for prog in ("helloworld", "bounce"):
        assert open(os.path.join(T, "run", prog, prog + ".go"), "rb").read() == open(f"/root/reference/examples/{prog}/{prog}.go", "rb").read()
        reg = open(os.path.join(T, "run", prog, "register_xgmi.go")).read()
        assert "func init() { mpi.Register(&xgmi.Backend{}) }" in reg and reg.startswith("package main")
    mod = open(os.path.join(T, "run", "go.mod")).read()
    assert f"replace github.com/btracey/mpi => {T}/ref" in mod and f"replace github.com/btracey/mpi-xgmi => {T}/go" in mod
    assert "go test ./xgmi skipped" in r.stdout and "the programs were built, not run" in r.stdout and "[pin_parity] done" in r.stdout
    assert "parity stays unpinned" in r.stdout or "parity PINNED" in r.stdout  # (the stub writes no fixtures; a tree that has them says so)
    shutil.rmtree(T, ignore_errors=True)


def test_pin_parity_script_says_what_is_missing(tmp_path):
    """without a Go toolchain (this image) the script stops at once with the sentence, touching nothing"""
    env = {k: v for k, v in os.environ.items()}
    env["PATH"] = ":".join(d for d in env["PATH"].split(":") if not os.path.exists(os.path.join(d, "go")))
    r = subprocess.run(["bash", os.path.join(ROOT, "scripts", "pin_parity.sh")], capture_output=True, text=True, timeout=60, env=env, cwd=str(tmp_path))
    assert r.returncode == 1 and "no Go toolchain on PATH" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 4])
def test_helloworld_program(n):
    """BASELINE config 1: same text as examples/helloworld/helloworld.go:51,59-62,78"""
    r = subprocess.run([os.path.join(BIN, "xmpirun"), str(n), os.path.join(BIN, "helloworld")], capture_output=True,
                       text=True, timeout=300, env={**os.environ, "XMPI_BASEPORT": str(6100 + 10 * n)})
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().split("\n")
    for rank in range(n):
        assert f"Hello world, I'm node {rank} in a land with {n} nodes" in lines
        for src in range(n):
            msg = f"\"I'm just node {rank} talking to myself\"" if src == rank else f"\"Hello node {rank}, I'm node {src}\""
            assert f"I, node {rank}, received a message: {msg}" in lines
    assert len(lines) == n + n * n


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [[], ["--host"]])
def test_bounce_program(mode):
    """BASELINE config 2 / examples/bounce/bounce.go: lossless echo at every message length"""
    r = subprocess.run([os.path.join(BIN, "xmpirun"), "2", os.path.join(BIN, "bounce"), *mode], capture_output=True,
                       text=True, timeout=600, env={**os.environ, "XMPI_BASEPORT": "6200"})
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Number of nodes =  2" in r.stdout
    assert "Average byte trip time in µs between node 0 and 1: [" in r.stdout
    assert "Average float64 trip time in µs between node 0 and 1: [" in r.stdout
    assert "message not the same" not in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("n,elems", [(2, 1 << 20), (4, 100003), (8, 4099)])
def test_allreduce_program(n, elems):
    """examples/allreduce.cpp: the collectives through the C++ mirror's package-level functions, launched like a
    reference program (xmpirun = gompirun); the program checks every result against its closed form"""
    r = subprocess.run([os.path.join(BIN, "xmpirun"), str(n), os.path.join(BIN, "allreduce"), str(elems)],
                       capture_output=True, text=True, timeout=300, env={**os.environ, "XMPI_BASEPORT": str(6300 + 10 * n)})
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"allreduce of {elems} float32 over {n} nodes" in r.stdout and "every result exact" in r.stdout


def _build_cgo_shape_check(tmp_path):
    exe = str(tmp_path / "cgo_shape_check")
    subprocess.check_call(["gcc", "-O1", "-std=c11", "-Wall", "-Wextra", "-Werror", "-D_GNU_SOURCE", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cgo_shape_check.c"), "-o", exe, "-L", os.path.join(ROOT, "mpi_amd"),
                           "-lxmpi", "-Wl,-rpath," + os.path.join(ROOT, "mpi_amd"), "-lpthread"])
    return exe


def test_cgo_shape_check_compiles_as_plain_c(tmp_path):
    """include/xmpi.h is consumable by a C compiler exactly as cgo would see it (no C++-isms), and the program that
    drives the ABI the way go/xgmi/xgmi.go does links against libxmpi.so"""
    assert os.path.exists(_build_cgo_shape_check(tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 4])
def test_cgo_calling_pattern(n, tmp_path):
    """tests/cgo_shape_check.c: every ABI call from a fresh OS thread that never selected a device, stack
    out-parameters, (NULL, 0) for empty slices, concurrent Send / Receive with distinct {peer, tag} -- the contract
    cgo imposes, tested without a Go toolchain"""
    exe = _build_cgo_shape_check(tmp_path)
    r = subprocess.run([os.path.join(BIN, "xmpirun"), str(n), exe], capture_output=True, text=True, timeout=300,
                       env={**os.environ, "XMPI_BASEPORT": str(6500 + 10 * n), "XMPI_TIMEOUT_S": "60"})
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"cgo shape check: {n} ranks, every call from a fresh OS thread: ok" in r.stdout
