"""bench.py prints exactly one JSON line with the keys the driver reads, and smoke() passes."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]


def test_bench_json_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--size-mib", "16",
                        "--no-extras", "--cpu-count", "65536"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().split("\n") if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "GB/s" and d["value"] > 0 and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["dtype"] == "f32" and "workload" in d["config"] and "model" not in d["config"]
    # the value IS BASELINE.json's metric: algbw = bytes per rank / time per step (GB = 1e9 B), no rank multiplier
    assert "algbw" in d["metric"] and "aggregate" not in d["metric"]
    assert abs(d["value"] - d["config"]["bytes_per_rank"] / d["ms_per_step"] / 1e6) < 1e-6 * d["value"]
    r = d["config"]["ranks"]
    assert abs(d["busbw_GBps"] - d["value"] * 2 * (r - 1) / r) < 1e-6 * d["value"]
    assert "xGMI" not in d["config"]["transport"]  # eight ranks on one GPU never cross a link
    ro = d["roofline"]
    assert ro["bound"] == "hbm" and ro["unit"] == "GB/s" and ro["peak"] == 8000.0
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-9 and ro["launches"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert abs(cb["value"] - 65536 * 4 / cb["seconds_per_allreduce"] / 1e9) < 1e-6 * cb["value"]  # algbw too, not x ranks
    assert d["parity"]["checked"] and d["parity"]["ok"]
    assert not d["parity_failures"]


def test_bench_extras_tables():
    """the untimed extras at a small size: busbw against message size at 1 / 2 / 4 / 8 ranks, cfg 3 at its 4 ranks,
    and the one-process-per-rank sweep (ranks meeting on the device)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--size-mib", "16",
                        "--no-cpu"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.strip().split("\n") if ln.startswith("{")][0])
    rows = d["extras"]["busbw_table"]["rows"]
    assert {x["ranks"] for x in rows} == {1, 2, 4, 8}
    for x in rows:
        assert x["us"] > 0 and abs(x["busbw_GBps"] - x["algbw_GBps"] * 2 * (x["ranks"] - 1) / x["ranks"]) < 1e-9
    assert {x["bytes"] for x in rows if x["ranks"] == 8} == {1 << 10, 1 << 12, 1 << 14, 1 << 16, 1 << 18, 1 << 20, 1 << 22, 1 << 24}
    c3 = d["extras"]["cfg3_allgather_i64_16MiB_4ranks"]
    assert c3["ranks"] == 4 and c3["auto"]["ms"] > 0 and c3["ring"]["ms"] > 0
    mp = d["extras"]["multiprocess_sweep"]  # eight PROCESSES on this GPU, meeting on the device
    assert mp["ranks"] == 8 and mp["exact"] is True and "device" in mp["meet"], mp
    assert mp["rows"][0]["bytes"] == 1024 and mp["rows"][0]["queued_us"] < 1000, mp["rows"][0]


def test_coll_sweep_one_process_per_rank():
    """examples/coll_sweep under the launcher: 4 processes, blocking and stream-queued allreduce, exact results"""
    env = dict(os.environ, XMPI_TIMEOUT_S="60", XMPI_NGPUS="1", XMPI_BASEPORT="7400")
    r = subprocess.run([os.path.join(ROOT, "mpi_amd", "bin", "xmpirun"), "4", os.path.join(ROOT, "mpi_amd", "bin", "coll_sweep"),
                        str(1 << 20), "50"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(r.stdout.strip().split("\n")[-1])
    assert d["ranks"] == 4 and d["exact"] is True and "device" in d["meet"]
    assert [x["bytes"] for x in d["rows"]] == [1 << 10, 1 << 12, 1 << 14, 1 << 16, 1 << 18, 1 << 20]
    assert all(x["blocking_us"] > 0 and x["queued_us"] > 0 for x in d["rows"])


def test_smoke_entry():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "__graft_entry__.py"), "--smoke"], capture_output=True,
                       text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "smoke ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
