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
    ro = d["roofline"]
    assert ro["bound"] == "hbm" and ro["unit"] == "GB/s" and ro["peak"] == 8000.0
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-9 and ro["launches"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert d["parity"]["checked"] and d["parity"]["ok"]
    assert not d["parity_failures"]


def test_smoke_entry():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "__graft_entry__.py"), "--smoke"], capture_output=True,
                       text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "smoke ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
