"""bench.py's launch layout (host logic, no GPU): how the 8 ranks of the job are spread over the processes
and GPUs of `--gpus N` under torch.distributed.run, and what the CPU baseline helpers return."""
import argparse
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _job(m, gpus, world=None, rank=0, local=0, monkeypatch=None):
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "XMPI_BENCH_DEVICE", "XMPI_BENCH_KEY"):
        monkeypatch.delenv(k, raising=False)
    if world is not None:
        monkeypatch.setenv("WORLD_SIZE", str(world))
        monkeypatch.setenv("RANK", str(rank))
        monkeypatch.setenv("LOCAL_RANK", str(local))
        monkeypatch.setenv("MASTER_PORT", "29500")
    for k in ("XMPI_SLOT_BYTES", "XMPI_FIFO_DEPTH", "GPU_MAX_HW_QUEUES"):  # Job() sets defaults for these: keep them
        monkeypatch.setenv(k, os.environ.get(k, "0"))                       # from leaking into later tests
    args = argparse.Namespace(gpus=gpus, ranks=0)
    return m.Job(args)


@pytest.mark.parametrize("gpus", [1, 2, 4, 8])
def test_ranks_are_spread_evenly_one_process_per_gpu(gpus, monkeypatch):
    m = _bench()
    seen, keys = [], set()
    for proc in range(gpus):
        job = _job(m, gpus, world=gpus if gpus > 1 else None, rank=proc, local=proc, monkeypatch=monkeypatch)
        assert job.ranks == 8 and job.ranks_per_proc == 8 // gpus
        mine = job.my_ranks()
        assert len(mine) == 8 // gpus and mine == list(range(mine[0], mine[0] + len(mine)))  # consecutive: adjacent chunks
        assert all(job.device_of(g) == (proc if gpus > 1 else 0) for g in mine)
        seen += mine
        keys.add(job.key)
    assert sorted(seen) == list(range(8))
    assert len(keys) == 1 or gpus == 1  # every process of a run derives the same job key


def test_lone_process_drives_all_gpus(monkeypatch):
    m = _bench()
    job = _job(m, 4, monkeypatch=monkeypatch)
    assert job.my_ranks() == list(range(8))
    assert [job.device_of(g) for g in range(8)] == [0, 0, 1, 1, 2, 2, 3, 3]


def test_mismatched_world_size_is_refused(monkeypatch):
    m = _bench()
    with pytest.raises(SystemExit):
        _job(m, 4, world=2, monkeypatch=monkeypatch)


def test_cpu_baseline_reports_what_it_used():
    m = _bench()
    r = m.cpu_baseline(4, 1 << 16, 2)
    if r is None:
        pytest.skip("oracle/refpath_bin not built")
    assert "error" not in r, r
    assert r["kind"] == "port" and r["unit"] == "GB/s" and r["value"] > 0
    assert 1.0 <= r["cores"] <= r["host_cores"] and r["threads"] == 4 * 2 * 4  # busy cores measured, not the host's count
    assert "4 ranks" in r["sample"] and "2 repetitions" in r["sample"]
    assert abs(r["value"] - (1 << 18) / r["seconds_per_allreduce"] / 1e9) < 1e-9  # algbw = S / t, no rank multiplier


def test_cpu_reference_bounce_runs_the_reference_lengths():
    m = _bench()
    rows = m.cpu_bounce()
    if rows is None:
        pytest.skip("oracle/refpath_bin not built")
    assert [r["bytes"] for r in rows] == [0, 1, 10, 100, 1000, 10**4, 10**5, 10**6, 10**7]  # bounce.go:33
    assert all(r["round_trip_us"] > 0 for r in rows)


def test_cpu_baseline_at_every_rank_count_of_the_metric():
    """north_star: bus bandwidth at 1 / 2 / 4 / 8 ranks next to the reference's loopback-TCP path timed in the same run -- the
    cpu_baseline object carries `by_ranks` (2, 4 and the full communicator; one rank sends no message)"""
    m = _bench()
    cb = m.cpu_baseline(8, 1 << 14, 1)
    if cb is None:
        pytest.skip("oracle/refpath_bin not built")
    cb = m.cpu_by_ranks(cb, 8, 1 << 14)
    assert sorted(cb["by_ranks"]) == ["2", "4", "8"], cb
    for r, row in cb["by_ranks"].items():
        assert "error" not in row, row
        assert row["algbw_GBps"] > 0 and abs(row["busbw_GBps"] - row["algbw_GBps"] * 2 * (int(r) - 1) / int(r)) < 0.02 * row["busbw_GBps"] + 3e-5  # (both rounded to 5 places)
    assert cb["by_ranks"]["8"]["algbw_GBps"] == round(cb["value"], 5)
    assert m.cpu_by_ranks({"error": "x"}, 8, 16) == {"error": "x"}  # (a failed sample stays what it is)


def test_the_quoted_pmc_profile_is_not_stale():
    """bench.py reads roofline.traffic from profiles/pmc_traffic.json (separate rocprofv3 --pmc passes, committed): the file must
    come from the newest round that has a profile directory, and every kernel it names must still be a kernel of the sources --
    a stale profile quoted beside a live time would be a figure for another binary"""
    import glob
    import json
    import re
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
        pmc = json.load(f)
    rounds = sorted(int(m.group(1)) for d in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]")) for m in [re.search(r"r(\d\d)$", d)] if m)
    assert pmc["round"] == rounds[-1], f"profiles/pmc_traffic.json is round {pmc['round']}, the newest profile directory is r{rounds[-1]:02d}: re-run the --pmc passes"
    src = "".join(open(f).read() for f in glob.glob(os.path.join(ROOT, "mpi_amd", "csrc", "*.hip")))
    for row in pmc["rows"]:
        stem = row["kernel"].split("<")[0]
        assert re.search(r"__global__[^;{]*\b" + re.escape(stem) + r"\s*\(", src), f"{stem}: named by profiles/pmc_traffic.json, not a kernel of mpi_amd/csrc/*.hip"
        assert row["traffic_bytes_per_launch"] > 0 and row["algorithmic_bytes_per_launch"] > 0
