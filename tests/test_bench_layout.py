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
