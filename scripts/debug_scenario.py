"""Run one tests/scenarios.py scenario on N rank processes with a trace line before every collective case
(development tool: tells which case a crashing run was in).  python scripts/debug_scenario.py NAME SIZE [json-args]"""
import json
import os
import subprocess
import sys
import uuid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(name, rank, size, key, args):
    from mpi_amd import xmpi
    from tests import scenarios

    def traced(fn):
        def wrap(comm, *a, **k):
            if comm.rank() == 0:
                print(f"[trace] {fn.__name__} {a} {k}", flush=True)
            return fn(comm, *a, **k)
        return wrap

    for fname in ("allreduce_case", "allgather_case"):
        setattr(scenarios, fname, traced(getattr(scenarios, fname)))
    comm = xmpi.Comm(rank, size, -1, key)
    scenarios.SCENARIOS[name](comm, args)
    comm.barrier()
    comm.finalize()
    print(f"rank {rank} ok", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 4 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], json.loads(sys.argv[6]))
        sys.exit(0)
    name, size = sys.argv[1], int(sys.argv[2])
    args = sys.argv[3] if len(sys.argv) > 3 else "{}"
    key = f"dbg{os.getpid()}-{uuid.uuid4().hex[:6]}"
    procs = [subprocess.Popen([sys.executable, __file__, "--child", name, str(r), str(size), key, args], cwd=ROOT,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(size)]
    for r, p in enumerate(procs):
        out, _ = p.communicate(timeout=300)
        print(f"--- rank {r} exit {p.returncode}\n" + "\n".join(out.strip().split("\n")[-6:]))
