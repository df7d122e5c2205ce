"""One rank of a multi-process GPU scenario:  python tests/rank_worker.py <scenario> <rank> <size> <key> [json]"""
import json
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _binding():
    """mpi_amd.xmpi -- bound to libxmpi.so, or, for the CPU suite's tests/devsim runs ONLY, to the stand-in the test named
    (the same sources compiled for the host over a simulated HIP runtime; the product binding has no such switch)"""
    from mpi_amd import xmpi
    if os.environ.get("XMPI_DEVSIM_LIB"):
        xmpi.LIB_PATH = os.environ["XMPI_DEVSIM_LIB"]
    return xmpi


def threads_main():
    """python tests/rank_worker.py --threads <scenario> <size> [json]: every rank a thread of this process"""
    import threading
    import uuid
    name, size = sys.argv[2], int(sys.argv[3])
    args = json.loads(sys.argv[4]) if len(sys.argv) > 4 else {}
    xmpi = _binding()
    from tests import scenarios
    key = f"th{os.getpid()}-{uuid.uuid4().hex[:8]}"
    errors = []

    def body(r):
        try:
            comm = xmpi.Comm(r, size, args.get("device", -1), key)
            for k, v in args.get("params", {}).items():
                comm.set_param(k, v)
            if scenarios.SCENARIOS[name](comm, args) == "aborted":
                return  # (the scenario drove the job into an error on purpose: no barrier, no finalize to meet the others in)
            comm.barrier()
            comm.finalize()
        except BaseException:  # noqa: BLE001
            errors.append((r, traceback.format_exc()))

    ts = [threading.Thread(target=body, args=(r,)) for r in range(size)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errors:
        print("\n".join(f"rank {r}:\n{tb}" for r, tb in errors))
        sys.stdout.flush()
        os._exit(1)
    print(f"{size} rank threads {name}: ok")


def main():
    if sys.argv[1] == "--threads":
        return threads_main()
    name, rank, size, key = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    if os.environ.get("XMPI_TEST_DUMP_AFTER"):  # where is a rank that hangs?  (python stacks of all its threads)
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["XMPI_TEST_DUMP_AFTER"]), exit=False)
    args = json.loads(sys.argv[5]) if len(sys.argv) > 5 else {}
    xmpi = _binding()
    from tests import scenarios
    comm = xmpi.Comm(rank, size, args.get("device", -1), key)
    try:
        if name not in ("degraded", "corrupt"):  # nothing a test machine should refuse to map or get wrong: a job that quietly ran a level down would still pass
            assert comm.get_param("degraded") & 14 == 0, f"the job is degraded: {comm.degraded()}"
        for k, v in args.get("params", {}).items():
            comm.set_param(k, v)
        for k, v in args.get("expect_params", {}).items():
            assert comm.get_param(k) == v, f"{k} = {comm.get_param(k)}, expected {v}"
        scenarios.SCENARIOS[name](comm, args)
        comm.barrier()
    except BaseException:
        traceback.print_exc()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(1)  # do not wait in finalize for peers that may be stuck
    comm.finalize()
    print(f"rank {rank}/{size} {name}: ok")


if __name__ == "__main__":
    main()
