"""One rank of a multi-process GPU scenario:  python tests/rank_worker.py <scenario> <rank> <size> <key> [json]"""
import json
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    name, rank, size, key = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    args = json.loads(sys.argv[5]) if len(sys.argv) > 5 else {}
    from mpi_amd import xmpi
    from tests import scenarios
    comm = xmpi.Comm(rank, size, args.get("device", -1), key)
    try:
        for k, v in args.get("params", {}).items():
            comm.set_param(k, v)
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
