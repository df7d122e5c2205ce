"""Standalone HBM roofline of the local kernels (the per-chunk reduction of every ring step).
  python scripts/kernel_bench.py [--reps 20]
Timing = HIP events recorded by libxmpi around each launch on the stream it runs on."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mpi_amd import xmpi  # noqa: E402

HBM_PEAK = 8.0e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--sizes-mib", type=str, default="1,8,32,128,256,1024")
    a = ap.parse_args()
    comm = xmpi.Comm(0, 1, 0, f"kb{os.getpid()}")
    comm.prof_enable(True)
    rows = []
    for mib in [int(x) for x in a.sizes_mib.split(",")]:
        nbytes = mib << 20
        bufs = [comm.alloc(nbytes) for _ in range(9)]
        for i, b in enumerate(bufs):
            comm.fill(b, nbytes // 4, xmpi.F32, xmpi.PAT_UNIFORM, i)
        for name, dtype in (("f32", xmpi.F32), ("f16", xmpi.F16), ("f64", xmpi.F64), ("i64", xmpi.I64), ("bf16", xmpi.BF16)):
            cnt = nbytes // xmpi.DTYPE_SIZE[dtype]
            for _ in range(3):
                comm.reduce_local(bufs[0], bufs[1], bufs[2], cnt, dtype, xmpi.SUM)
            comm.prof_reset()
            for _ in range(a.reps):
                comm.reduce_local(bufs[0], bufs[1], bufs[2], cnt, dtype, xmpi.SUM)
            n, ms, by = comm.prof_get(xmpi.PROF_REDUCE2)
            rows.append({"kernel": f"reduce2_{name}_sum", "MiB": mib, "us": 1e3 * ms / n, "GBps": by / (ms * 1e-3) / 1e9,
                         "frac_hbm": by / (ms * 1e-3) / HBM_PEAK})
        comm.prof_reset()
        for _ in range(a.reps):
            comm.copy_local(bufs[0], bufs[1], nbytes)
        n, ms, by = comm.prof_get(xmpi.PROF_COPY)
        rows.append({"kernel": "copy16", "MiB": mib, "us": 1e3 * ms / n, "GBps": by / (ms * 1e-3) / 1e9,
                     "frac_hbm": by / (ms * 1e-3) / HBM_PEAK})
        for nsrc in (4, 8):
            comm.prof_reset()
            for _ in range(a.reps):
                comm.reduce_local_n(bufs[0], bufs[1:1 + nsrc], nbytes // 4, xmpi.F32, xmpi.SUM)
            n, ms, by = comm.prof_get(xmpi.PROF_REDUCEN)
            rows.append({"kernel": f"reduce_n{nsrc}_f32", "MiB": mib, "us": 1e3 * ms / n, "GBps": by / (ms * 1e-3) / 1e9,
                         "frac_hbm": by / (ms * 1e-3) / HBM_PEAK})
        for b in bufs:
            b.free()
    for r in rows:
        print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}))
    comm.finalize()


if __name__ == "__main__":
    main()
