"""Where does the time of a small allreduce go?  R rank threads on one GPU; prints per-call wall time
(python side), engine time and final-sync time as seen by rank 0."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpi_amd import xmpi

R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sys.setswitchinterval(1e-4)
key = f"lat{os.getpid()}"
out = {}

def body(r):
    c = xmpi.Comm(r, R, 0, key)
    n = 256
    a, b = c.alloc(4 * n), c.alloc(4 * n)
    c.fill(a, n, xmpi.F32, 0, r)
    for algo in (xmpi.ALGO_DIRECT, xmpi.ALGO_RING):
        for _ in range(20):
            c.allreduce(a, b, n, xmpi.F32, xmpi.SUM, algo)
        c.barrier()
        t0 = time.perf_counter()
        runs, syncs = [], []
        for _ in range(200):
            c.allreduce(a, b, n, xmpi.F32, xmpi.SUM, algo)
            runs.append(c.get_param("last_run_us")); syncs.append(c.get_param("last_sync_us"))
        dt = (time.perf_counter() - t0) / 200
        if r == 0:
            out[algo] = (dt * 1e6, sum(runs) / len(runs), sum(syncs) / len(syncs))
        c.barrier()
    c.finalize()

ts = [threading.Thread(target=body, args=(r,)) for r in range(R)]
[t.start() for t in ts]; [t.join() for t in ts]
for algo, (wall, run, sync) in out.items():
    print(f"R={R} algo={algo}: python wall {wall:.0f} us/call, engine {run:.0f} us, of which final sync {sync:.0f} us")
