"""World-size 2 / 4 / 8 runs of the control plane on the CPU (plain OS processes, no GPU, no HIP
call): the shared-memory bootstrap that replaces the reference's TCP handshake
(network.go:122-351), the barrier, the pipe counters and the mail-entry states behind
Send/Receive, and host-resident payloads of 1 byte ... 3 rings streaming through the entries' host
lanes (every rank sending to the next and receiving from the one before at once).  Also a stale control block of a crashed job with the same key must not confuse a
new job, and a world-size-2 torch.distributed (gloo) run executes the ring schedule's step tables
over real inter-process messaging."""
import os
import subprocess
import sys
import uuid

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = """
import sys
sys.path.insert(0, %r)
from mpi_amd import xmpi
rc = xmpi.lib().xmpi_ctl_selftest(sys.argv[1].encode(), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
print("rc", rc, xmpi.lib().xmpi_last_error().decode())
sys.exit(0 if rc == 0 else 1)
""" % ROOT


def run_world(size, rounds=50, key=None):
    key = key or f"ctl-{uuid.uuid4().hex[:10]}"
    procs = [subprocess.Popen([sys.executable, "-c", WORKER, key, str(r), str(size), str(rounds)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(size)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    return [p.returncode for p in procs], outs


@pytest.mark.parametrize("size", [1, 2, 4, 8])
def test_control_plane_rounds(size):
    rcs, outs = run_world(size)
    assert rcs == [0] * size, outs


def test_stale_block_is_replaced():
    key = f"stale-{uuid.uuid4().hex[:8]}"
    # a "crashed" job: a block with this key whose creator is gone
    path = f"/dev/shm/xmpi-{os.getuid()}-{key}"
    with open(path, "wb") as f:
        f.write(b"\0" * (1 << 20))
    try:
        rcs, outs = run_world(2, rounds=5, key=key)
        assert rcs == [0, 0], outs
    finally:
        if os.path.exists(path):
            os.unlink(path)


def test_missing_rank_times_out_cleanly():
    """only rank 1 of 2 shows up: it must give up with an error, not hang (30 s join timeout is
    shortened here through the creator never appearing -> XMPI_ERR_TIMEOUT)"""
    key = f"lonely-{uuid.uuid4().hex[:8]}"
    env = dict(os.environ, XMPI_INIT_TIMEOUT_S="4")  # (the bootstrap's own clock, api.cpp: default 60 s)
    p = subprocess.Popen([sys.executable, "-c", WORKER, key, "1", "2", "1"], stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, env=env)
    out, _ = p.communicate(timeout=120)
    assert p.returncode == 1 and "rc -4" in out, out


GLOO_WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from mpi_amd import xmpi
from tests import plan_sim
from oracle import oracle

rank, size = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=size)
count, es = 5000, 8
for algo in (xmpi.ALGO_RING, xmpi.ALGO_RHD, xmpi.ALGO_DIRECT):
    plan = plan_sim.parse_plan(xmpi.plan_text(xmpi.COLL_ALLREDUCE, algo, size, rank, 0, count, es, 2, 512))
    send = oracle.fill(count, oracle.I64, 0, 1000 + rank)
    recv = np.zeros(count, dtype=np.int64)
    bufs = {0: send, 1: recv}
    held = {}
    reqs = []
    for i, s in enumerate(plan.steps):          # in table order: a legal schedule for blocking messaging
        if s.kind == 0:
            t = torch.from_numpy(bufs[s.src_buf][s.src_off // es:(s.src_off + s.nbytes) // es].copy())
            reqs.append(dist.isend(t, s.peer, tag=s.lane))
        elif s.kind in (1, 2, 3):
            t = torch.empty(s.nbytes // es, dtype=torch.int64)
            dist.recv(t, s.peer, tag=s.lane)
            x = t.numpy()
            if s.kind == 1:
                a = bufs[s.src_buf][s.src_off // es:(s.src_off + s.nbytes) // es]
                bufs[s.dst_buf][s.dst_off // es:(s.dst_off + s.nbytes) // es] = a + x
            elif s.kind == 2:
                bufs[s.dst_buf][s.dst_off // es:(s.dst_off + s.nbytes) // es] = x
            else:
                held[i] = x
        elif s.kind in (6, 7):                   # fused ring steps: pop, (reduce,) store?, forward
            t = torch.empty(s.nbytes // es, dtype=torch.int64)
            dist.recv(t, s.peer, tag=s.lane)
            x = t.numpy()
            if s.kind == 6:
                x = bufs[s.src_buf][s.src_off // es:(s.src_off + s.nbytes) // es] + x
            if s.kind == 7 or s.keep:
                bufs[s.dst_buf][s.dst_off // es:(s.dst_off + s.nbytes) // es] = x
            reqs.append(dist.isend(torch.from_numpy(x.copy()), s.peer2, tag=s.lane2))
        elif s.kind == 4:
            acc = None
            for h in s.srcs:
                x = bufs[s.src_buf][s.src_off // es:(s.src_off + s.nbytes) // es] if h < 0 else held[h]
                acc = x.copy() if acc is None else acc + x
            bufs[s.dst_buf][s.dst_off // es:(s.dst_off + s.nbytes) // es] = acc
        else:
            bufs[s.dst_buf][s.dst_off // es:(s.dst_off + s.nbytes) // es] = bufs[s.src_buf][s.src_off // es:(s.src_off + s.nbytes) // es]
    for r in reqs:
        r.wait()
    want = oracle.reduce_ranks([oracle.fill(count, oracle.I64, 0, 1000 + q) for q in range(size)], oracle.I64, oracle.SUM)
    assert np.array_equal(recv, want), f"algo {algo} rank {rank}"
    dist.barrier()

# The zero-copy allreduce's data flow (zcopy.cpp) with gloo standing in for the xGMI loads and stores:
# rank j "reads" chunk j of every rank's send buffer (gather to j), folds in rank order, and "writes"
# the result into chunk j of every rank's receive buffer (broadcast from j).  Floats: bit-exact.
for count, es, code, npdt, tdt in ((4099, 4, oracle.F32, np.float32, torch.float32), (17, 8, oracle.F64, np.float64, torch.float64)):
    send = oracle.fill(count, code, 3, 500 + rank)
    recv = np.zeros(count, dtype=npdt)
    for j in range(size):
        off, cnt = xmpi.zc_chunk(count, es, size, j)
        mine = torch.from_numpy(send[off:off + cnt].copy())
        parts = [torch.empty(cnt, dtype=tdt) for _ in range(size)] if rank == j else None
        dist.gather(mine, parts, dst=j)
        out = torch.empty(cnt, dtype=tdt)
        if rank == j:
            out = torch.from_numpy(oracle.reduce_ranks([p.numpy() for p in parts], code, oracle.SUM).copy())
        dist.broadcast(out, src=j)
        recv[off:off + cnt] = out.numpy()
    want = oracle.reduce_ranks([oracle.fill(count, code, 3, 500 + q) for q in range(size)], code, oracle.SUM)
    assert recv.tobytes() == want.tobytes(), f"zero-copy flow, rank {rank}"
dist.destroy_process_group()
print("ok")
""" % ROOT


@pytest.mark.parametrize("size", [2])
def test_step_tables_over_gloo(size):
    """torch.distributed, backend gloo, world_size 2 on the CPU: each process runs ITS rank's step table
    (the one libxmpi's executor runs) with gloo send/recv as the pipes, and checks the oracle."""
    port = 20000 + (os.getpid() % 10000)  # (below the ephemeral range: no outgoing connection of another test sits there)
    procs = []
    for r in range(size):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(size), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", GLOO_WORKER], stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True, env=env))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    _ = np
