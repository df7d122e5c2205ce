"""CPU simulator of the per-rank step tables libxmpi's executor runs (csrc/plan.cpp).

Test infrastructure: it executes every rank's table over in-memory FIFOs with a bounded depth
and a randomised schedule, so the N = 2..8 data flow, the FIFO ordering on every pipe, the
hazard dependencies and deadlock freedom are checked without a GPU.  It is NOT a fallback of
the product: nothing under mpi_amd/ imports it.
"""
from __future__ import annotations

import random
from collections import deque
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np

from mpi_amd import xmpi

KINDS = {"SEND": 0, "RECV_REDUCE": 1, "RECV_COPY": 2, "RECV_HOLD": 3, "REDUCE_N": 4, "LOCAL_COPY": 5,
         "RECV_REDUCE_SEND": 6, "RECV_COPY_SEND": 7}


@dataclass
class Step:
    kind: int
    peer: int
    lane: int
    src_buf: int
    src_off: int
    dst_buf: int
    dst_off: int
    nbytes: int
    deps: List[int]
    srcs: List[int]
    peer2: int = -1
    lane2: int = 0
    keep: int = 0


@dataclass
class Plan:
    algo: int
    channels: int
    temp_bytes: int
    steps: List[Step] = field(default_factory=list)


def parse_plan(text: str) -> Plan:
    lines = text.strip().split("\n")
    hdr = dict(kv.split("=") for kv in lines[0].split()[1:])
    plan = Plan(int(hdr["algo"]), int(hdr["channels"]), int(hdr["temp_bytes"]))
    for ln in lines[1:]:
        tok = ln.split()
        kv = dict(t.split("=", 1) for t in tok[2:])
        sb, so = kv["src"].split(":")
        db, do = kv["dst"].split(":")
        deps = [int(x) for x in kv["deps"].split(",") if x]
        srcs = [int(x) for x in kv["srcs"].split(",") if x]
        plan.steps.append(Step(KINDS[tok[1]], int(kv["peer"]), int(kv["lane"]), int(sb), int(so), int(db), int(do),
                               int(kv["bytes"]), deps, srcs, int(kv.get("peer2", -1)), int(kv.get("lane2", 0)),
                               int(kv.get("keep", 0))))
    assert len(plan.steps) == int(hdr["steps"])
    return plan


def get_plans(coll, algo, size, root, count, elem_size, channels, piece_elems, fifo_depth: int = 2,
              oneshot_bytes: int = 0) -> List[Plan]:
    """Step tables of every rank, built for pipes of `fifo_depth` slots (valid for any deeper FIFO).
    oneshot_bytes = 0 keeps the direct allreduce in its two-phase form whatever the message size."""
    return [parse_plan(xmpi.plan_text(coll, algo, size, r, root, count, elem_size, channels, piece_elems, fifo_depth, oneshot_bytes))
            for r in range(size)]


def np_combine(a: np.ndarray, b: np.ndarray, op: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        if op == xmpi.SUM:
            return a + b
        if op == xmpi.PROD:
            return a * b
        if op == xmpi.MIN:
            return np.where(b < a, b, a)
        return np.where(a < b, b, a)


def simulate(plans: List[Plan], sendbufs: List[np.ndarray], recv_elems: int, dtype, op: int, fifo_depth: int,
             seed: int = 0, inplace: bool = False, send_shift_bytes: int = 0) -> List[np.ndarray]:
    """Run all ranks' tables; returns each rank's recvbuf.  Raises on deadlock / protocol error."""
    n = len(plans)
    rng = random.Random(seed)
    dt = np.dtype(dtype)
    es = dt.itemsize
    recv = [np.zeros(recv_elems, dtype=dt) for _ in range(n)]
    if inplace:  # sendbuf aliases recvbuf at byte offset r * send_shift_bytes (allgather convention)
        for r in range(n):
            o = r * send_shift_bytes // es
            recv[r][o: o + sendbufs[r].size] = sendbufs[r]
    temp = [np.zeros(max(1, p.temp_bytes // es), dtype=dt) for p in plans]

    def view(r: int, buf: int, off: int, nbytes: int) -> np.ndarray:
        assert off % es == 0 and nbytes % es == 0
        if inplace and buf == 0:
            off += r * send_shift_bytes
        base = (recv[r] if inplace else sendbufs[r]) if buf == 0 else (recv[r] if buf == 1 else temp[r])
        v = base[off // es: (off + nbytes) // es]
        assert v.size == nbytes // es, "step runs off the end of its buffer"
        return v

    pipes: Dict[Tuple[int, int, int], deque] = {}
    occupied: Dict[Tuple[int, int, int], int] = {}
    held: List[Dict[int, Tuple[Tuple[int, int, int], np.ndarray]]] = [dict() for _ in range(n)]
    state = [[0] * len(p.steps) for p in plans]
    # per rank, per pipe direction: FIFO order of the steps
    sendq: List[Dict[Tuple[int, int], deque]] = [dict() for _ in range(n)]
    recvq: List[Dict[Tuple[int, int], deque]] = [dict() for _ in range(n)]
    localq: List[List[int]] = [[] for _ in range(n)]
    for r, p in enumerate(plans):
        for i, s in enumerate(p.steps):
            if s.kind == 0:
                sendq[r].setdefault((s.peer, s.lane), deque()).append(i)
            elif s.kind in (1, 2, 3):
                recvq[r].setdefault((s.peer, s.lane), deque()).append(i)
            elif s.kind in (6, 7):  # fused: pops one pipe and pushes another, in FIFO order on both
                recvq[r].setdefault((s.peer, s.lane), deque()).append(i)
                sendq[r].setdefault((s.peer2, s.lane2), deque()).append(i)
            else:
                localq[r].append(i)
    remaining = sum(len(p.steps) for p in plans)

    def deps_ok(r: int, s: Step) -> bool:
        return all(state[r][d] == 2 for d in s.deps)

    while remaining:
        ready = []
        for r, p in enumerate(plans):
            for (peer, lane), q in sendq[r].items():
                if q:
                    s = p.steps[q[0]]
                    key = (r, peer, lane)
                    if s.kind == 0 and deps_ok(r, s) and occupied.get(key, 0) < fifo_depth:
                        ready.append((r, q[0]))
            for (peer, lane), q in recvq[r].items():
                if q:
                    s = p.steps[q[0]]
                    if not (deps_ok(r, s) and pipes.get((peer, r, lane))):
                        continue
                    if s.kind in (6, 7):  # must also be next in line on its outgoing pipe, with room there
                        oq = sendq[r][(s.peer2, s.lane2)]
                        if oq[0] != q[0] or occupied.get((r, s.peer2, s.lane2), 0) >= fifo_depth:
                            continue
                    ready.append((r, q[0]))
            for i in localq[r]:
                s = p.steps[i]
                if state[r][i] == 0 and deps_ok(r, s) and all(h < 0 or state[r][h] == 2 for h in s.srcs):
                    ready.append((r, i))
        if not ready:
            raise RuntimeError(f"deadlock: {remaining} steps left, no step can run")
        r, i = rng.choice(ready)
        s = plans[r].steps[i]
        if s.kind == 0:
            key = (r, s.peer, s.lane)
            pipes.setdefault(key, deque()).append(view(r, s.src_buf, s.src_off, s.nbytes).copy())
            occupied[key] = occupied.get(key, 0) + 1
            sendq[r][(s.peer, s.lane)].popleft()
        elif s.kind in (1, 2, 3):
            key = (s.peer, r, s.lane)
            data = pipes[key].popleft()
            assert data.size * es == s.nbytes, f"rank {r} step {i}: slot holds {data.size * es} B, step expects {s.nbytes}"
            recvq[r][(s.peer, s.lane)].popleft()
            if s.kind == 1:
                a = view(r, s.src_buf, s.src_off, s.nbytes)
                view(r, s.dst_buf, s.dst_off, s.nbytes)[:] = np_combine(a, data, op)
                occupied[key] -= 1
            elif s.kind == 2:
                view(r, s.dst_buf, s.dst_off, s.nbytes)[:] = data
                occupied[key] -= 1
            else:
                held[r][i] = (key, data)
        elif s.kind in (6, 7):
            key = (s.peer, r, s.lane)
            data = pipes[key].popleft()
            assert data.size * es == s.nbytes
            recvq[r][(s.peer, s.lane)].popleft()
            sendq[r][(s.peer2, s.lane2)].popleft()
            if s.kind == 6:
                val = np_combine(view(r, s.src_buf, s.src_off, s.nbytes), data, op)
                if s.keep:
                    view(r, s.dst_buf, s.dst_off, s.nbytes)[:] = val
            else:
                val = data
                view(r, s.dst_buf, s.dst_off, s.nbytes)[:] = val
            occupied[key] -= 1
            okey = (r, s.peer2, s.lane2)
            pipes.setdefault(okey, deque()).append(np.array(val, copy=True))
            occupied[okey] = occupied.get(okey, 0) + 1
        elif s.kind == 4:
            acc = None
            for h in s.srcs:
                x = view(r, s.src_buf, s.src_off, s.nbytes).copy() if h < 0 else held[r][h][1]
                acc = x.copy() if acc is None else np_combine(acc, x, op)
            view(r, s.dst_buf, s.dst_off, s.nbytes)[:] = acc
            for h in s.srcs:
                if h >= 0:
                    occupied[held[r][h][0]] -= 1
                    del held[r][h]
        else:
            src = view(r, s.src_buf, s.src_off, s.nbytes).copy()
            view(r, s.dst_buf, s.dst_off, s.nbytes)[:] = src
        state[r][i] = 2
        remaining -= 1
    for key, q in pipes.items():
        assert not q, f"pipe {key} still holds {len(q)} slots at the end"
    return recv
