"""Executes the step programs of the stepped kernels (mpi_amd/csrc/sched.hip: ring allreduce, recursive halving +
doubling -- for any number of ranks --, ring allgather, binary-tree broadcast and reduce) on the CPU: every rank's program comes from the function the kernel
itself calls (`xmpi_sched_dump` -> sched_steps.h `sched_step`), every worker (a block of the kernel) of every rank is a
little sequential machine -- wait for the flag, move its tiles element by element, raise the flags -- and a seeded
random scheduler interleaves all of them at the granularity of ONE element moved, with ranks and workers running at
any relative speed (a rank may be a whole phase ahead of another).

What the real thing does with flag words in uncached HBM the model does with a dictionary, and remote loads read the
peer's memory as it is at that moment, so a schedule that lets somebody read a region before it is complete, or
overwrite one that somebody still has to read, produces a wrong result under some interleaving -- which is what the
tests look for (tests/test_sched_sim.py), for N = 2 ... 9, in place and out of place, several channels and workers.
The tile -> worker rule (tile t of the buffer belongs to worker t % W on every rank) is restated here from
`range_apply`; everything else is the library's own output."""
from __future__ import annotations

import random
import re

import numpy as np

from mpi_amd import xmpi

LINE = re.compile(r"(\d+) wait=(-?\d+):(\d+) sig=(-?\d+),(-?\d+):(\d+) ns=(\d) D=(\S+) A=(\S+) B=(\S+) lo=(\d+) hi=(\d+)")


def parse_ref(s):
    if s == "-":
        return None
    m = re.fullmatch(r"(\d+)\.([sr])([+-]\d+)", s)
    return int(m.group(1)), m.group(2), int(m.group(3))


def program(sched, size, rank, root, pieces, count, es, nchan, ch):
    steps = []
    for line in xmpi.sched_text(sched, size, rank, root, pieces, count, es, nchan, ch).strip().split("\n"):
        m = LINE.fullmatch(line.strip())
        assert m, line
        g, wr, wv, s0, s1, sv, ns, d, a, b, lo, hi = m.groups()
        steps.append(dict(g=int(g), wait=(int(wr), int(wv)), sig=[int(s0), int(s1)], sig_val=int(sv), ns=int(ns),
                          D=parse_ref(d), A=parse_ref(a), B=parse_ref(b), lo=int(lo), hi=int(hi)))
    return steps


class Violation(AssertionError):
    pass


def run(sched, size, count, es=4, nchan=1, gx=2, tile=16, root=0, pieces=1, inplace=False, seed=0, bias=None):
    """Simulate one collective.  `tile`: bytes per tile (the kernel's is 16 KiB; small here so that small buffers spread
    over several workers).  bias: a rank that gets scheduled 20x as often as the others (it runs ahead).  Returns the
    ranks' receive buffers (lists of element values)."""
    rng = random.Random(seed)
    W = nchan * gx
    nelem_send = count
    nelem_recv = count * size if sched == xmpi.SCHED_RING_ALLGATHER else count
    # element values: distinct powers so that a sum identifies exactly which contributions it holds
    send = [np.array([(1 << (4 * r)) * (1 + (i % 7)) for i in range(nelem_send)], dtype=object) for r in range(size)]
    if sched == xmpi.SCHED_TREE_BCAST:
        recv = [np.array([(i * 31 + 7) if r == root else -1 for i in range(count)], dtype=object) for r in range(size)]
        send = recv  # one buffer
    elif sched == xmpi.SCHED_TREE_REDUCE:
        # only the root's receive buffer is the caller's (in place: its send buffer); an inner node's is the accumulator the
        # library lends it (dsync.cpp), a leaf's is never touched
        recv = [send[r] if (inplace and r == root) else np.array([-1] * nelem_recv, dtype=object) for r in range(size)]
    elif inplace:
        recv = send
    else:
        recv = [np.array([-1] * nelem_recv, dtype=object) for r in range(size)]
    orig = [s.copy() for s in send]
    mem = {"s": send, "r": recv}
    flags = {}  # (owner page, sender, worker) -> value

    # worker machines
    class Worker:
        def __init__(self, rank, w):
            self.rank, self.w = rank, w
            self.prog = program(sched, size, rank, root, pieces, count, es, nchan, w % nchan)  # (sched.hip: w = x * channels + channel)
            self.pc = 0
            self.phase = "wait"
            self.todo = None

        def elements(self, st):
            out = []
            for x in range(st["lo"], st["hi"], es):
                if (x // tile) % W == self.w:
                    out.append(x)
            return out

        def done(self):
            return self.pc >= len(self.prog)

        def step(self):
            """one atomic action; returns False if blocked"""
            st = self.prog[self.pc]
            if self.phase == "wait":
                wr, wv = st["wait"]
                if wr >= 0 and flags.get((self.rank, wr, self.w), 0) < wv:
                    return False
                self.todo = self.elements(st) if st["ns"] else []
                rng.shuffle(self.todo)  # lanes of a block run in any order
                self.phase = "move"
                return True
            if self.phase == "move":
                if self.todo:
                    x = self.todo.pop()
                    rd, kd, od = st["D"]
                    ra, ka, oa = st["A"]
                    v = mem[ka][ra][(x + oa) // es]
                    if st["ns"] == 2:
                        rb, kb, ob = st["B"]
                        v = v + mem[kb][rb][(x + ob) // es]
                    assert rd == self.rank, "a step writes local memory only"
                    mem[kd][rd][(x + od) // es] = v
                    return True
                self.phase = "signal"
                return True
            for target in st["sig"]:
                if target >= 0:
                    key = (target, self.rank, self.w)
                    if flags.get(key, 0) >= st["sig_val"]:
                        raise Violation(f"flag {key} not monotonic")
                    flags[key] = st["sig_val"]
            self.pc += 1
            self.phase = "wait"
            return True

    workers = [Worker(r, w) for r in range(size) for w in range(W)]
    idle = 0
    while any(not k.done() for k in workers):
        live = [k for k in workers if not k.done()]
        if bias is not None and rng.random() < 0.95:
            cand = [k for k in live if k.rank == bias]
            k = rng.choice(cand) if cand else rng.choice(live)
        else:
            k = rng.choice(live)
        if k.step():
            idle = 0
        else:
            idle += 1
            if idle > 20000:
                # is really nobody able to move?
                if not any(x.step() for x in live):
                    raise Violation("deadlock: no worker can make progress")
                idle = 0
    return recv, orig


def expected(sched, size, count, orig, root=0):
    if sched in (xmpi.SCHED_RING_ALLREDUCE, xmpi.SCHED_RHD_ALLREDUCE):
        tot = [sum(orig[r][i] for r in range(size)) for i in range(count)]
        return [tot] * size
    if sched == xmpi.SCHED_RING_ALLGATHER:
        cat = [orig[r][i] for r in range(size) for i in range(count)]
        return [cat] * size
    if sched == xmpi.SCHED_TREE_REDUCE:
        return [[sum(orig[r][i] for r in range(size)) for i in range(count)] if q == root else None for q in range(size)]
    return [list(orig[root])] * size
