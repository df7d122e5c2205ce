"""Executes the step programs of the stepped kernels (mpi_amd/csrc/sched.hip: ring allreduce, recursive halving +
doubling -- for any number of ranks --, ring allgather, binary-tree broadcast and reduce) on the CPU: every rank's program comes from the function the kernel
itself calls (`xmpi_sched_dump` -> sched_steps.h `sched_step`), every worker (a block of the kernel) of every rank is a
little sequential machine -- wait for the flag, move its tiles element by element, raise the flags -- and a seeded
random scheduler interleaves all of them at the granularity of ONE element moved, with ranks and workers running at
any relative speed (a rank may be a whole phase ahead of another).

Every schedule comes in a PULL form (a step loads the peer's memory and stores locally) and a PUSH form (a step loads local
memory only and stores into the peer's receive buffer or the landing block the peer lends): the model checks which memory a
step touches, and that no address leaves the block it names.

What the real thing does with flag words in uncached HBM the model does with a dictionary, and remote loads and stores
touch the peer's memory as it is at that moment, so a schedule that lets somebody read a region before it is complete, or
overwrite one that somebody still has to read, produces a wrong result under some interleaving -- which is what the
tests look for (tests/test_sched_sim.py), for N = 2 ... 9, in place and out of place, several channels and workers.
The tile -> worker rule (tile t of the buffer belongs to worker t % W on every rank) is restated here from
`range_apply`; everything else is the library's own output."""
from __future__ import annotations

import random
import re

import numpy as np

from mpi_amd import xmpi

HEAD = re.compile(r"(\d+) wait=(-?\d+):(\d+) sig=(-?\d+),(-?\d+):(\d+) nmv=(\d)")
MOVE = re.compile(r"ns=(\d) D=(\S+) D2=(\S+) A=(\S+) B=(\S+) C=(\S+) lo=(\d+) hi=(\d+)")


def parse_ref(s):
    if s == "-":
        return None
    m = re.fullmatch(r"(\d+)\.([srl])([+-]\d+)", s)
    return int(m.group(1)), m.group(2), int(m.group(3))


def program(sched, size, rank, root, pieces, count, es, nchan, ch, push=False, inplace=False):
    steps = []
    for line in xmpi.sched_text(sched, size, rank, root, pieces, count, es, nchan, ch, push=push, inplace=inplace).strip().split("\n"):
        parts = [p.strip() for p in line.split("|")]
        m = HEAD.fullmatch(parts[0])
        assert m, line
        g, wr, wv, s0, s1, sv, nmv = m.groups()
        moves = []
        for p in parts[1:]:
            mm = MOVE.fullmatch(p)
            assert mm, line
            ns, d, d2, a, b, c, lo, hi = mm.groups()
            moves.append(dict(ns=int(ns), D=parse_ref(d), D2=parse_ref(d2), A=parse_ref(a), B=parse_ref(b), C=parse_ref(c), lo=int(lo), hi=int(hi)))
        assert len(moves) == int(nmv), line
        steps.append(dict(g=int(g), wait=(int(wr), int(wv)), sig=[int(s0), int(s1)], sig_val=int(sv), moves=moves))
    return steps


class Violation(AssertionError):
    pass


def run(sched, size, count, es=4, nchan=1, gx=2, tile=16, root=0, pieces=1, inplace=False, seed=0, bias=None, push=False, symbolic=False):
    """Simulate one collective.  `tile`: bytes per tile (the kernel's is 16 KiB; small here so that small buffers spread
    over several workers).  bias: a rank that gets scheduled 20x as often as the others (it runs ahead).  push: the push form
    (a step reads local memory only and stores into the peer's receive buffer or landing block).  symbolic: the elements are
    terms -- ("x", rank, index) combined as ("op", a, b) -- instead of numbers: the result spells out in which order and
    association a schedule combined its operands (floating-point addition is not associative: two forms of a schedule give the
    same BITS only if they build the same term).  Returns the ranks' receive buffers (lists of element values)."""
    rng = random.Random(seed)
    W = nchan * gx
    nelem_send = count
    nelem_recv = count * size if sched == xmpi.SCHED_RING_ALLGATHER else count
    # element values: distinct powers so that a sum identifies exactly which contributions it holds
    send = [np.array([(1 << (4 * r)) * (1 + (i % 7)) for i in range(nelem_send)], dtype=object) for r in range(size)]
    combine = (lambda x, y: x + y)
    if symbolic:
        send = [np.empty(nelem_send, dtype=object) for r in range(size)]
        for r in range(size):
            for i in range(nelem_send):
                send[r][i] = ("x", r, i)
        combine = (lambda x, y: ("op", x, y))
    if sched == xmpi.SCHED_TREE_BCAST:
        recv = [np.array([(i * 31 + 7) if r == root else -1 for i in range(count)], dtype=object) for r in range(size)]
        send = recv  # one buffer
    elif sched == xmpi.SCHED_TREE_REDUCE:
        # only the root's receive buffer is the caller's (in place: its send buffer); pull form: an inner node's is the accumulator
        # the library lends it (dsync.cpp), a leaf's is never touched; push form: nobody's but the root's is touched
        recv = [send[r] if (inplace and r == root) else np.array([-1] * nelem_recv, dtype=object) for r in range(size)]
    elif sched == xmpi.SCHED_RING_ALLGATHER and inplace:
        # the rank's block of its receive buffer is its send buffer (the program names it r.r+offset)
        recv = [np.array([send[r][i - r * count] if r * count <= i < (r + 1) * count else -1 for i in range(nelem_recv)], dtype=object)
                for r in range(size)]
    elif inplace:
        recv = send
    else:
        recv = [np.array([-1] * nelem_recv, dtype=object) for r in range(size)]
    orig = [s.copy() for s in send]
    # landing blocks (push forms): as many bytes as the library would lend, poisoned
    land = [np.array([-7] * (xmpi.sched_land_bytes(sched, size, r, root, count, es, inplace) // es + 1 if push else 0), dtype=object)
            for r in range(size)]
    mem = {"s": send, "r": recv, "l": land}
    flags = {}  # (owner page, sender, worker) -> value

    def cell(ref, x):
        r, k, o = ref
        i, rem = divmod(x + o, es)
        if rem or i < 0 or i >= len(mem[k][r]):
            raise Violation(f"address {ref}+{x} outside rank {r}'s {k} block of {len(mem[k][r])} elements")
        return mem[k][r], i

    # worker machines
    class Worker:
        def __init__(self, rank, w):
            self.rank, self.w = rank, w
            self.prog = program(sched, size, rank, root, pieces, count, es, nchan, w % nchan, push, inplace)  # (sched.hip: w = x * channels + channel)
            self.pc = 0
            self.phase = "wait"
            self.todo = None

        def elements(self, st):
            out = []
            for k, m in enumerate(st["moves"]):
                for x in range(m["lo"], m["hi"], es):
                    if (x // tile) % W == self.w:
                        out.append((k, x))
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
                self.todo = self.elements(st)
                rng.shuffle(self.todo)  # lanes of a block run in any order
                self.phase = "move"
                return True
            if self.phase == "move":
                if self.todo:
                    k, x = self.todo.pop()
                    m = st["moves"][k]
                    if push:
                        for name in ("A", "B", "C"):
                            assert m[name] is None or m[name][0] == self.rank, "a push step reads local memory only"
                    else:
                        assert m["D"][0] == self.rank and m["D2"] is None, "a pull step writes local memory only"
                    arr, i = cell(m["A"], x)
                    v = arr[i]
                    if m["ns"] >= 2:
                        arr, i = cell(m["B"], x)
                        v = combine(v, arr[i])
                    if m["ns"] == 3:
                        arr, i = cell(m["C"], x)
                        v = combine(arr[i], v)
                    for name in ("D", "D2"):
                        if m[name] is not None:
                            arr, i = cell(m[name], x)
                            arr[i] = v
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
