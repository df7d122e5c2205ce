"""A CPU model of the LL small-collective protocol (mpi_amd/csrc/ll.hip, layout in kernels.h): N ranks, each running the
same sequence of collectives -- LL allreduce / allgather / broadcast / reduce and the announce-and-done kind of the
zero-copy forms ("fold") -- every lane of every kernel a little state machine whose 8-byte stores and loads are the
atomic steps, all interleaved by a seeded random scheduler (ranks run at any relative speed; the kernels of one rank run
one at a time, in order).  What it checks is what the header claims:

  I1  a half-line is never overwritten while the kernel it was written for may still read it (slots are double-buffered
      by epoch parity; the argument needs every collective to complete on a rank only after every peer has STARTED it --
      the `here` words give the broadcast and the reduce that property);
  I2  what a lane accepts as rank p's contribution to epoch e is what p pushed for epoch e, both halves (a 16-byte load may
      see the two 8-byte halves at different times: each half carries its own flag);
  I3  every kernel ends, whatever the interleaving;
  I4  pages are pooled and never cleared: a later communicator (epochs above everything its pages have seen) is not
      confused by stale lines.

`bugs` switches known-bad variants on so the tests can show the checker notices them: "no_here" (broadcast / reduce
without the `here` words), "one_slot" (no parity), "flag_on_first_half_only", "no_epoch_base".
"""
from __future__ import annotations

import random


class Violation(AssertionError):
    pass


class Page:
    def __init__(self, n, lines):
        # slot[src][parity][line][half] = (flag epoch, payload tag)
        self.slot = [[[[(0, None), (0, None)] for _ in range(lines)] for _ in range(2)] for _ in range(n)]
        self.here = [0] * n
        self.ready = [0] * n
        self.done = [0] * n


class Lane:
    """one lane of an LL kernel: push my line to the targets (two stores each), then poll the sources (two loads each)"""

    def __init__(self, targets, sources):
        self.todo = [(p, h) for p in targets for h in (0, 1)]
        self.need = {(p, h) for p in sources for h in (0, 1)}
        self.got = {}


class Kernel:
    def __init__(self, rank, epoch, kind, root, n, lines):
        self.rank, self.epoch, self.kind, self.root = rank, epoch, kind, root
        peers = [p for p in range(n) if p != rank]
        self.started = self.ended = False
        self.said_here = self.announced = self.said_done = False
        self.lanes = []
        if kind in ("ar", "ag"):
            self.lanes = [Lane(peers, peers) for _ in range(lines)]
        elif kind == "bc":
            self.lanes = [Lane(peers if rank == root else [], [] if rank == root else [root]) for _ in range(lines)]
        elif kind == "rd":
            self.lanes = [Lane([] if rank == root else [root], peers if rank == root else []) for _ in range(lines)]
        self.uses_here = kind in ("bc", "rd")


def run(n, lines, program, seed, comms=1, bugs=()):
    """program: list of (kind, root) every rank runs per communicator.  Raises Violation; returns the scheduler steps."""
    rng = random.Random(seed)
    pages = [Page(n, lines) for _ in range(n)]
    last_epoch = [0] * n
    steps = 0
    for comm in range(comms):
        base = max(last_epoch) if "no_epoch_base" not in bugs else 0
        queue = [[Kernel(r, base + 1 + k, kind, root, n, lines) for k, (kind, root) in enumerate(program)] for r in range(n)]
        cur = [0] * n
        kernels = {(r, k.epoch): k for r in range(n) for k in queue[r]}
        while any(cur[r] < len(program) for r in range(n)):
            steps += 1
            if steps > 3_000_000:
                raise Violation("I3: no termination")
            r = rng.randrange(n)
            if cur[r] >= len(program):
                continue
            k = queue[r][cur[r]]
            k.started = True
            e = k.epoch
            par = 0 if "one_slot" in bugs else e & 1
            mine = pages[r]
            peers = [p for p in range(n) if p != r]
            if k.kind == "fold":  # announce, wait for everybody's, (move), done exchange: touches no LL slot
                if not k.announced:
                    for p in peers:
                        pages[p].ready[r] = e
                    k.announced = True
                elif not k.said_done:
                    if all(mine.ready[p] >= e for p in peers):
                        for p in peers:
                            pages[p].done[r] = e
                        k.said_done = True
                elif all(mine.done[p] >= e for p in peers):
                    k.ended = True
            else:
                if k.uses_here and not k.said_here and "no_here" not in bugs:
                    for p in peers:
                        pages[p].here[r] = e
                    k.said_here = True
                    continue
                open_lanes = [i for i, ln in enumerate(k.lanes) if ln.todo or ln.need]
                if open_lanes:
                    i = rng.choice(open_lanes)
                    ln = k.lanes[i]
                    if ln.todo:  # one 8-byte store into a peer's page
                        p, h = ln.todo.pop(rng.randrange(len(ln.todo)) if rng.random() < 0.3 else 0)
                        old_e, _ = pages[p].slot[r][par][i][h]
                        victim = kernels.get((p, old_e))  # the kernel of rank p this half was written for
                        if victim is not None and old_e != e and not victim.ended and victim.lanes and (r, h) in victim.lanes[i].need:
                            raise Violation(f"I1: rank {r} (epoch {e}) overwrites line {i}.{h} that rank {p}'s epoch {old_e} has not read")
                        pages[p].slot[r][par][i][h] = (e, (r, e, i, h))
                    else:  # one 8-byte load from my own page
                        p, h = rng.choice(sorted(ln.need))
                        f, tag = mine.slot[p][par][i][h]
                        trust = f == e
                        if "flag_on_first_half_only" in bugs and h == 1:  # the reader trusts the first half's flag for both
                            trust = mine.slot[p][par][i][0][0] == e
                        if trust:
                            if tag != (p, e, i, h):
                                raise Violation(f"I2: rank {r} epoch {e} accepted {tag} as rank {p}'s line {i}.{h}")
                            ln.need.discard((p, h))
                elif k.uses_here and "no_here" not in bugs and not all(mine.here[p] >= e for p in peers):
                    pass  # block 0 still waits for somebody's `here`
                else:
                    k.ended = True
            if k.ended:
                cur[r] += 1
                last_epoch[r] = e
    return steps
