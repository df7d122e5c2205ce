"""A model of the device-synchronised collective protocol (mpi_amd/csrc/kernels.hip `dsync_begin` / `dsync_end`,
mpi_amd/csrc/dsync.cpp) that runs on the CPU: N ranks, each launching a sequence of kernels of B blocks, every block an
independent little state machine, all of them interleaved by a seeded random scheduler (blocks of one kernel become
resident in any order, ranks run at any relative speed).  What the real thing does with flag words in uncached HBM the
model does with dictionaries; what it checks is what the protocol promises:

  I1  a block touches a peer's buffers for epoch e only while that peer's kernel e is running (it has announced them and
      has not ended) -- nobody reads an input that is not ready or writes an output that has been handed back;
  I2  the buffer reference a block reads from a peer's slot is the one that peer published for THIS epoch (a slot is
      never overwritten while somebody may still read it);
  I3  every kernel of every rank ends (no deadlock), whatever the interleaving;
  I4  pages are pooled and never cleared: a new communicator whose epochs start above the highest epoch any of its ranks'
      pages has seen (dsync_connect) is not confused by what an earlier communicator left behind.

`bugs` switches known-bad variants on, so the tests can show that the checker notices them.
"""
from __future__ import annotations

import random


class Violation(AssertionError):
    pass


class Page:
    """one rank's flag page: slot p is written by rank p only"""

    def __init__(self, n):
        self.ready = [(0, None)] * n  # (epoch, buffer reference)
        self.done = [0] * n
        self.ticket = 0


class Kernel:
    def __init__(self, rank, epoch, blocks, bufref):
        self.rank, self.epoch, self.bufref = rank, epoch, bufref
        self.phase = ["new"] * blocks  # new -> waiting -> moving -> finished ; the last finisher: -> closing -> ended
        self.started = False
        self.ended = False
        self.waited = [set() for _ in range(blocks)]  # peers whose slot this block has already seen
        self.closing_block = None


def run(n, blocks, epochs_per_comm, seed, comms=1, bugs=()):
    """Simulate `comms` communicators one after the other on the same (pooled, uncleared) pages; each runs
    `epochs_per_comm` collectives.  Raises Violation.  Returns the number of scheduler steps."""
    rng = random.Random(seed)
    pages = [Page(n) for _ in range(n)]
    last_epoch = [0] * n  # what pool_release remembered per page
    steps = 0
    for comm in range(comms):
        base = max(last_epoch) if "no_epoch_base" not in bugs else 0
        # every rank's kernels of this communicator, in stream order
        queue = [[Kernel(r, base + 1 + k, blocks, ("buf", comm, r, k)) for k in range(epochs_per_comm)] for r in range(n)]
        current = [0] * n  # index of the kernel a rank is running (stream order: the next starts when this one ended)
        running = lambda r: queue[r][current[r]] if current[r] < epochs_per_comm else None  # noqa: E731
        in_flight = {}  # (rank, epoch) -> Kernel, for I1

        def owner_running(p, e):
            k = in_flight.get((p, e))
            return k is not None and k.started and not k.ended

        idle_rounds = 0
        while any(current[r] < epochs_per_comm for r in range(n)):
            steps += 1
            if steps > 2_000_000:
                raise Violation("I3: no termination (livelock?)")
            r = rng.randrange(n)
            k = running(r)
            if k is None:
                continue
            b = rng.randrange(blocks)
            ph = k.phase[b]
            progressed = True
            me_page = pages[r]
            if ph == "new":
                # a block becomes resident: the kernel has started (its buffers now belong to the collective)
                if not k.started:
                    k.started = True
                    in_flight[(r, k.epoch)] = k
                if b == 0:  # one block announces this rank to everybody
                    for p in range(n):
                        if p != r:
                            pages[p].ready[r] = (k.epoch, k.bufref)
                k.phase[b] = "waiting"
            elif ph == "waiting":
                # poll one peer's slot in the own page
                todo = [p for p in range(n) if p != r and p not in k.waited[b]]
                if todo:
                    p = rng.choice(todo)
                    ep, ref = me_page.ready[p]
                    ok = ep >= k.epoch
                    if ok:
                        want = in_flight.get((p, k.epoch))
                        if want is None or ref != want.bufref:
                            raise Violation(f"I2: rank {r} block {b} epoch {k.epoch} read slot of rank {p}: {ref!r}")
                        k.waited[b].add(p)
                    else:
                        progressed = False
                if len(k.waited[b]) == n - 1:
                    k.phase[b] = "moving"
            elif ph == "moving":
                for p in range(n):  # touches every peer's buffers
                    if p != r and not owner_running(p, k.epoch):
                        raise Violation(f"I1: rank {r} block {b} touches rank {p}'s buffers of epoch {k.epoch} outside its kernel")
                me_page.ticket += 1
                k.phase[b] = "finished"
                last = me_page.ticket == blocks
                if "early_done" in bugs and b == 0:
                    last = True  # a block says "done" for the whole kernel without counting tickets
                if last and k.closing_block is None:
                    k.closing_block = b
                    if me_page.ticket == blocks:
                        me_page.ticket = 0
                    for p in range(n):
                        if p != r:
                            pages[p].done[r] = k.epoch
                    k.phase[b] = "closing"
            elif ph == "closing":
                if "no_done_wait" in bugs or all(me_page.done[p] >= k.epoch for p in range(n) if p != r):
                    if "early_done" in bugs:
                        me_page.ticket = 0
                    k.phase[b] = "ended"
                else:
                    progressed = False
            elif ph in ("finished", "ended"):
                progressed = False
            # the kernel ends when every block has finished and the closing block has seen every peer's "done"
            if not k.ended and all(x in ("finished", "ended") for x in k.phase) and "ended" in k.phase:
                k.ended = True
                current[r] += 1
            idle_rounds = 0 if progressed else idle_rounds + 1
            if idle_rounds > 200_000:
                raise Violation("I3: deadlock -- no block can make progress")
        for r in range(n):  # dsync_finalize: the page goes back to the pool with the last epoch written into it
            last_epoch[r] = base + epochs_per_comm if "no_epoch_base" not in bugs else epochs_per_comm
    return steps
