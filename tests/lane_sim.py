"""A model of the host lanes (mpi_amd/csrc/engine.cpp p2p_send / p2p_recv, DIRECT_HOST; mpi_amd/csrc/ctl.h host_lane): a
host-resident payload travels through a ring of S pieces in the job's shared segment.  The sender fills as many pieces as
the ring holds BEFORE it posts the message, then one piece whenever the receiver's tail has made room; the receiver takes
pieces as the head shows them -- with memcpy one by one (host destination), or in RUNS that a kernel pulls (device
destination: a run ends where the ring wraps, one run in flight, the tail moves when its completion word arrives) -- and
acknowledges (state DONE) after the last; a receiver whose buffer is too short says so at once (verdict + DONE) and the
sender stops streaming; a sender nobody answers withdraws the message (POSTED -> FREE) unless a receiver matched it first.

Actors are generators that yield before every access to shared state; a seeded scheduler interleaves them.
Checked: the bytes arrive whole and in order, no piece is overwritten before it was taken, both sides end with the same
verdict, the entry is FREE and its counters zero afterwards, nothing waits for ever.  `bugs`: known-bad variants."""
from __future__ import annotations

import random

FREE, CLAIMED, POSTED, MATCHED, DONE = range(5)
OK, TRUNCATE, TIMEOUT = 0, -7, -5


class Violation(AssertionError):
    pass


class Hang(AssertionError):
    pass


def run(length, slots=4, piece=8, capacity=None, device_dst=False, receiver_delay=0, sender_patience=None, seed=0, bugs=(),
        kernel_turns=6, max_steps=200000):
    """One message of `length` units through a ring of `slots` pieces of `piece` units.  capacity: the receiver's buffer
    (None = large enough).  sender_patience: turns without progress after which the sender tries to withdraw (None = never).
    Returns (sender's verdict, receiver's verdict, what the receiver holds)."""
    rng = random.Random(seed)
    payload = [(7 * i + 3) & 0xFF for i in range(length)]
    ring = [None] * (slots * piece)
    e = dict(state=FREE, status=OK, head=0, tail=0, bytes=0)
    out = dict(sender=None, receiver=None, got=[None] * length)
    npieces = (length + piece - 1) // piece

    def fill(k):
        for i in range(k * piece, min((k + 1) * piece, length)):
            idx = (k % slots) * piece + (i - k * piece)
            yield
            ring[idx] = (k, payload[i])

    def sender():
        yield
        assert e["state"] == FREE
        e["state"] = CLAIMED
        e["bytes"] = length
        filled = 0
        while filled < npieces and filled < slots:  # eager: in place before the message is visible
            yield from fill(filled)
            filled += 1
        yield
        e["head"] = filled
        yield
        e["state"] = POSTED
        idle = 0
        while filled < npieces:
            yield
            room = filled - e["tail"] < slots or "no_room_check" in bugs
            if room:
                yield from fill(filled)
                filled += 1
                yield
                e["head"] = filled
                idle = 0
                continue
            yield
            if e["state"] == DONE:
                break  # the receiver gave up (truncate)
            idle += 1
            if sender_patience is not None and idle > sender_patience:
                yield
                if e["state"] == POSTED:  # withdraw: nobody matched it
                    e["state"] = CLAIMED
                    yield
                    e["head"] = e["tail"] = 0
                    yield
                    e["state"] = FREE
                    out["sender"] = TIMEOUT
                    return
                idle = 0
        idle = 0
        while True:  # await_ack
            yield
            if e["state"] == DONE:
                break
            idle += 1
            if sender_patience is not None and idle > sender_patience:
                yield
                if e["state"] == POSTED:
                    e["state"] = CLAIMED
                    yield
                    e["head"] = e["tail"] = 0
                    yield
                    e["state"] = FREE
                    out["sender"] = TIMEOUT
                    return
                idle = 0
        yield
        out["sender"] = e["status"]
        yield
        e["head"] = e["tail"] = 0
        yield
        e["state"] = FREE

    def receiver():
        for _ in range(receiver_delay):
            yield
        waited = 0
        while True:
            yield
            if e["state"] == POSTED:
                e["state"] = MATCHED  # (a compare-and-swap in the real thing: one step)
                break
            waited += 1
            if waited > 50000:
                out["receiver"] = TIMEOUT
                return
        yield
        n = e["bytes"]
        if capacity is not None and n > capacity:
            yield
            e["status"] = TRUNCATE
            yield
            e["state"] = DONE
            out["receiver"] = TRUNCATE
            return
        taken = 0
        pending = None  # (first piece, pieces) of the run a kernel is pulling
        while taken < npieces:
            yield
            head = e["head"]
            if head > taken and not device_dst:
                for k in range(taken, head):
                    for i in range(k * piece, min((k + 1) * piece, length)):
                        yield
                        cell = ring[(k % slots) * piece + (i - k * piece)]
                        if cell is None or cell[0] != k:
                            raise Violation(f"piece {k}: the ring holds {cell} at unit {i}")
                        out["got"][i] = cell[1]
                taken = head
                yield
                e["tail"] = taken
            elif head > taken and pending is None:
                first = taken % slots
                run = min(head - taken, slots - first) if "run_wraps" not in bugs else head - taken
                pending = (taken, run, rng.randrange(1, kernel_turns))  # the kernel takes a few turns
            elif pending is not None:
                k0, run, left = pending
                if left > 0:
                    pending = (k0, run, left - 1)
                    continue
                lo = (k0 % slots) * piece  # ONE contiguous copy out of the lane, as the kernel does it
                for j in range(min(run * piece, length - k0 * piece)):
                    idx = lo + j
                    if idx >= len(ring):
                        raise Violation("the kernel read past the end of the lane")
                    cell = ring[idx]
                    k = k0 + j // piece
                    if cell is None or cell[0] != k:
                        raise Violation(f"piece {k}: the ring holds {cell}")
                    out["got"][k0 * piece + j] = cell[1]
                taken += run
                pending = None
                yield
                e["tail"] = taken
        yield
        e["status"] = OK
        yield
        e["state"] = DONE
        out["receiver"] = OK

    actors = [sender(), receiver()]
    steps = 0
    while actors:
        steps += 1
        if steps > max_steps:
            raise Hang(f"no end after {max_steps} turns (state {e})")
        a = rng.choice(actors)
        try:
            next(a)
        except StopIteration:
            actors.remove(a)
    if out["sender"] != TIMEOUT and (e["state"] != FREE or e["head"] or e["tail"]):
        raise Violation(f"the entry was not given back clean: {e}")
    return out["sender"], out["receiver"], out["got"]


def expected(length):
    return [(7 * i + 3) & 0xFF for i in range(length)]
