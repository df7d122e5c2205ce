"""A model of the receive agent (mpi_amd/csrc/sched.hip `p2p_agent_kernel`, mpi_amd/csrc/engine.cpp `agent_submit` /
`p2p_agent_stop`) that runs on the CPU.  Every actor -- the host thread of the Receives, block 0 of the agent, the
blocks that only watch -- is a generator that yields before every access to memory somebody else can see; a seeded
random scheduler picks who moves next, so one run is one interleaving and a few thousand seeds are a decent search.

What the real thing keeps where, the model keeps in two lists:
  cmd[0..7]   the command record in pinned host memory: w0 = doorbell | bytes << 2 | seq << 24, w1 = source, w2 =
              destination, w3 = mail entry | seq << 32 (written w1, w2, w3, then w0; read as two 16-byte halves, either
              half first), cmd[6] = number of the last command served, cmd[7] = "the agent has gone"
  rec[0..6]   device memory, NOT cleared between launches: [0] block 0's n-th word to the others, keyed by the launch,
              [1..3] what to copy, [5] ticket, [6] "every block has finished with word n"
The agent's patience is a number of polls; launches run in stream order (a launch starts when the one before has ended).

Checked: every message is copied exactly once, whole, to the right place, by the blocks that should (block 0 alone when
short); both acks follow the copy of EVERY block; the host never waits for ever, whatever the timing of its Receives
against the agent's patience; a stop is obeyed.  `bugs` switches known-bad variants on: "stale_key" is the hang of
round 3, session 7 (the key of block 0's word did not carry the launch number: the "go away" word of the launch before,
still in `rec`, was taken by the next launch's watchers for its first word)."""
from __future__ import annotations

import random

ALONE = 4  # messages of at most this many units are copied by block 0 alone (alone_bytes)


class Violation(AssertionError):
    pass


class Hang(AssertionError):
    pass


def run(messages, blocks=4, patience=6, seed=0, gaps=None, bugs=(), stop_at_end=True, max_steps=200000):
    """messages: list of lengths (units).  gaps[i]: scheduler turns the host idles before message i (None: random).
    Returns dict(launches=..., served=...)."""
    rng = random.Random(seed)
    cmd = [0] * 8
    rec = [0] * 7
    rec[0] = rng.getrandbits(20)  # whatever an earlier communicator's agent left there
    mem = {}  # (message index, unit) -> how many times written
    acked = {}  # message index -> True once MAIL_DONE was written
    copied_by = {}  # message index -> set of blocks that finished their share
    state = dict(launches=0, served=0, running=False, seq=0, host_done=False)
    pending = []  # launches not started yet: (launch number, seq0)
    actors = []  # live generators: [gen, name, launch]
    live_blocks = {}  # launch -> number of blocks still running

    def key_of(launch, n):
        return n if "stale_key" in bugs else (launch << 40) | n

    def share(i, length, b, nb):  # the units block b of nb copies
        return [u for u in range(length) if u % nb == b]

    def block(launch, seq0, b):
        seq, told = seq0, 0
        while True:
            go, wide = 0, True
            if b == 0:
                polls, have, w = 0, False, [0, 0, 0, 0]
                while True:
                    halves = [(0, 1), (2, 3)]
                    if rng.random() < 0.5:
                        halves.reverse()
                    for lo, hi in halves:  # two 16-byte loads, each atomic, in either order
                        yield
                        w[lo], w[hi] = cmd[lo], cmd[hi]
                    have = (w[0] & 3) != 0 and (w[0] >> 24) == seq and (w[3] >> 32) == seq
                    polls += 1
                    if have or polls > patience:
                        break
                if have and (w[0] & 3) == 1:
                    msg, length, go = w[1], (w[0] >> 2) & 0x3FFFFF, 1
                    if w[2] != msg or (w[3] & 0xFFFFFFFF) != msg:
                        raise Violation(f"command {seq}: torn record {w}")
                else:
                    msg, length, go = 0, 0, 0
                    yield
                    cmd[7] = seq + 1
                served_seq = seq
                if blocks > 1 and (not go or length > ALONE):
                    yield
                    rec[1] = msg
                    yield
                    rec[3] = length
                    yield
                    rec[0] = (key_of(launch, told + 1) << 1) | go
            else:
                want = key_of(launch, told + 1)
                while True:
                    yield
                    v = rec[0]
                    if v >> 1 == want:
                        break
                go = v & 1
                yield
                msg = rec[1]
                yield
                length = rec[3]
            if not go:
                return
            wide = blocks > 1 and length > ALONE
            if wide or b != 0:
                told += 1
            if b == 0 and not wide:
                mine = list(range(length))
            elif wide:
                mine = share(msg, length, b, blocks)
            else:
                raise Violation(f"block {b} woken for a short message")
            for u in mine:
                yield
                if acked.get(msg):
                    raise Violation(f"message {msg}: block {b} still copying after the ack")
                mem[(msg, u)] = mem.get((msg, u), 0) + 1
            copied_by.setdefault(msg, set()).add(b)
            key = key_of(launch, told)
            last = True
            if wide:
                yield
                rec[5] += 1
                last = rec[5] == blocks
                if last:
                    yield
                    rec[5] = 0
                    yield
                    rec[6] = key
            if b == 0:
                if wide and "no_wait_for_all" not in bugs:
                    while True:
                        yield
                        if rec[6] == key:
                            break
                yield
                acked[msg] = True
                yield
                cmd[6] = served_seq
                seq += 1

    def launch_agent(seq0):
        state["launches"] += 1
        pending.append((state["launches"], seq0))

    def host():
        for i, length in enumerate(messages):
            gap = gaps[i] if gaps is not None else rng.choice([0, 0, 1, 3, patience * 2, patience * 8, patience * 40])
            for _ in range(gap):
                yield
            state["seq"] += 1
            seq, msg = state["seq"], i + 1
            yield
            cmd[1] = msg
            yield
            cmd[2] = msg
            yield
            cmd[3] = msg | (seq << 32)
            yield
            cmd[0] = 1 | (length << 2) | (seq << 24)
            if not state["running"]:
                yield
                cmd[7] = 0
                launch_agent(seq)
                state["running"] = True
            while True:
                yield
                if cmd[6] == seq:
                    break
                yield
                if cmd[7] != 0:
                    yield
                    if cmd[6] == seq:
                        break
                    yield
                    cmd[7] = 0
                    launch_agent(seq)
            state["served"] += 1
            if not acked.get(msg):
                raise Violation(f"message {msg}: the host went on before the sender's ack")
        if stop_at_end and state["running"]:
            state["seq"] += 1
            seq = state["seq"]
            yield
            cmd[3] = seq << 32
            yield
            cmd[0] = 2 | (seq << 24)
            while True:
                yield
                if cmd[7] != 0:
                    break
        state["host_done"] = True

    actors.append([host(), "host", 0])
    steps = 0
    while True:
        # stream order: the next launch starts when no block of an earlier one is left
        if pending and not any(n > 0 for n in live_blocks.values()):
            launch, seq0 = pending.pop(0)
            live_blocks[launch] = blocks
            for b in range(blocks):
                actors.append([block(launch, seq0, b), f"L{launch}b{b}", launch])
        if not actors:
            break
        steps += 1
        if steps > max_steps:
            raise Hang(f"no end after {max_steps} turns: " + ", ".join(a[1] for a in actors))
        a = rng.choice(actors)
        try:
            next(a[0])
        except StopIteration:
            actors.remove(a)
            if a[2]:
                live_blocks[a[2]] -= 1
    if not state["host_done"]:
        raise Hang("the host never finished")
    for i, length in enumerate(messages):
        msg = i + 1
        for u in range(length):
            if mem.get((msg, u), 0) != 1:
                raise Violation(f"message {msg} unit {u} written {mem.get((msg, u), 0)} times")
        want = set(range(blocks)) if (blocks > 1 and length > ALONE) else {0}
        if copied_by.get(msg, set()) != want and length > 0:
            raise Violation(f"message {msg} (length {length}) copied by blocks {sorted(copied_by.get(msg, set()))}, expected {sorted(want)}")
    return dict(launches=state["launches"], served=state["served"], steps=steps)
