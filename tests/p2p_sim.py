"""A model of the stream-ordered Send / Receive protocol (mpi_amd/csrc/sched.hip `p2p_send_kernel` / `p2p_recv_kernel`,
mpi_amd/csrc/dsync.cpp `dsync_send` / `dsync_recv`) that runs on the CPU: kernels are little state machines, streams run
their kernels in order, a seeded random scheduler interleaves everything that can move.

What the real thing keeps in the flag allocations the model keeps in dictionaries:
  box[(src, dst)][b]   {seq, tag, payload}   written by the sender's kernel, b = (n - 1) % K for message number n of the pair
  ack[(src, dst)][b]   seq                   written by the receiver's kernel when it has consumed the message of that box
  taken[(src, dst)][b] seq                   the receiver's own note of the last message it consumed from the box
Message numbers carry the communicator number in their high bits (pages are pooled and never cleared).

Checked: every receive gets the payload of the send with its tag (oldest first when a tag repeats); a box is never
overwritten before its message was consumed; everything terminates whatever the interleaving -- as long as the program
obeys the documented rule (no stream waits for work queued behind it); stale records of an earlier communicator in the
same pages are never matched.  `bugs` switches known-bad variants on."""
from __future__ import annotations

import random

K = 8  # kP2PBoxes


class Violation(AssertionError):
    pass


def run(ops, seed=0, comm_tag=1, stale=None, bugs=()):
    """ops: {rank: [stream, ...]}, a stream = list of ("send", peer, tag, payload) / ("recv", peer, tag).
    Returns {(rank, stream index, op index): received payload}."""
    rng = random.Random(seed)
    box, ack, taken = {}, {}, {}

    def tab(t, pair):
        return t.setdefault(pair, [dict(seq=0, tag=None, payload=None) if t is box else 0 for _ in range(K)])

    if stale:  # what an earlier communicator left behind in the (uncleared) pages
        for pair, recs in stale.items():
            for b, (seq, tag) in enumerate(recs):
                tab(box, pair)[b] = dict(seq=seq, tag=tag, payload="STALE")
                tab(taken, pair)[b] = 0 if "stale_unconsumed" in bugs else seq
    out_seq = {}
    # the host numbers the messages of an ordered pair when it ENQUEUES the send (dsync_send)
    kernels = {}
    for rank, streams in ops.items():
        for si, stream in enumerate(streams):
            for oi, op in enumerate(stream):
                k = dict(rank=rank, op=op, state="new", got=None)
                if op[0] == "send":
                    n = out_seq[(rank, op[1])] = out_seq.get((rank, op[1]), 0) + 1
                    k["n"], k["seq"] = n, (comm_tag << 32) | n
                kernels[(rank, si, oi)] = k
    pos = {(rank, si): 0 for rank, streams in ops.items() for si in range(len(streams))}
    results, idle = {}, 0
    while any(pos[(r, si)] < len(ops[r][si]) for (r, si) in pos):
        live = [(r, si) for (r, si) in pos if pos[(r, si)] < len(ops[r][si])]
        r, si = rng.choice(live)
        key = (r, si, pos[(r, si)])
        k = kernels[key]
        op = k["op"]
        moved = True
        if op[0] == "send":
            pair = (r, op[1])
            b = (k["n"] - 1) % K
            if k["state"] == "new":
                if k["n"] > K and tab(ack, pair)[b] < k["seq"] - K and "no_box_wait" not in bugs:
                    moved = False  # the message that used the box before has not been answered yet
                else:
                    old = tab(box, pair)[b]
                    if old["seq"] >> 32 == comm_tag and old["seq"] > tab(taken, pair)[b]:
                        raise Violation(f"box {b} of {pair} overwritten before message {old['seq'] & 0xffffffff} was consumed")
                    tab(box, pair)[b] = dict(seq=k["seq"], tag=op[2], payload=op[3])
                    k["state"] = "posted"
            elif tab(ack, pair)[b] >= k["seq"]:
                k["state"] = "done"
            else:
                moved = False
        else:
            pair = (op[1], r)
            if k["state"] == "new":
                best = None
                for b in range(K):
                    rec = tab(box, pair)[b]
                    if rec["seq"] >> 32 != comm_tag and "no_comm_check" not in bugs:
                        continue
                    if rec["seq"] == 0 or rec["seq"] <= tab(taken, pair)[b] or rec["tag"] != op[2]:
                        continue
                    if best is None or rec["seq"] < tab(box, pair)[best]["seq"]:
                        best = b
                if best is None:
                    moved = False
                else:
                    rec = tab(box, pair)[best]
                    k["got"], k["b"], k["seq"] = rec["payload"], best, rec["seq"]
                    tab(taken, pair)[best] = rec["seq"]
                    k["state"] = "copied"
            else:
                tab(ack, pair)[k["b"]] = k["seq"]
                results[key] = k["got"]
                k["state"] = "done"
        if k["state"] == "done":
            pos[(r, si)] += 1
        idle = 0 if moved else idle + 1
        if idle > 20000:
            raise Violation("deadlock: no kernel can make progress")
    return results
