"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel count / mean / total and the fraction of the
traced wall time during which at least one kernel was running (GPU busy union).
    python scripts/trace_busy.py <dir-or-csv>"""
import csv
import glob
import os
import sys
from collections import defaultdict


def find_csv(path):
    if os.path.isfile(path):
        return path
    c = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))
    if not c:
        raise SystemExit("no *kernel_trace.csv under " + path)
    return c[-1]


def main():
    p = find_csv(sys.argv[1])
    rows = list(csv.DictReader(open(p)))
    iv = []
    per = defaultdict(lambda: [0, 0.0])
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0][-60:]
        iv.append((s, e))
        per[name][0] += 1
        per[name][1] += (e - s)
    iv.sort()
    busy, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        busy += cur_e - cur_s
    wall = iv[-1][1] - iv[0][0] if iv else 0
    print(f"{p}: {len(rows)} dispatches, wall {wall / 1e6:.2f} ms, busy-union {busy / 1e6:.2f} ms ({100.0 * busy / max(1, wall):.1f}%)")
    for name, (n, tot) in sorted(per.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"  {n:7d} x {tot / n / 1e3:9.2f} us = {tot / 1e6:9.2f} ms  {name}")


if __name__ == "__main__":
    main()
