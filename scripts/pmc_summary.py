"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as
MI355X_MICROARCH.md §HBM prescribes for gfx950: both counters are in KiB; FETCH_SIZE reports half of
the bytes of a wide coalesced read (x2); WRITE_SIZE matched the algorithmic bytes in calibration.

    python scripts/pmc_summary.py <fetch_dir> <write_dir> [substring-of-kernel-name ...] > out.json
Rows are grouped by (kernel, grid size): one group per launch geometry."""
import collections
import csv
import glob
import json
import os
import sys


def load(d, counter):
    agg = collections.defaultdict(list)
    # one file per traced process (a job of 8 rank processes leaves 8): all of them
    for p in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(p)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"]
            short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            agg[(short, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return agg


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    want = sys.argv[3:]
    out = []
    for key in sorted(set(fetch) & set(write)):
        if want and not any(w in key[0] for w in want):
            continue
        f, w = fetch[key], write[key]
        fm, wm = sum(f) / len(f), sum(w) / len(w)
        out.append({"kernel": key[0], "grid_threads": key[1], "launches": len(f),
                    "FETCH_SIZE_KiB_mean": fm, "WRITE_SIZE_KiB_mean": wm,
                    "traffic_bytes_per_launch": (2.0 * fm + wm) * 1024.0,
                    "correction": "traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024  [gfx950: FETCH_SIZE counts 64 B per 128-B request]"})
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
