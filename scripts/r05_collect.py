#!/usr/bin/env python3
"""Copies what scripts/r05_profile.sh left in gpurun_out/r05/ (scratch) into profiles/r05/ (tracked): the JSON lines, logs and
text files as they are, one kernel_stats.csv per profiled process renamed by rank (processes in PID order = ranks in launch
order), the PMC summaries; and rewrites profiles/pmc_traffic.json (what bench.py reads for `roofline.traffic`) from this run's
passes."""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r05")
DST = os.path.join(ROOT, "profiles", "r05")


def main():
    if not os.path.isdir(SRC):
        sys.exit(f"{SRC}: run scripts/r05_profile.sh through gpurun first")
    os.makedirs(DST, exist_ok=True)
    for f in glob.glob(os.path.join(DST, "*")):
        if os.path.isfile(f):
            os.remove(f)
    for f in sorted(glob.glob(os.path.join(SRC, "*.json")) + glob.glob(os.path.join(SRC, "*.log"))):
        if os.path.getsize(f):
            shutil.copy(f, os.path.join(DST, os.path.basename(f)))
    n1 = sorted(glob.glob(os.path.join(SRC, "stats_n1", "*", "*_kernel_stats.csv")), key=os.path.getsize)
    if n1:
        shutil.copy(n1[-1], os.path.join(DST, "bench_zcopy_kernel_stats.csv"))
    for mode in ("ring", "ring_push"):
        files = [f for f in glob.glob(os.path.join(SRC, f"stats_prod_{mode}", "*", "*_kernel_stats.csv")) if os.path.getsize(f) > 0]
        files.sort(key=lambda f: int(os.path.basename(f).split("_")[0]))
        for r, f in enumerate(files):
            shutil.copy(f, os.path.join(DST, f"prod_{mode}_rank{r}_kernel_stats.csv"))
    rows = []
    want = {"reduce_n_multi_kernel<float, 0, 8, 2>": ("zcopy, 8 rank threads, one launch folds all chunks", 4294967296)}
    try:
        for row in json.load(open(os.path.join(DST, "pmc_bench_zcopy.json"))):
            k = row["kernel"].replace("xmpi::", "")
            if k in want:
                rows.append({"kernel": k, "schedule": want[k][0], "launches": row["launches"], "grid_threads": row["grid_threads"],
                             "traffic_bytes_per_launch": row["traffic_bytes_per_launch"], "FETCH_SIZE_KiB_mean": row["FETCH_SIZE_KiB_mean"],
                             "WRITE_SIZE_KiB_mean": row["WRITE_SIZE_KiB_mean"], "algorithmic_bytes_per_launch": want[k][1]})
    except (OSError, ValueError) as e:
        print("no N = 1 PMC passes:", e)
    if rows:
        old = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        old["rows"] = rows + [r for r in old["rows"] if not r["kernel"].startswith("reduce_n_multi_kernel")]
        old["round"] = 5
        old["source"] = "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with --kernel-trace only (scripts/r05_profile.sh; dsync rows: scripts/r04_profile.sh), traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024"
        json.dump(old, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print(len(os.listdir(DST)), "files in", DST)
    for r in rows:
        print(r["kernel"], r["traffic_bytes_per_launch"], r["traffic_bytes_per_launch"] / r["algorithmic_bytes_per_launch"])
    try:
        for row in json.load(open(os.path.join(DST, "pmc_sched_8proc.json"))):
            print(row["kernel"], row["grid_threads"], row["launches"], round(row["traffic_bytes_per_launch"] / 1e9, 3), "GB chip-wide during one rank's kernel")
    except (OSError, ValueError):
        pass


if __name__ == "__main__":
    main()
