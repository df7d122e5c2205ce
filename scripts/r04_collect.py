#!/usr/bin/env python3
"""Copies what scripts/r04_profile.sh left in gpurun_out/r04/ (scratch) into profiles/r04/ (tracked): the JSON lines and text
files as they are, one kernel_stats.csv per profiled process renamed by rank (processes in PID order = ranks in launch order),
the PMC summaries; and rewrites profiles/pmc_traffic.json (what bench.py reads for `roofline.traffic`) from this run's passes."""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r04")
DST = os.path.join(ROOT, "profiles", "r04")


def main():
    if not os.path.isdir(SRC):
        sys.exit(f"{SRC}: run scripts/r04_profile.sh through gpurun first")
    os.makedirs(DST, exist_ok=True)
    for f in glob.glob(os.path.join(DST, "*")):  # (the directories -- cpx/, guard/ -- are other calls' evidence: kept)
        if os.path.isfile(f):
            os.remove(f)
    for f in sorted(glob.glob(os.path.join(SRC, "*.json")) + glob.glob(os.path.join(SRC, "*.txt"))):
        if os.path.getsize(f):
            shutil.copy(f, os.path.join(DST, os.path.basename(f)))
    os.replace(os.path.join(DST, "pmc_prod.json"), os.path.join(DST, "pmc_prod_8proc.json"))
    n1 = sorted(glob.glob(os.path.join(SRC, "stats_n1", "*", "*_kernel_stats.csv")), key=os.path.getsize)
    if n1:
        shutil.copy(n1[-1], os.path.join(DST, "bench_zcopy_kernel_stats.csv"))
    for mode in ("split", "fused", "ring", "rhd"):
        files = [f for f in glob.glob(os.path.join(SRC, f"stats_prod_{mode}", "*", "*_kernel_stats.csv")) if os.path.getsize(f) > 0]
        files.sort(key=lambda f: int(os.path.basename(f).split("_")[0]))
        for r, f in enumerate(files):
            shutil.copy(f, os.path.join(DST, f"prod_{mode}_rank{r}_kernel_stats.csv"))
    # roofline.traffic: this run's PMC passes
    rows = []
    want = {"reduce_n_multi_kernel<float, 0, 8, 2>": ("zcopy, 8 rank threads, one launch folds all chunks", 4294967296),
            "dsync_fold_kernel<float, 0, 8, 1>": ("one process per rank, one-kernel form: chip-wide traffic during one rank's kernel = the whole step (8 kernels x 512 MiB)", 4294967296),
            "dsync_body_kernel<float, 0, 8, 2>": ("one process per rank, meet / body / done: chip-wide traffic during one rank's data kernel (the ranks' data kernels overlap only partly, so this is NOT a per-kernel figure: between 1x and 8x of 512 MiB)", 536870912)}
    for name in ("pmc_bench_zcopy.json", "pmc_prod_8proc.json"):
        for row in json.load(open(os.path.join(DST, name))):
            k = row["kernel"].replace("xmpi::", "")
            if k in want:
                rows.append({"kernel": k, "schedule": want[k][0], "launches": row["launches"], "grid_threads": row["grid_threads"],
                             "traffic_bytes_per_launch": row["traffic_bytes_per_launch"], "FETCH_SIZE_KiB_mean": row["FETCH_SIZE_KiB_mean"],
                             "WRITE_SIZE_KiB_mean": row["WRITE_SIZE_KiB_mean"], "algorithmic_bytes_per_launch": want[k][1]})
    old = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    old["rows"] = rows
    old["round"] = 4
    old["source"] = old["source"].replace("r03_profile", "r04_profile")
    json.dump(old, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print(len(os.listdir(DST)), "files in", DST)
    for r in rows:
        print(r["kernel"], r["traffic_bytes_per_launch"], r["traffic_bytes_per_launch"] / r["algorithmic_bytes_per_launch"])


if __name__ == "__main__":
    main()
