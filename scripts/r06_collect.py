#!/usr/bin/env python3
"""Copies what scripts/r06_profile.sh left in gpurun_out/r06/ (scratch) into profiles/r06/final/ (tracked): the JSON lines, logs, the
kernel_stats.csv of the N = 1 bench under rocprofv3, the PMC summaries; and rewrites profiles/pmc_traffic.json (what bench.py reads
for `roofline.traffic`; tests/test_bench_layout.py fails when it is older than the newest profiles/rNN/) from this run's passes."""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r06")
DST = os.path.join(ROOT, "profiles", "r06", "final")
ROUND = 6


def main():
    if not os.path.isdir(SRC):
        sys.exit(f"{SRC}: run scripts/r06_profile.sh through gpurun first")
    os.makedirs(DST, exist_ok=True)
    for f in glob.glob(os.path.join(DST, "*")):
        if os.path.isfile(f):
            os.remove(f)
    for f in sorted(glob.glob(os.path.join(SRC, "*.json")) + glob.glob(os.path.join(SRC, "*.log"))):
        if os.path.getsize(f):
            shutil.copy(f, os.path.join(DST, os.path.basename(f)))
    # (gpurun MERGES a call's files into gpurun_out/: the csv of every earlier call is still there under its own pid -- the newest is this call's)
    n1 = sorted(glob.glob(os.path.join(SRC, "stats_n1", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    if n1:
        shutil.copy(n1[-1], os.path.join(DST, "bench_zcopy_kernel_stats.csv"))
    want = {"reduce_n_multi_kernel<float, 0, 8, 2>": ("zcopy, 8 rank threads, one launch folds all chunks", 4294967296),
            "copy_pairs_kernel<8, 2>": ("the fold's access pattern without the arithmetic (8 sources -> 8 destinations, one launch): bench.py's box copy", 4294967296)}
    prod = {"dsync_body_kernel<float, 0, 8, 2>": ("one process per rank, meet / body / done: chip-wide traffic during one rank's data kernel (the ranks' data kernels overlap only "
                                                  "partly, so this is NOT a per-kernel figure: between 1x and 8x of 512 MiB)", 536870912),
            "dsync_fold_kernel<float, 0, 8, 1>": ("one process per rank, one-kernel form: chip-wide traffic during one rank's kernel = the whole step (8 kernels x 512 MiB)", 4294967296)}
    rows = []
    for name, table in (("pmc_bench_zcopy.json", want), ("pmc_prod_8proc.json", prod)):
        try:
            for row in json.load(open(os.path.join(DST, name))):
                k = row["kernel"].replace("xmpi::", "")
                if k in table and not any(r["kernel"] == k for r in rows):
                    rows.append({"kernel": k, "schedule": table[k][0], "launches": row["launches"], "grid_threads": row["grid_threads"],
                                 "traffic_bytes_per_launch": row["traffic_bytes_per_launch"], "FETCH_SIZE_KiB_mean": row["FETCH_SIZE_KiB_mean"],
                                 "WRITE_SIZE_KiB_mean": row["WRITE_SIZE_KiB_mean"], "algorithmic_bytes_per_launch": table[k][1]})
        except (OSError, ValueError) as e:
            print("no PMC passes in", name, e)
    if any(r["kernel"].startswith("reduce_n_multi_kernel") for r in rows):
        out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with --kernel-trace only (scripts/r06_profile.sh: `python bench.py --algo zcopy "
                         "--no-extras --no-cpu --no-production --steps 5`, 8 ranks as threads; `xmpirun 8 allreduce_bench 268435456 5 2 split fused`, 8 processes); "
                         "traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction, MI355X_MICROARCH.md HBM section).  The TCC counters are chip-wide: with 8 "
                         "processes a kernel's figure is what ALL ranks' kernels moved while it ran",
               "round": ROUND, "rows": rows}
        json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print(len(os.listdir(DST)), "files in", DST)
    for r in rows:
        print(r["kernel"], r["traffic_bytes_per_launch"], r["traffic_bytes_per_launch"] / r["algorithmic_bytes_per_launch"])


if __name__ == "__main__":
    main()
