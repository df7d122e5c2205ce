#!/usr/bin/env python3
"""What scripts/profile_8gpu.sh left behind, read in the order DESIGN.md section 0 says to read it:

    python scripts/first_hour_report.py [profiles/r06_8gpu]

One screen: did the job degrade; what one link gives each engine, reads against writes; what the library's tuner chose; every
schedule by name against the time its busiest link direction needs at 76.8 GB/s (fold 0.25 S, ring 0.292 S,
halving 1.0 S, push-only 0.25 S); the ring in both forms against north_star's target; the XCD masks of the split form.  Reads
files only; rehearsed on virtual GPUs by tests/test_devsim.py::test_the_8gpu_script_rehearsal."""
import glob
import json
import os
import sys

DIR_GBPS = 76.8  # one direction of one xGMI link, nominal
SHARE = {"auto": None, "fused": 0.25, "fused2": 0.25, "split": 0.25, "zpush": 0.25, "ring": 1.75 / 6, "ring_push": 1.75 / 6, "rhd": 1.0, "rhd_push": 1.0}


def last_json(path):
    try:
        text = open(path).read().strip()
    except OSError:
        return None
    for piece in (text.split("\n")[-1], text):  # one JSON line after whatever was printed before it, or one (indented) document
        try:
            return json.loads(piece)
        except ValueError:
            pass
    return None


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "profiles", "r06_8gpu")
    lines = glob.glob(os.path.join(d, "bench_n*.json"))
    line = next((x for x in (last_json(f) for f in sorted(lines) if not f.endswith("_extras.json")) if x), None)
    print(f"== {d}")
    log = os.path.join(d, "pytest_split_sched_ll.log")
    if os.path.exists(log):
        print("1. GPU tests over the links (split form, stepped kernels in both forms, LL):", open(log).read().strip().split("\n")[-1])
    if line:
        n = line["n_gpus"]
        print(f"2. bench --gpus {n}: value {line['value']:.1f} GB/s algbw, {line['ms_per_step']:.3f} ms per step, busbw {line['busbw_GBps']:.1f}; "
              f"{line.get('ranks_meet')}; parity ok = {line['parity'].get('ok')}")
        print("   degraded:", line.get("degraded", "no (every rank mapped every peer's window and flag page, and the flag words arrive)"))
        rf = line["roofline"]
        if rf.get("bound") == "xgmi":
            print(f"   link roofline: schedule {rf['schedule']} ({rf['direction']}), busiest link direction {rf['achieved']:.1f} GB/s = "
                  f"{rf['frac']:.2f} of {rf['peak']} nominal" + (f", {rf['frac_of_measured']:.2f} of the probed {rf['peak_measured']:.1f}" if rf.get("peak_measured") else ""))
        probe = (line.get("xgmi") or {}).get("link_probe") or {}
        for eng in ("hipMemcpyAsync", "copy_kernel", "sys_kernel"):
            w, r = probe.get(eng + "_write_GBps"), probe.get(eng + "_read_GBps")
            if w and r:
                verdict = "reads and writes alike" if 0.85 < r / w < 1.18 else ("READS SLOWER: the push forms / push-only should win" if r < w else "writes slower: the pull forms should win")
                print(f"   link probe {eng:15s} write {w:7.1f}  read {r:7.1f}  both ways {probe.get(eng + '_bidir_each_GBps', 0):7.1f} GB/s  -> {verdict}")
        print("   the tuner chose:", line["config"].get("tuned"))
        # (xmpi_tune checks every candidate's ANSWER on patterned inputs before it believes its time: a schedule wrong on any rank is out on every rank)
        print("   schedules whose answers the library found WRONG on this node:", (line["config"].get("tuned") or {}).get("rejected", "none"))
        ex = last_json(os.path.join(d, f"bench_n{n}_extras.json")) or {}
        for coll, row in ((ex.get("autotune") or {}).get("tables_other") or {}).items():
            print(f"      {coll:9s} 1 KiB, 4 KiB ... :", " ".join(row))
        for form, v in (line.get("ring") or {}).items():
            if isinstance(v, dict) and "frac_of_link_peak" in v:
                ok = "MEETS" if v["frac_of_link_peak"] >= 0.7 else "below"
                print(f"   ring by name, {form:4s}: {v['ms_per_step']:.3f} ms, busbw {v['busbw_GBps']:.1f} GB/s, link direction {v['link_direction_GBps']:.1f} GB/s = "
                      f"{v['frac_of_link_peak']:.2f} of link peak ({ok} north_star's 0.70); parity ok = {v['parity_ok']}")
    for f in sorted(glob.glob(os.path.join(d, "prod_n*_*.json"))):
        p = last_json(f)
        if not p or "rows" not in p:
            continue
        S = p["bytes_per_rank"]
        print(f"3. {os.path.basename(f)}: {p['ranks']} ranks x {S >> 20 if S >= 1 << 20 else S / 1048576:g} MiB, exact = {p.get('exact')}, sharers {p.get('sharers')}, "
              f"XCD masks meet / done {p.get('xcd_meet_mask'):#x} / {p.get('xcd_done_mask'):#x} (short launches: {p.get('xcd_short')}), body_sys {p.get('body_sys')}; "
              f"self-check at init {p.get('init_selfcheck_us')} us, schedules rejected {p.get('tune_rejected')}, degraded {p.get('degraded')}")
        for row in p["rows"]:
            share = SHARE.get(row["mode"])
            bound_us = share * S / (DIR_GBPS * 1e9) * 1e6 if share else None
            tail = f"  link bound {bound_us:8.1f} us -> {bound_us / row['us_per_step']:.2f} of it" if bound_us and row["us_per_step"] else ""
            extra = f"  (tuned: {row['tuned']})" if row["mode"] == "auto" and row.get("tuned") else ""
            print(f"      {row['mode']:10s} {row['us_per_step']:10.1f} us per step  algbw {row.get('algbw_GBps', 0):7.1f} GB/s{tail}{extra}")
    for name in ("split_body_sys0", "split_body_sys1", "ring_channels_1", "ring_channels_2", "ring_channels_0"):
        for f in sorted(glob.glob(os.path.join(d, name + "_n*.json"))):
            p = last_json(f)
            if p and "rows" in p:
                print(f"4. {os.path.basename(f)}:", {r["mode"]: round(r["us_per_step"], 1) for r in p["rows"]})
    for f in sorted(glob.glob(os.path.join(d, "coll_sweep_n*.json"))):
        p = last_json(f)
        if p and "rows" in p:
            print(f"5. {os.path.basename(f)}: blocking / enqueued us:", [(r["bytes"], round(r["blocking_us"], 1), round(r["queued_us"], 1)) for r in p["rows"][:4]])
    c5 = last_json(next(iter(glob.glob(os.path.join(d, "cfg5_n*.json"))), ""))
    if c5:
        print("6. cfg 5 (fp16, 1 MiB ... 1 GiB): all schedules bit-identical to the rank-order result =", c5.get("all_bit_identical"))
    return 0


if __name__ == "__main__":
    sys.exit(main())
