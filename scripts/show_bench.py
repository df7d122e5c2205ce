"""Pretty-print the compact JSON line bench.py wrote (and bench_extras.json beside it):  python scripts/show_bench.py file [file...]"""
import json
import os
import sys


def r(v, n=2):
    return round(v, n) if isinstance(v, float) else v


for f in sys.argv[1:]:
    txt = open(f).read().strip()
    if not txt:
        print(f, "EMPTY")
        continue
    d = json.loads(txt.split("\n")[-1])
    cfg = d["config"]
    print(f"== {f}: n_gpus={d['n_gpus']} ranks={cfg['ranks']} value={r(d['value'])} {d['unit']} "
          f"ms/step={r(d['ms_per_step'], 3)} algbw={r(d['algbw_GBps'])} busbw={r(d['busbw_GBps'])}  [{len(txt.split(chr(10))[-1])} bytes]")
    print("   config:", {k: v for k, v in cfg.items() if k != "workload"}, "|", d.get("ranks_meet"))
    ro = d["roofline"]
    if ro.get("bound") == "xgmi":  # one rank per GPU: the link roofline of the schedule that was timed
        print(f"   roofline (LINK): schedule={ro['schedule']} ({ro['direction']}) busiest link direction {r(ro['achieved'])} GB/s of {ro['peak']} = "
              f"frac {r(ro['frac'], 3)}; measured link {r(ro.get('peak_measured'))} GB/s -> frac_of_measured {r(ro.get('frac_of_measured'), 3)}; "
              f"{ro['busiest_link_direction_bytes_over_S']:.3f} S per link direction")
        ro = d.get("roofline_hbm", {})
    if ro:
        print(f"   roofline (HBM): {ro['kernel'][:48]} achieved={r(ro['achieved'])} frac={r(ro['frac'], 3)} launches={ro.get('launches')} "
              f"avg_us={r(ro.get('avg_launch_us'))} bytes/launch={ro.get('algorithmic_bytes_per_launch')} traffic={ro.get('traffic')}")
    if d.get("degraded"):
        print("   DEGRADED:", d["degraded"])
    if d.get("ring"):
        print("   ring by name (north_star's target: frac_of_link_peak >= 0.7):",
              {k: ({x: r(y, 3) for x, y in v.items()} if isinstance(v, dict) else v) for k, v in d["ring"].items()})
    for k in ("roofline_production", "roofline_isolated", "xgmi", "busbw_at_size", "cfg5_f16_us", "parity"):
        if d.get(k):
            v = d[k]
            print(f"   {k}:", {a: (r(b, 3) if not isinstance(b, dict) else {x: r(y, 2) for x, y in b.items()}) for a, b in v.items()} if isinstance(v, dict) else v)
    if d.get("cpu_baseline"):
        print("   cpu:", {k: r(v, 4) for k, v in d["cpu_baseline"].items() if k != "sample"})
    ex = os.path.join(os.path.dirname(os.path.abspath(f)), os.path.basename(f).replace(".json", "_extras.json"))
    if os.path.exists(ex):
        e = json.load(open(ex))
        print("   autotune:", {k: v for k, v in e.get("autotune", {}).items() if not k.startswith("table")})
        for k, v in (e.get("extras") or {}).items():
            if isinstance(v, dict) and "rows" in v:
                print("    ", k)
                for row in v["rows"]:
                    print("       ", {a: (r(b, 1) if not isinstance(b, dict) else {x: r(y, 1) for x, y in b.items()}) for a, b in row.items()})
            elif isinstance(v, list):
                print("    ", k)
                for row in v:
                    print("       ", {a: r(b, 1) for a, b in row.items()} if isinstance(row, dict) else row)
            else:
                print("    ", k, {a: (r(b) if not isinstance(b, dict) else {x: r(y) for x, y in b.items()}) for a, b in v.items()} if isinstance(v, dict) else v)
