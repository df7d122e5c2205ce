"""Pretty-print the JSON line(s) bench.py wrote:  python scripts/show_bench.py file [file...]"""
import json
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
          f"ms/step={r(d['ms_per_step'], 3)} algbw={r(d['algbw_GBps'])} busbw={r(d['busbw_GBps'])}")
    print("   config:", {k: v for k, v in cfg.items() if k != "workload"})
    ro = d["roofline"]
    print(f"   roofline: {ro['kernel'][:40]} achieved={r(ro['achieved'])} frac={r(ro['frac'], 3)} launches={ro['launches']} "
          f"avg_us={r(ro['avg_launch_us'])} bytes/launch={ro['algorithmic_bytes_per_launch']}")
    if d.get("roofline_isolated"):
        i = d["roofline_isolated"]
        print(f"   isolated: achieved={r(i['achieved'])} frac={r(i['frac'], 3)} avg_us={r(i['avg_launch_us'])} bytes/launch={i['bytes_per_launch']}")
    print("   parity:", d["parity"])
    if d.get("cpu_baseline"):
        c = d["cpu_baseline"]
        print("   cpu:", {k: r(v, 4) for k, v in c.items() if k != "sample"})
    for t in d.get("autotune", []):
        print("     tune", {k: r(v) for k, v in t.items()})
    ex = d.get("extras") or {}
    for k, v in ex.items():
        if k == "size_sweep":
            for row in v:
                print("      ", {a: r(b, 1) for a, b in row.items()})
        else:
            print("    ", k, {a: (r(b) if not isinstance(b, dict) else {x: r(y) for x, y in b.items()}) for a, b in v.items()} if isinstance(v, dict) else v)
