"""What the links and each device's memory would carry, per schedule, counted on tests/devsim's virtual devices (no GPU).

python scripts/r04_traffic.py [--ranks 2 4 8] [--bytes-per-chunk 262144]
  -> profiles/r04/devsim/traffic_N<ranks>.json  (the scenario's TRAFFIC line: every figure in bytes)
  -> profiles/r04/devsim/traffic.md             (the same in units of S = bytes per rank)

One process per virtual device, the library's own kernel sources with every load and store traced (tests/devsim/build.py
--traffic), the scenario tests/scenarios.py sc_traffic, which ASSERTS the payload matrix against each schedule's plan before it
prints.  These are byte counts, not rates: what a link carries, not how fast."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, nargs="*", default=[2, 4, 8])
    ap.add_argument("--chunk-bytes", type=int, default=256 << 10, help="S / N: one rank's chunk")
    a = ap.parse_args()
    from tests.devsim import build
    from tests.gpu_harness import run_ranks
    lib = build.build_traffic_lib()
    out_dir = os.path.join(ROOT, "profiles", "r04", "devsim")
    os.makedirs(out_dir, exist_ok=True)
    md = ["# Link and memory traffic per schedule, counted on virtual devices (`scripts/r04_traffic.py`; bytes, not rates)", "",
          "S = bytes per rank.  `link max` = the most payload any one link direction carries; `links` = directed links that carry "
          "payload (of N (N - 1)); `HBM / device` = bytes one device's memory serves (its own kernels' and its peers' accesses); "
          "`flag stores` = bytes of flag words / boxes / LL lines written into peers' flag pages, all ranks together.", ""]
    for n in a.ranks:
        count = n * a.chunk_bytes // 8
        outs = run_ranks("traffic", n, {"count": count}, timeout=900, env={"XMPI_DEVSIM_LIB": lib, "DEVSIM_TRAFFIC": "1"})
        line = next(l for o in outs for l in o.splitlines() if l.startswith("TRAFFIC "))
        d = json.loads(line[8:])
        with open(os.path.join(out_dir, f"traffic_N{n}.json"), "w") as f:
            json.dump(d, f, indent=1)
        S = d["bytes_per_rank"]
        md += [f"## N = {n}, S = {S} bytes", "", "| schedule | remote loads / rank | remote stores / rank | links | link max | HBM / device | flag stores |",
               "|---|---|---|---|---|---|---|"]
        for name, v in d["schedules"].items():
            hb = v["hbm_per_device_max"] / S
            hbm = f"{hb:.3f} S" if v["hbm_per_device_max"] == v["hbm_per_device_min"] else f"{v['hbm_per_device_min'] / S:.3f} … {hb:.3f} S"
            md.append(f"| {name} | {v['remote_loads'] / S / n:.3f} S | {v['remote_stores'] / S / n:.3f} S | {v['links_used']} / {n * (n - 1)} | "
                      f"{v['busiest_link_direction'] / S:.4f} S | {hbm} | {v['flag_page_remote_stores']} B |")
        md.append("")
    with open(os.path.join(out_dir, "traffic.md"), "w") as f:
        f.write("\n".join(md))
    print("\n".join(md))


if __name__ == "__main__":
    main()
