#!/bin/bash
# HBM traffic of the stepped kernels FORM BY FORM from the PMC counters (separate FETCH_SIZE / WRITE_SIZE passes, --kernel-trace only;
# one pair of passes per form: all four forms are the same kernel name and grid, one pass over all of them averages them), 8 processes
# on the one GPU, 256 MiB f32 -- beside what tests/devsim counted for the same kernels: ring pull 4.375 S per device (35 S), ring push
# 4.5 S (36 S), halving pull 4.375 S, halving push 5.25 S (42 S).  -> gpurun_out/r05_pmc_forms/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 XMPI_TIMEOUT_S=30 XMPI_NGPUS=1
O=$GRAFT_REPO_ROOT/gpurun_out/r05_pmc_forms
rm -rf $O; mkdir -p $O
BIN=$GRAFT_REPO_ROOT/mpi_amd/bin
PRODS="$BIN/xmpirun 8 $BIN/allreduce_bench 268435456 5 2"
cd /tmp
port=7300
for m in ring ring_push rhd rhd_push; do
  for c in FETCH_SIZE WRITE_SIZE; do
    port=$((port + 20))
    XMPI_BASEPORT=$port timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${m}_$c -- $PRODS $m > $O/under_pmc_${m}_$c.json 2> $O/${m}_$c.err; echo "$m $c rc=$?"
  done
done
cd $GRAFT_REPO_ROOT
for m in ring ring_push rhd rhd_push; do
  python scripts/pmc_summary.py $O/${m}_FETCH_SIZE $O/${m}_WRITE_SIZE dsync_sched > $O/pmc_$m.json
done
find $O -name "*.csv" -delete; find $O -name "*.db" -delete
python - <<'PY'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05_pmc_forms"
S = 268435456
plan = {"ring": 35, "ring_push": 36, "rhd": 35, "rhd_push": 42}
out = {}
for m, units in plan.items():
    rows = json.load(open(f"{O}/pmc_{m}.json"))
    if not rows:
        print(m, "no rows"); continue
    r = max(rows, key=lambda x: x["launches"])
    t = r["traffic_bytes_per_launch"]
    out[m] = {"launches": r["launches"], "traffic_bytes_per_step": t, "counted_on_virtual_devices": units * S, "ratio": t / (units * S),
              "FETCH_bytes": 2 * r["FETCH_SIZE_KiB_mean"] * 1024, "WRITE_bytes": r["WRITE_SIZE_KiB_mean"] * 1024}
    print(m, out[m])
json.dump({"note": "chip-wide TCC counters during one rank's kernel = the whole step (every rank's kernel spans it); traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024; "
                   "counted: tests/devsim traffic (sc_traffic), S = 256 MiB, 8 ranks", "forms": out}, open(f"{O}/pmc_forms.json", "w"), indent=1)
PY
du -sh $O
