#!/bin/bash
# Round-4 GPU session 2 (one MI355X, through gpurun from the repo root) -> gpurun_out/r04b/:
#   1. the split form's XCD guard + the system-scope data kernel: parity tests, then what the fallback costs on one GPU
#      (8 and 2 processes, 256 MiB and 16 MiB per rank)
#   2. the pull kernel's grid when sender and receiver share the GPU (XMPI_P2P_GRID_CAP sweep at 16 MiB)
#   3. the 750-vs-686 us question: rocprofv3 --kernel-trace of the N = 1 command, every dispatch in order
#   4. roctx ranges: rocprofv3 --marker-trace --kernel-trace of coll_sweep (2 processes)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 XMPI_TIMEOUT_S=60 XMPI_NGPUS=1
O=$GRAFT_REPO_ROOT/gpurun_out/r04b
rm -rf $O; mkdir -p $O
BIN=$GRAFT_REPO_ROOT/mpi_amd/bin
timeout 900 python -m pytest tests/test_gpu_collectives.py -k "split_form" -x -q > $O/pytest_split.log 2>&1; echo "pytest split rc=$?"
tail -n 15 $O/pytest_split.log
port=7100
for N in 8 2; do for SZ in 268435456 16777216; do for SYS in 0 1; do
  port=$((port + 20))
  XMPI_BODY_SYS=$SYS XMPI_BASEPORT=$port timeout 200 $BIN/xmpirun $N $BIN/allreduce_bench $SZ 20 5 split fused > $O/prod_${N}proc_${SZ}_sys$SYS.json 2> $O/prod_${N}proc_${SZ}_sys$SYS.err
  echo "N=$N SZ=$SZ BODY_SYS=$SYS rc=$?"
done; done; done
for CAP in 32 64 128 256 512 1024; do
  port=$((port + 20))
  XMPI_P2P_GRID_CAP=$CAP XMPI_BASEPORT=$port timeout 100 $BIN/xmpirun 2 $BIN/allreduce_bench 16777216 5 2 fused > $O/bounce_cap$CAP.json 2> $O/bounce_cap$CAP.err
done
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --algo zcopy --no-extras --no-cpu --no-production"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_n1 -- $B > $O/bench_zcopy_under_rocprof.json 2> $O/trace_n1.err
XMPI_BASEPORT=7900 timeout 200 rocprofv3 --marker-trace --kernel-trace --output-format csv -d $O/markers -- $BIN/xmpirun 2 $BIN/coll_sweep 1048576 20 > $O/coll_sweep_markers.json 2> $O/markers.err
echo "marker trace rc=$?"
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r04b")
# 3. every reduce_n_multi dispatch of the N = 1 command, in order: duration and the gap to the one before
rows = []
for f in glob.glob(O + "/trace_n1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "reduce_n_multi" in r.get("Kernel_Name", ""):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
seq = [{"i": i, "us": (e - s) / 1e3, "gap_us": (s - rows[i - 1][1]) / 1e3 if i else None} for i, (s, e) in enumerate(rows)]
json.dump(seq, open(O + "/n1_dispatch_sequence.json", "w"), indent=0)
print("dispatches:", " ".join(f"{x['us']:.0f}" + (f"(+{x['gap_us']:.0f})" if x["gap_us"] is not None else "") for x in seq))
# 4. marker names
names = {}
for f in glob.glob(O + "/markers/**/*marker*trace*.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r.get("Function") or r.get("Name") or r.get("Message") or str(r))[:60]
        names[k] = names.get(k, 0) + 1
print("marker ranges:", len(names), list(names.items())[:12])
json.dump(names, open(O + "/marker_names.json", "w"), indent=0)
for f in sorted(glob.glob(O + "/prod_*proc_*_sys*.json")) + sorted(glob.glob(O + "/bounce_cap*.json")):
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
        if "bounce_cap" in f:
            print(os.path.basename(f), {b["bytes"]: b["half_round_trip_us"] for b in d.get("bounce", [])})
        else:
            print(os.path.basename(f), [(r["mode"], round(r["us_per_step"], 1), r.get("kernel_avg_us")) for r in d["rows"]], "exact", d.get("exact"))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e, open(f.replace(".json", ".err")).read()[-600:])
PY
find $O -name "*.csv" ! -name "*kernel_stats.csv" ! -name "*marker*" -size +2M -delete; find $O -name "*.db" -delete
du -sh $O
