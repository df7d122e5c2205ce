#!/bin/bash
# Round 6, first call: the GPU suite + smoke + the N = 1 line at the round's starting HEAD, and what xmpi_tune costs today
# (8 processes, to 256 MiB) -> gpurun_out/r06_baseline/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06_baseline
rm -rf $O; mkdir -p $O
BIN=$GRAFT_REPO_ROOT/mpi_amd/bin
(time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > $O/gpusuite.log 2>&1
tail -6 $O/gpusuite.log
timeout 120 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
cp bench_extras.json $O/bench_n1_extras.json
cut -c1-600 $O/bench_n1.json; echo
export XMPI_TIMEOUT_S=40 XMPI_NGPUS=1
for i in 1 2 3; do
XMPI_BASEPORT=$((7100 + i * 20)) timeout 300 $BIN/xmpirun 8 $BIN/allreduce_bench 268435456 5 2 auto > $O/prod_tune_$i.json 2>> $O/prod.err
done
python - <<'PY'
import json, glob, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06_baseline"
for f in sorted(glob.glob(O + "/prod_tune_*.json")):
    d = json.loads(open(f).read().strip().split("\n")[-1])
    print(os.path.basename(f), d.get("exact"), [(r["mode"], round(r["us_per_step"], 1), r.get("tuned")) for r in d["rows"]])
PY
