#!/bin/bash
# Round 6: xmpi_tune with every candidate's ANSWER checked before its time is believed -- the GPU tests of the tuner, what the check
# costs (8 processes to 256 MiB, three runs; tune_us / tune_check_us), the self-check forced on one GPU (XMPI_SELFCHECK=1)
# -> gpurun_out/r06_tune/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06_tune
rm -rf $O; mkdir -p $O
BIN=$GRAFT_REPO_ROOT/mpi_amd/bin
(time timeout 900 python -m pytest tests -m gpu -x -q -k "tuner or selfcheck or degraded or smoke" 2>&1 | tail -15) > $O/pytest_tune.log 2>&1
tail -5 $O/pytest_tune.log
export XMPI_TIMEOUT_S=40 XMPI_NGPUS=1
for i in 1 2 3; do
XMPI_BASEPORT=$((7100 + i * 20)) timeout 300 $BIN/xmpirun 8 $BIN/allreduce_bench 268435456 5 2 auto > $O/prod_tune_$i.json 2>> $O/prod.err
done
XMPI_SELFCHECK=1 XMPI_BASEPORT=7300 timeout 300 $BIN/xmpirun 8 $BIN/allreduce_bench 1048576 20 5 auto fused split > $O/prod_selfcheck_8.json 2>> $O/prod.err
XMPI_SELFCHECK=1 XMPI_BASEPORT=7320 timeout 300 $BIN/xmpirun 2 $BIN/allreduce_bench 1048576 20 5 auto fused split > $O/prod_selfcheck_2.json 2>> $O/prod.err
python - <<'PY'
import json, glob, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06_tune"
for f in sorted(glob.glob(O + "/prod_*.json")):
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    print(os.path.basename(f), d.get("exact"), {k: d.get(k) for k in ("tune_rejected", "tune_check_ms", "init_selfcheck_us", "degraded")}, [(r["mode"], round(r["us_per_step"], 1), r.get("tuned")) for r in d["rows"]])
PY
cat $O/prod.err | head -20
