#!/bin/bash
# the push forms with the communicator's own landing block (kept across collectives): the GPU suite's stepped-kernel tests, cfg 5 up
# to 1 GiB and cfg 4 by name -> gpurun_out/r05_land/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05_land
rm -rf $O; mkdir -p $O
BIN=$GRAFT_REPO_ROOT/mpi_amd/bin
(time timeout 900 python -m pytest tests/test_gpu_collectives.py -k "sched" -x -q 2>&1 | tail -15) > $O/pytest_sched.log 2>&1
tail -4 $O/pytest_sched.log
export XMPI_TIMEOUT_S=40 XMPI_NGPUS=1
cd /tmp
XMPI_BASEPORT=7190 timeout 300 $BIN/xmpirun 8 $BIN/cfg5_sweep 1073741824 5 > $O/cfg5_8proc.json 2> $O/prod.err
XMPI_BASEPORT=7100 timeout 300 $BIN/xmpirun 8 $BIN/allreduce_bench 268435456 20 5 ring ring_push rhd rhd_push > $O/prod_8proc_256MiB.json 2>> $O/prod.err
XMPI_BASEPORT=7120 timeout 300 $BIN/xmpirun 8 $BIN/allreduce_bench 1073741824 10 3 ring ring_push rhd rhd_push > $O/prod_8proc_1GiB.json 2>> $O/prod.err
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05_land"
d = json.loads(open(O + "/cfg5_8proc.json").read().strip().split("\n")[-1])
print(d["all_bit_identical"], [(r["bytes"], {k: round(v["us"], 1) for k, v in r.items() if isinstance(v, dict)}) for r in d["rows"]])
for f in ("prod_8proc_256MiB.json", "prod_8proc_1GiB.json"):
    d = json.loads(open(O + "/" + f).read().strip().split("\n")[-1])
    print(f, d["exact"], {r["mode"]: round(r["us_per_step"], 1) for r in d["rows"]})
PY
tail -5 $O/prod.err
