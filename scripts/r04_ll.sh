#!/bin/bash
# LL small collectives on the GPU: the parity tests, then what a small allreduce costs with and without them
# (examples/coll_sweep: blocking and enqueued, 2 and 8 processes on the one GPU).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=$ROOT/gpurun_out/ll
mkdir -p $O
cd $ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 XMPI_TIMEOUT_S=60 XMPI_NGPUS=1
BIN=$ROOT/mpi_amd/bin
if [ "${1:-tests}" = tests ]; then
  timeout 1500 python -m pytest tests/test_gpu_collectives.py -k "ll_" -x -q > $O/pytest_ll.log 2>&1; echo "pytest ll: rc=$?"
  tail -n 30 $O/pytest_ll.log
fi
port=7500
for N in 2 8; do
  for LL in 0 32768; do
    port=$((port + 20))
    XMPI_LL_BYTES=$LL XMPI_BASEPORT=$port timeout 300 $BIN/xmpirun $N $BIN/coll_sweep 1048576 200 > $O/coll_sweep_${N}proc_ll$LL.json 2> $O/coll_sweep_${N}proc_ll$LL.err
    echo "coll_sweep N=$N XMPI_LL_BYTES=$LL rc=$?"
    python - <<PY
import json
try:
    row = json.loads(open("$O/coll_sweep_${N}proc_ll$LL.json").read().strip().split("\n")[-1])
    print("  exact", row.get("exact"), " ".join(f"{r['bytes']}B:{r['queued_us']:.1f}/{r['blocking_us']:.1f}" for r in row["rows"]))
except Exception as e:
    print("  unreadable:", e); print(open("$O/coll_sweep_${N}proc_ll$LL.err").read()[-1500:])
PY
  done
done
