#!/bin/bash
# the LL agent's patience against the caller's own work between two blocking collectives (COLL_SWEEP_THINK_US, not part of the figure)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=$ROOT/gpurun_out/agent_patience
rm -rf $O; mkdir -p $O
cd $ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 XMPI_TIMEOUT_S=60 XMPI_NGPUS=1
BIN=$ROOT/mpi_amd/bin
port=8100
for THINK in 0 20 100 500; do
  for PAT in 40 200 1000; do
    port=$((port + 20))
    F=$O/think${THINK}_patience${PAT}
    COLL_SWEEP_THINK_US=$THINK XMPI_LL_AGENT_US=$PAT XMPI_BASEPORT=$port timeout 100 $BIN/xmpirun 2 $BIN/coll_sweep 4096 200 4 > $F.json 2> $F.err
    python - <<PY
import json
try:
    row = json.loads(open("$F.json").read().strip().split("\n")[-1])
    print("think $THINK us, patience $PAT us: exact", row.get("exact"), "agent launches", row.get("ll_agent_launches"), " ".join(f"{r['bytes']}B:{r['blocking_us']:.1f}" for r in row["rows"]))
except Exception as e:
    print("  unreadable:", e); print(open("$F.err").read()[-800:])
PY
  done
done
