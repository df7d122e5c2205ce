#!/bin/bash
# The LL agent (ll.hip ll_agent_kernel) on the GPU: the LL parity tests (they drive the agent with every dtype / operation /
# collective), then what a BLOCKING small allreduce costs with and without it (examples/coll_sweep, 2 and 8 processes on the one
# GPU; the enqueued figure beside it).  -> gpurun_out/agent/
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=$ROOT/gpurun_out/agent
rm -rf $O; mkdir -p $O
cd $ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 XMPI_TIMEOUT_S=60 XMPI_NGPUS=1
BIN=$ROOT/mpi_amd/bin
if [ "${1:-tests}" = tests ]; then
  (time timeout 900 python -m pytest tests/test_gpu_collectives.py -k "ll_" -x -q) > $O/pytest_ll.log 2>&1; echo "pytest ll: rc=$?"
  tail -n 25 $O/pytest_ll.log
fi
port=7500
for N in 2 8; do
  for AG in 1 0 1 0; do
    port=$((port + 20))
    F=$O/coll_sweep_${N}proc_agent${AG}_$port
    XMPI_AGENT_LL=$AG XMPI_LL_BYTES=32768 XMPI_BASEPORT=$port timeout 200 $BIN/xmpirun $N $BIN/coll_sweep 65536 300 > $F.json 2> $F.err
    echo "coll_sweep N=$N XMPI_AGENT_LL=$AG rc=$?"
    python - <<PY
import json
try:
    row = json.loads(open("$F.json").read().strip().split("\n")[-1])
    print("  exact", row.get("exact"), "agent ran", row.get("run_by_the_ll_agent"), "of", row.get("ll_collectives"), "launches", row.get("ll_agent_launches"),
          " ".join(f"{r['bytes']}B:{r['queued_us']:.1f}/{r['blocking_us']:.1f}" for r in row["rows"]))
except Exception as e:
    print("  unreadable:", e); print(open("$F.err").read()[-1500:])
PY
  done
done
# the agent's payload limit: blocking figures with everything up to the slot limit handed to it
for N in 2; do
  port=$((port + 20))
  F=$O/coll_sweep_${N}proc_agent_32k
  XMPI_AGENT_LL=1 XMPI_AGENT_LL_BYTES=32768 XMPI_LL_BYTES=32768 XMPI_BASEPORT=$port timeout 200 $BIN/xmpirun $N $BIN/coll_sweep 65536 300 > $F.json 2> $F.err
  echo "coll_sweep N=$N agent up to 32 KiB rc=$?"; cut -c1-900 $F.json
done
# Send / Receive beside it (the receive agent is untouched: its figures must be what they were)
XMPI_BASEPORT=7900 timeout 120 $BIN/xmpirun 2 $BIN/bounce > $O/bounce.txt 2> $O/bounce.err; tail -12 $O/bounce.txt
