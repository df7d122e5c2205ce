#!/bin/bash
# where the LL agent stops paying: blocking allreduce at 1 ... 32 KiB in steps of two, by the agent (up to 32 KiB) and launched
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=$ROOT/gpurun_out/agent_limit
rm -rf $O; mkdir -p $O
cd $ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 XMPI_TIMEOUT_S=60 XMPI_NGPUS=1 XMPI_LL_BYTES=32768
BIN=$ROOT/mpi_amd/bin
port=7800
for N in 2 8; do
  for AG in 32768 0 32768 0; do
    port=$((port + 20))
    F=$O/limit_${N}proc_agent${AG}_$port
    XMPI_AGENT_LL_BYTES=$AG XMPI_BASEPORT=$port timeout 200 $BIN/xmpirun $N $BIN/coll_sweep 32768 300 2 > $F.json 2> $F.err
    python - <<PY
import json
try:
    row = json.loads(open("$F.json").read().strip().split("\n")[-1])
    print("N=$N agent_ll_bytes=$AG exact", row.get("exact"), " ".join(f"{r['bytes']}B:{r['queued_us']:.1f}/{r['blocking_us']:.1f}/{r['host_slices_blocking_us']:.1f}" for r in row["rows"]))
except Exception as e:
    print("  unreadable:", e); print(open("$F.err").read()[-1500:])
PY
  done
done
