#!/bin/bash
# quick GPU check of the Send / Receive path: the tests, then the ping-pong with the pre-launched kernel on / off
cd "$(dirname "$0")/.." || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp XMPI_TIMEOUT_S=60 XMPI_NGPUS=1
OUT=gpurun_out/p2p
rm -rf $OUT; mkdir -p $OUT
BIN=mpi_amd/bin
timeout 200 python -m pytest tests/test_gpu_collectives.py -x -q -k "bounce or p2p_semantics or helloworld or send_recv" > $OUT/tests.txt 2>&1; echo "tests rc=$?" >> $OUT/tests.txt
for us in 40 0 100; do
  XMPI_P2P_AGENT_US=$us XMPI_BASEPORT=$((7100 + us)) timeout 50 $BIN/xmpirun 2 $BIN/allreduce_bench $((16<<20)) 5 2 fused > $OUT/bounce_agent_$us.txt 2>&1
done
XMPI_P2P_KERNEL_ACK=0 XMPI_BASEPORT=7400 timeout 50 $BIN/xmpirun 2 $BIN/allreduce_bench $((16<<20)) 5 2 fused > $OUT/bounce_round2_path.txt 2>&1
tail -n 3 $OUT/tests.txt
for f in $OUT/bounce_*.txt; do echo $f; python3 -c "
import json,sys
for ln in open('$f'):
    if ln.startswith('{'):
        d=json.loads(ln); print('  ', [(b['bytes'], b['half_round_trip_us']) for b in d['bounce']])
"; done
