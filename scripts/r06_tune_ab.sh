# A/B on ONE box: xmpi_tune to 256 MiB, 8 processes, answer check with one pass / two passes, alternating
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 XMPI_TIMEOUT_S=60 XMPI_NGPUS=1
BIN=$GRAFT_REPO_ROOT/mpi_amd/bin
for i in 1 2 3 4; do
 for passes in 1 2; do
  XMPI_CHECK_PASSES=$passes XMPI_BASEPORT=7100 timeout 300 $BIN/xmpirun 8 $BIN/allreduce_bench 268435456 3 1 auto 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('passes $passes tune_ms', d['rows'][0]['tuned']['tune_ms'], 'check_ms', d['tune_check_ms'])"
 done
done
