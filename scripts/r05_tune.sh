#!/bin/bash
# the tuner over all four collectives on the GPU: its tests, what it costs, what it chose -> gpurun_out/r05_tune/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05_tune
rm -rf $O; mkdir -p $O
(time timeout 900 python -m pytest tests/test_gpu_collectives.py -m gpu -x -q -k "tuner or bcast or reduce" 2>&1 | tail -8) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
BIN=mpi_amd/bin
for n in 2 8; do
  XMPI_BASEPORT=7500 timeout 300 $BIN/xmpirun $n $BIN/allreduce_bench 268435456 10 3 auto > $O/prod_${n}proc.json 2> $O/prod_${n}proc.err
  grep -o '"tuned": {[^}]*}' $O/prod_${n}proc.json | head -1
done
python - <<'PY' > $O/tables.txt 2>&1
import tests.gpu_harness as h
for n in (2, 8):
    out = h.run_ranks("tune", n, {"max_bytes": 64 << 20}, timeout=300)
    print(n, "processes:", out[0] if isinstance(out, list) else out)
PY
cat $O/tables.txt
ls gpurun_out/fail_* 2>/dev/null
