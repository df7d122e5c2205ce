#!/bin/bash
# ranks in different calls, on the GPU -> gpurun_out/r05_mismatch/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05_mismatch
rm -rf $O; mkdir -p $O
(time timeout 600 python -m pytest tests/test_gpu_collectives.py -m gpu -x -q -k "different_calls or dies" 2>&1 | tail -8) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
ls gpurun_out/fail_* 2>/dev/null; true
