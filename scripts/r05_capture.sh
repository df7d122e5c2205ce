#!/bin/bash
# the captured forms and the tuner on the GPU -> gpurun_out/r05_capture/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05_capture
rm -rf $O; mkdir -p $O
(time timeout 1200 python -m pytest tests/test_gpu_collectives.py -m gpu -x -q -k "stepped or tuner or stream or soak" 2>&1 | tail -8) > $O/pytest.log 2>&1
tail -6 $O/pytest.log
ls gpurun_out/fail_* 2>/dev/null; true
