#!/bin/bash
# Round 6: the whole GPU suite with xmpi_init's self-check FORCED on (XMPI_SELFCHECK=1) -- what every init does on a node whose ranks
# sit on different GPUs, and the 1-GPU box leaves off by default -> gpurun_out/r06_suite_selfcheck/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 XMPI_SELFCHECK=1
O=$GRAFT_REPO_ROOT/gpurun_out/r06_suite_selfcheck
rm -rf $O; mkdir -p $O
(time timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40) > $O/gpusuite_selfcheck_on.log 2>&1
tail -12 $O/gpusuite_selfcheck_on.log
ls gpurun_out/fail_* 2>/dev/null | head
