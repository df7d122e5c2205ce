#!/bin/bash
# Round 6: a peer that dies is an error at once with default settings (the watchdog), no false positive over many lifetimes
# -> gpurun_out/r06_watchdog/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06_watchdog
rm -rf $O; mkdir -p $O
(time timeout 1200 python -m pytest tests -m gpu -x -q -k "peer_that_dies or lifecycle or selfcheck or tuner" 2>&1 | tail -15) > $O/pytest_watchdog.log 2>&1
tail -6 $O/pytest_watchdog.log
timeout 600 python -m pytest tests -m gpu -q -k "peer_that_dies_is_an_error_at_once" -s 2>&1 | grep "ok (error after" > $O/peer_dies_times.log
cat $O/peer_dies_times.log
XMPI_TRACE=1 XMPI_SELFCHECK=1 XMPI_NGPUS=1 XMPI_BASEPORT=7300 timeout 300 mpi_amd/bin/xmpirun 2 mpi_amd/bin/allreduce_bench 1048576 5 2 fused 2>&1 | grep "xmpi 0 " > $O/selfcheck_trace_2proc.log
cat $O/selfcheck_trace_2proc.log
ls gpurun_out/fail_* 2>/dev/null
