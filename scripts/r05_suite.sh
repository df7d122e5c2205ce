#!/bin/bash
# the GPU suite + smoke + the N = 1 line at HEAD -> gpurun_out/r05_suite/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05_suite
rm -rf $O; mkdir -p $O
(time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > $O/gpusuite.log 2>&1
tail -6 $O/gpusuite.log
timeout 120 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
cp bench_extras.json $O/bench_n1_extras.json
cut -c1-400 $O/bench_n1.json; echo
ls gpurun_out/fail_* 2>/dev/null
