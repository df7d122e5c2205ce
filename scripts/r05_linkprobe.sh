#!/bin/bash
# the link probe's three engines between two processes on the one GPU (its HBM stands in for the link) -> gpurun_out/r05_linkprobe/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05_linkprobe
rm -rf $O; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_collectives.py -k "bounce" -x -q 2>&1 | tail -4) > $O/pytest_bounce.log 2>&1; tail -2 $O/pytest_bounce.log
python - <<'PY' > $O/linkprobe.log 2>&1
from tests.gpu_harness import run_ranks
for b in (1 << 20, 16 << 20, 64 << 20, 256 << 20):
    outs = run_ranks("linkprobe", 2, {"bytes": b, "iters": 10}, timeout=300)
    print([ln for ln in outs[0].splitlines() if ln.startswith("LINKPROBE")][0], flush=True)
PY
cat $O/linkprobe.log
