#!/bin/bash
# bcast and reduce by every name (the fold, push-only, the tree kernel in both forms), 2 and 8 processes on the GPU -> gpurun_out/r05_tree/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05_tree
rm -rf $O; mkdir -p $O
python - <<'PY' > $O/rooted.log 2>&1
from tests.gpu_harness import run_ranks
for size in (2, 8):
    outs = run_ranks("rooted_bench", size, {"sizes": [65536, 1 << 20, 16 << 20, 256 << 20]}, timeout=300)
    print([ln for ln in outs[0].splitlines() if ln.startswith("ROOTED")][0], flush=True)
PY
cat $O/rooted.log
