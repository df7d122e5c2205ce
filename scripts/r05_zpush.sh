#!/bin/bash
# push-only allreduce as two kernels through the communicators' own blocks: parity tests that draw it, timings beside the fold -> gpurun_out/r05_zpush/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05_zpush
rm -rf $O; mkdir -p $O
BIN=$GRAFT_REPO_ROOT/mpi_amd/bin
(time timeout 900 python -m pytest tests/test_gpu_collectives.py -k "allreduce_small or soak or stepped or zero_copy" -x -q 2>&1 | tail -8) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 120 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
export XMPI_TIMEOUT_S=40 XMPI_NGPUS=1
cd /tmp
for n in 8 4 2; do
XMPI_BASEPORT=7100 timeout 300 $BIN/xmpirun $n $BIN/allreduce_bench 268435456 20 5 auto split zpush > $O/prod_${n}proc_256MiB.json 2>> $O/prod.err
done
XMPI_BASEPORT=7120 timeout 200 $BIN/xmpirun 8 $BIN/allreduce_bench 16777216 50 5 auto zpush > $O/prod_8proc_16MiB.json 2>> $O/prod.err
XMPI_BASEPORT=7150 timeout 200 $BIN/xmpirun 8 $BIN/allreduce_bench 1048576 200 10 auto zpush > $O/prod_8proc_1MiB.json 2>> $O/prod.err
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json, glob, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05_zpush"
for f in sorted(glob.glob(O + "/prod_*.json")):
    d = json.loads(open(f).read().strip().split("\n")[-1])
    print(os.path.basename(f), d["exact"], {r["mode"]: round(r["us_per_step"], 1) for r in d["rows"]})
PY
tail -3 $O/prod.err
