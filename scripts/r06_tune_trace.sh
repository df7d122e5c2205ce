cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 XMPI_TIMEOUT_S=60 XMPI_NGPUS=1 XMPI_TRACE=1
BIN=$GRAFT_REPO_ROOT/mpi_amd/bin
mkdir -p gpurun_out/r06_tune_trace; rm -f gpurun_out/r06_tune_trace/*
for i in 1 2; do
XMPI_BASEPORT=7100 timeout 300 $BIN/xmpirun 8 $BIN/allreduce_bench 268435456 3 1 auto > gpurun_out/r06_tune_trace/run_$i.log 2>&1
python - <<PY
import re
txt = open("gpurun_out/r06_tune_trace/run_$i.log").read()
slow = [ln for ln in txt.splitlines() if "tune:   " in ln and float(re.search(r"all ([0-9.]+) ms", ln).group(1)) > 1000]
print("run $i", "slow candidates:", len(slow))
for ln in slow[:40]: print("   ", ln[:260])
import os
if not slow: os.remove("gpurun_out/r06_tune_trace/run_$i.log")
PY
done
