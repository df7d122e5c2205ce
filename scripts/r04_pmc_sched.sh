#!/bin/bash
# HBM traffic of the stepped ring / halving kernels from the PMC counters (separate FETCH_SIZE / WRITE_SIZE passes, --kernel-trace only),
# 8 processes on the one GPU, 256 MiB f32 -- to set beside what tests/devsim counted for the same kernels (4.375 S per device: 35 S).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 XMPI_TIMEOUT_S=30 XMPI_NGPUS=1
O=$GRAFT_REPO_ROOT/gpurun_out/r04_pmc_sched
rm -rf $O; mkdir -p $O
BIN=$GRAFT_REPO_ROOT/mpi_amd/bin
PRODS="$BIN/xmpirun 8 $BIN/allreduce_bench 268435456 5 2"
cd /tmp
XMPI_BASEPORT=7300 timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $PRODS ring rhd > $O/under_pmc_fetch.json 2> $O/fetch.err; echo "fetch rc=$?"
XMPI_BASEPORT=7350 timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $PRODS ring rhd > $O/under_pmc_write.json 2> $O/write.err; echo "write rc=$?"
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py $O/fetch $O/write dsync_sched > $O/pmc_sched.json
find $O -name "*.csv" -delete; find $O -name "*.db" -delete
cat $O/pmc_sched.json | head -60
