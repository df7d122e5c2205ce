#!/bin/bash
# Round 4, last GPU call: the GPU suite at HEAD, the N = 1 line, the same command under rocprofv3 --kernel-trace --stats, and the
# production layout (one process per rank) with every schedule by name -- the ring / halving kernels after their workers were
# renumbered through the channels (sched.hip) -- plain and, for the ring, under rocprofv3.  -> gpurun_out/r04_final/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r04_final
rm -rf $O; mkdir -p $O
BIN=$GRAFT_REPO_ROOT/mpi_amd/bin
(time timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > $O/gpusuite.log 2>&1
tail -4 $O/gpusuite.log
timeout 120 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
cp bench_extras.json $O/bench_n1_extras.json
B="python $GRAFT_REPO_ROOT/bench.py --algo zcopy --no-extras --no-cpu --no-production"
PROD="$BIN/xmpirun 8 $BIN/allreduce_bench 268435456 20 5"
export XMPI_TIMEOUT_S=40 XMPI_NGPUS=1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_n1 -- $B > $O/bench_zcopy_under_rocprof.json 2> $O/stats_n1.err
XMPI_BASEPORT=7100 timeout 200 $PROD auto fused split ring rhd > $O/prod_8proc_256MiB.json 2> $O/prod.err
XMPI_BASEPORT=7150 timeout 200 $BIN/xmpirun 8 $BIN/allreduce_bench 1048576 200 10 auto ring rhd > $O/prod_8proc_1MiB.json 2>> $O/prod.err
XMPI_BASEPORT=7200 timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_prod_ring -- $PROD ring > $O/prod_ring_under_rocprof.json 2> $O/stats_prod_ring.err
# the LL agent under the profiler: a handful of ll_agent_kernel launches serve the blocking small collectives, the enqueued ones are ll_reduce_kernel launches
XMPI_BASEPORT=7250 XMPI_LL_BYTES=32768 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_agent -- $BIN/xmpirun 2 $BIN/coll_sweep 65536 300 > $O/coll_sweep_2proc_under_rocprof.json 2> $O/stats_agent.err
XMPI_BASEPORT=7300 timeout 200 $BIN/xmpirun 2 $BIN/coll_sweep 1048576 300 > $O/coll_sweep_2proc.json 2>> $O/prod.err
XMPI_BASEPORT=7350 timeout 200 $BIN/xmpirun 8 $BIN/coll_sweep 1048576 300 > $O/coll_sweep_8proc.json 2>> $O/prod.err
cd $GRAFT_REPO_ROOT
find $O -name "*.csv" ! -name "*kernel_stats.csv" -delete; find $O -name "*.db" -delete
cut -c1-400 $O/bench_n1.json; echo; cut -c1-600 $O/prod_8proc_256MiB.json; echo; du -sh $O
