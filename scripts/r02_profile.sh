# Round-2 evidence on one MI355X (run through gpurun from the repo root):
#   1. the default bench line (N=1: 8 ranks as threads of one process, zero-copy, ranks meet on the host)
#   2. rocprofv3 kernel stats of the same command, --algo zcopy (the dominant kernel; must agree with roofline.avg_launch_us)
#   3. one process per rank (the production layout): examples/coll_sweep under the launcher, plain and under
#      rocprofv3 --kernel-trace --stats (one dsync_fold_kernel launch per rank per collective)
#   4. PMC traffic of the zero-copy kernel (separate --pmc passes, kernel-trace only)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r02
rm -rf $O; mkdir -p $O
timeout 600 python bench.py > $O/bench_n1_default.json 2> $O/bench_n1_default.err
tail -c 600 $O/bench_n1_default.err
B="python $GRAFT_REPO_ROOT/bench.py --algo zcopy --no-extras --no-cpu"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/bench_zcopy_under_rocprof.json 2> $O/stats.err
RUN="$GRAFT_REPO_ROOT/mpi_amd/bin/xmpirun 8 $GRAFT_REPO_ROOT/mpi_amd/bin/coll_sweep 16777216 200"
XMPI_TIMEOUT_S=60 timeout 200 $RUN > $O/coll_sweep_8proc.json 2> $O/coll_sweep_8proc.err
XMPI_TIMEOUT_S=60 timeout 200 $GRAFT_REPO_ROOT/mpi_amd/bin/xmpirun 2 $GRAFT_REPO_ROOT/mpi_amd/bin/coll_sweep 16777216 200 > $O/coll_sweep_2proc.json 2>> $O/coll_sweep_8proc.err
XMPI_TIMEOUT_S=60 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dsync_stats -- $RUN > $O/coll_sweep_8proc_under_rocprof.json 2> $O/dsync_stats.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $B --steps 5 > /dev/null 2> $O/fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $B --steps 5 > /dev/null 2> $O/write.err
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py $O/fetch $O/write reduce_n_multi copy_multi > $O/pmc_bench_zcopy.json
find $O -name "*kernel_stats.csv" | head -20; find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
du -sh $O
