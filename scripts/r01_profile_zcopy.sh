set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 400 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 400 gpurun_out/bench_default.err
B="python $GRAFT_REPO_ROOT/bench.py --algo zcopy --no-extras --no-cpu"
O=$GRAFT_REPO_ROOT/gpurun_out/zcprof
rm -rf $O; mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/bench_under_rocprof.json 2> $O/stats.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $B --steps 5 > /dev/null 2> $O/fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $B --steps 5 > /dev/null 2> $O/write.err
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py $O/fetch $O/write reduce_n_multi copy_multi > $O/pmc_bench_zcopy.json
find $O -name "*kernel_stats.csv" | head; find $O -name "*.csv" -size +2M -delete
du -sh $O
