#!/bin/bash
# Round 5, the last GPU call at HEAD: the GPU suite, smoke, the N = 1 line, the same command under rocprofv3 --kernel-trace --stats,
# the production layout with every schedule by name (pull and push forms, push-only in two kernels), small collectives.
# -> gpurun_out/r05_final/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05_final
rm -rf $O; mkdir -p $O
BIN=$GRAFT_REPO_ROOT/mpi_amd/bin
(time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > $O/gpusuite.log 2>&1
tail -6 $O/gpusuite.log
timeout 120 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
cp bench_extras.json $O/bench_n1_extras.json
B="python $GRAFT_REPO_ROOT/bench.py --algo zcopy --no-extras --no-cpu --no-production"
export XMPI_TIMEOUT_S=40 XMPI_NGPUS=1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_n1 -- $B > $O/bench_zcopy_under_rocprof.json 2> $O/stats_n1.err
XMPI_BASEPORT=7100 timeout 300 $BIN/xmpirun 8 $BIN/allreduce_bench 268435456 20 5 auto fused split zpush ring ring_push rhd rhd_push > $O/prod_8proc_256MiB.json 2> $O/prod.err
XMPI_BASEPORT=7150 timeout 200 $BIN/xmpirun 8 $BIN/allreduce_bench 1048576 200 10 auto zpush ring ring_push rhd rhd_push > $O/prod_8proc_1MiB.json 2>> $O/prod.err
XMPI_BASEPORT=7170 timeout 200 $BIN/xmpirun 2 $BIN/allreduce_bench 268435456 20 5 auto zpush ring ring_push rhd rhd_push > $O/prod_2proc_256MiB.json 2>> $O/prod.err
XMPI_BASEPORT=7190 timeout 300 $BIN/xmpirun 8 $BIN/cfg5_sweep 1073741824 5 > $O/cfg5_8proc.json 2>> $O/prod.err
XMPI_BASEPORT=7195 timeout 200 $BIN/xmpirun 4 $BIN/cfg3_allgather 2097152 20 > $O/cfg3_4proc.json 2>> $O/prod.err
XMPI_BASEPORT=7400 timeout 200 $BIN/xmpirun 2 $BIN/coll_sweep 1048576 300 > $O/coll_sweep_2proc.json 2>> $O/prod.err
XMPI_BASEPORT=7450 timeout 200 $BIN/xmpirun 8 $BIN/coll_sweep 1048576 300 > $O/coll_sweep_8proc.json 2>> $O/prod.err
cd $GRAFT_REPO_ROOT
find $O -name "*.csv" ! -name "*kernel_stats.csv" -delete; find $O -name "*.db" -delete
n1=$(ls -S $O/stats_n1/*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$n1" ] && cp $n1 $O/bench_zcopy_kernel_stats.csv
python scripts/show_bench.py $O/bench_n1.json | head -8
python - <<'PY'
import json, glob, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05_final"
for f in sorted(glob.glob(O + "/prod_*proc_*.json")):
    d = json.loads(open(f).read().strip().split("\n")[-1])
    print(os.path.basename(f), "exact", d.get("exact"), {r["mode"]: round(r["us_per_step"], 1) for r in d["rows"]})
PY
ls gpurun_out/fail_* 2>/dev/null; du -sh $O
