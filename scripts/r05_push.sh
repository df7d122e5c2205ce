#!/bin/bash
# Round 5, first GPU call: the GPU suite at HEAD (the stepped kernels' push forms beside the pull forms, bit for bit), then the
# production layout (one process per rank) with both forms of the ring and halving kernels by name at 256 MiB / 16 MiB / 1 MiB,
# cfg 5 and cfg 3 with the push forms, the ring in both forms under rocprofv3 --kernel-trace --stats, and the N = 1 line.
# -> gpurun_out/r05_push/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05_push
rm -rf $O; mkdir -p $O
BIN=$GRAFT_REPO_ROOT/mpi_amd/bin
(time timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > $O/gpusuite.log 2>&1
tail -6 $O/gpusuite.log
timeout 120 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
export XMPI_TIMEOUT_S=40 XMPI_NGPUS=1
PROD="$BIN/xmpirun 8 $BIN/allreduce_bench 268435456 20 5"
cd /tmp
XMPI_BASEPORT=7100 timeout 300 $PROD auto split ring ring_push rhd rhd_push zpush > $O/prod_8proc_256MiB.json 2> $O/prod.err
XMPI_BASEPORT=7120 timeout 200 $BIN/xmpirun 8 $BIN/allreduce_bench 16777216 50 5 auto ring ring_push rhd rhd_push > $O/prod_8proc_16MiB.json 2>> $O/prod.err
XMPI_BASEPORT=7150 timeout 200 $BIN/xmpirun 8 $BIN/allreduce_bench 1048576 200 10 auto ring ring_push rhd rhd_push > $O/prod_8proc_1MiB.json 2>> $O/prod.err
XMPI_BASEPORT=7170 timeout 200 $BIN/xmpirun 2 $BIN/allreduce_bench 268435456 20 5 auto ring ring_push rhd rhd_push > $O/prod_2proc_256MiB.json 2>> $O/prod.err
XMPI_BASEPORT=7180 timeout 200 $BIN/xmpirun 4 $BIN/allreduce_bench 268435456 20 5 auto ring ring_push rhd rhd_push > $O/prod_4proc_256MiB.json 2>> $O/prod.err
XMPI_BASEPORT=7190 timeout 300 $BIN/xmpirun 8 $BIN/cfg5_sweep 1073741824 5 > $O/cfg5_8proc.json 2>> $O/prod.err
XMPI_BASEPORT=7195 timeout 200 $BIN/xmpirun 4 $BIN/cfg3_allgather 2097152 20 > $O/cfg3_4proc.json 2>> $O/prod.err
for m in ring ring_push; do
  XMPI_BASEPORT=7200 timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_prod_$m -- $PROD $m > $O/prod_${m}_under_rocprof.json 2> $O/stats_prod_$m.err
done
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
cp bench_extras.json $O/bench_n1_extras.json
find $O -name "*.csv" ! -name "*kernel_stats.csv" -delete; find $O -name "*.db" -delete
cut -c1-300 $O/bench_n1.json; echo; python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r05_push"
for f in sorted(glob.glob(O + "/prod_*proc_*.json")):
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
        print(os.path.basename(f), "exact", d.get("exact"), {r["mode"]: round(r["us_per_step"], 1) for r in d["rows"]})
    except Exception as e:
        print(f, "??", e)
PY
du -sh $O
