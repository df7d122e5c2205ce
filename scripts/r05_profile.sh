#!/bin/bash
# Round 5: the profile call at HEAD -> gpurun_out/r05/ (scripts/r05_collect.py copies the summaries into profiles/r05/).
#  1. the GPU suite, smoke, the N = 1 line (plain and under rocprofv3 --kernel-trace --stats), its PMC passes;
#  2. the production layout (one process per rank): every schedule by name -- the stepped kernels in their pull AND push forms --
#     at 256 / 16 / 1 MiB with 8 processes, 256 MiB with 2 and 4; cfg 3, cfg 5;
#  3. the ring kernel in both forms under rocprofv3 --kernel-trace --stats (per rank) and under --pmc FETCH_SIZE / WRITE_SIZE
#     (separate passes, --kernel-trace only): the push form's HBM traffic beside the pull form's;
#  4. small collectives, 2 and 8 processes (LL lines, the LL agent) with the device side sized for 8 ranks.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05
rm -rf $O; mkdir -p $O
BIN=$GRAFT_REPO_ROOT/mpi_amd/bin
(time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > $O/gpusuite.log 2>&1
tail -6 $O/gpusuite.log
timeout 120 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
cp bench_extras.json $O/bench_n1_extras.json
export XMPI_TIMEOUT_S=40 XMPI_NGPUS=1
B="python $GRAFT_REPO_ROOT/bench.py --algo zcopy --no-extras --no-cpu --no-production"
PROD="$BIN/xmpirun 8 $BIN/allreduce_bench 268435456 20 5"
PRODS="$BIN/xmpirun 8 $BIN/allreduce_bench 268435456 5 2"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_n1 -- $B > $O/bench_zcopy_under_rocprof.json 2> $O/stats_n1.err
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_n1_fetch -- $B --steps 5 --warmup 2 > /dev/null 2> $O/pmc_n1_fetch.err
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_n1_write -- $B --steps 5 --warmup 2 > /dev/null 2> $O/pmc_n1_write.err
XMPI_BASEPORT=7100 timeout 300 $PROD auto fused split zpush ring ring_push rhd rhd_push > $O/prod_8proc_256MiB.json 2> $O/prod.err
XMPI_BASEPORT=7120 timeout 200 $BIN/xmpirun 8 $BIN/allreduce_bench 16777216 50 5 auto ring ring_push rhd rhd_push > $O/prod_8proc_16MiB.json 2>> $O/prod.err
XMPI_BASEPORT=7150 timeout 200 $BIN/xmpirun 8 $BIN/allreduce_bench 1048576 200 10 auto ring ring_push rhd rhd_push > $O/prod_8proc_1MiB.json 2>> $O/prod.err
XMPI_BASEPORT=7170 timeout 200 $BIN/xmpirun 2 $BIN/allreduce_bench 268435456 20 5 auto ring ring_push rhd rhd_push > $O/prod_2proc_256MiB.json 2>> $O/prod.err
XMPI_BASEPORT=7180 timeout 200 $BIN/xmpirun 4 $BIN/allreduce_bench 268435456 20 5 auto ring ring_push rhd rhd_push > $O/prod_4proc_256MiB.json 2>> $O/prod.err
XMPI_BASEPORT=7190 timeout 300 $BIN/xmpirun 8 $BIN/cfg5_sweep 1073741824 5 > $O/cfg5_8proc.json 2>> $O/prod.err
XMPI_BASEPORT=7195 timeout 200 $BIN/xmpirun 4 $BIN/cfg3_allgather 2097152 20 > $O/cfg3_4proc.json 2>> $O/prod.err
for m in ring ring_push; do
  XMPI_BASEPORT=7200 timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_prod_$m -- $PROD $m > $O/prod_${m}_under_rocprof.json 2> $O/stats_prod_$m.err
done
XMPI_BASEPORT=7300 timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $PRODS ring ring_push rhd rhd_push > $O/under_pmc_fetch.json 2> $O/pmc_fetch.err; echo "fetch rc=$?"
XMPI_BASEPORT=7350 timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $PRODS ring ring_push rhd rhd_push > $O/under_pmc_write.json 2> $O/pmc_write.err; echo "write rc=$?"
XMPI_BASEPORT=7400 timeout 200 $BIN/xmpirun 2 $BIN/coll_sweep 1048576 300 > $O/coll_sweep_2proc.json 2>> $O/prod.err
XMPI_BASEPORT=7450 timeout 200 $BIN/xmpirun 8 $BIN/coll_sweep 1048576 300 > $O/coll_sweep_8proc.json 2>> $O/prod.err
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py $O/pmc_n1_fetch $O/pmc_n1_write reduce_n_multi > $O/pmc_bench_zcopy.json
python scripts/pmc_summary.py $O/pmc_fetch $O/pmc_write dsync_sched > $O/pmc_sched_8proc.json
find $O -name "*.csv" ! -name "*kernel_stats.csv" -delete; find $O -name "*.db" -delete
cut -c1-300 $O/bench_n1.json; echo; python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r05"
for f in sorted(glob.glob(O + "/prod_*proc_*.json")):
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
        print(os.path.basename(f), "exact", d.get("exact"), {r["mode"]: round(r["us_per_step"], 1) for r in d["rows"]})
    except Exception as e:
        print(f, "??", e)
print(open(O + "/pmc_sched_8proc.json").read()[:1500])
PY
du -sh $O
