#!/bin/bash
# Round 6: the profile call at HEAD -> gpurun_out/r06/ (scripts/r06_collect.py copies the summaries into profiles/r06/ and rewrites
# profiles/pmc_traffic.json):
#  1. the GPU suite, smoke, the N = 1 line with the new fields (box_copy_us, frac_of_box, kernel_us_min / _max, frac_of_achievable,
#     cpu_baseline.by_ranks), plain and under rocprofv3 --kernel-trace --stats (bench --algo zcopy: the fold AND the box copy behind it);
#  2. its PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs, --kernel-trace only) -- the fold and the copy with the fold's pattern;
#  3. the production layout: xmpi_tune with every candidate's answer checked (tune_ms, tune_check_ms, tune_rejected), the split and
#     the one-kernel fold under PMC (one process per rank, 8 on the GPU).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06
rm -rf $O; mkdir -p $O
BIN=$GRAFT_REPO_ROOT/mpi_amd/bin
if [ -z "$R06_SKIP_SUITE" ]; then
(time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > $O/gpusuite.log 2>&1
tail -6 $O/gpusuite.log
timeout 120 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
fi
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
cp bench_extras.json $O/bench_n1_extras.json
export XMPI_TIMEOUT_S=40 XMPI_NGPUS=1
B="python $GRAFT_REPO_ROOT/bench.py --algo zcopy --no-extras --no-cpu --no-production"
PRODS="$BIN/xmpirun 8 $BIN/allreduce_bench 268435456 5 2"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_n1 -- $B > $O/bench_zcopy_under_rocprof.json 2> $O/stats_n1.err
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_n1_fetch -- $B --steps 5 --warmup 2 > /dev/null 2> $O/pmc_n1_fetch.err
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_n1_write -- $B --steps 5 --warmup 2 > /dev/null 2> $O/pmc_n1_write.err
for i in 1 2 3; do
XMPI_BASEPORT=$((7100 + i * 20)) timeout 300 $BIN/xmpirun 8 $BIN/allreduce_bench 268435456 20 5 auto > $O/prod_tune_$i.json 2>> $O/prod.err
done
XMPI_BASEPORT=7500 timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $PRODS split fused > $O/under_pmc_fetch.json 2> $O/pmc_fetch.err; echo "fetch rc=$?"
XMPI_BASEPORT=7520 timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $PRODS split fused > $O/under_pmc_write.json 2> $O/pmc_write.err; echo "write rc=$?"
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py $O/pmc_n1_fetch $O/pmc_n1_write reduce_n_multi copy_pairs > $O/pmc_bench_zcopy.json
python scripts/pmc_summary.py $O/pmc_fetch $O/pmc_write dsync_ > $O/pmc_prod_8proc.json
find $O -name "*.csv" ! -name "*kernel_stats.csv" -delete; find $O -name "*.db" -delete
python scripts/show_bench.py $O/bench_n1.json | head -6
python - <<'PY'
import json, glob, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06"
d = json.loads(open(O + "/bench_n1.json").read().strip().split("\n")[-1])
print("line bytes", len(open(O + "/bench_n1.json").read().strip().split("\n")[-1]))
print("roofline", {k: d["roofline"].get(k) for k in ("avg_launch_us", "kernel_us_min", "kernel_us_max", "frac", "frac_of_achievable", "box_copy_us", "frac_of_box", "traffic")})
print("cpu by_ranks", d["cpu_baseline"].get("by_ranks"))
for f in sorted(glob.glob(O + "/prod_tune_*.json")):
    x = json.loads(open(f).read().strip().split("\n")[-1])
    print(os.path.basename(f), x.get("exact"), {k: x.get(k) for k in ("tune_rejected", "tune_check_ms", "degraded")}, [(r["mode"], round(r["us_per_step"], 1), r.get("tuned")) for r in x["rows"]])
for f in ("pmc_bench_zcopy.json", "pmc_prod_8proc.json"):
    for r in json.load(open(O + "/" + f)):
        print(f, r["kernel"], r["grid_threads"], r["launches"], round(r["traffic_bytes_per_launch"] / 1e9, 4), "GB")
PY
du -sh $O
