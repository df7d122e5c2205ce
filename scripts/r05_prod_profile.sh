#!/bin/bash
# the production layout's fold (one process per rank, 8 on the one GPU, 256 MiB f32) at HEAD: rocprofv3 --kernel-trace --stats per
# rank for the split and the one-kernel form and push-only, PMC passes (FETCH_SIZE / WRITE_SIZE apart, --kernel-trace only)
# -> gpurun_out/r05_prod/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 XMPI_TIMEOUT_S=40 XMPI_NGPUS=1
O=$GRAFT_REPO_ROOT/gpurun_out/r05_prod
rm -rf $O; mkdir -p $O
BIN=$GRAFT_REPO_ROOT/mpi_amd/bin
PROD="$BIN/xmpirun 8 $BIN/allreduce_bench 268435456 20 5"
PRODS="$BIN/xmpirun 8 $BIN/allreduce_bench 268435456 5 2"
cd /tmp
port=7400
for m in split fused zpush; do
  port=$((port + 20))
  XMPI_BASEPORT=$port timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_prod_$m -- $PROD $m > $O/prod_${m}_under_rocprof.json 2> $O/stats_prod_$m.err
done
XMPI_BASEPORT=7500 timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $PRODS split fused > $O/under_pmc_fetch.json 2> $O/pmc_fetch.err; echo "fetch rc=$?"
XMPI_BASEPORT=7520 timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $PRODS split fused > $O/under_pmc_write.json 2> $O/pmc_write.err; echo "write rc=$?"
XMPI_BASEPORT=7540 timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_zpush_fetch -- $PRODS zpush > /dev/null 2> $O/pmc_zpush_fetch.err
XMPI_BASEPORT=7560 timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_zpush_write -- $PRODS zpush > /dev/null 2> $O/pmc_zpush_write.err
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py $O/pmc_fetch $O/pmc_write dsync_ > $O/pmc_prod_8proc.json
python scripts/pmc_summary.py $O/pmc_zpush_fetch $O/pmc_zpush_write dsync_ > $O/pmc_zpush_8proc.json
find $O -name "*.csv" ! -name "*kernel_stats.csv" -delete; find $O -name "*.db" -delete
python - <<'PY'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05_prod"
for f in ("pmc_prod_8proc.json", "pmc_zpush_8proc.json"):
    for r in json.load(open(O + "/" + f)):
        print(f, r["kernel"], r["grid_threads"], r["launches"], round(r["traffic_bytes_per_launch"] / 1e9, 3), "GB")
PY
du -sh $O
