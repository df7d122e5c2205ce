#!/bin/bash
# a two-minute look at the production layout (8 processes on one GPU): every schedule at 256 / 64 / 16 / 1 MiB, the ring
# kernel with 64 and 128 workers per rank
cd "$(dirname "$0")/.." || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp XMPI_TIMEOUT_S=30
OUT=gpurun_out/quick
rm -rf $OUT; mkdir -p $OUT
BIN=mpi_amd/bin
port=7100
go() { name=$1; shift; port=$((port + 11)); XMPI_BASEPORT=$port timeout 120 "$@" > $OUT/$name.json 2>> $OUT/err.txt || echo "rc=$? $name" >> $OUT/err.txt; }
go p256 $BIN/xmpirun 8 $BIN/allreduce_bench $((256<<20)) 20 5 auto fused split ring rhd
go p64 $BIN/xmpirun 8 $BIN/allreduce_bench $((64<<20)) 40 5 auto fused split
go p16 $BIN/xmpirun 8 $BIN/allreduce_bench $((16<<20)) 50 5 auto fused split ring rhd
go p1 $BIN/xmpirun 8 $BIN/allreduce_bench $((1<<20)) 200 10 auto fused split ring rhd
XMPI_SCHED_GRID=64 go ring64_256 $BIN/xmpirun 8 $BIN/allreduce_bench $((256<<20)) 20 5 ring rhd
XMPI_SCHED_GRID=64 go ring64_1 $BIN/xmpirun 8 $BIN/allreduce_bench $((1<<20)) 200 10 ring rhd
XMPI_SCHED_GRID=32 go ring32_1 $BIN/xmpirun 8 $BIN/allreduce_bench $((1<<20)) 200 10 ring rhd
python - <<P
import json, glob, os
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), [(r["mode"], round(r["us_per_step"], 1), round(r["kernel_avg_us"], 1)) for r in d["rows"]])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
P
tail -n 5 $OUT/err.txt 2>/dev/null
