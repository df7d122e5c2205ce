#!/bin/bash
# How many blocks should the one-kernel fold use between "small" and "split"?  8 processes on one GPU.
#   bash scripts/r03_tiles.sh   (writes gpurun_out/tiles/)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=$ROOT/gpurun_out/tiles
mkdir -p $O
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 XMPI_TIMEOUT_S=20
BIN=$ROOT/mpi_amd/bin
port=7600
for bytes in ${SIZES:-1048576 4194304 16777216 67108864}; do
  for tiles in ${TILES:-8 4 2 1}; do
    for grid in ${GRIDS:-0}; do
      port=$((port + 7))
      XMPI_DSYNC_TILES=$tiles XMPI_DSYNC_GRID=$grid XMPI_BASEPORT=$port timeout 60 $BIN/xmpirun 8 $BIN/allreduce_bench $bytes 50 10 fused fused2 \
        > $O/b${bytes}_t${tiles}_g${grid}.json 2>> $O/err.txt || echo "rc=$? $bytes $tiles $grid" >> $O/err.txt
    done
  done
done
python - <<P
import json, glob, os
for f in sorted(glob.glob("$O/b*.json"), key=lambda p: (int(os.path.basename(p)[1:].split("_")[0]), p)):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), [(r["mode"], round(r["us_per_step"], 1)) for r in d["rows"]])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
P
