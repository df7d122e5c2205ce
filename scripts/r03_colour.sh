#!/bin/bash
# heap colouring on / off, same box, interleaved: the N = 1 line's kernel and the production layout
cd "$(dirname "$0")/.." || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp XMPI_TIMEOUT_S=30
O=gpurun_out/colour; rm -rf $O; mkdir -p $O
BIN=mpi_amd/bin
port=7100
for i in 1 2 3; do
  for col in 1 0; do
    XMPI_HEAP_COLOUR=$col timeout 120 python bench.py --algo zcopy --no-extras --no-cpu --no-production > $O/n1_c${col}_$i.json 2>> $O/err.txt
    port=$((port + 13))
    XMPI_HEAP_COLOUR=$col XMPI_BASEPORT=$port timeout 120 $BIN/xmpirun 8 $BIN/allreduce_bench $((256<<20)) 20 5 split fused ring > $O/p8_c${col}_$i.json 2>> $O/err.txt
  done
done
for col in 1 0; do
  port=$((port + 13))
  XMPI_HEAP_COLOUR=$col XMPI_BASEPORT=$port timeout 120 $BIN/xmpirun 2 $BIN/allreduce_bench $((256<<20)) 20 5 split fused > $O/p2_c${col}.json 2>> $O/err.txt
  port=$((port + 13))
  XMPI_HEAP_COLOUR=$col XMPI_BASEPORT=$port timeout 120 $BIN/xmpirun 8 $BIN/allreduce_bench $((16<<20)) 50 5 split fused > $O/p8_16M_c${col}.json 2>> $O/err.txt
done
python - <<P
import json, glob, os
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        if "rows" in d: print(os.path.basename(f), [(r["mode"], round(r["us_per_step"], 1), round(r["kernel_avg_us"], 1)) for r in d["rows"]])
        else: print(os.path.basename(f), d["value"], round(d["ms_per_step"], 4), round(d["roofline"]["avg_launch_us"], 1), round(d["roofline"]["frac"], 3))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
P
tail -n 3 $O/err.txt
