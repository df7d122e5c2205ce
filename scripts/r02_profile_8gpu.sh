# One command for an 8-GPU MI355X node (profiles/README.md, "8 GPUs"): the SCALE line plus the link roofline and
# per-rank kernel stats.  Run from the repo root; writes under profiles/r02_8gpu/ (small files only).
#   bash scripts/r02_profile_8gpu.sh [N=8]
# 1. bench.py --gpus N under torch.distributed.run (one process per GPU): JSON line with value = algbw @ 256 MiB f32,
#    busbw, xgmi.{link_probe (taken BEFORE tuning: SDMA vs copy kernel, write / read / both directions),
#    wire GB/s per rank and per link against 76.8 / 153 GB/s}, autotune (ring / halving / direct / zero-copy with the
#    ranks meeting on the device, 1 or 2 packets in flight / on the host / push-only), busbw_table, cfg 5 sweep.
# 2. rocprofv3 --kernel-trace --stats of the same job with the schedule the tuner chose: one kernel_stats.csv PER RANK
#    (rocprofv3 writes one directory per process id).
# 3. the one-process-per-rank size sweep through the launcher (examples/coll_sweep), plain and under rocprofv3.
set -x
N=${1:-8}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=$ROOT/profiles/r02_8gpu
mkdir -p $O
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $ROOT
LAUNCH="python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29533"
timeout 1500 $LAUNCH bench.py --gpus $N --steps 20 --warmup 5 > $O/bench_n$N.json 2> $O/bench_n$N.err
tail -c 800 $O/bench_n$N.err
ALGO=$(python -c "import json,sys; print(json.load(open('$O/bench_n$N.json'))['config']['algo'])" 2>/dev/null || echo auto)
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $LAUNCH $ROOT/bench.py --gpus $N --algo $ALGO --no-extras --no-cpu --no-probe \
    > $O/bench_n${N}_under_rocprof.json 2> $O/stats.err
RUN="$ROOT/mpi_amd/bin/xmpirun $N $ROOT/mpi_amd/bin/coll_sweep 268435456 100"
XMPI_TIMEOUT_S=120 timeout 600 $RUN > $O/coll_sweep_n$N.json 2> $O/coll_sweep_n$N.err
XMPI_TIMEOUT_S=120 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sweep_stats -- $RUN > /dev/null 2> $O/sweep_stats.err
cd $ROOT
find $O -name "*.csv" ! -name "*kernel_stats.csv" -delete; find $O -name "*.db" -delete
find $O -name "*kernel_stats.csv" | head -20
python scripts/show_bench.py $O/bench_n$N.json 2>/dev/null | head -40
du -sh $O
