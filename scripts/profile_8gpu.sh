#!/bin/bash
# One command for an 8-GPU MI355X node: the SCALE line plus the link roofline and per-rank kernel evidence.
# Run from the repo root; writes under profiles/r06_8gpu/ (small files only).
#   bash scripts/profile_8gpu.sh [N=8]
# 1. bench.py --gpus N under torch.distributed.run (one process per GPU, ranks meet on the device): the compact JSON line with
#    value = algbw @ 256 MiB f32, busbw, `xgmi` {link_probe taken before anything is tuned: SDMA vs copy kernel, write / read /
#    both directions; wire GB/s per rank and per link against 76.8 / 153 GB/s; frac_of_link_peak}, `config.tuned` = what the
#    LIBRARY's tuner (xmpi_tune) chose on these links; bench_extras.json beside it (per-algorithm times, sweeps).
# 2. the production-layout program on the links: examples/allreduce_bench with every schedule by name -- the library's choice,
#    the one-kernel fold (1 / 2 packets in flight), meet / body / done, push-only, the ring kernel (all ring channels = all
#    links), the halving kernel, and the PUSH forms of both (ring_push / rhd_push: every payload byte a posted store over its link
#    where ring / rhd load it -- which of the two a link moves faster is the first thing to read) -- at 256 MiB and 1 MiB, plain and under rocprofv3 --kernel-trace --stats (one
#    kernel_stats.csv PER RANK); and the Send / Receive ping-pong between GPU 0 and GPU 1 (half round trip, through the C ABI).
# 3. BASELINE cfg 5 on the links: examples/cfg5_sweep (fp16, ring vs halving vs the library's choice, 1 MiB ... 1 GiB).
# 4. the ring kernel's channel count on real links: 1, 2, all.
# 5. (round 4) FIRST of all: the GPU suite's split-form tests -- on real links this is where the XCD guard's masks and the
#    system-scope data kernel meet another GPU's memory for the first time; then the split form with and without body_sys, small
#    collectives with and without LL lines, and a marker trace (named ranges between the kernels).
# Rehearsal (tests/test_devsim.py, no GPU): XMPI_8GPU_REHEARSAL=1 shrinks every size, skips the GPU suite and the rocprofv3 steps and
# writes under $XMPI_8GPU_OUT (bench.py without its child-process probe of every schedule: tests/test_devsim.py rehearses that with the scale command itself) -- every OTHER command line below runs as written, on tests/devsim's virtual GPUs, so that a typo in a
# mode name or an environment variable is found before the one chance on a real node is spent on it.
set -x
N=${1:-8}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=${XMPI_8GPU_OUT:-$ROOT/profiles/r06_8gpu}
mkdir -p $O
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 XMPI_TIMEOUT_S=${XMPI_TIMEOUT_S:-120}
BIN=$ROOT/mpi_amd/bin
REH=${XMPI_8GPU_REHEARSAL:-0}
if [ "$REH" = 1 ]; then L=131072; M=65536; S=16384; C5=1048576; C3=16384; K="2 1"; KS="2 1"; CS="65536 2"; BARGS="--steps 2 --warmup 1 --size-mib 0.25 --no-cpu --no-probe"
else L=268435456; M=16777216; S=1048576; C5=1073741824; C3=2097152; K="20 5"; KS="200 10"; CS="1048576 200"; BARGS="--steps 20 --warmup 5"; fi
cd $ROOT
if [ "$REH" != 1 ]; then
timeout 1200 python -m pytest tests/test_gpu_collectives.py -k "split_form or stepped or ll_" -x -q > $O/pytest_split_sched_ll.log 2>&1; echo "pytest rc=$?"
tail -n 5 $O/pytest_split_sched_ll.log
fi
LAUNCH="python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29533"
XMPI_BENCH_EXTRAS_DIR=$O timeout 1500 $LAUNCH bench.py --gpus $N $BARGS > $O/bench_n$N.json 2> $O/bench_n$N.err
mv $O/bench_extras.json $O/bench_n${N}_extras.json 2>/dev/null
tail -c 800 $O/bench_n$N.err
MODES="auto fused fused2 split zpush ring ring_push rhd rhd_push"
XMPI_BASEPORT=7100 timeout 600 $BIN/xmpirun $N $BIN/allreduce_bench $L $K $MODES > $O/prod_n${N}_256MiB.json 2> $O/prod.err
XMPI_BASEPORT=7150 timeout 600 $BIN/xmpirun $N $BIN/allreduce_bench $M $K $MODES > $O/prod_n${N}_16MiB.json 2>> $O/prod.err
XMPI_BASEPORT=7200 timeout 600 $BIN/xmpirun $N $BIN/allreduce_bench $S $KS $MODES > $O/prod_n${N}_1MiB.json 2>> $O/prod.err
XMPI_BASEPORT=7250 timeout 900 $BIN/xmpirun $N $BIN/cfg5_sweep $C5 ${K%% *} > $O/cfg5_n$N.json 2>> $O/prod.err
XMPI_BASEPORT=7280 timeout 300 $BIN/xmpirun 4 $BIN/cfg3_allgather $C3 ${K%% *} > $O/cfg3_4gpu.json 2>> $O/prod.err
for ch in 1 2 0; do
  XMPI_SCHED_CHANNELS=$ch XMPI_BASEPORT=7300 timeout 600 $BIN/xmpirun $N $BIN/allreduce_bench $L $K ring ring_push > $O/ring_channels_${ch}_n$N.json 2>> $O/prod.err
done
for SYS in 0 1; do
  XMPI_BODY_SYS=$SYS XMPI_BASEPORT=7320 timeout 600 $BIN/xmpirun $N $BIN/allreduce_bench $L $K split > $O/split_body_sys${SYS}_n$N.json 2>> $O/prod.err
done
for LL in 0 32768; do
  XMPI_LL_BYTES=$LL XMPI_BASEPORT=7340 timeout 300 $BIN/xmpirun $N $BIN/coll_sweep $CS > $O/coll_sweep_n${N}_ll$LL.json 2>> $O/prod.err
done
# the LL agent over links: blocking small collectives launched (0) / by the lingering kernel up to its default limit (8192) / up to the
# slot limit (32768) -- where its limit lies on a node is one of the things only a node can say
for AG in 0 8192 32768; do
  XMPI_LL_BYTES=32768 XMPI_AGENT_LL_BYTES=$AG XMPI_BASEPORT=7350 timeout 300 $BIN/xmpirun $N $BIN/coll_sweep 32768 ${CS##* } 2 > $O/coll_sweep_n${N}_agent$AG.json 2>> $O/prod.err
done
if [ "$REH" = 1 ]; then python scripts/show_bench.py $O/bench_n$N.json | head -40; python scripts/first_hour_report.py $O; exit 0; fi
cd /tmp
XMPI_BASEPORT=7360 timeout 300 rocprofv3 --marker-trace --kernel-trace --output-format csv -d $O/markers -- $BIN/xmpirun $N $BIN/coll_sweep ${CS%% *} 20 > $O/coll_sweep_under_marker_trace.json 2> $O/markers.err
for f in $O/markers/*/*marker_api_trace.csv; do head -n 120 $f > $O/marker_trace_$(basename $f | cut -d_ -f1)_head.txt; done
for m in auto ring ring_push rhd rhd_push; do
  XMPI_BASEPORT=7400 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$m -- $BIN/xmpirun $N $BIN/allreduce_bench $L $K $m \
      > $O/prod_${m}_under_rocprof.json 2> $O/stats_$m.err
done
cd $ROOT
find $O -name "*.csv" ! -name "*kernel_stats.csv" -delete; find $O -name "*.db" -delete
find $O -name "*kernel_stats.csv" | head -40
python scripts/show_bench.py $O/bench_n$N.json 2>/dev/null | head -40
python scripts/first_hour_report.py $O   # the same files in the order DESIGN.md section 0 reads them
du -sh $O
