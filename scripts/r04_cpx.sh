#!/bin/bash
# Compute partitioning as the only way to see more than one HIP device on a 1-GPU box: an MI355X in CPX mode presents its 8 XCDs
# as 8 logical devices (32 CUs each; with NPS1 they all see the whole HBM).  NOT xGMI: "peers" are slices of one chip -- but
# rank i <-> device i, hipDeviceEnablePeerAccess, cross-DEVICE hipIpcOpenMemHandle and system-scope loads of another device's
# memory execute for the first time.  SPX is restored on exit whatever happens.
#   bash scripts/r04_cpx.sh probe          what the box says, try CPX, device count, peer matrix, peer copies; quick 2 / 8 rank runs
#   bash scripts/r04_cpx.sh suite          the same, then pytest -m gpu + bench.py --gpus 8 on the partitions
# Everything lands in gpurun_out/cpx/.
MODE=${1:-probe}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=$ROOT/gpurun_out/cpx
mkdir -p $O
cd $ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 XMPI_TIMEOUT_S=60
BIN=$ROOT/mpi_amd/bin
SMI=/opt/rocm/bin/rocm-smi
AMDSMI=/opt/rocm/bin/amd-smi

restore() {
  echo "== restore SPX" >> $O/partition.log
  timeout 120 $SMI --setcomputepartition SPX >> $O/partition.log 2>&1 || timeout 120 $AMDSMI set --gpu 0 --compute-partition SPX >> $O/partition.log 2>&1
  timeout 30 $SMI --showcomputepartition >> $O/partition.log 2>&1
}
trap restore EXIT
trap 'exit 143' TERM INT HUP
LIMIT=${CPX_LIMIT_S:-700}   # no new step starts after this many seconds: the restore must run before gpurun's own limit
late() { [ $SECONDS -gt $LIMIT ] && echo "past ${LIMIT}s: skipping $1"; }

{
  echo "== before"; date
  timeout 30 $SMI --showcomputepartition --showmemorypartition
  timeout 30 $AMDSMI partition 2>&1 | head -60
  ls -l /dev/kfd /dev/dri 2>&1
  echo "== devprobe in the mode the box came in"
  timeout 120 scripts/devprobe_bin
  echo "== set CPX"
  timeout 120 $SMI --setcomputepartition CPX; echo "rocm-smi rc=$?"
  timeout 30 $SMI --showcomputepartition --showmemorypartition
} > $O/partition.log 2>&1
if ! timeout 30 $SMI --showcomputepartition 2>/dev/null | grep -q CPX; then
  { echo "== rocm-smi did not switch; amd-smi"; timeout 120 $AMDSMI set --gpu 0 --compute-partition CPX; echo "amd-smi rc=$?"
    timeout 30 $SMI --showcomputepartition --showmemorypartition; } >> $O/partition.log 2>&1
fi
{ echo "== after"; ls -l /dev/dri 2>&1; timeout 60 /opt/rocm/bin/rocminfo | grep -c "gfx950" ; } >> $O/partition.log 2>&1
timeout 180 scripts/devprobe_bin > $O/devprobe_cpx.json 2> $O/devprobe_cpx.err
cat $O/partition.log | tail -60
cat $O/devprobe_cpx.json
NDEV=$(python -c "import json,sys; print(json.load(open('$O/devprobe_cpx.json')).get('devices',0))" 2>/dev/null || echo 0)
echo "devices visible: $NDEV"
if [ "${NDEV:-0}" -lt 2 ]; then
  echo "no second device: nothing more to do" | tee -a $O/partition.log
  exit 0
fi
export XMPI_NGPUS=$NDEV
# first contact of the multi-device path: 2 ranks on 2 devices, then one rank per device
late coll2 || XMPI_BASEPORT=7100 timeout 120 $BIN/xmpirun 2 $BIN/coll_sweep 1048576 20 > $O/coll_sweep_2dev.json 2> $O/coll_sweep_2dev.err; echo "coll_sweep 2: rc=$?"
tail -c 600 $O/coll_sweep_2dev.json; tail -c 1500 $O/coll_sweep_2dev.err
late bounce || XMPI_BASEPORT=7120 timeout 120 $BIN/xmpirun 2 $BIN/bounce > $O/bounce_2dev.txt 2> $O/bounce_2dev.err; echo "bounce 2: rc=$?"
tail -n 14 $O/bounce_2dev.txt; tail -c 800 $O/bounce_2dev.err
R=$NDEV; [ $R -gt 8 ] && R=8
late collR || XMPI_BASEPORT=7140 timeout 180 $BIN/xmpirun $R $BIN/coll_sweep 1048576 20 > $O/coll_sweep_${R}dev.json 2> $O/coll_sweep_${R}dev.err; echo "coll_sweep $R: rc=$?"
tail -c 600 $O/coll_sweep_${R}dev.json; tail -c 1500 $O/coll_sweep_${R}dev.err
MODES="auto fused split zpush ring rhd"
late prod256 || XMPI_BASEPORT=7160 timeout 300 $BIN/xmpirun $R $BIN/allreduce_bench 268435456 10 3 $MODES > $O/prod_${R}dev_256MiB.json 2> $O/prod_${R}dev_256MiB.err; echo "allreduce_bench $R x 256 MiB: rc=$?"
tail -c 1500 $O/prod_${R}dev_256MiB.json; tail -c 1500 $O/prod_${R}dev_256MiB.err
late prod1 || XMPI_BASEPORT=7180 timeout 300 $BIN/xmpirun $R $BIN/allreduce_bench 1048576 100 10 $MODES > $O/prod_${R}dev_1MiB.json 2> $O/prod_${R}dev_1MiB.err; echo "allreduce_bench $R x 1 MiB: rc=$?"
tail -c 1200 $O/prod_${R}dev_1MiB.json
if [ "$MODE" = suite ]; then
  timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_cpx.log 2>&1; echo "pytest -m gpu on $NDEV partitions: rc=$?"
  tail -n 25 $O/pytest_gpu_cpx.log
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $R --master-addr 127.0.0.1 --master-port 29541 \
      bench.py --gpus $R --steps 20 --warmup 5 > $O/bench_gpus$R.json 2> $O/bench_gpus$R.err; echo "bench --gpus $R: rc=$?"
  cp bench_extras.json $O/bench_gpus${R}_extras.json 2>/dev/null
  tail -c 3500 $O/bench_gpus$R.json; tail -c 1500 $O/bench_gpus$R.err
fi
