#!/bin/bash
# quick GPU session of round 3: the new device-side schedules (functional), then the production-layout figures
cd "$(dirname "$0")/.." || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
OUT=gpurun_out/s1
rm -rf $OUT; mkdir -p $OUT
run() { name=$1; shift; echo "=== $name: $*" | tee -a $OUT/log.txt; ( time timeout "${T:-600}" "$@" ) > $OUT/$name.txt 2>&1; echo "rc=$? $name" | tee -a $OUT/log.txt; tail -n 5 $OUT/$name.txt >> $OUT/log.txt; }
T=900 run t_new python -m pytest tests/test_gpu_collectives.py -x -q -k "stepped or split or several_streams or send_recv or tuner or staged_schedules or stream_ordered or zero_copy_processes"
T=600 run t_p2p python -m pytest tests/test_gpu_collectives.py -x -q -k "bounce or p2p_semantics or helloworld"
BIN=mpi_amd/bin
export XMPI_TIMEOUT_S=60 XMPI_NGPUS=1
T=300 XMPI_BASEPORT=7100 run prod256 $BIN/xmpirun 8 $BIN/allreduce_bench $((256<<20)) 20 5 auto fused split ring rhd
T=200 XMPI_BASEPORT=7300 run prod16 $BIN/xmpirun 8 $BIN/allreduce_bench $((16<<20)) 50 5 auto fused split ring rhd
T=200 XMPI_BASEPORT=7500 run prod1 $BIN/xmpirun 8 $BIN/allreduce_bench $((1<<20)) 200 10 auto fused split ring rhd
T=200 XMPI_BASEPORT=7700 run prod2p $BIN/xmpirun 2 $BIN/allreduce_bench $((16<<20)) 50 5 auto fused split ring rhd
T=200 XMPI_BASEPORT=7800 run prod2p256 $BIN/xmpirun 2 $BIN/allreduce_bench $((256<<20)) 20 5 auto fused split ring rhd
T=200 XMPI_BASEPORT=7400 run overlap256 $BIN/xmpirun 2 scripts/overlap_probe_bin $((256<<20)) 5 3
T=200 XMPI_BASEPORT=7450 run overlap16 $BIN/xmpirun 2 scripts/overlap_probe_bin $((16<<20)) 8 3
T=120 XMPI_BASEPORT=7900 run sweep2 $BIN/xmpirun 2 $BIN/coll_sweep $((1<<20)) 200
T=120 XMPI_BASEPORT=7950 run sweep8 $BIN/xmpirun 8 $BIN/coll_sweep $((1<<20)) 200
T=600 run bench python bench.py
cp bench_extras.json $OUT/ 2>/dev/null
echo done >> $OUT/log.txt
