#!/bin/bash
# Round-4 evidence on one MI355X (run through gpurun from the repo root); everything lands in gpurun_out/r04/:
#   1. the default bench line (N=1: 8 ranks as threads of one process; roofline_production from 8 processes)
#   2. rocprofv3 --kernel-trace --stats of the N=1 line's command (reduce_n_multi_kernel)
#   3. THE PRODUCTION LAYOUT: one OS process per rank, 8 x 256 MiB f32 (examples/allreduce_bench under the launcher), modes
#      auto (the library's tuned choice: meet / body / done), fused, ring, rhd -- plain, and under rocprofv3 --kernel-trace --stats
#      (one kernel_stats.csv per rank)
#   4. PMC traffic (separate --pmc passes, kernel-trace only) of the N=1 command and of the production layout
#   5. what a collective costs the caller's other streams (scripts/overlap_probe.hip, 2 processes); BASELINE cfg 5 with one
#      process per rank (examples/cfg5_sweep); the production program at 2 / 4 processes and at 16 MiB / 1 MiB
#   6. (round 4) small collectives with and without LL lines; the split form with the system-scope data kernel; roctx ranges
#      under rocprofv3 --marker-trace; the fold on different buffer sets (scripts/r04_gap.py)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r04
rm -rf $O; mkdir -p $O
BIN=$GRAFT_REPO_ROOT/mpi_amd/bin
timeout 600 python bench.py > $O/bench_n1_default.json 2> $O/bench_n1_default.err
cp bench_extras.json $O/bench_n1_default_extras.json
tail -c 400 $O/bench_n1_default.err
B="python $GRAFT_REPO_ROOT/bench.py --algo zcopy --no-extras --no-cpu --no-production"
PROD="$BIN/xmpirun 8 $BIN/allreduce_bench 268435456 20 5"
export XMPI_TIMEOUT_S=40 XMPI_NGPUS=1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_n1 -- $B > $O/bench_zcopy_under_rocprof.json 2> $O/stats_n1.err
XMPI_BASEPORT=7100 timeout 200 $PROD auto fused fused2 split ring rhd zpush > $O/prod_8proc_256MiB.json 2> $O/prod.err
XMPI_BASEPORT=7150 timeout 200 $BIN/xmpirun 8 $BIN/allreduce_bench 1073741824 5 2 auto ring rhd > $O/prod_8proc_1GiB.json 2>> $O/prod.err
for m in split fused ring rhd; do
  XMPI_BASEPORT=7200 timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_prod_$m -- $PROD $m > $O/prod_${m}_under_rocprof.json 2> $O/stats_prod_$m.err
done
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_n1 -- $B --steps 5 > /dev/null 2> $O/fetch_n1.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_n1 -- $B --steps 5 > /dev/null 2> $O/write_n1.err
PRODS="$BIN/xmpirun 8 $BIN/allreduce_bench 268435456 5 2"
XMPI_TIMEOUT_S=20 XMPI_BASEPORT=7300 timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_prod -- $PRODS split fused > $O/prod_under_pmc_fetch.json 2> $O/fetch_prod.err
echo "pmc fetch rc=$?" >> $O/prod.err
XMPI_TIMEOUT_S=20 XMPI_BASEPORT=7350 timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_prod -- $PRODS split fused > $O/prod_under_pmc_write.json 2> $O/write_prod.err
echo "pmc write rc=$?" >> $O/prod.err
XMPI_BASEPORT=7400 timeout 200 $BIN/xmpirun 2 $GRAFT_REPO_ROOT/scripts/overlap_probe_bin 268435456 8 3 > $O/overlap_2proc_256MiB.json 2> $O/overlap.err
XMPI_BASEPORT=7450 timeout 200 $BIN/xmpirun 2 $GRAFT_REPO_ROOT/scripts/overlap_probe_bin 16777216 8 3 > $O/overlap_2proc_16MiB.json 2>> $O/overlap.err
for us in 40 0; do XMPI_P2P_AGENT_US=$us XMPI_BASEPORT=$((7900 + us)) timeout 60 $BIN/xmpirun 2 $BIN/allreduce_bench 16777216 5 2 fused > $O/bounce_2proc_agent_${us}us.json 2>> $O/prod.err; done
XMPI_P2P_KERNEL_ACK=0 XMPI_BASEPORT=7990 timeout 60 $BIN/xmpirun 2 $BIN/allreduce_bench 16777216 5 2 fused > $O/bounce_2proc_round2_path.json 2>> $O/prod.err
XMPI_BASEPORT=7500 timeout 300 $BIN/xmpirun 8 $BIN/cfg5_sweep 1073741824 5 > $O/cfg5_8proc.json 2> $O/cfg5.err
XMPI_BASEPORT=7550 timeout 200 $BIN/xmpirun 4 $BIN/cfg3_allgather 2097152 20 > $O/cfg3_4proc.json 2>> $O/cfg5.err
for n in 2 4; do XMPI_BASEPORT=7600 timeout 200 $BIN/xmpirun $n $BIN/allreduce_bench 268435456 20 5 auto fused split ring rhd > $O/prod_${n}proc_256MiB.json 2>> $O/prod.err; done
XMPI_BASEPORT=7700 timeout 200 $BIN/xmpirun 8 $BIN/allreduce_bench 16777216 50 5 auto fused split ring rhd > $O/prod_8proc_16MiB.json 2>> $O/prod.err
XMPI_BASEPORT=7800 timeout 200 $BIN/xmpirun 8 $BIN/allreduce_bench 1048576 200 10 auto fused split ring rhd > $O/prod_8proc_1MiB.json 2>> $O/prod.err
# Send / Receive in the reference's own loop shape (examples/bounce.cpp = bounce.go): buffers in HBM, Go-style host slices
# (host lanes of the shared segment; XMPI_HOST_LANES=0: staged through both GPUs), the reference's TCP + gob protocol
for v in "" "--host" "--tcp"; do XMPI_BASEPORT=8100 timeout 100 $BIN/xmpirun 2 $BIN/bounce $v --repeats 50 > "$O/bounce_example${v#-}.txt" 2>> $O/prod.err; done
XMPI_HOST_LANES=0 XMPI_BASEPORT=8150 timeout 100 $BIN/xmpirun 2 $BIN/bounce --host --repeats 50 > $O/bounce_example-host_staged_through_hbm.txt 2>> $O/prod.err
# blocking vs enqueued collectives, 1 KiB ... 1 MiB (completion word instead of an event)
XMPI_BASEPORT=8200 timeout 100 $BIN/xmpirun 2 $BIN/coll_sweep 1048576 200 > $O/coll_sweep_2proc.json 2>> $O/prod.err
XMPI_BASEPORT=8250 timeout 100 $BIN/xmpirun 8 $BIN/coll_sweep 1048576 200 > $O/coll_sweep_8proc.json 2>> $O/prod.err
# round 4: LL lines on / off, the system-scope data kernel, named ranges, buffer sets
for N in 2 8; do for LL in 0 32768; do
  XMPI_LL_BYTES=$LL XMPI_BASEPORT=$((8300 + N * 10 + LL / 8192)) timeout 100 $BIN/xmpirun $N $BIN/coll_sweep 1048576 200 > $O/coll_sweep_${N}proc_ll$LL.json 2>> $O/prod.err
done; done
XMPI_BODY_SYS=1 XMPI_BASEPORT=8400 timeout 200 $PROD split > $O/prod_8proc_256MiB_body_sys.json 2>> $O/prod.err
XMPI_BODY_SYS=1 XMPI_BASEPORT=8420 timeout 200 $BIN/xmpirun 2 $BIN/allreduce_bench 268435456 20 5 split > $O/prod_2proc_256MiB_body_sys.json 2>> $O/prod.err
XMPI_BASEPORT=8440 timeout 200 rocprofv3 --marker-trace --kernel-trace --output-format csv -d $O/markers -- $BIN/xmpirun 2 $BIN/coll_sweep 1048576 20 > $O/coll_sweep_under_marker_trace.json 2> $O/markers.err
for f in $O/markers/*/*marker_api_trace.csv; do head -n 120 $f > $O/marker_trace_$(basename $f | cut -d_ -f1)_head.txt; done
timeout 200 python $GRAFT_REPO_ROOT/scripts/r04_gap.py 256 > $O/gap_buffer_sets.json 2>> $O/prod.err
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py $O/fetch_n1 $O/write_n1 reduce_n_multi > $O/pmc_bench_zcopy.json
python scripts/pmc_summary.py $O/fetch_prod $O/write_prod dsync_ > $O/pmc_prod.json 2>> $O/prod.err
find $O -name "*kernel_stats.csv" | head -40; find $O -name "*.csv" ! -name "*kernel_stats.csv" -size +1M -delete; find $O -name "*.db" -delete
du -sh $O
