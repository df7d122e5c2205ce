cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 XMPI_TIMEOUT_S=60 XMPI_NGPUS=1 XMPI_CHECK_PASSES=1
BIN=$GRAFT_REPO_ROOT/mpi_amd/bin
for i in 1 2 3 4 5 6; do
  s=$(date +%s)
  XMPI_BASEPORT=7100 timeout 300 $BIN/xmpirun 8 $BIN/allreduce_bench 268435456 3 1 auto 2>&1 >/dev/null | grep "xmpi 0 " | cut -c1-250
  e=$(date +%s); echo "run $i wall $((e - s)) s"
done
