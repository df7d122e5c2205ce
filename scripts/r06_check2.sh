#!/bin/bash
# Round 6: the self-check with Send / Receive, the bench with live PMC passes, the bench contract tests -> gpurun_out/r06_check2/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r06_check2
rm -rf $O; mkdir -p $O
(time timeout 1500 python -m pytest tests -m gpu -x -q -k "selfcheck or bench_json_contract or bench_extras or degraded or smoke" 2>&1 | tail -15) > $O/pytest.log 2>&1
tail -6 $O/pytest.log
timeout 200 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
(time timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err) 2>&1 | grep real
python - <<'PY'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06_check2"
line = open(O + "/bench_n1.json").read().strip().split("\n")[-1]
d = json.loads(line)
print("line bytes", len(line), "value", d["value"])
print("roofline", {k: d["roofline"].get(k) for k in ("avg_launch_us", "kernel_us_min", "kernel_us_max", "frac", "frac_of_achievable", "box_copy_us", "frac_of_box", "traffic", "traffic_from_profile", "traffic_over_algorithmic")})
PY
tail -3 $O/bench_n1.err
XMPI_TRACE=1 XMPI_SELFCHECK=1 XMPI_NGPUS=1 XMPI_BASEPORT=7300 timeout 300 mpi_amd/bin/xmpirun 8 mpi_amd/bin/allreduce_bench 1048576 5 2 fused 2>&1 | grep "xmpi 0 " | grep -i "self-check\|init: done\|init: final" > $O/selfcheck_trace_8proc.log
cat $O/selfcheck_trace_8proc.log
