#!/bin/bash
# Round 6 (the watchdog on by default, XMPI_SELFCHECK=1 forced): a longer walk than the GPU suite's (the push forms of the stepped kernels and the kept landing block among the forms it draws): the soak scenario (tests/scenarios.py sc_soak) with several seeds, 2 ... 8 processes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 XMPI_SELFCHECK=1
mkdir -p gpurun_out/r06_soak
python - <<'PY' > gpurun_out/r06_soak/soak_long.log 2>&1
import time
from tests.gpu_harness import run_ranks, run_threads
for size, seed, steps in ((2, 31, 2500), (3, 32, 2000), (4, 33, 2000), (5, 34, 1500), (6, 35, 1500), (7, 36, 1200), (8, 37, 1200), (8, 38, 1200)):
    t0 = time.time()
    try:
        outs = run_ranks("soak", size, {"seed": seed, "steps": steps}, timeout=600)
        print(f"size {size} seed {seed} steps {steps}: ok in {time.time() - t0:.1f} s", [ln for ln in outs[0].splitlines() if ln.startswith("soak:")], flush=True)
    except BaseException as e:  # noqa: BLE001
        print(f"size {size} seed {seed} steps {steps}: FAILED after {time.time() - t0:.1f} s\n{str(e)[-3000:]}", flush=True)
t0 = time.time()
try:
    run_threads("soak", 8, {"seed": 29, "steps": 1000}, timeout=600)
    print(f"8 threads: ok in {time.time() - t0:.1f} s")
except BaseException as e:  # noqa: BLE001
    print(f"8 threads: FAILED\n{str(e)[-3000:]}")
PY
cat gpurun_out/r06_soak/soak_long.log | tail -80
