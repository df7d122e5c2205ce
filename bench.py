#!/usr/bin/env python3
"""bench.py -- allreduce bandwidth of the xmpi hot path (BASELINE.json metric), ONE compact JSON line.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json config 4, the one the metric is quoted on): allreduce-sum of 256 MiB of
float32 per rank over 8 ranks.  The job always has 8 ranks; they are spread over the N GPUs the
run was given (N = 8: one rank per MI355X, the north-star layout, ranks meeting on the device; N = 4 / 2 / 1:
2 / 4 / 8 ranks share each GPU, hosted as threads of the per-GPU process).  Total work is fixed as N grows,
so "scaling" is "strong".  A step = one allreduce; inputs are generated on the device before the timed
region (nothing crosses PCIe while timing).

value = algbw = S / t, the nccl-tests convention BASELINE.json's metric names (S = bytes per rank, GB = 1e9 B),
with t = max over ranks of the barrier-bracketed time of K steps / K; busbw = algbw x 2(R-1)/R alongside, and
`busbw_at_size` = busbw at the workload's size for 1 / 2 / 4 / 8 ranks.

The schedule is the LIBRARY's (XMPI_ALGO_AUTO: its own tuned table where ranks meet on the device -- xmpi_tune --
and the zero-copy fold otherwise); `--algo` names one instead.  The result of the schedule that is timed is checked
against the CPU oracle over the WHOLE buffer before timing (rank r its chunk and the chunk boundaries, all ranks'
buffers shown identical by checksums).

roofline: the dominant kernel of the timed region with HIP events attached to sampled dispatches (on the stream the
kernel runs on).  N = 1 hosts the 8 ranks as threads: reduce_n_multi_kernel folds all chunks.  `roofline_production`
(N = 1 only) is the same workload with ONE OS PROCESS PER RANK (examples/allreduce_bench under the launcher): the kernels
the north-star layout runs -- dsync_fold_kernel, or dsync_body_kernel between the meet and done kernels.

cpu_baseline (N = 1, rank 0 only): the reference path -- mpi.Network over loopback TCP with gob framing -- restated in
C++ (oracle/refpath.cpp, "kind": "port": the image has no Go toolchain), 8 OS processes on the host cores.

Everything else (per-algorithm times, size sweeps, cfg 2 / 3 / 5, the one-process-per-rank sweeps) goes to
bench_extras.json beside this file (and gpurun_out/ when that exists), not to stdout.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from mpi_amd import xmpi  # noqa: E402

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s
HBM_ACHIEVABLE_GBPS = 6300.0  # ... what a streaming kernel achieves of it (the guide's float4-copy figure)
XGMI_LINK_GBPS = 153.6   # per-link peak, both directions together (task statement: 7 links x ~153 GB/s per GPU)
XGMI_DIR_GBPS = 76.8     # ... one direction, nominal: what bounds a schedule's busiest link direction
ALGO_NAME = {xmpi.ALGO_RING: "ring", xmpi.ALGO_RHD: "rhd", xmpi.ALGO_DIRECT: "direct", xmpi.ALGO_ZCOPY: "zcopy",
             xmpi.ALGO_ZPUSH: "zpush", xmpi.ALGO_LL: "ll", xmpi.ALGO_RING_PUSH: "ring_push", xmpi.ALGO_RHD_PUSH: "rhd_push"}
ZC_ALGOS = (xmpi.ALGO_ZCOPY, xmpi.ALGO_ZPUSH)
DT = {"f32": xmpi.F32, "f16": xmpi.F16, "f64": xmpi.F64, "bf16": xmpi.BF16, "i64": xmpi.I64}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--ranks", type=int, default=0, help="total ranks (default 8 when it divides by --gpus)")
    ap.add_argument("--size-mib", type=float, default=256.0, help="bytes per rank")
    ap.add_argument("--dtype", default="f32", choices=sorted(DT))
    ap.add_argument("--algo", default="auto", choices=["auto", "ring", "rhd", "direct", "zcopy", "zpush", "ring_push", "rhd_push"])
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed sweeps after the timed region")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU reference-path baseline")
    ap.add_argument("--no-production", action="store_true",
                    help="skip the one-process-per-rank run of the same workload (roofline_production, N = 1 only)")
    ap.add_argument("--probe", action="store_true",
                    help="(internal) a short zero-copy allreduce in a job of its own; the exit status is the verdict")
    ap.add_argument("--no-probe", action="store_true", help="skip the zero-copy probe before a multi-GPU run")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the headline's call on HOST slices (pcie_inclusive; one GPU only)")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the live HBM-traffic measurement (two rocprofv3 --pmc child passes of this bench, N = 1 only): roofline.traffic then "
                         "comes from the committed profile")
    ap.add_argument("--cpu-count", type=int, default=0,
                    help="float32 elements per rank of the CPU sample (default: the workload's own size, 256 MiB per rank)")
    ap.add_argument("--cpu-reps", type=int, default=2, help="repetitions of the CPU sample")
    return ap.parse_args()


class Job:
    """Everything the rank threads of this process share."""

    def __init__(self, args):
        self.args = args
        self.world_procs = int(os.environ.get("WORLD_SIZE", "1"))
        self.proc_rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        n = args.gpus
        if self.world_procs > 1 and self.world_procs != n:
            raise SystemExit(f"--gpus {n} but WORLD_SIZE={self.world_procs}")
        self.ranks = args.ranks or (8 if 8 % n == 0 else n)
        if self.ranks % n:
            raise SystemExit("--ranks must be a multiple of --gpus")
        self.single_process = self.world_procs == 1
        # one process per GPU under torchrun; a lone process hosts every rank (and drives every GPU)
        self.procs = self.world_procs
        self.ranks_per_proc = self.ranks // self.procs
        port = os.environ.get("MASTER_PORT", "0")
        self.key = os.environ.get("XMPI_BENCH_KEY") or f"bench-{port}-{os.getppid() if self.procs > 1 else os.getpid()}"
        self.zero_copy_ok = True
        self.probe_ok = None  # the schedules the probe saw working (None: not probed)
        self.probe = "not run"
        # ranks sharing a GPU have no link to pipeline against: large pieces (one launch per chunk)
        # keep every kernel at full-chip bandwidth; one rank per GPU keeps the library defaults
        if self.ranks // n >= 4:
            os.environ.setdefault("XMPI_SLOT_BYTES", str(32 << 20))
            os.environ.setdefault("XMPI_FIFO_DEPTH", "4")
        if self.ranks // n == 1 and "XMPI_BENCH_DEVICE" not in os.environ:
            # one rank per GPU drives up to 7 peer links at once, each on its own stream (staged schedules): give the
            # HIP runtime more hardware queues than its default of 4 so those streams do not serialise
            os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
        if "XMPI_BENCH_DEVICE" in os.environ:
            # rehearsal of the multi-process launch on ONE GPU: its processes share that GPU's hardware queues, and
            # beyond a few dozen the scheduler time-slices them (22 ms per collective instead of 40 us)
            os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
        # a wait that makes no progress for two minutes is an error with a message, not a bench that never returns
        # (the library's own default is to wait for ever, like the reference's blocking calls)
        os.environ.setdefault("XMPI_TIMEOUT_S", "120")
        self.result = {}
        self.bufs = {}  # rank -> (send, recv) of the timed region (ranks hosted by this process)
        self.errors = []
        self.lock = threading.Lock()

    def device_of(self, grank: int) -> int:
        if "XMPI_BENCH_DEVICE" in os.environ:  # rehearsal of the multi-process launch on a 1-GPU box
            return int(os.environ["XMPI_BENCH_DEVICE"])
        if self.procs > 1:  # one node: the local rank of the process that hosts `grank` (NOT this process' own: every rank is asked about)
            return grank // self.ranks_per_proc
        return grank * self.args.gpus // self.ranks  # lone process: ranks spread over the visible GPUs

    def my_ranks(self):
        base = self.proc_rank * self.ranks_per_proc
        return list(range(base, base + self.ranks_per_proc))


def all_max(comm, value: float) -> float:
    a = np.array([value], dtype=np.float64)
    out = np.zeros(1, dtype=np.float64)
    comm.allreduce(a, out, 1, xmpi.F64, xmpi.MAX, xmpi.ALGO_DIRECT)
    return float(out[0])


def timed(comm, fn, iters: int, prof: bool = False, batch=None) -> float:
    """barrier + device sync on both sides; max over ranks of the per-iteration time.  batch(iters), when
    given, runs all the iterations inside one native call (xmpi_allreduce_repeat) instead of fn() x iters:
    the rank threads of this process then do not queue for the interpreter lock between steps."""
    comm.barrier()
    comm.sync()
    if prof:
        comm.prof_enable(True)
    t0 = time.perf_counter()
    if batch is not None:
        batch(iters)
    else:
        for _ in range(iters):
            fn()
    comm.sync()
    comm.barrier()
    dt = (time.perf_counter() - t0) / iters
    if prof:
        comm.prof_enable(False)
    return all_max(comm, dt)


def check_whole(comm, recv, count, dtype, seed0):
    """This rank's share of the WHOLE result against the rank-order CPU oracle (oracle_check_allreduce regenerates every
    rank's input block by block): chunk `rank` and 4 Ki elements either side of it -- all chunks and all chunk boundaries
    are covered between the ranks -- plus a checksum of the entire buffer, which the caller compares across ranks.
    f32 / f16 inputs are such that the rank-order sum is what every schedule must reproduce within 1e-6 * sum|x| (f32) or
    exactly (f16: k/64)."""
    from oracle import oracle
    R, r = comm.size(), comm.rank()
    es = xmpi.DTYPE_SIZE[dtype]
    per = (count + R - 1) // R
    lo, hi = max(0, r * per - 4096), min(count, (r + 1) * per + 4096)
    got = recv.download(xmpi.NUMPY_DTYPE[dtype], hi - lo, byte_offset=lo * es)
    bad, first = oracle.check_allreduce(got, lo, dtype, xmpi.PAT_UNIFORM, seed0, R, oracle.SUM)
    worst = 0.0
    ok = bad == 0
    if bad and dtype == xmpi.F32:  # another summation order: the per-element tolerance rule over the same range
        ins = [oracle.fill_range(lo, hi - lo, dtype, xmpi.PAT_UNIFORM, seed0 + q).astype(np.float64) for q in range(R)]
        want = oracle.reduce_ranks([x.astype(np.float32) for x in ins], dtype, oracle.SUM).astype(np.float64)
        err = np.abs(got.astype(np.float64) - want)
        worst = float(err.max())
        ok = bool(np.all(err <= 1e-6 * np.sum(np.abs(ins), axis=0)))
    return ok, bad, worst, comm.checksum(recv, count * es), hi - lo


def rank_main(job: Job, grank: int):
    a = job.args
    dtype = DT[a.dtype]
    es = xmpi.DTYPE_SIZE[dtype]
    count = int(a.size_mib * (1 << 20)) // es
    nbytes = count * es
    comm = xmpi.Comm(grank, job.ranks, job.device_of(grank), job.key)
    R = comm.size()
    lead = grank == 0
    seed0 = 1000 if dtype != xmpi.F16 else 2000
    # every process probed the zero-copy path in a job of its own (multi-GPU runs): one veto keeps all staged
    zc_ok = all_max(comm, 0.0 if job.zero_copy_ok else 1.0) == 0.0
    if not zc_ok:
        comm.set_param("zero_copy", 0)
    ZC = (xmpi.ALGO_ZCOPY,) if zc_ok else ()
    send, recv = comm.alloc(nbytes), comm.alloc(nbytes)
    if os.environ.get("XMPI_BENCH_DEBUG"):
        print(f"[bench] rank {grank}: send {send.ptr:#x} (4 KiB slot {(send.ptr >> 12) & 15}) recv {recv.ptr:#x} (slot {(recv.ptr >> 12) & 15})", file=sys.stderr)
    comm.fill(send, count, dtype, xmpi.PAT_UNIFORM, seed0 + grank)
    comm.memset(recv, 0, nbytes)
    with job.lock:
        job.bufs[grank] = (send, recv)

    dsync_can = comm.get_param("dsync") == 1  # one rank per (process, GPU): the ranks may meet on the device
    ndev_used = len({job.device_of(r) for r in range(R)})
    # ---- the links first (multi-GPU only), before anything is tuned: what one xGMI link gives, per engine ----------
    link = None
    other = next((r for r in range(R) if job.device_of(r) != job.device_of(0)), None)
    if other is not None and not a.no_extras and comm.get_param("windows_ok") == 1:  # (the probe copies between the windows)
        probe = {}
        # (2: a kernel that moves data as a step of the stepped kernels does -- system-scope loads / written-through stores: read
        # against write there is the pull form against the push form)
        for eng, name in ((0, "hipMemcpyAsync"), (1, "copy_kernel"), (2, "sys_kernel")):
            comm.barrier()
            if grank == 0:
                probe[name + "_write_GBps"] = comm.link_probe(other, 64 << 20, eng, 10, 0)
                probe[name + "_read_GBps"] = comm.link_probe(other, 64 << 20, eng, 10, 1)
            comm.barrier()
            if grank in (0, other):  # both directions at once
                v = comm.link_probe(other if grank == 0 else 0, 64 << 20, eng, 10, 0)
                if grank == 0:
                    probe[name + "_bidir_each_GBps"] = v
            comm.barrier()
        link = {"between_ranks": [0, other], **probe}

    def run(algo, cnt=count, s=send, r=recv, dt=dtype):
        comm.allreduce(s, r, cnt, dt, xmpi.SUM, algo)

    def run_n(algo, iters):
        comm.allreduce_repeat(send, recv, count, dtype, xmpi.SUM, algo, iters)

    # ---- the schedule: the LIBRARY's (AUTO), unless one was named ---------------------------------------------------
    # Ranks that meet on the device: xmpi_tune times the library's schedules on these very GPUs / links and AUTO follows
    # its table; ranks hosted by threads of one process meet on the host, where AUTO is the zero-copy fold.
    by_name = {v: k for k, v in ALGO_NAME.items()}
    tune = {"by": "library defaults (ranks meet on the host: one schedule)"}
    rejected_here = set()  # allreduce schedules xmpi_tune found wrong on this machine
    if a.algo == "auto":
        algo = xmpi.ALGO_AUTO
        if dsync_can and job.probe_ok is not None:  # what the probe saw failing on this machine is not a candidate
            mask = -1
            for name, bit in TUNE_BIT.items():
                if name not in job.probe_ok:
                    mask &= ~(1 << bit)
            comm.set_param("tune_mask", mask)
            if "split" not in job.probe_ok:
                comm.set_param("dsync_split_bytes", 0)
        if dsync_can:
            t0 = time.perf_counter()
            comm.tune(min(nbytes, 1 << 30))
            cls = max(0, min(23, nbytes.bit_length() - 9))
            tune = {"by": "xmpi_tune (library)", "ms": (time.perf_counter() - t0) * 1e3, "size_class": cls,
                    "algo": ALGO_NAME.get(comm.get_param(f"tune_algo_0_{cls}"), "default"),
                    "split": comm.get_param(f"tune_split_0_{cls}"), "unroll": comm.get_param(f"tune_unroll_0_{cls}"),
                    "table_algo": [comm.get_param(f"tune_algo_0_{k}") for k in range(24)],
                    "table_split": [comm.get_param(f"tune_split_0_{k}") for k in range(24)],
                    "tables_other": {w: [{**ALGO_NAME, xmpi.ALGO_TREE: "tree", xmpi.ALGO_TREE_PUSH: "tree_push"}.get(comm.get_param(f"tune_algo_{ci}_{k}"), "default")
                                         for k in range(0, 24, 2)]
                                     for ci, w in ((1, "allgather"), (2, "bcast"), (3, "reduce"))},  # (per measured size, 1 KiB x4 ...)
                    "probe_ok": sorted(job.probe_ok) if job.probe_ok is not None else None,
                    # what the LIBRARY found wrong on this machine while tuning (every candidate's answer is checked before its time is
                    # believed): left out of AUTO, refused by name -- so nothing below runs them by name either
                    "check_ms": comm.get_param("tune_check_us") / 1e3,
                    "rejected": {w: [CAND_NAME[k] for k in range(11) if comm.get_param(f"tune_rejected_{ci}") >> k & 1]
                                 for ci, w in ((0, "allreduce"), (1, "allgather"), (2, "bcast"), (3, "reduce"))}}
            rejected_here = set(tune["rejected"]["allreduce"])
        # should the library's choice not reproduce the oracle on this machine: the one-kernel fold (no tuned table, nothing
        # split), then the host-driven schedules
        order = [(xmpi.ALGO_AUTO, {})] + ([(xmpi.ALGO_ZCOPY, {"tuned": 0, "dsync_split_bytes": 0})] if zc_ok else []) + \
                [(xmpi.ALGO_DIRECT, {}), (xmpi.ALGO_RING, {"tuned": 0, "dsync": 0} if dsync_can else {})]
    else:
        forced = a.algo if (a.algo not in ("zcopy", "zpush") or zc_ok) else "ring"
        algo = by_name[forced]
        order = [(algo, {})]

    # ---- warmup + parity of the schedule that will be timed, over the WHOLE buffer ------------------------------------
    # (a schedule that is wrong on this machine is reported and the next one takes its place, never timed)
    parity, parity_failures, chosen = {"checked": False}, [], None
    for cand, cand_params in order:
        for k, v in cand_params.items():
            comm.set_param(k, v)
        comm.memset(recv, 0, nbytes)
        for _ in range(max(1, a.warmup)):
            run(cand)
        info, ok_here, csum = {"checked": False}, True, 0
        if dtype in (xmpi.F32, xmpi.F16) and count >= 1 << 16:
            ok_here, nbad, worst, csum, nchecked = check_whole(comm, recv, count, dtype, seed0)
            info = {"checked": True, "ok": ok_here, "elements_checked_by_this_rank": nchecked, "not_bit_identical": nbad,
                    "max_abs_err": worst, "bit_identical": nbad == 0,
                    "coverage": "whole buffer: rank r checks chunk r and the chunk boundaries against the rank-order oracle; "
                                "the ranks' buffers are identical (checksums)",
                    "rule": "bit-exact" if dtype == xmpi.F16 else "|delta| <= 1e-6 * sum_i|x_i| vs rank-order oracle"}
        sums = np.zeros(R, dtype=np.int64)
        comm.allgather(np.array([csum & 0x7FFFFFFFFFFFFFFF], dtype=np.int64), sums, 1, xmpi.I64, xmpi.ALGO_DIRECT)
        same = bool(np.all(sums == sums[0]))
        if all_max(comm, 0.0 if (ok_here and same) else 1.0) == 0.0:
            parity, chosen, algo = info, cand, cand
            break
        parity_failures.append({"algo": ALGO_NAME.get(cand, "auto"), "rank": grank, "local": info, "buffers_identical": same})
    if chosen is None:
        raise AssertionError(f"rank {grank}: no schedule reproduces the oracle: {parity_failures}")
    best = {"algo": ALGO_NAME.get(algo, "auto (library)"), "channels": comm.get_param("channels"),
            "copy_engine": comm.get_param("copy_engine"), "piece_bytes": comm.get_param("piece_bytes")}
    best_ring = None

    # ---- timed region: exactly K steps ----------------------------------------------------------------
    comm.prof_reset()
    # every 4th launch of a kind carries its own begin/end events (hipExtLaunchKernelGGL): the events are
    # exact per dispatch whatever runs around them, and sampling keeps their cost out of `value`
    # (a zero-copy step is ONE launch per GPU process: every 2nd carries events)
    comm.set_param("prof_every", 2 if algo in (xmpi.ALGO_ZCOPY, xmpi.ALGO_AUTO) else (1 if algo == xmpi.ALGO_ZPUSH else 4))
    sl0, sc0 = comm.get_param("dsync_split_launches"), comm.get_param("dsync_sched_launches")
    t_step = timed(comm, None, a.steps, prof=True, batch=lambda k: run_n(algo, k))
    form = ("stepped" if comm.get_param("dsync_sched_launches") > sc0 else
            "split" if comm.get_param("dsync_split_launches") > sl0 else "one kernel")
    prof = {k: comm.prof_get(k) for k in (xmpi.PROF_REDUCE2, xmpi.PROF_REDUCEN, xmpi.PROF_COPY, xmpi.PROF_PEER,
                                          xmpi.PROF_ZCOPY)}
    # the spread of the sampled launches of the timed region, by kind (ns -> us)
    prof_span = {k: (comm.get_param(f"prof_min_ns_{k}") / 1e3, comm.get_param(f"prof_max_ns_{k}") / 1e3)
                 for k in (xmpi.PROF_REDUCE2, xmpi.PROF_REDUCEN, xmpi.PROF_COPY, xmpi.PROF_PEER, xmpi.PROF_ZCOPY)}

    # ---- the box, right behind the timed region and on ITS sixteen buffers: the fold's access pattern without the arithmetic -- R sources
    # -> R destinations in one launch, same grid, same cache policy (xmpi_copy_local_pairs) -- by rank 0 alone, everybody else parked.
    # The same binary ran the fold in 669 ... 763 us on different boxes (r04 / r05): `box_copy_us` says what THIS box's memory gave the
    # pattern on THESE buffers, `frac_of_box` = box_copy_us / the fold's avg launch says how much of it the fold used.
    box = None
    comm.barrier()
    if lead and R > 1 and len(job.bufs) == R and prof[xmpi.PROF_ZCOPY][0]:
        try:
            comm.prof_reset()
            comm.prof_enable(True)
            comm.set_param("prof_every", 1)
            bs, bd = [job.bufs[k][0] for k in range(R)], [job.bufs[k][1] for k in range(R)]
            for j in range(8):
                comm.copy_local_pairs(bd, bs, nbytes)
            n_b, ms_b, by_b = comm.prof_get(xmpi.PROF_ZCOPY)
            box = {"box_copy_us": ms_b * 1e3 / n_b, "box_copy_us_min": comm.get_param(f"prof_min_ns_{xmpi.PROF_ZCOPY}") / 1e3,
                   "box_copy_us_max": comm.get_param(f"prof_max_ns_{xmpi.PROF_ZCOPY}") / 1e3, "box_copy_launches": n_b,
                   "box_copy_bytes_per_launch": by_b / n_b, "box_copy_GBps": by_b / (ms_b * 1e-3) / 1e9}
            comm.prof_enable(False)
        except Exception as e:  # noqa: BLE001  (the line does not depend on it)
            box = {"error": repr(e)[:200]}
    comm.barrier()

    # the same kernel with the GPU to itself (rank 0 only, everybody else parked at a barrier):
    # one ring-step chunk (S / R) per launch
    iso = None
    comm.barrier()
    if lead and R > 1:
        chunk = max(1, count // R)
        comm.prof_reset()
        comm.prof_enable(True)
        comm.set_param("prof_every", 1)
        if algo in (xmpi.ALGO_ZCOPY, xmpi.ALGO_AUTO) and prof[xmpi.PROF_ZCOPY][0]:  # the same N-source, N-destination launch, all operands local
            nz, _, bz = prof[xmpi.PROF_ZCOPY]
            per = int(bz / nz / (2 * R * es)) if nz else chunk  # elements one launch of the timed region folded
            # ON THE TIMED REGION'S OWN BUFFERS when this process hosts every rank: the same launch by rank 0 alone, everybody else
            # parked.  (r04, scripts/r04_gap.py: the SAME kernel on another set of sixteen 256 MiB buffers of the same process runs
            # up to 20 % faster or slower -- 693 vs 832 us with sources and destinations of one set swapped, 693 vs 783 fresh set vs
            # timed set on one box, 813 vs 814 on the next; where the sets lie in physical HBM, which the library cannot see.  A
            # pair "in the collective / isolated" on different buffers compares placements, not launch paths: round 3's 750 vs 686.)
            same = len(job.bufs) == R and per == count
            fresh = None
            if same:
                zs, zd = [job.bufs[k][0] for k in range(R)], [job.bufs[k][1] for k in range(R)]
            else:
                zs = [comm.alloc(per * es) for _ in range(R)]
                zd = [comm.alloc(per * es) for _ in range(R)]
                for k, b in enumerate(zs):
                    comm.fill(b, per, dtype, xmpi.PAT_UNIFORM, 77 + k)
            iso_slots = {"sources": [(b.ptr >> 12) & 15 for b in zs], "destinations": [(b.ptr >> 12) & 15 for b in zd], "timed_buffers": same}
            for j in range(6):
                comm.reduce_local_multi(zd, zs, per, dtype, xmpi.SUM)
            n_i, ms_i, by_i = comm.prof_get(xmpi.PROF_ZCOPY)
            if not same:
                for b in zs + zd:
                    b.free()
            elif not a.no_extras:  # the spread between buffer sets: a fresh set, and the same with the roles swapped
                fs = [comm.alloc(per * es) for _ in range(R)]
                fd = [comm.alloc(per * es) for _ in range(R)]
                for k, b in enumerate(fs):
                    comm.fill(b, per, dtype, xmpi.PAT_UNIFORM, 77 + k)
                fresh = {}
                for name, dd, ss in (("fresh_set_us", fd, fs), ("fresh_set_roles_swapped_us", fs, fd)):
                    comm.prof_reset()
                    for j in range(6):
                        comm.reduce_local_multi(dd, ss, per, dtype, xmpi.SUM)
                    n_f, ms_f, _ = comm.prof_get(xmpi.PROF_ZCOPY)
                    fresh[name] = ms_f * 1e3 / max(1, n_f)
                for b in fs + fd:
                    b.free()
            iso_name = ("reduce_n_multi_kernel (the timed region's own R sources -> R destinations, launched by rank 0 alone, GPU otherwise idle)"
                        if same else "reduce_n_multi_kernel (R local sources -> R local destinations on fresh buffers, GPU otherwise idle)")
        else:
            for j in range(24):  # walk the chunks: operands are cold, as they are inside the collective
                k = j % R
                comm.reduce_local(recv.at(k * chunk * es), send.at(k * chunk * es), send.at(((k + 1) % R) * chunk * es),
                                  chunk, dtype, xmpi.SUM)
            n_i, ms_i, by_i = comm.prof_get(xmpi.PROF_REDUCE2)
            iso_name = "reduce2_kernel (one ring-step chunk, GPU otherwise idle)"
        comm.prof_enable(False)
        iso = {"kernel": iso_name, "bytes_per_launch": by_i / n_i, "slots": locals().get("iso_slots"), "other_buffer_sets": locals().get("fresh"),
               "avg_launch_us": ms_i * 1e3 / n_i, "achieved": by_i / (ms_i * 1e-3) / 1e9, "unit": "GB/s",
               "frac": by_i / (ms_i * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    comm.barrier()

    # ---- north_star's target is quoted on RING: where the ranks talk over links, the ring kernel is timed by name beside the
    # library's own choice -- both forms (a step LOADS its operand over the link / a step STORES its result over the link),
    # `steps` iterations each, the result checked against the oracle like the headline's
    ring_named = None
    if ndev_used > 1 and R > 1 and dsync_can and dtype in (xmpi.F32, xmpi.F16) and count >= 1 << 16:
        ring_named = {}
        ok_forms = (set(job.probe_ok) if job.probe_ok is not None else {"ring", "ring_push"}) - rejected_here
        for name, al in (("pull", xmpi.ALGO_RING), ("push", xmpi.ALGO_RING_PUSH)):
            if ALGO_NAME[al] not in ok_forms:
                continue
            try:
                comm.memset(recv, 0, nbytes)
                run(al)
                ok_r, nbad, worst, csum, _ = check_whole(comm, recv, count, dtype, seed0)
                ok_r = all_max(comm, 0.0 if ok_r else 1.0) == 0.0
                t_r = timed(comm, None, a.steps, batch=lambda k, a2=al: run_n(a2, k))
                ring_named[name] = {"ms_per_step": t_r * 1e3, "algbw_GBps": nbytes / t_r / 1e9, "busbw_GBps": nbytes / t_r / 1e9 * 2 * (R - 1) / R,
                                    "parity_ok": ok_r, "max_abs_err": worst, "steps": a.steps}
            except Exception as e:  # noqa: BLE001  (the line does not depend on it)
                ring_named[name] = {"error": repr(e)[:200]}
        comm.barrier()

    # what the bytes actually cross: only ranks on DIFFERENT devices talk over xGMI (a multi-process rehearsal on
    # one GPU does not, whatever --gpus says)
    transport = ("xGMI (one rank per GPU)" if ndev_used == R else
                 "intra-HBM (all ranks share one GPU)" if ndev_used == 1 else f"mixed: {R} ranks on {ndev_used} GPUs (intra-HBM + xGMI)")
    slot = lambda b: (b.ptr >> 12) & 15  # noqa: E731 -- the 4 KiB slot of the 64 KiB frame a block starts in (heap.cpp colouring)
    out = {"degraded": {"level": comm.get_param("degraded"), "why": comm.degraded()},
           "slots": {"send": slot(send), "recv": slot(recv)}, "form": form, "link": link, "transport": transport, "devices": ndev_used, "dsync": comm.get_param("dsync"), "t_step": t_step, "prof": prof, "tune": tune, "best": best, "best_ring": best_ring, "ring_named": ring_named, "parity": parity,
           "algo": algo, "nbytes": nbytes, "count": count, "iso": iso, "box": box, "prof_span": prof_span, "parity_failures": parity_failures,
           "shared_stream": comm.get_param("shared_stream"), "slot_bytes": comm.get_param("slot_bytes"),
           "zero_copy_probe": {"dsync": "ok: ranks meet on the device" if comm.get_param("dsync") == 1 else "ok (ranks that share a process meet on the host)",
                               "host": "device rendezvous failed, host rendezvous ok",
                               "failed": "failed: staged schedules only", "not run": "ok"}[job.probe] if zc_ok else "failed: staged schedules only"}

    # ---- untimed extras: what the next round needs to tune blind multi-GPU runs -----------------------
    extras = {}
    out["extras"] = extras
    with job.lock:  # the line's figures are safe from here on: whatever the untimed extras do, rank 0 can print them
        job.result[grank] = out
    # On real links only what the probe saw working is run by name (a schedule that hangs there would take the run's result
    # with it); DIRECT -- a host-driven step table -- stays on one GPU, where round 1 and 2 validated it.
    multi = ndev_used > 1
    safe = (set(job.probe_ok) if job.probe_ok is not None else {"fused", "split", "zpush", "ring", "rhd", "ring_push", "rhd_push"}) - rejected_here
    named = [al for al, nm in ((xmpi.ALGO_RING, "ring"), (xmpi.ALGO_RHD, "rhd"), (xmpi.ALGO_ZPUSH, "zpush")) if not multi or nm in safe]
    try:
        if not a.no_extras and R > 1:
            extras["algos_at_size"] = {}
            for al in [x for x in named if x != xmpi.ALGO_ZPUSH] + ([] if multi else [xmpi.ALGO_DIRECT]) + ([x for x in ZC_ALGOS if x == xmpi.ALGO_ZCOPY or x in named] if zc_ok else []):
                if al == xmpi.ALGO_RHD and R & (R - 1):
                    continue
                run(al)
                t = timed(comm, None, 3, batch=lambda k, a2=al: run_n(a2, k))
                extras["algos_at_size"][ALGO_NAME[al]] = {"ms": t * 1e3, "algbw_GBps": nbytes / t / 1e9,
                                                          "busbw_GBps": nbytes / t / 1e9 * 2 * (R - 1) / R}
            sweep = []
            sz = 1 << 10
            while sz <= min(nbytes, 1 << 30):
                cnt = sz // es
                row = {"bytes": sz}
                for al in ([xmpi.ALGO_RING] if xmpi.ALGO_RING in named else []) + ([] if multi else [xmpi.ALGO_DIRECT]) + list(ZC):
                    run(al, cnt)
                    # inside one native call: what the library costs, not what 8 Python threads cost each other
                    t = timed(comm, None, 10 if sz <= (16 << 20) else 3,
                              batch=lambda k, a2=al, c2=cnt: comm.allreduce_repeat(send, recv, c2, dtype, xmpi.SUM, a2, k))
                    row[ALGO_NAME[al] + "_us"] = t * 1e6
                    row[ALGO_NAME[al] + "_busbw_GBps"] = sz / t / 1e9 * 2 * (R - 1) / R
                sweep.append(row)
                sz *= 4
            extras["size_sweep"] = sweep
            # BASELINE.json's metric: busbw against message size at 1 / 2 / 4 / 8 ranks.  Smaller communicators among the
            # first ranks of this job (the others wait at the barrier); the schedule is the library's own choice (AUTO).
            # One rank: an allreduce is a copy, busbw is defined 0, algbw is the figure.  cfg 3 rides along at 4 ranks.
            table = []
            sizes = []
            sz = 1 << 10
            while sz <= min(nbytes, 1 << 30):
                sizes.append(sz)
                sz *= 4
            for r2 in (1, 2, 4, R):
                if r2 > R:
                    continue
                comm.barrier()
                if grank < r2:
                    sub = comm if r2 == R else xmpi.Comm(grank, r2, job.device_of(grank), f"{job.key}-r{r2}")
                    s2, d2 = (send, recv) if r2 == R else (sub.alloc(nbytes), sub.alloc(nbytes))
                    if r2 != R:
                        sub.fill(s2, count, dtype, xmpi.PAT_UNIFORM, seed0 + grank)
                    for b in sizes:
                        cnt = b // es
                        sub.allreduce(s2, d2, cnt, dtype, xmpi.SUM, xmpi.ALGO_AUTO)
                        # (the same method as the headline: a.steps launches inside ONE native call between two rendezvous; with 3
                        # the start-up of the batch showed as a 10 % gap between the 8-rank row and `value`, r03)
                        it = max(10, a.steps)
                        t = timed(sub, None, it, batch=lambda k, c=cnt: sub.allreduce_repeat(s2, d2, c, dtype, xmpi.SUM, xmpi.ALGO_AUTO, k))
                        if grank == 0:
                            table.append({"ranks": r2, "bytes": b, "us": t * 1e6, "algbw_GBps": b / t / 1e9,
                                          "busbw_GBps": b / t / 1e9 * 2 * (r2 - 1) / r2})
                    if r2 == 4:  # BASELINE cfg 3: allgather int64, 16 MiB per rank, 4 ranks
                        cnt3 = min(2097152, nbytes // 8 // 4)
                        row3 = {"ranks": 4, "bytes_per_rank": cnt3 * 8}
                        for al in (xmpi.ALGO_AUTO, xmpi.ALGO_RING):
                            sub.allgather(s2, d2, cnt3, xmpi.I64, al)
                            t = timed(sub, lambda: sub.allgather(s2, d2, cnt3, xmpi.I64, al), 5)
                            row3["auto" if al == xmpi.ALGO_AUTO else "ring"] = {
                                "ms": t * 1e3, "algbw_GBps": cnt3 * 8 * 4 / t / 1e9, "busbw_GBps": cnt3 * 8 * 4 / t / 1e9 * 3 / 4}
                        if grank == 0:
                            extras["cfg3_allgather_i64_16MiB_4ranks"] = row3
                    if r2 != R:
                        s2.free()
                        d2.free()
                        sub.finalize()
                comm.barrier()
            extras["busbw_table"] = {"schedule": "AUTO (the library's choice)", "unit_note": "GB = 1e9 B; busbw = algbw x 2(R-1)/R; 1 rank: busbw is 0 by definition, algbw is a device copy",
                                     "rows": table}
            # BASELINE cfg 2: 1 MiB float32 ping-pong between ranks 0 and 1 (half round trip)
            n1 = 262144
            if grank in (0, 1):
                peer = 1 - grank
                iters = 50
                for w in range(5 + iters):
                    if w == 5:
                        t0 = time.perf_counter()
                    if grank == 0:
                        comm.send(send, n1, xmpi.F32, peer, 3)
                        comm.recv(recv, n1, xmpi.F32, peer, 3)
                    else:
                        comm.recv(recv, n1, xmpi.F32, peer, 3)
                        comm.send(recv, n1, xmpi.F32, peer, 3)
                half = (time.perf_counter() - t0) / iters / 2
                extras["bounce_1MiB_f32"] = {"half_round_trip_us": half * 1e6, "GBps": n1 * 4 / half / 1e9}
                # the reference's own sweep (examples/bounce/bounce.go:33: message lengths 0 ... 1e7 bytes, 10 repetitions,
                # bounce.go:140-151 prints the mean round trip per length)
                sweep_b = []
                for length in (0, 1, 10, 100, 1000, 10**4, 10**5, 10**6, 10**7):
                    if length > nbytes:
                        break
                    reps = 10
                    for w in range(2 + reps):
                        if w == 2:
                            t0 = time.perf_counter()
                        if grank == 0:
                            comm.send(send, length, xmpi.U8, peer, 4)
                            comm.recv(recv, length, xmpi.U8, peer, 4)
                        else:
                            comm.recv(recv, length, xmpi.U8, peer, 4)
                            comm.send(recv, length, xmpi.U8, peer, 4)
                    rt = (time.perf_counter() - t0) / reps
                    sweep_b.append({"bytes": length, "round_trip_us": rt * 1e6, "GBps": 2 * length / rt / 1e9})
                extras["bounce_sweep_u8"] = sweep_b
                # the same with HOST slices on both sides -- what bounce.go itself passes (bounce.go:96) -- through the host lanes
                # of the job's shared segment (never `value`: no HBM in it)
                hs, hr = np.arange(10**7, dtype=np.uint8), np.zeros(10**7, dtype=np.uint8)
                sweep_h = []
                for length in (1, 10, 100, 1000, 10**4, 10**5, 10**6, 10**7):
                    reps = 10
                    for w in range(2 + reps):
                        if w == 2:
                            t0 = time.perf_counter()
                        if grank == 0:
                            comm.send(hs[:length], length, xmpi.U8, peer, 6)
                            comm.recv(hr[:length], length, xmpi.U8, peer, 6)
                        else:
                            comm.recv(hr[:length], length, xmpi.U8, peer, 6)
                            comm.send(hr[:length], length, xmpi.U8, peer, 6)
                    rt = (time.perf_counter() - t0) / reps
                    sweep_h.append({"bytes": length, "round_trip_us": rt * 1e6, "GBps": 2 * length / rt / 1e9})
                if grank == 0 and not np.array_equal(hs, hr):
                    parity_failures.append("bounce of host slices: echo differs")
                extras["bounce_sweep_u8_host_slices"] = sweep_h
                extras["bounce_sweep_note"] = ("both sweeps run between two PYTHON threads of this process (N = 1) or two Python processes: below 100 KB the "
                                               "figures are interpreter hand-offs, not the library -- examples/bounce.cpp through the launcher: 2 us (host "
                                               "slices), 8-9 us (HBM, ping-pong proper), profiles/r03/bounce_example*.txt, bounce_2proc_*.json")
            comm.barrier()
            # BASELINE cfg 5: allreduce-sum fp16 up to 1 GiB per rank, recursive halving vs ring (and the library's own
            # choice) over sizes 1 MiB ... 1 GiB; exactly summable inputs k/64: every schedule must be bit-identical to
            # the rank-order result.  >= 5 timed iterations per point.
            if dtype == xmpi.F32 and nbytes >= (256 << 20):
                n5 = (1 << 30) // 2
                s5, r5, ref5 = comm.alloc(n5 * 2), comm.alloc(n5 * 2), comm.alloc(n5 * 2)
                comm.fill(s5, n5, xmpi.F16, xmpi.PAT_UNIFORM, 2000 + grank)
                rows5 = []
                b5 = 1 << 20
                while b5 <= (1 << 30):
                    c5 = b5 // 2
                    comm.allreduce(s5, ref5, c5, xmpi.F16, xmpi.SUM, xmpi.ALGO_ZCOPY if multi else xmpi.ALGO_DIRECT)  # rank order either way
                    row = {"bytes": b5}
                    for al, name in ((xmpi.ALGO_RING, "ring"), (xmpi.ALGO_RHD, "rhd"), (xmpi.ALGO_AUTO, "auto")):
                        if (al == xmpi.ALGO_RHD and R & (R - 1)) or (al != xmpi.ALGO_AUTO and al not in named):
                            continue
                        comm.allreduce(s5, r5, c5, xmpi.F16, xmpi.SUM, al)
                        same = comm.count_mismatch(r5, ref5, b5) == 0
                        t = timed(comm, None, 5, batch=lambda k, a5=al, cc=c5: comm.allreduce_repeat(s5, r5, cc, xmpi.F16, xmpi.SUM, a5, k))
                        row[name] = {"ms": t * 1e3, "algbw_GBps": b5 / t / 1e9, "busbw_GBps": b5 / t / 1e9 * 2 * (R - 1) / R,
                                     "bit_identical_to_rank_order": same}
                    rows5.append(row)
                    b5 *= 4
                extras["cfg5_allreduce_f16_sweep"] = {"ranks": R, "iterations": 5, "rows": rows5,
                                                      "layout": "rank threads of one process: ring / rhd are HOST-DRIVEN step tables here (the stepped kernels "
                                                                "need ranks that meet on the device: cfg5_allreduce_f16_sweep_one_process_per_rank)"}
                for b in (s5, r5, ref5):
                    b.free()
            # The headline's call with HOST slices -- what a program written against the reference passes (a []float32 in host memory,
            # mpi.go:126-128 / helloworld.go:53-81): every rank's buffer goes up and its result comes down over the GPU's PCIe link INSIDE
            # the call.  The PCIe-inclusive rate, reported beside `value` and never as `value`.  Pageable memory, as a Go
            # slice is; all R ranks share this GPU's one link.
            if dtype == xmpi.F32 and not multi and not a.no_host_leg:
                run(algo)
                want_h = recv.download(np.float32, count)
                hx = send.download(np.float32, count)
                ho = np.zeros(count, dtype=np.float32)
                comm.allreduce(hx, ho, count, dtype, xmpi.SUM, xmpi.ALGO_AUTO)  # (warm: staging blocks, first touch of `ho`)
                t_h = timed(comm, lambda: comm.allreduce(hx, ho, count, dtype, xmpi.SUM, xmpi.ALGO_AUTO), 2)
                err_h = float(np.max(np.abs(ho - want_h))) if count else 0.0
                ok_h = all_max(comm, 0.0 if err_h <= 1e-6 * R else 1.0) == 0.0  # (|delta| <= 1e-6 x sum_i |x_i|, x in [0, 1): at most 1e-6 x R)
                if not ok_h:
                    parity_failures.append({"host_slices_allreduce": {"rank": grank, "max_abs_err": err_h}})
                extras["host_slices_allreduce"] = {"bytes_per_rank": nbytes, "ranks": R, "ms_per_step": t_h * 1e3, "algbw_GBps": nbytes / t_h / 1e9,
                                                   "pcie_bytes_per_step_each_direction": nbytes * R, "pcie_GBps_each_direction": nbytes * R / t_h / 1e9,
                                                   "memory": "pageable (numpy), as a Go slice", "steps": 2, "parity_ok": ok_h, "max_abs_err": err_h,
                                                   "bit_identical_to_the_device_run": bool(np.array_equal(ho, want_h))}
                del hx, ho, want_h
            if link is not None:
                extras["xgmi_link_probe"] = link
    except Exception as e:  # noqa: BLE001  (an extra that fails must not cost the line)
        extras["error"] = repr(e)[:300]

    try:  # (a peer that failed in its extras no longer arrives: the figures above are stored already)
        comm.barrier()
        send.free()
        recv.free()
        comm.finalize()
    except Exception as e:  # noqa: BLE001
        extras.setdefault("error", repr(e)[:300])
    _ = lead


def free_ports(n: int):
    """n TCP ports nobody listens on right now, below the ephemeral range (where other processes' outgoing connections come and go)"""
    import random
    import socket
    ports = []
    for _ in range(400):
        p = random.randint(10000, 30000)
        if p in ports:
            continue
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            try:
                sk.bind(("", p))
            except OSError:
                continue
        ports.append(p)
        if len(ports) == n:
            break
    return [f":{p}" for p in ports]


def refpath_ranks(mode: str, ranks: int, args, timeout: int):
    """`ranks` processes of oracle/refpath_bin on localhost; (outputs, None) or (None, what went wrong).  A port lost between the
    check and a rank's listen is another set of ports, not a failure."""
    binp = os.path.join(ROOT, "oracle", "refpath_bin")
    outs = []
    for attempt in range(3):
        ports = free_ports(ranks)
        procs = [subprocess.Popen([binp, mode, "-mpi-addr", p, "-mpi-alladdr", ",".join(ports), *args],
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for p in ports]
        outs = []
        for p in procs:
            try:
                outs.append(p.communicate(timeout=timeout)[0])
            except subprocess.TimeoutExpired:
                p.kill()
                outs.append(p.communicate()[0] + " (timed out)")
        if all(p.returncode == 0 for p in procs):
            return outs, None
        if not any("listen failed" in o for o in outs):
            break
    return None, next((o for o, p in zip(outs, procs) if p.returncode != 0), "")[-300:]


def cpu_baseline(ranks: int, count: int, reps: int = 2):
    """oracle/refpath_bin: the reference's TCP+gob path, `ranks` processes on localhost"""
    if not os.path.exists(os.path.join(ROOT, "oracle", "refpath_bin")):
        return None
    import resource
    ru0 = resource.getrusage(resource.RUSAGE_CHILDREN)
    t0 = time.perf_counter()
    outs, err = refpath_ranks("allreduce_f32", ranks, [str(count), str(reps)], 900)
    if outs is None:
        return {"error": err}
    rows = [json.loads(out.strip().split("\n")[-1]) for out in outs]
    wall = time.perf_counter() - t0
    ru1 = resource.getrusage(resource.RUSAGE_CHILDREN)
    cpu_s = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
    t = max(r["mean_s"] for r in rows)
    s = count * 4
    # value: the same figure as the line's `value` -- algbw = S / t.  `cores`: what the sample actually kept busy
    # (CPU seconds of the rank processes / wall seconds); the processes are not pinned and have 2 x ranks threads
    # each (one per concurrent Send / Receive, as the reference's goroutines), most of them blocked on their sockets
    return {"value": s / t / 1e9, "unit": "GB/s", "cores": round(max(1.0, cpu_s / wall), 1), "kind": "port",
            "host_cores": os.cpu_count(), "threads": ranks * 2 * ranks, "cpu_seconds": cpu_s,
            "busbw_GBps": s / t / 1e9 * 2 * (ranks - 1) / ranks, "ranks": ranks, "seconds_per_allreduce": t, "wall_s": wall,
            "sample": f"allreduce-sum f32, {s >> 20} MiB per rank, {ranks} ranks (one OS process each, unpinned), "
                      f"{reps} repetitions; loopback TCP + gob framing, all-to-all exchange + rank-order host sum "
                      f"(oracle/refpath.cpp restating network.go:518-625; no Go toolchain in the image)"}


def cpu_by_ranks(cb, ranks: int, count: int, sub=(2, 4)):
    """cpu_baseline + `by_ranks`: the reference path at every rank count of the metric (1 rank: no message, nothing to time) -- the
    full communicator's entry is the sample already taken; the smaller ones one repetition each"""
    if not isinstance(cb, dict) or "value" not in cb:
        return cb
    by = {}
    for r in sorted({x for x in sub if 1 < x < ranks}):
        row = cpu_baseline(r, count, 1)
        by[str(r)] = ({"algbw_GBps": round(row["value"], 5), "busbw_GBps": round(row["busbw_GBps"], 5), "seconds_per_allreduce": round(row["seconds_per_allreduce"], 4),
                       "cores": row["cores"], "repetitions": 1} if isinstance(row, dict) and "value" in row else {"error": (row or {}).get("error", "no result")[:120]})
    by[str(ranks)] = {"algbw_GBps": round(cb["value"], 5), "busbw_GBps": round(cb["busbw_GBps"], 5)}  # (the sample above: its other figures stand there)
    cb["by_ranks"] = by
    return cb


def live_pmc_traffic(args, kernel_stem: str):
    """HBM bytes per launch of the dominant kernel, MEASURED IN THIS RUN: two child passes of this very bench (the same workload, the
    zero-copy fold by name, no extras) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- separate passes with
    --kernel-trace only, as MI355X_MICROARCH.md's HBM section prescribes -- and its gfx950 correction: traffic = (2 x FETCH_SIZE +
    WRITE_SIZE) x 1024.  None where rocprofv3 is not installed, a profiler is already attached to this process, or a pass fails:
    the line then quotes the committed profile (profiles/pmc_traffic.json) and says so."""
    import csv
    import glob
    import shutil
    import tempfile
    if shutil.which("rocprofv3") is None or os.environ.get("ROCP_TOOL_LIBRARIES"):
        return None
    out = {}
    work = tempfile.mkdtemp(prefix="xmpi_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(work, counter)
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                   "--algo", "zcopy", "--no-extras", "--no-cpu", "--no-production", "--no-pmc", "--steps", "5", "--warmup", "2",
                   "--size-mib", f"{args.size_mib:g}", "--dtype", args.dtype]
            env = dict(os.environ, TMPDIR="/tmp", XMPI_BENCH_KEY=f"bench-pmc-{counter}-{os.getpid()}", XMPI_BENCH_EXTRAS_DIR=work)
            p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
            if p.returncode != 0:
                return {"error": f"rocprofv3 --pmc {counter}: exit {p.returncode}: {(p.stderr or p.stdout)[-200:]}"}
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] == counter and kernel_stem in r["Kernel_Name"]:
                        vals.append(float(r["Counter_Value"]))
            if not vals:
                return {"error": f"rocprofv3 --pmc {counter}: no launch of {kernel_stem} in the trace"}
            out[counter] = (sum(vals) / len(vals), len(vals))
    except (OSError, subprocess.TimeoutExpired, KeyError, ValueError) as e:
        return {"error": repr(e)[:200]}
    finally:
        shutil.rmtree(work, ignore_errors=True)
    (f, nf), (w, nw) = out["FETCH_SIZE"], out["WRITE_SIZE"]
    return {"traffic_bytes_per_launch": (2.0 * f + w) * 1024.0, "FETCH_SIZE_KiB_mean": f, "WRITE_SIZE_KiB_mean": w, "launches": min(nf, nw)}


def multiprocess_sweep(ranks: int):
    """examples/coll_sweep through the launcher: ONE OS PROCESS PER RANK (the production layout), all on this box's
    GPU(s).  Processes meet on the device (flag words in HBM) -- the figure the rank-threads of this bench cannot give."""
    run = os.path.join(ROOT, "mpi_amd", "bin", "xmpirun")
    prog = os.path.join(ROOT, "mpi_amd", "bin", "coll_sweep")
    if not (os.path.exists(run) and os.path.exists(prog)):
        return None
    env = dict(os.environ, XMPI_TIMEOUT_S="60", XMPI_BASEPORT=str(7000 + os.getpid() % 1000 * 16))
    env.pop("XMPI_SLOT_BYTES", None)
    env.pop("XMPI_FIFO_DEPTH", None)
    try:
        p = subprocess.run([run, str(ranks), prog, str(16 << 20), "200"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           text=True, timeout=240)
    except subprocess.TimeoutExpired:
        return {"error": "timed out"}
    if p.returncode != 0:
        return {"error": (p.stderr or p.stdout)[-400:]}
    try:
        return json.loads(p.stdout.strip().split("\n")[-1])
    except ValueError:
        return {"error": p.stdout[-400:]}


def cpu_bounce():
    """oracle/refpath_bin bounce: the reference's ping-pong (bounce.go:83-151) over loopback TCP + gob, 2 processes"""
    if not os.path.exists(os.path.join(ROOT, "oracle", "refpath_bin")):
        return None
    outs, err = refpath_ranks("bounce", 2, ["10000000", "10"], 300)
    if outs is None:
        return {"error": err}
    row = json.loads(outs[0].strip().split("\n")[-1])
    lens = (0, 1, 10, 100, 1000, 10**4, 10**5, 10**6, 10**7)
    return [{"bytes": n, "round_trip_us": us, "GBps": 2 * n / (us * 1e-6) / 1e9 if us > 0 else 0.0}
            for n, us in zip(lens, row["bytes_us"])]


PROBE_FORMS = (("fused", xmpi.ALGO_ZCOPY, {"dsync_split_bytes": 0}), ("split", xmpi.ALGO_ZCOPY, {"dsync_split_bytes": 1}),
               ("zpush", xmpi.ALGO_ZPUSH, {}), ("ring", xmpi.ALGO_RING, {}), ("rhd", xmpi.ALGO_RHD, {}),
               ("ring_push", xmpi.ALGO_RING_PUSH, {}), ("rhd_push", xmpi.ALGO_RHD_PUSH, {}))
TUNE_BIT = {"split": 2, "zpush": 3, "ring": 4, "rhd": 5, "ring_push": 7, "rhd_push": 8}  # xmpi_set_param("tune_mask"): candidate numbers of xmpi_tune
CAND_NAME = {0: "fused", 1: "fused2", 2: "split", 3: "zpush", 4: "ring", 5: "rhd", 6: "ll", 7: "ring_push", 8: "rhd_push", 9: "tree", 10: "tree_push"}
STEPPED = ("ring", "rhd", "ring_push", "rhd_push")


def probe_rank(job: Job, grank: int):
    """--probe: every schedule the library may choose, ONE communicator each in a child process: a schedule that FAULTS or HANGS on
    this machine takes its own job with it, not the bench's.  Three 16 MiB allreduces per form, no look at the bits: whether a
    schedule's ANSWERS are right here is the library's own business (xmpi_tune checks every candidate on patterned inputs and the
    run reads `tune_rejected`; a caller behind mpi.Init() has no bench.py).  The first form (the one-kernel fold) must run: its
    failure is the probe's exit status.  The others are reported on stdout -- `PROBE_OK fused,split,...` -- after all ranks agreed."""
    if os.environ.get("XMPI_BENCH_FAIL_PROBE"):  # rehearsal of the fallback
        raise AssertionError("probe failure forced by XMPI_BENCH_FAIL_PROBE")
    count = 4 << 20
    good = []
    for name, algo, params in PROBE_FORMS:
        if algo in (xmpi.ALGO_RHD, xmpi.ALGO_RHD_PUSH) and job.ranks & (job.ranks - 1):
            continue
        if name == "ring" and os.environ.get("XMPI_BENCH_FAIL_RING"):  # rehearsal: a schedule that does not work here
            continue
        ok = False
        try:
            comm = xmpi.Comm(grank, job.ranks, job.device_of(grank), f"{job.key}-{name}")
            if name != "fused" and comm.get_param("dsync") != 1:
                comm.finalize()
                break  # ranks meet on the host: there is one schedule
            for k, v in params.items():
                comm.set_param(k, v)
            send, recv = comm.alloc(count * 4), comm.alloc(count * 4)
            comm.fill(send, count, xmpi.F32, xmpi.PAT_SIGNED, 4000 + grank)
            comm.memset(recv, 0, count * 4)
            for _ in range(3):
                comm.allreduce(send, recv, count, xmpi.F32, xmpi.SUM, algo)
            went_staged = comm.get_param("zc_fallbacks_unregistered") + comm.get_param("zc_fallbacks_unmappable")
            comm.sync()
            ok = name in STEPPED or went_staged == 0  # (it ran, and as the form it was named as)
            ok = all_max(comm, 0.0 if ok else 1.0) == 0.0  # every rank, or nobody
            comm.barrier()
            comm.finalize()
        except Exception as e:  # noqa: BLE001
            sys.stderr.write(f"[probe] rank {grank}: {name}: {e}\n")
            ok = False
        if ok:
            good.append(name)
        elif name == "fused":
            raise AssertionError(f"rank {grank}: the zero-copy fold failed the probe")
    if grank == job.my_ranks()[0]:
        print("PROBE_OK " + ",".join(good), flush=True)


def probe_zero_copy(job: Job) -> str:
    """The zero-copy kernels load and store through xGMI-mapped peer memory, and with one process per GPU the ranks
    meet inside those kernels (flag words in HBM).  A MAPPING the runtime refuses is the library's business: xmpi_init votes
    and runs the job at the best level every rank reached (the line's `degraded`).  A schedule that gives WRONG BITS on this node is
    the library's business too: xmpi_tune checks every candidate's answer before it believes its time and the run reads
    `tune_rejected` (config.tuned.rejected).  What is left for the bench is a schedule that FAULTS or HANGS on a node this code has
    not run on before: every schedule is run in a job of its own (child processes) first; if the one-kernel fold fails with the
    ranks meeting on the device, once more with them meeting on the host (XMPI_DSYNC=0); if that fails too, this run keeps to the
    staged schedules instead of dying without a result.
    Returns "dsync" | "host" | "failed"; job.probe_ok = the schedules that ran (the library's tuner is told to leave the others
    out)."""
    for attempt, extra in (("dsync", {}), ("host", {"XMPI_DSYNC": "0"})):
        env = dict(os.environ, XMPI_BENCH_KEY=f"{job.key}-probe-{attempt}", XMPI_TIMEOUT_S="30", **extra)
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(job.args.gpus), "--ranks", str(job.ranks), "--probe"]
        try:
            p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=400)
        except subprocess.TimeoutExpired:
            sys.stderr.write(f"[bench] zero-copy probe ({attempt}) timed out\n")
            continue
        if p.returncode == 0:
            for ln in (p.stdout or "").split("\n"):
                if ln.startswith("PROBE_OK "):
                    job.probe_ok = set(ln[len("PROBE_OK "):].strip().split(","))
            return attempt
        sys.stderr.write(f"[bench] zero-copy probe ({attempt}) failed (exit {p.returncode}):\n{(p.stdout or '')[-1500:]}\n")
    sys.stderr.write("[bench] keeping to the staged schedules\n")
    return "failed"


def main():
    args = parse_args()
    sys.setswitchinterval(1e-4)  # rank threads hand the GIL over quickly between their (GIL-free) C calls
    job = Job(args)
    if args.probe:
        threads = [threading.Thread(target=_guard, args=(job, g, probe_rank)) for g in job.my_ranks()]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for g, tb in job.errors:
            sys.stderr.write(f"[probe] rank {g}:\n{tb}\n")
        sys.exit(1 if job.errors else 0)
    # One OS process per rank on this box's GPU, before THIS process opens the GPU: a GPU schedules the queues of at most
    # 8 processes at once, and a ninth (this one, with its rank threads) would have the other eight time-sliced
    # (afterwards would not do -- measured 862 us instead of 35: hipDeviceReset does not give the queues back)
    job.mp_sweep = job.production = None
    if args.gpus == 1 and job.procs == 1 and not args.probe and not args.no_production:
        job.production = production_layout(job.ranks, int(args.size_mib * (1 << 20)), args.steps, args.warmup)
    job.cfg5 = job.cfg3 = None
    if args.gpus == 1 and job.procs == 1 and not args.no_extras and not args.probe:
        job.mp_sweep = multiprocess_sweep(job.ranks)
        job.cfg3 = cfg3_production()
        if args.size_mib >= 256:
            job.cfg5 = cfg5_production(job.ranks, 1 << 30)
    if (args.gpus > 1 or job.procs > 1) and not args.no_probe:
        job.probe = probe_zero_copy(job)
        job.zero_copy_ok = job.probe != "failed"
        if job.probe == "host":
            os.environ["XMPI_DSYNC"] = "0"  # the communicators of this run are created after this point
    threads = [threading.Thread(target=_guard, args=(job, g)) for g in job.my_ranks()]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for g, tb in job.errors:
        sys.stderr.write(f"[bench] rank {g} failed:\n{tb}\n")
    if any(g not in job.result for g in job.my_ranks()):
        sys.exit(1)  # a rank of this process never got through the timed region: there is no figure to report
    if 0 not in job.result:
        return  # not the process hosting rank 0
    r0 = job.result[0]
    R, S, t = job.ranks, r0["nbytes"], r0["t_step"]
    algbw = S / t / 1e9
    busbw = algbw * 2 * (R - 1) / R
    n2, ms2, b2 = r0["prof"][xmpi.PROF_REDUCE2]
    nn, msn, bn = r0["prof"][xmpi.PROF_REDUCEN]
    nz, msz, bz = r0["prof"][xmpi.PROF_ZCOPY]
    if nz and msz >= max(ms2, msn) and r0["best"]["algo"] == "zpush" and r0["dsync"] == 1:
        kname, launches, ms, by = ("dsync_fold_kernel<uint8_t,SUM,1> (push-only, kernel 1: contributions stored into the peers' own blocks) + "
                                   f"dsync_fold_kernel / dsync_body_kernel<float,SUM,{R}> with local sources (kernel 2: rank-order fold of what landed, "
                                   "stored into every receive buffer); bytes and time summed over the two"), nz, msz, bz
    elif nz and msz >= max(ms2, msn) and r0["best"]["algo"] == "zpush":
        kname, launches, ms, by = ("zero-copy push pipeline: copy_batch_kernel (contributions into the peers' receive buffers) + "
                                   f"reduce_n_multi_kernel<float,SUM,{R}> (local rank-order fold) + copy_multi_kernel (results to "
                                   "all peers); bytes and time summed over the three"), nz, msz, bz
    elif nz and msz >= max(ms2, msn) and r0["dsync"] == 1 and r0["form"] == "stepped":
        kname, launches, ms, by = (f"dsync_sched_kernel<float,SUM> (a stepped schedule -- ring or recursive halving + doubling -- inside ONE "
                                   "kernel per rank: every step released by the peer's flag word; its duration includes waiting)"), nz, msz, bz
    elif nz and msz >= max(ms2, msn) and r0["dsync"] == 1 and r0["form"] == "split":
        kname, launches, ms, by = (f"dsync_body_kernel<float,SUM,{R}> (the fold of chunk j of the {R} send buffers in rank order into the {R} "
                                   "receive buffers, between the one-block meet and done kernels: no waiting inside)"), nz, msz, bz
    elif nz and msz >= max(ms2, msn) and r0["dsync"] == 1:
        kname, launches, ms, by = (f"dsync_fold_kernel<float,SUM,{R}> (one kernel per rank IS the allreduce: rendezvous through flag "
                                   f"words in HBM, fold of chunk j of the {R} send buffers in rank order into the {R} receive buffers, "
                                   "completion exchange; its duration includes waiting for the slowest peer)"), nz, msz, bz
    elif nz and msz >= max(ms2, msn):
        kname, launches, ms, by = (f"reduce_n_multi_kernel<float,SUM,{R}> (zero-copy allreduce: folds chunk j of the {R} send "
                                   f"buffers in rank order, stores it into the {R} receive buffers)"), nz, msz, bz
    elif ms2 >= msn and n2:
        kname, launches, ms, by = ("reduce2_batch_kernel<float,SUM> (fused ring step: a + slot -> next rank's slot [+ local]) / "
                                   "reduce2_kernel<float,SUM>"), n2, ms2, b2
    elif nn:
        kname, launches, ms, by = "reduce_n_kernel<float,SUM,N> (rank-order fold)", nn, msn, bn
    else:
        nc, msc, bc = r0["prof"][xmpi.PROF_COPY]
        kname, launches, ms, by = "copy16_kernel", nc, msc, bc
    kind = (xmpi.PROF_ZCOPY if (launches, ms, by) == (nz, msz, bz) and nz else xmpi.PROF_REDUCE2 if (launches, ms, by) == (n2, ms2, b2) and n2 else
            xmpi.PROF_REDUCEN if (launches, ms, by) == (nn, msn, bn) and nn else xmpi.PROF_COPY)
    achieved = by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    # HBM traffic per launch comes from separate rocprofv3 --pmc passes (committed under profiles/); it is
    # attached only when this run launched the same kernel on the same number of bytes
    traffic, traffic_src = None, "PMC counters are collected in separate rocprofv3 passes (profiles/); none matches this launch size"
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:  # rewritten from each round's --pmc passes
            pmc = json.load(f)
        stem = kname.split("<")[0].split(" ")[0]
        for row in pmc["rows"]:
            if row["kernel"].startswith(stem) and launches and abs(row["traffic_bytes_per_launch"] / (by / launches) - 1) < 0.01:
                traffic = row["traffic_bytes_per_launch"]
                traffic_src = f"profiles/pmc_traffic.json round {pmc.get('round')} (" + row["kernel"] + ", " + row["schedule"] + "): " + pmc["source"]
                break
    except (OSError, ValueError, KeyError):
        pass
    # ... unless this run can measure it itself (N = 1: two rocprofv3 --pmc child passes of the same workload, ~10 s each)
    live = None
    if args.gpus == 1 and job.procs == 1 and not args.no_pmc and launches and kind == xmpi.PROF_ZCOPY and r0["dsync"] != 1:
        live = live_pmc_traffic(args, kname.split("<")[0].split(" ")[0])
        if live and "traffic_bytes_per_launch" in live:
            traffic = live["traffic_bytes_per_launch"]
            traffic_src = (f"live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child passes of this run (separate, --kernel-trace only; {live['launches']} launches "
                           f"each); traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction, MI355X_MICROARCH.md)")
    roof = {"bound": "hbm", "kernel": kname.split(" (")[0], "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_from_profile": not (live and "traffic_bytes_per_launch" in live),
            "traffic_over_algorithmic": (traffic / (by / launches)) if traffic and launches else None, "traffic_source": traffic_src,
            "kernel_detail": kname,
            "launches": launches, "avg_launch_us": (ms * 1e3 / launches) if launches else None,
            # the spread of the sampled launches; against what a streaming kernel achieves on this part (MI355X_MICROARCH.md: ~6.3 TB/s of the
            # 8 TB/s peak); and against THIS box on THESE buffers (the fold's access pattern without the arithmetic, right behind the timed region)
            "kernel_us_min": r0["prof_span"][kind][0], "kernel_us_max": r0["prof_span"][kind][1],
            "frac_of_achievable": achieved / HBM_ACHIEVABLE_GBPS,
            "box_copy_us": (r0.get("box") or {}).get("box_copy_us"),
            "frac_of_box": ((r0["box"]["box_copy_us"] / (ms * 1e3 / launches)) if launches and ms > 0 and (r0.get("box") or {}).get("box_copy_us") else None),
            "algorithmic_bytes_per_launch": (by / launches) if launches else None,
            "note": "live HIP events on the kernel's own stream inside the timed region (rank 0's launches); "
                    "algorithmic bytes: 12 B per output element for reduce2 (2 reads + 1 write), (N+1) x 4 B for the "
                    "N-way fold, 2N x 4 B for the zero-copy fold (N reads + N writes); sampled launches carry events "
                    "attached to their dispatch (every 2nd launch for zero-copy, every 4th otherwise)"}
    # the other kernels of the timed region (same per-dispatch events): slot drains and peer pushes
    others = {}
    for kind, label, factor in ((xmpi.PROF_COPY, "copy-out of receive slots (copy16 / copy_batch)", 1.0),
                                (xmpi.PROF_PEER, "peer pushes (copy_batch / hipMemcpyAsync), payload bytes x2 = read + write", 2.0)):
        n_k, ms_k, by_k = r0["prof"][kind]
        if n_k and ms_k > 0:
            others[label] = {"launches": n_k, "avg_launch_us": ms_k * 1e3 / n_k, "bytes_per_launch": factor * by_k / n_k,
                             "GBps": factor * by_k / (ms_k * 1e-3) / 1e9, "frac_of_hbm_peak": factor * by_k / (ms_k * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    size_rows = {}
    try:
        for x in r0["extras"].get("busbw_table", {}).get("rows", []):
            if x["bytes"] == min(S, 1 << 30) or x["bytes"] == max(y["bytes"] for y in r0["extras"]["busbw_table"]["rows"]):
                size_rows[str(x["ranks"])] = {"bytes": x["bytes"], "busbw_GBps": round(x["busbw_GBps"], 2), "algbw_GBps": round(x["algbw_GBps"], 2)}
                if r0["devices"] == R and 1 < x["ranks"] < R:
                    # one rank per GPU: the sub-communicators run the untuned library's fold, whose busiest link direction carries 2 S / ranks
                    # (the full communicator's row is the line's `roofline`)
                    size_rows[str(x["ranks"])]["frac_of_link_peak"] = round(2.0 / x["ranks"] * x["bytes"] / (x["us"] * 1e-6) / 1e9 / XGMI_DIR_GBPS, 4)
    except (KeyError, ValueError, TypeError):
        pass
    meaningful = r0["devices"] == R
    line = {
        "metric": f"allreduce_sum_{args.dtype}_{args.size_mib:g}MiB algbw",
        "value": algbw, "unit": "GB/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": t * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"BASELINE cfg 4: allreduce-sum {args.dtype} {args.size_mib:g} MiB/rank, {R} ranks",
                   "ranks": R, "ranks_per_gpu": R // args.gpus, "bytes_per_rank": S,
                   "algo": r0["best"]["algo"], "schedule_by": r0["tune"].get("by"),
                   "tuned": {k: r0["tune"][k] for k in ("algo", "split", "unroll", "ms", "check_ms") if k in r0["tune"]},
                   "transport": r0["transport"]},
        "algbw_GBps": algbw, "busbw_GBps": busbw,
        "ranks_meet": "on the device (dsync)" if r0["dsync"] == 1 else "on the host (control block)",
        "roofline": roof,
        "parity": {k: r0["parity"].get(k) for k in ("checked", "ok", "bit_identical", "max_abs_err", "rule", "coverage")},
        "parity_failures": len(r0["parity_failures"]),
        "busbw_at_size": size_rows,
    }
    if r0["iso"]:
        line["roofline_isolated"] = {k: r0["iso"][k] for k in ("avg_launch_us", "achieved", "frac")}
    ring_channels = (R - 2) if (R >= 4 and R % 2 == 0) else max(1, sum(1 for d in range(1, R) if np.gcd(d, R) == 1))
    ring_share = 2.0 * (R - 1) / R / min(8, ring_channels) if R > 1 else 0.0
    ndev = r0["devices"]
    mixed_share = None
    if 1 < ndev < R and r0["best"]["algo"] in ("auto (library)", "zcopy", "direct"):
        # Several ranks per GPU (--gpus 2 / 4): the one-hop schedules (the zero-copy fold with its host rendezvous, DIRECT) move
        # S / R between every ordered pair of RANKS twice (contribution in, result out), so a directed GPU pair -- one link -- carries
        # 2 x (R / ndev)^2 x S / R = 2 R / ndev^2 x S: at 2 GPUs FOUR times S through the one link between them.  That link, not
        # either GPU's HBM, bounds the step.
        mixed_share = 2.0 * R / (ndev * ndev)
    if (meaningful and R > 1) or mixed_share:
        # One rank per GPU: the LINKS bound the collective, not the HBM of any one GPU -- `roofline` is the link
        # roofline of the schedule that was timed, the kernel's HBM figure moves to `roofline_hbm`.  achieved = what the BUSIEST
        # link direction carries per step (the schedule's share of S, counted on virtual devices and asserted to the byte in the CPU
        # suite: DESIGN section 8) / the step time; peak = one direction of one link, nominal; peak_measured = what the copy kernel
        # wrote over one link in the probe before the timed region.
        sched = str(r0["tune"].get("algo") or r0["best"]["algo"])
        share = mixed_share or {"ring": ring_share, "ring_push": ring_share, "rhd": 1.0, "rhd_push": 1.0}.get(sched, 2.0 / R)
        measured = (r0["link"] or {}).get("copy_kernel_write_GBps")
        ach = share * S / t / 1e9
        line["roofline_hbm"] = roof
        line["roofline"] = {"bound": "xgmi", "kernel": roof["kernel"], "schedule": sched, "unit": "GB/s", "achieved": ach, "peak": XGMI_DIR_GBPS,
                            "frac": ach / XGMI_DIR_GBPS, "peak_measured": measured, "frac_of_measured": (ach / measured) if measured else None,
                            "busiest_link_direction_bytes_over_S": share, "algorithmic_bytes_per_link_direction": share * S,
                            "ranks_per_gpu": R // ndev,
                            "traffic": None, "direction": "stores" if sched in ("ring_push", "rhd_push", "zpush") else
                                                          "loads" if sched in ("ring", "rhd") else "loads and stores (S / R each per link)"}
    if r0.get("ring_named"):
        # north_star: "ring-allreduce bus-bandwidth on 256 MiB float32 at 8 ranks >= 70 % of per-link xGMI peak"
        forms = {k: v for k, v in r0["ring_named"].items() if "busbw_GBps" in v}
        line["ring"] = {k: {"busbw_GBps": round(v["busbw_GBps"], 2), "ms_per_step": round(v["ms_per_step"], 4), "parity_ok": v["parity_ok"],
                            "link_direction_GBps": ring_share * S / (v["ms_per_step"] * 1e-3) / 1e9,
                            "frac_of_link_peak": ring_share * S / (v["ms_per_step"] * 1e-3) / 1e9 / XGMI_DIR_GBPS,
                            "busbw_frac_of_link_both_directions": v["busbw_GBps"] / XGMI_LINK_GBPS} for k, v in forms.items()}
        if forms:
            line["ring"]["best"] = max(forms, key=lambda k: forms[k]["busbw_GBps"])
            line["ring"]["channels"] = min(8, ring_channels)
        for k, v in r0["ring_named"].items():
            if "error" in v:
                line["ring"][k] = v
    if meaningful or r0["link"]:
        # per rank 2(R-1)/R x S bytes leave (and arrive) per allreduce; the full-mesh schedules spread them over the R-1
        # links of a GPU, a ring channel puts its share on one.  153 GB/s per link is both directions together (76.8 each way).
        # What the BUSIEST link direction carries, by schedule, in units of S -- counted on virtual devices, every load and store
        # of the kernels traced (tests/devsim; asserted to the byte in the CPU suite): the fold and push-only
        # 2 / R; the ring kernel 2 (R - 1) / R over its channels (R - 2 Walecki rings on an even mesh); halving + doubling 1.
        sched = str(r0["tune"].get("algo") or r0["best"]["algo"])
        share = {"ring": ring_share, "ring_push": ring_share, "rhd": 1.0, "rhd_push": 1.0}.get(sched, 2.0 / R) if R > 1 else 0.0
        line["xgmi"] = {"per_link_peak_GBps": XGMI_LINK_GBPS, "per_link_direction_peak_GBps": XGMI_DIR_GBPS, "wire_GBps_per_rank_each_direction": busbw,
                        "wire_GBps_per_link_each_direction_if_spread": busbw / max(1, R - 1),
                        "schedule": sched, "busiest_link_direction_bytes_over_S": share,
                        "busiest_link_direction_GBps": share * S / t / 1e9,
                        "frac_of_link_peak": share * S / t / 1e9 / XGMI_DIR_GBPS,
                        "busbw_frac_of_one_link": busbw / XGMI_LINK_GBPS, "link_probe": r0["link"], "meaningful": meaningful}
    if any(r0["tune"].get("rejected", {}).values()):
        line["config"]["tuned"]["rejected"] = {k: v for k, v in r0["tune"]["rejected"].items() if v}
    if r0["degraded"]["level"] & 14:  # what xmpi_init's vote left out (ranks meet on the host / no windows) or a check found wrong here, and the first reason a rank gave
        line["degraded"] = r0["degraded"]
    if (args.gpus > 1 or job.procs > 1) and not args.no_probe:
        line["zero_copy_probe"] = r0["zero_copy_probe"]
    extras_out = {"timed_buffer_slots": {str(g): job.result[g].get("slots") for g in sorted(job.result)}, "box": r0.get("box"), "pmc_live": live,
                  "autotune": r0["tune"], "parity_failures": r0["parity_failures"], "other_kernels": others,
                  "roofline_isolated": r0["iso"], "extras": r0["extras"], "roofline_note": roof.pop("note", None),
                  "traffic_source": roof.pop("traffic_source", None), "kernel_detail": roof.pop("kernel_detail", None)}
    if args.gpus == 1 and not args.no_cpu and job.proc_rank == 0:
        cb = cpu_baseline(R, args.cpu_count or r0["count"], args.cpu_reps)
        # north_star: bus bandwidth "at 1, 2, 4 and 8 ranks ... next to the reference's loopback-TCP path timed ... in the same run": the
        # same bytes per rank at every rank count of the metric (one repetition at 2 and 4: they are the cheaper legs)
        cb = cpu_by_ranks(cb, R, args.cpu_count or r0["count"])
        line["cpu_baseline"] = cb
        for rk, row in (cb.get("by_ranks") or {}).items():
            if rk in size_rows and "algbw_GBps" in row and (args.cpu_count or r0["count"]) * 4 == size_rows[rk]["bytes"]:
                size_rows[rk]["cpu_algbw_GBps"] = row["algbw_GBps"]
        if isinstance(r0["extras"], dict) and "bounce_sweep_u8" in r0["extras"]:
            extras_out["extras"]["cpu_reference_bounce_u8"] = cpu_bounce()  # same lengths, the reference path on the host
    else:  # (not "unmeasured": the reference path is timed once, on the N = 1 run's host cores)
        line["cpu_baseline"] = {"see": "the N = 1 line: oracle/refpath_bin (the reference's loopback-TCP + gob path, kind \"port\") timed there on rank 0's host cores"}
    if job.production is not None:
        extras_out["production_layout"] = job.production
        rp = production_roofline(job.production)
        if rp:
            line["roofline_production"] = rp
            # the layout a real node runs (one process per rank, ranks meet on the device, the library's own schedule): the
            # same metric as `value`, first-class.  `value` stays the threads layout (the driver's contract since round 1)
            line["value_production"] = rp.get("algbw_GBps")
            line["value_production_exact"] = rp.get("exact")
    if job.mp_sweep is not None:
        extras_out["extras"]["multiprocess_sweep"] = job.mp_sweep
        try:  # the compact line carries the small end: what one BLOCKING allreduce of 1 KiB costs a caller (every call of the
            # reference's API blocks, mpi.go:47-48), one process per rank, and how many of them the lingering LL agent ran
            r1k = job.mp_sweep["rows"][0]
            line["small_allreduce_us"] = {"bytes": r1k["bytes"], "ranks": job.mp_sweep["ranks"], "blocking": round(r1k["blocking_us"], 1),
                                          "enqueued": round(r1k["queued_us"], 1), "by_ll_agent": r1k.get("by_agent")}
        except (KeyError, IndexError, TypeError):
            pass
    hs = r0["extras"].get("host_slices_allreduce") if isinstance(r0["extras"], dict) else None
    if hs:  # the PCIe-inclusive rate of the headline's call (host slices in, host slices out): beside `value`, never `value`
        line["pcie_inclusive"] = {"algbw_GBps": round(hs["algbw_GBps"], 2), "ms_per_step": round(hs["ms_per_step"], 1), "parity_ok": hs["parity_ok"]}
    if job.cfg3 is not None:
        extras_out["extras"]["cfg3_allgather_i64_16MiB_4ranks_one_process_per_rank"] = job.cfg3
    if job.cfg5 is not None:
        extras_out["extras"]["cfg5_allreduce_f16_sweep_one_process_per_rank"] = job.cfg5
        try:  # the compact line carries the two ends of the sweep
            rows = job.cfg5["rows"]
            line["cfg5_f16_us"] = {str(r["bytes"]): {k: round(r[k]["us"], 1) for k in ("ring", "rhd", "auto") if k in r} for r in (rows[0], rows[-1])}
            line["cfg5_f16_us"]["all_bit_identical"] = job.cfg5["all_bit_identical"]
        except (KeyError, IndexError, TypeError):
            pass
    line["extras_file"] = write_extras(extras_out)
    print(json.dumps(line, separators=(",", ":")))
    sys.stdout.flush()


def write_extras(obj) -> str:
    """everything that is not the contract line: bench_extras.json beside bench.py, and a copy in gpurun_out/ when present"""
    name = "bench_extras.json"
    where = []
    dirs = (ROOT, os.path.join(ROOT, "gpurun_out"))
    if os.environ.get("XMPI_BENCH_EXTRAS_DIR"):  # (a run that must not touch the checkout: the CPU suite's rehearsals)
        dirs = (os.environ["XMPI_BENCH_EXTRAS_DIR"],)
    for d in dirs:
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, name), "w") as f:
                    json.dump(obj, f, indent=1)
                where.append(os.path.relpath(os.path.join(d, name), ROOT))
            except OSError:
                pass
    return where[0] if where else ""


def production_layout(ranks: int, nbytes: int, steps: int, warmup: int):
    """examples/allreduce_bench under the launcher: ONE OS PROCESS PER RANK on this box's GPU -- the north-star layout,
    ranks meeting on the device.  The library tunes itself (xmpi_tune) and AUTO runs; the one-kernel and the
    meet / body / done form are timed by name beside it."""
    run = os.path.join(ROOT, "mpi_amd", "bin", "xmpirun")
    prog = os.path.join(ROOT, "mpi_amd", "bin", "allreduce_bench")
    if not (os.path.exists(run) and os.path.exists(prog)):
        return None
    env = dict(os.environ, XMPI_TIMEOUT_S="60", XMPI_BASEPORT=str(9000 + os.getpid() % 1000 * 16))
    env.pop("XMPI_SLOT_BYTES", None)
    env.pop("XMPI_FIFO_DEPTH", None)
    try:
        p = subprocess.run([run, str(ranks), prog, str(nbytes), str(steps), str(warmup), "auto", "fused", "split"], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    except subprocess.TimeoutExpired:
        return {"error": "timed out"}
    if p.returncode != 0:
        return {"error": (p.stderr or p.stdout)[-400:]}
    try:
        return json.loads(p.stdout.strip().split("\n")[-1])
    except ValueError:
        return {"error": p.stdout[-400:]}


def cfg3_production():
    """examples/cfg3_allgather under the launcher: BASELINE cfg 3 (allgather int64, 16 MiB per rank, 4 ranks, ring) with one
    OS process per rank -- the ring is the stepped kernel there"""
    run = os.path.join(ROOT, "mpi_amd", "bin", "xmpirun")
    prog = os.path.join(ROOT, "mpi_amd", "bin", "cfg3_allgather")
    if not (os.path.exists(run) and os.path.exists(prog)):
        return None
    env = dict(os.environ, XMPI_TIMEOUT_S="60", XMPI_BASEPORT=str(9800 + os.getpid() % 1000 * 16))
    env.pop("XMPI_SLOT_BYTES", None)
    env.pop("XMPI_FIFO_DEPTH", None)
    try:
        p = subprocess.run([run, "4", prog, "2097152", "20"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=200)
        return json.loads(p.stdout.strip().split("\n")[-1]) if p.returncode == 0 else {"error": (p.stderr or p.stdout)[-400:]}
    except (subprocess.TimeoutExpired, ValueError) as e:
        return {"error": repr(e)[:200]}


def cfg5_production(ranks: int, max_bytes: int):
    """examples/cfg5_sweep under the launcher: BASELINE cfg 5 (fp16, recursive halving + doubling vs ring, 1 MiB ... 1 GiB)
    with one OS process per rank -- ring and halving are the stepped KERNELS there, not host-driven step tables"""
    run = os.path.join(ROOT, "mpi_amd", "bin", "xmpirun")
    prog = os.path.join(ROOT, "mpi_amd", "bin", "cfg5_sweep")
    if not (os.path.exists(run) and os.path.exists(prog)):
        return None
    env = dict(os.environ, XMPI_TIMEOUT_S="60", XMPI_BASEPORT=str(9500 + os.getpid() % 1000 * 16))
    env.pop("XMPI_SLOT_BYTES", None)
    env.pop("XMPI_FIFO_DEPTH", None)
    try:
        p = subprocess.run([run, str(ranks), prog, str(max_bytes), "5"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           text=True, timeout=300)
    except subprocess.TimeoutExpired:
        return {"error": "timed out"}
    if p.returncode != 0:
        return {"error": (p.stderr or p.stdout)[-400:]}
    try:
        return json.loads(p.stdout.strip().split("\n")[-1])
    except ValueError:
        return {"error": p.stdout[-400:]}


def production_roofline(prod):
    """the compact line's `roofline_production`: what AUTO ran with one process per rank, and its dominant kernel"""
    try:
        row = next(x for x in prod["rows"] if x["mode"] == "auto")
    except (KeyError, StopIteration, TypeError):
        return {"error": (prod or {}).get("error", "no result")} if isinstance(prod, dict) else None
    split = row["tuned"]["split"] == 1 or (row["tuned"]["split"] < 0 and row["split_launches"] > 0)
    algo = ALGO_NAME.get(row["tuned"]["algo"], "zcopy")
    kernel = ("dsync_sched_kernel (stepped: ring / halving inside one kernel)" if algo in STEPPED else
              "dsync_body_kernel<float,SUM,8> between the meet and done kernels" if split else
              "dsync_fold_kernel<float,SUM,8> (rendezvous + fold + completion in one kernel: its duration includes waiting for peers)")
    ranks = prod["ranks"]
    # All ranks' kernels share this one GPU, overlapping as the scheduler lets them: per-kernel rates do not add up.  What the
    # chip delivered is every rank's algorithmic bytes over the STEP (kernels, their rendezvous and completion exchange);
    # the kernel's own duration (events attached to rank 0's sampled dispatches) is beside it.
    agg = row["kernel_bytes_per_launch"] * ranks / (row["us_per_step"] * 1e-6) / 1e9 if row["us_per_step"] else 0.0
    out = {"layout": f"{ranks} processes, one rank each, on this GPU; ranks meet {prod['meet']}", "algo": algo, "split": bool(split),
           "kernel": kernel, "ms_per_step": row["us_per_step"] / 1e3, "algbw_GBps": row["algbw_GBps"],
           "avg_launch_us": row["kernel_avg_us"], "algorithmic_bytes_per_launch": row["kernel_bytes_per_launch"],
           "achieved_all_ranks": agg, "peak": HBM_PEAK_GBPS,  # (achieved_all_ranks = ranks x algorithmic bytes per launch / time per step)
           "unit": "GB/s", "frac": agg / HBM_PEAK_GBPS, "exact": prod.get("exact")}
    for m in ("fused", "split"):
        try:
            r = next(x for x in prod["rows"] if x["mode"] == m)
            out[m + "_ms_per_step"] = r["us_per_step"] / 1e3
        except (StopIteration, KeyError):
            pass
    if prod.get("bounce"):
        out["bounce_half_round_trip_us"] = {str(b["bytes"]): b["half_round_trip_us"] for b in prod["bounce"]}
    return out


def _guard(job, g, fn=None):
    try:
        (fn or rank_main)(job, g)
    except BaseException:  # noqa: BLE001
        import traceback
        with job.lock:
            job.errors.append((g, traceback.format_exc()))


if __name__ == "__main__":
    main()
