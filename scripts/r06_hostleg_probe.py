#!/usr/bin/env python3
"""Where the 209 ms of `pcie_inclusive` go (bench.py: the headline's call on host slices): pageable uploads / downloads of 256 MiB through
xmpi_memcpy by 1 and by 8 rank threads at once, the staged allreduce on temporary device buffers alone, hipMalloc + hipFree of that size.
One GPU; prints one JSON object."""
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpi_amd import xmpi  # noqa: E402

R, N = 8, 64 << 20  # ranks, float32 elements per rank
out, lock = {}, threading.Lock()
bar = threading.Barrier(R)


def rank(r):
    c = xmpi.Comm(r, R, 0, f"hostleg-{os.getpid()}")
    dev = c.alloc(N * 4)
    dev2 = c.alloc(N * 4)
    c.fill(dev, N, xmpi.F32, xmpi.PAT_UNIFORM, 1000 + r)
    h = dev.download(np.float32, N)
    ho = np.zeros(N, dtype=np.float32)
    res = {}

    def t(name, fn, who):
        bar.wait()
        t0 = time.perf_counter()
        if r in who:
            fn()
        bar.wait()
        if r == 0:
            res[name] = round((time.perf_counter() - t0) * 1e3, 2)

    one, everybody = {0}, set(range(R))
    for rep in range(2):
        t("upload_256MiB_1_thread_ms", lambda: dev.upload(h), one)
        t("download_256MiB_1_thread_ms", lambda: c.memcpy(ho, dev, N * 4), one)
        t("upload_256MiB_8_threads_ms", lambda: dev.upload(h), everybody)
        t("download_256MiB_8_threads_ms", lambda: c.memcpy(ho, dev, N * 4), everybody)
        t("allreduce_device_buffers_auto_ms", lambda: c.allreduce(dev, dev2, N, xmpi.F32, xmpi.SUM, xmpi.ALGO_AUTO), everybody)
        t("allreduce_host_slices_auto_ms", lambda: c.allreduce(h, ho, N, xmpi.F32, xmpi.SUM, xmpi.ALGO_AUTO), everybody)
        t("allreduce_host_in_device_out_ms", lambda: c.allreduce(h, dev2, N, xmpi.F32, xmpi.SUM, xmpi.ALGO_AUTO), everybody)
        t("allreduce_device_in_host_out_ms", lambda: c.allreduce(dev, ho, N, xmpi.F32, xmpi.SUM, xmpi.ALGO_AUTO), everybody)
    if r == 0:
        with lock:
            out.update(res)
    bar.wait()
    dev.free()
    dev2.free()
    c.finalize()


ts = [threading.Thread(target=rank, args=(r,)) for r in range(R)]
[x.start() for x in ts]
[x.join() for x in ts]
print(json.dumps(out))
