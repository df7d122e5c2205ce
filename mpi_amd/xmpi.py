"""ctypes binding of the C ABI (include/xmpi.h) -- the harness tests and bench.py drive.

This is plumbing, not the product: every method is one call into libxmpi.so.  There is no CPU
fallback anywhere: if the shared library is missing, or no HIP device is visible, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libxmpi.so")

# enums of include/xmpi.h
U8, I32, I64, F16, F32, F64, BF16 = range(7)
SUM, PROD, MIN, MAX = range(4)
ALGO_AUTO, ALGO_RING, ALGO_RHD, ALGO_DIRECT, ALGO_TREE, ALGO_ZCOPY, ALGO_ZPUSH, ALGO_LL, ALGO_RING_PUSH, ALGO_RHD_PUSH, ALGO_TREE_PUSH = range(11)
COLL_ALLREDUCE, COLL_ALLGATHER, COLL_BCAST, COLL_REDUCE = range(4)
PAT_UNIFORM, PAT_INDEX, PAT_CONST, PAT_SIGNED = range(4)
PROF_REDUCE2, PROF_REDUCEN, PROF_COPY, PROF_PEER, PROF_ZCOPY = range(5)

OK = 0
ERR_ARG, ERR_HIP, ERR_BOOTSTRAP, ERR_TIMEOUT, ERR_TAG_EXISTS, ERR_TRUNCATE = -1, -2, -3, -4, -5, -6
ERR_NOMEM, ERR_STATE, ERR_UNSUPPORTED, ERR_NOGPU, ERR_PEER = -7, -8, -9, -10, -11

DTYPE_SIZE = {U8: 1, I32: 4, I64: 8, F16: 2, F32: 4, F64: 8, BF16: 2}
NUMPY_DTYPE = {U8: np.uint8, I32: np.int32, I64: np.int64, F16: np.float16, F32: np.float32, F64: np.float64,
               BF16: np.uint16}  # bf16 travels as raw bit patterns
DTYPE_NAME = {U8: "u8", I32: "i32", I64: "i64", F16: "f16", F32: "f32", F64: "f64", BF16: "bf16"}

# every symbol include/xmpi.h declares: (name, restype, argtypes)
_P, _I, _L, _Z = C.c_void_p, C.c_int, C.c_long, C.c_size_t
SYMBOLS = [
    ("xmpi_init", _I, [_I, _I, _I, C.c_char_p, C.POINTER(_P)]),
    ("xmpi_finalize", _I, [_P]),
    ("xmpi_rank", _I, [_P]),
    ("xmpi_size", _I, [_P]),
    ("xmpi_device", _I, [_P]),
    ("xmpi_barrier", _I, [_P]),
    ("xmpi_strerror", C.c_char_p, [_I]),
    ("xmpi_last_error", C.c_char_p, []),
    ("xmpi_version", C.c_char_p, []),
    ("xmpi_malloc", _P, [_P, _Z]),
    ("xmpi_free", _I, [_P, _P]),
    ("xmpi_memcpy", _I, [_P, _P, _P, _Z]),
    ("xmpi_memset", _I, [_P, _P, _I, _Z]),
    ("xmpi_sync", _I, [_P]),
    ("xmpi_send", _I, [_P, _P, _Z, _I, _I, _I]),
    ("xmpi_recv", _I, [_P, _P, _Z, _I, _I, _I, C.POINTER(_Z)]),
    ("xmpi_probe", _I, [_P, _I, _I, C.POINTER(_Z), C.POINTER(_I)]),
    ("xmpi_bcast", _I, [_P, _P, _Z, _I, _I, _I]),
    ("xmpi_reduce", _I, [_P, _P, _P, _Z, _I, _I, _I, _I]),
    ("xmpi_allreduce", _I, [_P, _P, _P, _Z, _I, _I, _I]),
    ("xmpi_allgather", _I, [_P, _P, _P, _Z, _I, _I]),
    ("xmpi_reduce_local", _I, [_P, _P, _P, _P, _Z, _I, _I]),
    ("xmpi_reduce_local_n", _I, [_P, _P, C.POINTER(_P), _I, _Z, _I, _I]),
    ("xmpi_copy_local", _I, [_P, _P, _P, _Z]),
    ("xmpi_count_mismatch", _I, [_P, _P, _P, _Z, C.POINTER(C.c_uint64)]),
    ("xmpi_checksum", _I, [_P, _P, _Z, C.POINTER(C.c_uint64)]),
    ("xmpi_diff_stats", _I, [_P, _P, _P, _Z, _I, C.POINTER(C.c_double)]),
    ("xmpi_diff_rel", _I, [_P, _P, _P, _Z, _I, C.POINTER(C.c_double)]),
    ("xmpi_fill_pattern", _I, [_P, _P, _Z, _I, _I, C.c_uint64]),
    ("xmpi_set_param", _I, [_P, C.c_char_p, _L]),
    ("xmpi_get_param", _L, [_P, C.c_char_p]),
    ("xmpi_prof_enable", _I, [_P, _I]),
    ("xmpi_prof_reset", _I, [_P]),
    ("xmpi_prof_get", _I, [_P, _I, C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    ("xmpi_link_probe", _I, [_P, _I, _Z, _I, _I, _I, C.POINTER(C.c_double)]),
    ("xmpi_ctl_selftest", _I, [C.c_char_p, _I, _I, _I]),
    ("xmpi_plan_dump", _I, [_I, _I, _I, _I, _I, _Z, _Z, _I, _Z, _I, _Z, C.c_char_p, _Z]),
    ("xmpi_dtype_size", _Z, [_I]),
    ("xmpi_allreduce_repeat", _I, [_P, _P, _P, _Z, _I, _I, _I, _I]),
    ("xmpi_heap_selftest", _I, [C.c_uint64, _I]),
    ("xmpi_iallreduce", _I, [_P, _P, _P, _Z, _I, _I, _I, C.POINTER(_P)]),
    ("xmpi_iallgather", _I, [_P, _P, _P, _Z, _I, _I, C.POINTER(_P)]),
    ("xmpi_ibcast", _I, [_P, _P, _Z, _I, _I, _I, C.POINTER(_P)]),
    ("xmpi_ireduce", _I, [_P, _P, _P, _Z, _I, _I, _I, _I, C.POINTER(_P)]),
    ("xmpi_request_test", _I, [_P, C.POINTER(_I)]),
    ("xmpi_request_wait", _I, [_P]),
    ("xmpi_send_nowait", _I, [_P, _P, _Z, _I, _I, _I]),
    ("xmpi_wait", _I, [_P, _I, _I]),
    ("xmpi_register", _I, [_P, _P, _Z]),
    ("xmpi_deregister", _I, [_P, _P]),
    ("xmpi_reduce_local_multi", _I, [_P, C.POINTER(_P), _I, C.POINTER(_P), _I, _Z, _I, _I]),
    ("xmpi_copy_local_multi", _I, [_P, C.POINTER(_P), _I, _P, _Z]),
    ("xmpi_copy_local_pairs", _I, [_P, C.POINTER(_P), C.POINTER(_P), _I, _Z]),
    ("xmpi_zc_chunk", _I, [_Z, _Z, _I, _I, C.POINTER(_Z), C.POINTER(_Z)]),
    ("xmpi_allreduce_on_stream", _I, [_P, _P, _P, _Z, _I, _I, _P]),
    ("xmpi_allgather_on_stream", _I, [_P, _P, _P, _Z, _I, _P]),
    ("xmpi_bcast_on_stream", _I, [_P, _P, _Z, _I, _I, _P]),
    ("xmpi_reduce_on_stream", _I, [_P, _P, _P, _Z, _I, _I, _I, _P]),
    ("xmpi_stream_create", _P, [_P]),
    ("xmpi_stream_destroy", _I, [_P, _P]),
    ("xmpi_stream_sync", _I, [_P, _P]),
    ("xmpi_graph_begin", _I, [_P, _P]),
    ("xmpi_graph_end", _I, [_P, _P, C.POINTER(_P)]),
    ("xmpi_graph_launch", _I, [_P, _P, _P]),
    ("xmpi_graph_destroy", _I, [_P, _P]),
    ("xmpi_send_on_stream", _I, [_P, _P, _Z, _I, _I, _I, _P]),
    ("xmpi_recv_on_stream", _I, [_P, _P, _Z, _I, _I, _I, _P]),
    ("xmpi_degraded", C.c_char_p, [_P]),
    ("xmpi_tune", _I, [_P, _Z]),
    ("xmpi_tune_decide", _I, [C.POINTER(C.c_double), _I, C.c_double]),
    ("xmpi_sched_dump", _I, [_I, _I, _I, _I, _I, _I, _I, _Z, _Z, _I, _I, C.c_char_p, _Z]),
    ("xmpi_sched_land_bytes", _Z, [_I, _I, _I, _I, _I, _Z, _Z]),
]

_lib: Optional[C.CDLL] = None


class XmpiError(RuntimeError):
    def __init__(self, code: int, where: str, detail: str = ""):
        self.code = code
        super().__init__(f"{where}: xmpi error {code} ({detail})")


def lib() -> C.CDLL:
    """Load libxmpi.so (built in-tree by mpi_amd.build).  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                f"{LIB_PATH} is missing: run `python -m mpi_amd.build` (hipcc, gfx950). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(rc: int, where: str) -> None:
    if rc != OK:
        L = lib()
        detail = L.xmpi_strerror(rc).decode() + "; " + L.xmpi_last_error().decode()
        raise XmpiError(rc, where, detail)


def plan_text(coll: int, algo: int, size: int, rank: int, root: int, count: int, elem_size: int, channels: int,
              piece_elems: int, fifo_depth: int = 0, oneshot_bytes: int = -1) -> str:
    """Step table the executor runs (host logic only: works without a GPU).  fifo_depth 0 / oneshot_bytes -1: the defaults."""
    L = lib()
    one = oneshot_bytes if oneshot_bytes >= 0 else (1 << 64) - 1
    n = L.xmpi_plan_dump(coll, algo, size, rank, root, count, elem_size, channels, piece_elems, fifo_depth, one, None, 0)
    if n < 0:
        raise XmpiError(n, "xmpi_plan_dump")
    buf = C.create_string_buffer(n + 1)
    L.xmpi_plan_dump(coll, algo, size, rank, root, count, elem_size, channels, piece_elems, fifo_depth, one, buf, n + 1)
    return buf.value.decode()


SCHED_RING_ALLREDUCE, SCHED_RHD_ALLREDUCE, SCHED_RING_ALLGATHER, SCHED_TREE_BCAST, SCHED_TREE_REDUCE = 1, 2, 3, 4, 5


def sched_text(sched: int, size: int, rank: int, root: int, pieces: int, count: int, elem_size: int, nchan: int,
               channel: int, push: bool = False, inplace: bool = False) -> str:
    """Step program of a stepped kernel (sched.hip) for one rank and channel, pull or push form (host logic only: works
    without a GPU)."""
    L = lib()
    n = L.xmpi_sched_dump(sched, int(push), int(inplace), size, rank, root, pieces, count, elem_size, nchan, channel, None, 0)
    if n < 0:
        raise XmpiError(n, "xmpi_sched_dump")
    buf = C.create_string_buffer(n + 1)
    L.xmpi_sched_dump(sched, int(push), int(inplace), size, rank, root, pieces, count, elem_size, nchan, channel, buf, n + 1)
    return buf.value.decode()


def sched_land_bytes(sched: int, size: int, rank: int, root: int, count: int, elem_size: int, inplace: bool = False) -> int:
    """bytes of the landing block `rank` lends to the push form of a stepped schedule"""
    return lib().xmpi_sched_land_bytes(sched, int(inplace), size, rank, root, count, elem_size)


def tune_decide(mean_us: Sequence[float], margin: float = 0.03) -> int:
    arr = (C.c_double * len(mean_us))(*mean_us)
    return lib().xmpi_tune_decide(arr, len(mean_us), margin)


def zc_chunk(count: int, elem_size: int, size: int, j: int):
    """(element offset, element count) of the chunk rank j folds / forwards in a zero-copy collective."""
    off, cnt = _Z(0), _Z(0)
    rc = lib().xmpi_zc_chunk(count, elem_size, size, j, C.byref(off), C.byref(cnt))
    if rc != OK:
        raise XmpiError(rc, "xmpi_zc_chunk")
    return off.value, cnt.value


class DeviceBuffer:
    """A span of this rank's HBM (xmpi_malloc)."""

    def __init__(self, comm: "Comm", nbytes: int):
        self.comm = comm
        self.nbytes = int(nbytes)
        self.ptr = lib().xmpi_malloc(comm.handle, max(1, self.nbytes))
        if not self.ptr:
            raise XmpiError(ERR_NOMEM, "xmpi_malloc", lib().xmpi_last_error().decode())

    def at(self, byte_offset: int) -> int:
        return self.ptr + int(byte_offset)

    def upload(self, arr: np.ndarray, byte_offset: int = 0) -> "DeviceBuffer":
        a = np.ascontiguousarray(arr)
        assert byte_offset + a.nbytes <= self.nbytes
        if a.nbytes:
            _check(lib().xmpi_memcpy(self.comm.handle, self.ptr + byte_offset, a.ctypes.data, a.nbytes), "upload")
        return self

    def download(self, dtype, count: Optional[int] = None, byte_offset: int = 0) -> np.ndarray:
        dt = np.dtype(dtype)
        if count is None:
            count = (self.nbytes - byte_offset) // dt.itemsize
        out = np.empty(count, dtype=dt)
        if out.nbytes:
            _check(lib().xmpi_memcpy(self.comm.handle, out.ctypes.data, self.ptr + byte_offset, out.nbytes), "download")
        return out

    def free(self) -> None:
        if self.ptr:
            lib().xmpi_free(self.comm.handle, self.ptr)
            self.ptr = None

    def __del__(self):  # best effort
        try:
            if self.ptr and self.comm.handle:
                self.free()
        except Exception:
            pass


def _ptr(x) -> int:
    if isinstance(x, DeviceBuffer):
        return x.ptr
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    if x is None:
        return 0
    return int(x)


class Comm:
    """One rank: one process (or thread), one MI355X.  Mirrors the six methods of the
    reference's mpi.Interface (mpi.go:163-170) plus the collectives the reference stubs out."""

    def __init__(self, rank: int, size: int, device: int = -1, job_key: str = "default"):
        self.handle = None
        h = _P()
        rc = lib().xmpi_init(rank, size, device, job_key.encode(), C.byref(h))
        _check(rc, "xmpi_init")
        self.handle = h

    # -- mpi.Interface -------------------------------------------------------------------------
    def rank(self) -> int:
        return lib().xmpi_rank(self.handle)

    def size(self) -> int:
        return lib().xmpi_size(self.handle)

    def device(self) -> int:
        return lib().xmpi_device(self.handle)

    def finalize(self) -> None:
        if self.handle:
            h, self.handle = self.handle, None
            _check(lib().xmpi_finalize(h), "xmpi_finalize")

    def send(self, buf, count: int, dtype: int, dest: int, tag: int) -> None:
        _check(lib().xmpi_send(self.handle, _ptr(buf), count, dtype, dest, tag), "xmpi_send")

    def send_nowait(self, buf, count: int, dtype: int, dest: int, tag: int) -> None:
        """Returns once the payload has left `buf`; wait(dest, tag) collects the receiver's confirmation."""
        _check(lib().xmpi_send_nowait(self.handle, _ptr(buf), count, dtype, dest, tag), "xmpi_send_nowait")

    def wait(self, dest: int, tag: int) -> None:
        _check(lib().xmpi_wait(self.handle, dest, tag), "xmpi_wait")

    def recv(self, buf, capacity: int, dtype: int, src: int, tag: int) -> int:
        got = _Z(0)
        _check(lib().xmpi_recv(self.handle, _ptr(buf), capacity, dtype, src, tag, C.byref(got)), "xmpi_recv")
        return got.value

    def probe(self, src: int, tag: int):
        """(count, dtype) of the message {src, tag} once it has been posted; does not consume it."""
        n, dt = _Z(0), _I(0)
        _check(lib().xmpi_probe(self.handle, src, tag, C.byref(n), C.byref(dt)), "xmpi_probe")
        return n.value, dt.value

    # -- collectives ---------------------------------------------------------------------------
    def barrier(self) -> None:
        _check(lib().xmpi_barrier(self.handle), "xmpi_barrier")

    def bcast(self, buf, count: int, dtype: int, root: int, algo: int = ALGO_AUTO) -> None:
        _check(lib().xmpi_bcast(self.handle, _ptr(buf), count, dtype, root, algo), "xmpi_bcast")

    def reduce(self, send, recv, count: int, dtype: int, op: int, root: int, algo: int = ALGO_AUTO) -> None:
        _check(lib().xmpi_reduce(self.handle, _ptr(send), _ptr(recv), count, dtype, op, root, algo), "xmpi_reduce")

    def allreduce(self, send, recv, count: int, dtype: int, op: int = SUM, algo: int = ALGO_AUTO) -> None:
        _check(lib().xmpi_allreduce(self.handle, _ptr(send), _ptr(recv), count, dtype, op, algo), "xmpi_allreduce")

    # non-blocking forms: return a request handle; request_wait() returns when the operation has completed
    def iallreduce(self, send, recv, count: int, dtype: int, op: int = SUM, algo: int = ALGO_AUTO) -> int:
        r = _P()
        _check(lib().xmpi_iallreduce(self.handle, _ptr(send), _ptr(recv), count, dtype, op, algo, C.byref(r)), "xmpi_iallreduce")
        return r.value

    def iallgather(self, send, recv, count: int, dtype: int, algo: int = ALGO_AUTO) -> int:
        r = _P()
        _check(lib().xmpi_iallgather(self.handle, _ptr(send), _ptr(recv), count, dtype, algo, C.byref(r)), "xmpi_iallgather")
        return r.value

    def ibcast(self, buf, count: int, dtype: int, root: int, algo: int = ALGO_AUTO) -> int:
        r = _P()
        _check(lib().xmpi_ibcast(self.handle, _ptr(buf), count, dtype, root, algo, C.byref(r)), "xmpi_ibcast")
        return r.value

    def ireduce(self, send, recv, count: int, dtype: int, op: int, root: int, algo: int = ALGO_AUTO) -> int:
        r = _P()
        _check(lib().xmpi_ireduce(self.handle, _ptr(send), _ptr(recv), count, dtype, op, root, algo, C.byref(r)),
               "xmpi_ireduce")
        return r.value

    @staticmethod
    def request_test(req: int) -> bool:
        d = _I(0)
        _check(lib().xmpi_request_test(req, C.byref(d)), "xmpi_request_test")
        return bool(d.value)

    @staticmethod
    def request_wait(req: int) -> None:
        _check(lib().xmpi_request_wait(req), "xmpi_request_wait")

    def allreduce_repeat(self, send, recv, count: int, dtype: int, op: int, algo: int, iters: int) -> None:
        """`iters` back-to-back allreduces inside one call (a benchmark's step loop without the interpreter)."""
        _check(lib().xmpi_allreduce_repeat(self.handle, _ptr(send), _ptr(recv), count, dtype, op, algo, iters),
               "xmpi_allreduce_repeat")

    # stream-ordered forms: enqueue on a HIP stream (None = the communicator's own), return at once
    def stream_create(self) -> int:
        s = lib().xmpi_stream_create(self.handle)
        if not s:
            raise XmpiError(ERR_HIP, "xmpi_stream_create", lib().xmpi_last_error().decode())
        return s

    def stream_destroy(self, stream) -> None:
        _check(lib().xmpi_stream_destroy(self.handle, stream), "xmpi_stream_destroy")

    def stream_sync(self, stream=None) -> None:
        _check(lib().xmpi_stream_sync(self.handle, stream), "xmpi_stream_sync")

    # hipGraph capture of the stream-ordered collectives enqueued on `stream` between graph_begin and graph_end
    def graph_begin(self, stream) -> None:
        _check(lib().xmpi_graph_begin(self.handle, stream), "xmpi_graph_begin")

    def graph_end(self, stream) -> int:
        g = _P()
        _check(lib().xmpi_graph_end(self.handle, stream, C.byref(g)), "xmpi_graph_end")
        return g.value

    def graph_launch(self, graph, stream=None) -> None:
        _check(lib().xmpi_graph_launch(self.handle, graph, stream), "xmpi_graph_launch")

    def graph_destroy(self, graph) -> None:
        _check(lib().xmpi_graph_destroy(self.handle, graph), "xmpi_graph_destroy")

    def allreduce_on_stream(self, send, recv, count: int, dtype: int, op: int = SUM, stream=None) -> None:
        _check(lib().xmpi_allreduce_on_stream(self.handle, _ptr(send), _ptr(recv), count, dtype, op, stream),
               "xmpi_allreduce_on_stream")

    def send_on_stream(self, buf, count: int, dtype: int, dest: int, tag: int, stream=None) -> None:
        _check(lib().xmpi_send_on_stream(self.handle, _ptr(buf), count, dtype, dest, tag, stream), "xmpi_send_on_stream")

    def recv_on_stream(self, buf, capacity: int, dtype: int, src: int, tag: int, stream=None) -> None:
        _check(lib().xmpi_recv_on_stream(self.handle, _ptr(buf), capacity, dtype, src, tag, stream), "xmpi_recv_on_stream")

    def tune(self, max_bytes: int) -> None:
        _check(lib().xmpi_tune(self.handle, max_bytes), "xmpi_tune")

    def allgather_on_stream(self, send, recv, count: int, dtype: int, stream=None) -> None:
        _check(lib().xmpi_allgather_on_stream(self.handle, _ptr(send), _ptr(recv), count, dtype, stream),
               "xmpi_allgather_on_stream")

    def bcast_on_stream(self, buf, count: int, dtype: int, root: int, stream=None) -> None:
        _check(lib().xmpi_bcast_on_stream(self.handle, _ptr(buf), count, dtype, root, stream), "xmpi_bcast_on_stream")

    def reduce_on_stream(self, send, recv, count: int, dtype: int, op: int, root: int, stream=None) -> None:
        _check(lib().xmpi_reduce_on_stream(self.handle, _ptr(send), _ptr(recv), count, dtype, op, root, stream),
               "xmpi_reduce_on_stream")

    def allgather(self, send, recv, count: int, dtype: int, algo: int = ALGO_AUTO) -> None:
        _check(lib().xmpi_allgather(self.handle, _ptr(send), _ptr(recv), count, dtype, algo), "xmpi_allgather")

    # -- local kernels -------------------------------------------------------------------------
    def reduce_local(self, dst, a, b, count: int, dtype: int, op: int = SUM) -> None:
        _check(lib().xmpi_reduce_local(self.handle, _ptr(dst), _ptr(a), _ptr(b), count, dtype, op), "reduce_local")

    def reduce_local_n(self, dst, srcs: Sequence, count: int, dtype: int, op: int = SUM) -> None:
        arr = (_P * len(srcs))(*[_ptr(s) for s in srcs])
        _check(lib().xmpi_reduce_local_n(self.handle, _ptr(dst), arr, len(srcs), count, dtype, op), "reduce_local_n")

    def copy_local(self, dst, src, nbytes: int) -> None:
        _check(lib().xmpi_copy_local(self.handle, _ptr(dst), _ptr(src), nbytes), "copy_local")

    def reduce_local_multi(self, dsts: Sequence, srcs: Sequence, count: int, dtype: int, op: int = SUM) -> None:
        """Every dsts[k] = left-to-right fold of srcs (the zero-copy allreduce kernel, on local buffers)."""
        d = (_P * len(dsts))(*[_ptr(x) for x in dsts])
        s = (_P * len(srcs))(*[_ptr(x) for x in srcs])
        _check(lib().xmpi_reduce_local_multi(self.handle, d, len(dsts), s, len(srcs), count, dtype, op),
               "reduce_local_multi")

    def copy_local_pairs(self, dsts: Sequence, srcs: Sequence, nbytes: int) -> None:
        """dsts[k] = srcs[k] in one launch with the fold's access pattern (reduce_local_multi without the arithmetic)"""
        d = (_P * len(dsts))(*[_ptr(x) for x in dsts])
        s_ = (_P * len(srcs))(*[_ptr(x) for x in srcs])
        _check(lib().xmpi_copy_local_pairs(self.handle, d, s_, len(srcs), nbytes), "xmpi_copy_local_pairs")

    def copy_local_multi(self, dsts: Sequence, src, nbytes: int) -> None:
        d = (_P * len(dsts))(*[_ptr(x) for x in dsts])
        _check(lib().xmpi_copy_local_multi(self.handle, d, len(dsts), _ptr(src), nbytes), "copy_local_multi")

    def register(self, ptr, nbytes: int) -> None:
        """Make device memory that did not come from alloc() reachable by the zero-copy collectives."""
        _check(lib().xmpi_register(self.handle, _ptr(ptr), nbytes), "xmpi_register")

    def deregister(self, ptr) -> None:
        _check(lib().xmpi_deregister(self.handle, _ptr(ptr)), "xmpi_deregister")

    def count_mismatch(self, a, b, nbytes: int) -> int:
        out = C.c_uint64(0)
        _check(lib().xmpi_count_mismatch(self.handle, _ptr(a), _ptr(b), nbytes, C.byref(out)), "count_mismatch")
        return out.value

    def checksum(self, buf, nbytes: int) -> int:
        out = C.c_uint64(0)
        _check(lib().xmpi_checksum(self.handle, _ptr(buf), nbytes, C.byref(out)), "checksum")
        return out.value

    def diff_stats(self, a, b, count: int, dtype: int):
        out = (C.c_double * 3)()
        _check(lib().xmpi_diff_stats(self.handle, _ptr(a), _ptr(b), count, dtype, out), "diff_stats")
        return out[0], out[1], out[2]

    def diff_rel(self, a, b, count: int, dtype: int) -> float:
        """max_i |a_i - b_i| / |b_i| over the whole buffer, on the device"""
        out = C.c_double(0)
        _check(lib().xmpi_diff_rel(self.handle, _ptr(a), _ptr(b), count, dtype, C.byref(out)), "diff_rel")
        return out.value

    def fill(self, buf, count: int, dtype: int, pattern: int, seed: int) -> None:
        _check(lib().xmpi_fill_pattern(self.handle, _ptr(buf), count, dtype, pattern, seed), "fill_pattern")

    # -- memory / tuning -----------------------------------------------------------------------
    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def memset(self, buf, byte: int, nbytes: int) -> None:
        _check(lib().xmpi_memset(self.handle, _ptr(buf), byte, nbytes), "memset")

    def memcpy(self, dst, src, nbytes: int) -> None:
        """blocking copy between any two of {host, this rank's HBM}"""
        _check(lib().xmpi_memcpy(self.handle, _ptr(dst), _ptr(src), nbytes), "memcpy")

    def sync(self) -> None:
        _check(lib().xmpi_sync(self.handle), "xmpi_sync")

    def set_param(self, name: str, value: int) -> None:
        _check(lib().xmpi_set_param(self.handle, name.encode(), value), "set_param " + name)

    def get_param(self, name: str) -> int:
        return lib().xmpi_get_param(self.handle, name.encode())

    def degraded(self) -> str:
        """what xmpi_init's vote left the job with, in words ("" = nothing degraded); get_param("degraded") is the level"""
        return (lib().xmpi_degraded(self.handle) or b"").decode()

    def link_probe(self, peer: int, nbytes: int, engine: int, iters: int = 10, direction: int = 0) -> float:
        out = C.c_double(0)
        _check(lib().xmpi_link_probe(self.handle, peer, nbytes, engine, iters, direction, C.byref(out)), "link_probe")
        return out.value

    def prof_enable(self, on: bool = True) -> None:
        _check(lib().xmpi_prof_enable(self.handle, 1 if on else 0), "prof_enable")

    def prof_reset(self) -> None:
        _check(lib().xmpi_prof_reset(self.handle), "prof_reset")

    def prof_get(self, kind: int):
        n, ms, b = C.c_uint64(0), C.c_double(0), C.c_uint64(0)
        _check(lib().xmpi_prof_get(self.handle, kind, C.byref(n), C.byref(ms), C.byref(b)), "prof_get")
        return n.value, ms.value, b.value
