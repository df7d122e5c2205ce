"""ctypes loader of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; nothing
under mpi_amd/ may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_lib = None

U8, I32, I64, F16, F32, F64, BF16 = range(7)
SUM, PROD, MIN, MAX = range(4)
NP = {U8: np.uint8, I32: np.int32, I64: np.int64, F16: np.float16, F32: np.float32, F64: np.float64, BF16: np.uint16}


def build() -> str:
    src = os.path.join(_HERE, "xmpi_oracle.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
                               src, "-o", _SO, "-lm"])
    return _SO


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        P, I, Z, U64 = C.c_void_p, C.c_int, C.c_size_t, C.c_uint64
        L.oracle_dtype_size.restype, L.oracle_dtype_size.argtypes = Z, [I]
        L.oracle_hash.restype, L.oracle_hash.argtypes = U64, [U64, U64]
        L.oracle_fill.restype, L.oracle_fill.argtypes = I, [P, Z, I, I, U64]
        L.oracle_fill_range.restype, L.oracle_fill_range.argtypes = I, [P, Z, Z, I, I, U64]
        L.oracle_check_allreduce.restype = U64
        L.oracle_check_allreduce.argtypes = [P, Z, Z, I, I, U64, I, I, C.POINTER(U64)]
        L.oracle_reduce2.restype, L.oracle_reduce2.argtypes = I, [P, P, P, Z, I, I]
        L.oracle_reduce_ranks.restype, L.oracle_reduce_ranks.argtypes = I, [P, C.POINTER(P), I, Z, I, I]
        L.oracle_allgather.restype, L.oracle_allgather.argtypes = I, [P, C.POINTER(P), I, Z, I]
        L.oracle_half_to_float.restype, L.oracle_half_to_float.argtypes = C.c_float, [C.c_uint16]
        L.oracle_double_to_half.restype, L.oracle_double_to_half.argtypes = C.c_uint16, [C.c_double]
        L.oracle_bf16_to_float.restype, L.oracle_bf16_to_float.argtypes = C.c_float, [C.c_uint16]
        L.oracle_float_to_bf16.restype, L.oracle_float_to_bf16.argtypes = C.c_uint16, [C.c_float]
        L.oracle_count_mismatch.restype, L.oracle_count_mismatch.argtypes = U64, [P, P, Z]
        L.oracle_checksum.restype, L.oracle_checksum.argtypes = U64, [P, Z]
        L.oracle_diff_stats.restype, L.oracle_diff_stats.argtypes = I, [P, P, Z, I, C.POINTER(C.c_double)]
        _lib = L
    return _lib


def fill(count: int, dtype: int, pattern: int, seed: int) -> np.ndarray:
    out = np.empty(count, dtype=NP[dtype])
    rc = lib().oracle_fill(out.ctypes.data, count, dtype, pattern, seed)
    assert rc == 0
    return out


def fill_range(start: int, count: int, dtype: int, pattern: int, seed: int) -> np.ndarray:
    out = np.empty(count, dtype=NP[dtype])
    rc = lib().oracle_fill_range(out.ctypes.data, start, count, dtype, pattern, seed)
    assert rc == 0
    return out


def check_allreduce(got: np.ndarray, start: int, dtype: int, pattern: int, seed0: int, nranks: int, op: int):
    """(number of elements of `got` = result[start : start + got.size] whose bits differ from the rank-order fold of
    the ranks' regenerated inputs, index of the first one)"""
    got = np.ascontiguousarray(got)
    first = C.c_uint64(0)
    bad = lib().oracle_check_allreduce(got.ctypes.data, start, got.size, dtype, pattern, seed0, nranks, op, C.byref(first))
    return int(bad), int(first.value)


def reduce2(a: np.ndarray, b: np.ndarray, dtype: int, op: int) -> np.ndarray:
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    out = np.empty_like(a)
    rc = lib().oracle_reduce2(out.ctypes.data, a.ctypes.data, b.ctypes.data, a.size, dtype, op)
    assert rc == 0
    return out


def reduce_ranks(inputs: Sequence[np.ndarray], dtype: int, op: int) -> np.ndarray:
    """Rank-order left-to-right fold: what a reference user computes after gathering everything."""
    ins = [np.ascontiguousarray(x) for x in inputs]
    out = np.empty_like(ins[0])
    ptrs = (C.c_void_p * len(ins))(*[x.ctypes.data for x in ins])
    rc = lib().oracle_reduce_ranks(out.ctypes.data, ptrs, len(ins), ins[0].size, dtype, op)
    assert rc == 0
    return out


def allgather(inputs: Sequence[np.ndarray], dtype: int) -> np.ndarray:
    ins = [np.ascontiguousarray(x) for x in inputs]
    out = np.empty(ins[0].size * len(ins), dtype=ins[0].dtype)
    ptrs = (C.c_void_p * len(ins))(*[x.ctypes.data for x in ins])
    rc = lib().oracle_allgather(out.ctypes.data, ptrs, len(ins), ins[0].size, dtype)
    assert rc == 0
    return out


def checksum(a: np.ndarray) -> int:
    a = np.ascontiguousarray(a)
    return int(lib().oracle_checksum(a.ctypes.data, a.nbytes))


def count_mismatch(a: np.ndarray, b: np.ndarray) -> int:
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    assert a.nbytes == b.nbytes
    return int(lib().oracle_count_mismatch(a.ctypes.data, b.ctypes.data, a.nbytes))


def diff_stats(a: np.ndarray, b: np.ndarray, dtype: int):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    out = (C.c_double * 3)()
    rc = lib().oracle_diff_stats(a.ctypes.data, b.ctypes.data, a.size, dtype, out)
    assert rc == 0
    return out[0], out[1], out[2]


def as_float64(a: np.ndarray, dtype: int) -> np.ndarray:
    """Decode any float dtype (bf16 given as uint16 bit patterns) to float64 for error analysis."""
    if dtype == BF16:
        return (a.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    return a.astype(np.float64)
