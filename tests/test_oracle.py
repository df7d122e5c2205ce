"""Pins the CPU oracle (oracle/xmpi_oracle.c): against the committed known-answer file
(tests/golden/collectives_kat.json, produced by an independent numpy restatement) and against
numpy arithmetic for every dtype and operator.  The reference itself holds no vectors for the
collectives (they do not exist upstream: mpi.go:130) -- parity for them is "unpinned" upstream
and defined by this project as the rank-order fold."""
import json
import os

import numpy as np
import pytest

from oracle import oracle
from tests.scenarios import np_hash

HERE = os.path.dirname(os.path.abspath(__file__))
NPDT = {"f32": (oracle.F32, np.float32), "f64": (oracle.F64, np.float64), "f16": (oracle.F16, np.float16),
        "i64": (oracle.I64, np.int64), "i32": (oracle.I32, np.int32)}


def kat():
    with open(os.path.join(HERE, "golden", "collectives_kat.json")) as f:
        return json.load(f)["cases"]


def test_golden_allreduce_and_allgather():
    n = 0
    for c in kat():
        if c["kind"] == "allreduce_sum":
            code, npdt = NPDT[c["dtype"]]
            pat = 0 if c["pattern"] == "uniform" else 2
            ins = [oracle.fill(c["count"], code, pat, c["seed0"] + r) for r in range(c["ranks"])]
            if "inputs_rank0_hex" in c:
                assert ins[0].tobytes().hex() == c["inputs_rank0_hex"], "oracle_fill differs from the golden input"
            got = oracle.reduce_ranks(ins, code, oracle.SUM)
            assert got.tobytes().hex() == c["result_hex"], c
            n += 1
        elif c["kind"] == "allgather":
            ins = [oracle.fill(c["count"], oracle.I64, 1, r) for r in range(c["ranks"])]
            assert oracle.allgather(ins, oracle.I64).tobytes().hex() == c["result_hex"]
            n += 1
        elif c["kind"] == "fold_order":
            vals = [np.array([v], dtype=np.float32) for v in c["values"]]
            got = oracle.reduce_ranks(vals, oracle.F32, oracle.SUM)
            assert float(got[0]) == c["rank_order_result"] == 1.0  # (1e8 + 1) - 1e8 + 1 in f32, left to right
            n += 1
    assert n >= 20


def test_hash_matches_numpy_restatement():
    L = oracle.lib()
    idx = np.array([0, 1, 2, 12345, 2 ** 40 + 7, 2 ** 63], dtype=np.uint64)
    for seed in (0, 1, 1000, 2 ** 33 + 5):
        want = np_hash(seed, idx)
        got = np.array([L.oracle_hash(seed, int(i)) for i in idx], dtype=np.uint64)
        assert np.array_equal(got, want)


def test_half_conversions_against_numpy():
    L = oracle.lib()
    allh = np.arange(65536, dtype=np.uint16)
    f = allh.view(np.float16).astype(np.float32)
    for h in range(0, 65536, 7):
        x = L.oracle_half_to_float(h)
        assert (np.isnan(x) and np.isnan(f[h])) or np.float32(x).tobytes() == f[h].tobytes(), h
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.standard_normal(3000) * 10.0 ** rng.integers(-9, 6, 3000),
                         np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e9, -1e9, 2.0 ** -24, 2.0 ** -25,
                                   2.0 ** -25 * 1.0000001, 2.0 ** -26, 6.1e-5, 6.0975e-5, np.inf, -np.inf])])
    for x in xs:
        want = np.float64(x).astype(np.float16).view(np.uint16)
        assert L.oracle_double_to_half(float(x)) == int(want), x
    # every exact midpoint between adjacent halves rounds to even
    for h in range(0x0001, 0x7BFF, 97):
        lo, hi = np.uint16(h).view(np.float16), np.uint16(h + 1).view(np.float16)
        mid = (float(lo) + float(hi)) / 2
        assert L.oracle_double_to_half(mid) == (h if h % 2 == 0 else h + 1)


def test_bf16_conversions():
    L = oracle.lib()
    rng = np.random.default_rng(1)
    xs = (rng.standard_normal(5000) * 10.0 ** rng.integers(-20, 20, 5000)).astype(np.float32)
    for x in xs:
        u = int(np.float32(x).view(np.uint32))
        want = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF
        assert L.oracle_float_to_bf16(float(x)) == want
        assert np.float32(L.oracle_bf16_to_float(want)).view(np.uint32) == np.uint32(want << 16)
    assert L.oracle_float_to_bf16(float("nan")) & 0x7FC0 == 0x7FC0


@pytest.mark.parametrize("code,npdt", [(oracle.F32, np.float32), (oracle.F64, np.float64), (oracle.F16, np.float16),
                                       (oracle.I32, np.int32), (oracle.I64, np.int64), (oracle.U8, np.uint8)])
def test_reduce2_against_numpy(code, npdt):
    n = 5003
    for pat in (0, 3):
        a, b = oracle.fill(n, code, pat, 1), oracle.fill(n, code, pat, 2)
        with np.errstate(over="ignore"):
            assert oracle.reduce2(a, b, code, oracle.SUM).tobytes() == (a + b).astype(npdt).tobytes()
            assert oracle.reduce2(a, b, code, oracle.PROD).tobytes() == (a * b).astype(npdt).tobytes()
        assert oracle.reduce2(a, b, code, oracle.MIN).tobytes() == np.where(b < a, b, a).tobytes()
        assert oracle.reduce2(a, b, code, oracle.MAX).tobytes() == np.where(a < b, b, a).tobytes()


def test_reduce2_bf16_against_float32_math():
    n = 4001
    a, b = oracle.fill(n, oracle.BF16, 3, 1), oracle.fill(n, oracle.BF16, 3, 2)
    fa = (a.astype(np.uint32) << 16).view(np.float32)
    fb = (b.astype(np.uint32) << 16).view(np.float32)
    s = (fa + fb).astype(np.float32).view(np.uint32)
    want = ((s + 0x7FFF + ((s >> 16) & 1)) >> 16).astype(np.uint16)
    assert oracle.reduce2(a, b, oracle.BF16, oracle.SUM).tobytes() == want.tobytes()


def test_fill_patterns_are_exactly_representable_and_summable():
    """BASELINE cfg 5: fp16 inputs k/64 make every 8-way partial sum exact, so ANY summation order
    gives the same bits -- the basis of the bit-exact fp16 ring / halving parity tests."""
    n = 20000
    ins = [oracle.fill(n, oracle.F16, 0, 2000 + r) for r in range(8)]
    for x in ins:
        assert np.all(x.astype(np.float64) * 64 == np.round(x.astype(np.float64) * 64))
        assert x.min() >= 0 and x.max() < 1
    exact = np.sum([x.astype(np.float64) for x in ins], axis=0)
    got = oracle.reduce_ranks(ins, oracle.F16, oracle.SUM).astype(np.float64)
    assert np.array_equal(got, exact)
    rev = oracle.reduce_ranks(ins[::-1], oracle.F16, oracle.SUM)
    assert rev.tobytes() == oracle.reduce_ranks(ins, oracle.F16, oracle.SUM).tobytes()
    i64 = oracle.fill(16, oracle.I64, 1, 3)
    assert np.array_equal(i64, (np.int64(3) << 40) | np.arange(16, dtype=np.int64))
    u = oracle.fill(100000, oracle.F32, 0, 1000)
    assert 0 <= u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.01


def test_f32_order_tolerance_bound_holds():
    """|ring-like order - rank order| <= 1e-6 * sum|x_i| for positive uniform inputs (BASELINE.md cfg 4)"""
    n, ranks = 50000, 8
    ins = [oracle.fill(n, oracle.F32, 0, 1000 + r) for r in range(ranks)]
    want = oracle.reduce_ranks(ins, oracle.F32, oracle.SUM).astype(np.float64)
    scale = np.sum([x.astype(np.float64) for x in ins], axis=0)
    for shift in range(1, ranks):
        other = oracle.reduce_ranks(ins[shift:] + ins[:shift], oracle.F32, oracle.SUM).astype(np.float64)
        assert np.all(np.abs(other - want) <= 1e-6 * scale)


def test_checksum_and_mismatch_helpers():
    a = oracle.fill(1003, oracle.U8, 0, 4)
    b = a.copy()
    b[[0, 500, 1002]] ^= 1
    assert oracle.count_mismatch(a, b) == 3
    words = np.frombuffer(a[:1000].tobytes(), dtype="<u4").astype(np.uint64).sum() + int(a[1000:].astype(np.uint64).sum())
    assert oracle.checksum(a) == int(words)
