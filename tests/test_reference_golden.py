"""Fixtures produced BY THE REFERENCE (go/golden/gen_golden.go: encoding/gob as network.go uses it, and mpi.Network itself
over localhost TCP) against this repository's restatements.  The image this repository is built in has no Go toolchain, so
the fixtures are absent there and these tests SKIP; wherever Go exists one command turns them on:

    cd go && go run ./golden -out ../tests/golden && cd .. && python -m pytest tests/test_reference_golden.py

What they pin: the gob codec of the oracle (oracle/gob_codec.h) and of the product (mpi_amd/host/gobwire.hpp) byte for
byte against real gob streams -- values, the handshake struct, the message and ack frames; the oracle's rank-order fold
(oracle_reduce_ranks) bit for bit against Go's own arithmetic on the oracle's own inputs; lossless Send / Receive.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOB = os.path.join(ROOT, "tests", "golden", "ref_gob.json")
TRANS = os.path.join(ROOT, "tests", "golden", "ref_transcripts.json")
needs_gob = pytest.mark.skipif(not os.path.exists(GOB), reason="tests/golden/ref_gob.json absent: generate it with go/golden/gen_golden.go (needs Go)")
needs_trans = pytest.mark.skipif(not os.path.exists(TRANS), reason="tests/golden/ref_transcripts.json absent: generate it with go/golden/gen_golden.go (needs Go)")

NP = {"[]float64": np.float64, "[]float32": np.float32, "[]int64": np.int64, "[]byte": np.uint8}
ENC = {"[]float64": "gobx_encode_f64", "[]float32": "gobx_encode_f32", "[]int64": "gobx_encode_i64", "[]byte": "gobx_encode_bytes"}
DEC = {"[]float64": "gobx_decode_f64", "[]float32": "gobx_decode_f32", "[]int64": "gobx_decode_i64", "[]byte": "gobx_decode_bytes"}


def _oracle_codec():
    L = C.CDLL(os.path.join(ROOT, "oracle", "librefpath.so"))
    for name in ENC.values():
        getattr(L, name).restype = C.c_size_t
        getattr(L, name).argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    for name in DEC.values():
        getattr(L, name).restype = C.c_long
        getattr(L, name).argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.gobx_encode_message.restype = C.c_size_t
    L.gobx_encode_message.argtypes = [C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    return L


@needs_gob
def test_oracle_codec_matches_real_gob():
    L = _oracle_codec()
    fx = json.load(open(GOB))
    checked = 0
    for case in fx["values"]:
        if case["go_type"] not in NP:
            continue
        raw, want = bytes.fromhex(case["raw_le_hex"]), bytes.fromhex(case["gob_hex"])
        arr = np.frombuffer(raw, dtype=NP[case["go_type"]]).copy()
        cap = len(want) + len(raw) + 256
        buf = (C.c_uint8 * cap)()
        n = getattr(L, ENC[case["go_type"]])(arr.ctypes.data, arr.size, buf, cap)
        assert bytes(buf[:n]) == want, f"{case['name']}: the oracle's encoding differs from gob's"
        out = np.empty(arr.size + 1, dtype=arr.dtype)
        got = getattr(L, DEC[case["go_type"]])(want, len(want), out.ctypes.data, out.size)
        assert got == arr.size and out[:arr.size].tobytes() == raw, f"{case['name']}: decoding gob's bytes"
        checked += 1
    for fr in fx["frames"]:
        if fr["name"] not in ("message", "ack"):
            continue
        payload, want = bytes.fromhex(fr.get("payload_hex", "")), bytes.fromhex(fr["gob_hex"])
        buf = (C.c_uint8 * (len(want) + 256))()
        n = L.gobx_encode_message(fr["tag"], payload or None, len(payload), buf, len(buf))
        assert bytes(buf[:n]) == want, f"{fr['name']} tag {fr['tag']}: the oracle's frame differs from the reference's"
        checked += 1
    assert checked > 10


@needs_gob
def test_product_codec_matches_real_gob(tmp_path):
    exe = str(tmp_path / "golden_gobwire_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "mpi_amd", "host"),
                           os.path.join(ROOT, "tests", "golden_gobwire_check.cpp"), "-o", exe])
    fx = json.load(open(GOB))
    lines = []
    for case in fx["values"]:
        lines.append("\t".join([case["go_type"], "", "", case["raw_le_hex"], case["gob_hex"]]))
    for fr in fx["frames"]:
        if fr["name"] == "initialMessage":
            lines.append("\t".join(["initialMessage", str(fr["id"]), fr["password"], "", fr["gob_hex"]]))
        else:
            lines.append("\t".join([fr["name"], str(fr["tag"]), "", fr.get("payload_hex", ""), fr["gob_hex"]]))
    out = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and out.stdout.strip().startswith("ok"), out.stdout[-2000:]


@needs_trans
def test_send_receive_is_lossless_in_the_reference():
    """bounce.go:105,133 and helloworld.go:59-62,78 as the reference itself ran them: the property the oracle assumes"""
    t = json.load(open(TRANS))
    for row in t["bounce_echo"]:
        assert row["sent_bytes_hex"] == row["echo_bytes_hex"] and row["sent_f64_le_hex"] == row["echo_f64_le_hex"], row["length"]
    n = len(t["helloworld_received"])
    for rank, got in t["helloworld_received"].items():
        want = [f"\"I'm just node {rank} talking to myself\"" if src == int(rank) else f"\"Hello node {rank}, I'm node {src}\"" for src in range(n)]
        assert got == want


@needs_trans
def test_oracle_fold_is_the_reference_users_arithmetic():
    """an allreduce composed from the reference's Send / Receive and folded in rank order IN GO == oracle_reduce_ranks on the
    same (oracle_fill) inputs, bit for bit: pins `one rounding per operation, in the element type, rank order`"""
    from oracle import oracle
    t = json.load(open(TRANS))
    kinds = {"f32_uniform": (oracle.F32, 0), "f32_signed": (oracle.F32, 3), "f64_uniform": (oracle.F64, 0), "i64_uniform": (oracle.I64, 0)}
    for row in t["allreduce_rank_order"]:
        dtype, pattern = kinds[row["name"]]
        ranks, count, seed0 = int(row["ranks"]), int(row["count"]), int(row["seed0"])
        ins = [oracle.fill(count, dtype, pattern, seed0 + r) for r in range(ranks)]
        assert ins[0].tobytes() == bytes.fromhex(row["input_rank0_le_hex"]), f"{row['name']}: the generators differ (Go vs oracle_fill)"
        assert oracle.reduce_ranks(ins, dtype, oracle.SUM).tobytes() == bytes.fromhex(row["result_le_hex"]), row["name"]
