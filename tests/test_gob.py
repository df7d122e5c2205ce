"""Pins the gob codec restatement (oracle/gob_codec.h) against the known-answer vectors of gob's
published format document, and checks that the codec is lossless for the payload types the
reference's programs send (bounce.go: []byte, []float64; helloworld.go: string).  The reference's
own checks for this path are exactly "round trip is lossless" (bounce.go:105,133)."""
import ctypes as C
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(os.path.dirname(HERE), "oracle", "librefpath.so")


@pytest.fixture(scope="module")
def g():
    L = C.CDLL(SO)
    for name in ("gobx_uint", "gobx_int", "gobx_float", "gobx_point", "gobx_encode_f64", "gobx_encode_f32",
                 "gobx_encode_i64", "gobx_encode_bytes", "gobx_encode_message"):
        getattr(L, name).restype = C.c_size_t
    for name in ("gobx_decode_f64", "gobx_decode_f32", "gobx_decode_i64", "gobx_decode_bytes", "gobx_decode_message"):
        getattr(L, name).restype = C.c_long
    L.gobx_uint.argtypes = [C.c_uint64, C.c_void_p, C.c_size_t]
    L.gobx_int.argtypes = [C.c_int64, C.c_void_p, C.c_size_t]
    L.gobx_float.argtypes = [C.c_double, C.c_void_p, C.c_size_t]
    L.gobx_point.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_size_t]
    for name in ("gobx_encode_f64", "gobx_encode_f32", "gobx_encode_i64", "gobx_encode_bytes"):
        getattr(L, name).argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.gobx_encode_message.argtypes = [C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    for name in ("gobx_decode_f64", "gobx_decode_f32", "gobx_decode_i64", "gobx_decode_bytes"):
        getattr(L, name).argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.gobx_decode_message.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_int64), C.c_void_p, C.c_size_t]
    return L


def call(fn, *args, cap=1 << 16):
    buf = (C.c_uint8 * cap)()
    n = fn(*args, buf, cap)
    assert n <= cap
    return bytes(buf[:n])


def test_document_known_answers(g):
    assert call(g.gobx_uint, 7) == bytes.fromhex("07")
    assert call(g.gobx_uint, 127) == bytes.fromhex("7f")
    assert call(g.gobx_uint, 128) == bytes.fromhex("ff80")
    assert call(g.gobx_uint, 256) == bytes.fromhex("fe0100")
    assert call(g.gobx_int, -129) == bytes.fromhex("fe0101")
    assert call(g.gobx_int, 0) == bytes.fromhex("00")
    assert call(g.gobx_int, -1) == bytes.fromhex("01")
    assert call(g.gobx_int, 1) == bytes.fromhex("02")
    assert call(g.gobx_float, 17.0) == bytes.fromhex("fe3140")
    assert call(g.gobx_float, 0.0) == bytes.fromhex("00")
    # type Point struct{X, Y int}; Point{22, 33}: the 40-byte stream of the format document
    want = bytes.fromhex("1fff81030101" "05506f696e74" "01ff8200" "0102" "010158010400" "010159010400" "0000"
                         "07ff82012c014200")
    assert call(g.gobx_point, 22, 33) == want


def encode(g, fn, arr):
    a = np.ascontiguousarray(arr)
    cap = a.nbytes * 2 + 256
    buf = (C.c_uint8 * cap)()
    n = fn(a.ctypes.data, a.size, buf, cap)
    assert 0 < n <= cap
    return bytes(buf[:n])


@pytest.mark.parametrize("n", [0, 1, 7, 1000, 100000])
def test_float64_slice_round_trip_is_lossless(g, n):
    rng = np.random.default_rng(n)
    x = rng.random(n)  # rand.Float64(), bounce.go:75
    if n > 10:
        x[:6] = [0.0, -0.0, np.inf, -np.inf, 5e-324, 1.7976931348623157e308]
    wire = encode(g, g.gobx_encode_f64, x)
    out = np.empty(n + 1, dtype=np.float64)
    got = g.gobx_decode_f64(wire, len(wire), out.ctypes.data, out.size)
    assert got == n and out[:n].tobytes() == x.tobytes()
    if n >= 1000:  # random float64 costs ~9 bytes/element on the wire
        assert 8.5 < len(wire) / n < 9.5


def test_float32_slice_is_widened_exactly(g):
    x = np.random.default_rng(0).random(50000).astype(np.float32)
    wire = encode(g, g.gobx_encode_f32, x)
    out = np.empty(x.size, dtype=np.float32)
    assert g.gobx_decode_f32(wire, len(wire), out.ctypes.data, out.size) == x.size
    assert out.tobytes() == x.tobytes()
    assert 5.0 < len(wire) / x.size < 6.5  # ~6 bytes per float32


def test_int64_and_bytes_round_trip(g):
    rng = np.random.default_rng(1)
    x = rng.integers(-2 ** 63, 2 ** 63 - 1, 20000, dtype=np.int64)
    x[:4] = [0, -1, 2 ** 63 - 1, -2 ** 63]
    wire = encode(g, g.gobx_encode_i64, x)
    out = np.empty(x.size, dtype=np.int64)
    assert g.gobx_decode_i64(wire, len(wire), out.ctypes.data, out.size) == x.size
    assert np.array_equal(out, x)
    b = rng.integers(0, 256, 100001, dtype=np.uint8)
    wire = encode(g, g.gobx_encode_bytes, b)
    assert len(wire) < b.size + 16  # raw bytes + a small header
    outb = np.empty(b.size, dtype=np.uint8)
    assert g.gobx_decode_bytes(wire, len(wire), outb.ctypes.data, outb.size) == b.size
    assert outb.tobytes() == b.tobytes()


def test_message_envelope(g):
    """message{Tag int; Bytes Raw} (network.go:511-514): tag and payload survive; the ack is the
    same struct with no payload (network.go:616-621)."""
    payload = bytes(range(256)) * 40
    for tag in (0, 1, -5, 123456789):
        cap = len(payload) + 256
        buf = (C.c_uint8 * cap)()
        n = g.gobx_encode_message(tag, payload, len(payload), buf, cap)
        wire = bytes(buf[:n])
        t = C.c_int64(99)
        out = (C.c_uint8 * cap)()
        got = g.gobx_decode_message(wire, len(wire), C.byref(t), out, cap)
        assert got == len(payload) and t.value == tag and bytes(out[:got]) == payload
        n = g.gobx_encode_message(tag, None, 0, buf, cap)
        ack = bytes(buf[:n])
        got = g.gobx_decode_message(ack, len(ack), C.byref(t), out, cap)
        assert got == 0 and t.value == tag
