"""ctypes binding of oracle/libminlz_oracle.so (CPU restatement of the reference's pure-Go
MinLZ block codec; reference file:line citations live in minlz_oracle.c).

TEST INFRASTRUCTURE ONLY: never imported by minlz_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libminlz_oracle.so")

OK, ERR_CORRUPT, ERR_TOO_LARGE, ERR_UNSUPPORTED, ERR_INVALID_LEVEL, ERR_CRC, ERR_DST_TOO_SMALL = range(7)
MAX_BLOCK_SIZE = 8 << 20


class OracleError(Exception):
    def __init__(self, code):
        super().__init__("oracle error %d" % code)
        self.code = code


def build(force=False):
    src = os.path.join(_HERE, "minlz_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p, sz, szp = C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)
        L.mlzo_decode_body.argtypes = [u8p, sz, u8p, sz]; L.mlzo_decode_body.restype = C.c_int
        L.mlzo_decode.argtypes = [u8p, sz, u8p, sz, szp]; L.mlzo_decode.restype = C.c_int
        L.mlzo_decoded_len.argtypes = [u8p, sz, szp]; L.mlzo_decoded_len.restype = C.c_int
        L.mlzo_emit_literal.argtypes = [u8p, u8p, sz]; L.mlzo_emit_literal.restype = sz
        L.mlzo_emit_repeat.argtypes = [u8p, sz]; L.mlzo_emit_repeat.restype = sz
        L.mlzo_emit_copy.argtypes = [u8p, sz, sz]; L.mlzo_emit_copy.restype = sz
        L.mlzo_emit_copy_lits2.argtypes = [u8p, u8p, sz, sz, sz]; L.mlzo_emit_copy_lits2.restype = sz
        L.mlzo_emit_copy_lits3.argtypes = [u8p, u8p, sz, sz, sz]; L.mlzo_emit_copy_lits3.restype = sz
        L.mlzo_max_encoded_len.argtypes = [sz]; L.mlzo_max_encoded_len.restype = C.c_long
        L.mlzo_encode_block_l0.argtypes = [u8p, u8p, sz]; L.mlzo_encode_block_l0.restype = sz
        L.mlzo_encode_block_l1.argtypes = [u8p, u8p, sz]; L.mlzo_encode_block_l1.restype = sz
        L.mlzo_encode_block_l2.argtypes = [u8p, u8p, sz]; L.mlzo_encode_block_l2.restype = sz
        L.mlzo_encode_block_l3.argtypes = [u8p, u8p, sz]; L.mlzo_encode_block_l3.restype = sz
        L.mlzo_encode.argtypes = [u8p, sz, u8p, sz, C.c_int]; L.mlzo_encode.restype = C.c_long
        L.mlzo_crc.argtypes = [u8p, sz]; L.mlzo_crc.restype = C.c_uint32
        L.mlzo_stream_bound.argtypes = [sz, sz]; L.mlzo_stream_bound.restype = sz
        L.mlzo_stream_encode.argtypes = [u8p, sz, u8p, sz, C.c_int, sz]; L.mlzo_stream_encode.restype = C.c_long
        L.mlzo_stream_decode.argtypes = [u8p, sz, u8p, sz, szp]; L.mlzo_stream_decode.restype = C.c_int
        L.mlzo_bench_encode.argtypes = [u8p, sz, sz, C.c_int, C.c_int, C.c_int, szp]; L.mlzo_bench_encode.restype = C.c_double
        L.mlzo_bench_decode.argtypes = [u8p, sz, sz, C.c_int, C.c_int, C.c_int]; L.mlzo_bench_decode.restype = C.c_double
        _lib = L
    return _lib


def _buf(b):
    """bytes / bytearray / numpy -> (keepalive, pointer, length)."""
    a = np.frombuffer(b, dtype=np.uint8) if not isinstance(b, np.ndarray) else np.ascontiguousarray(b, dtype=np.uint8)
    return a, a.ctypes.data if a.size else None, a.size


def max_encoded_len(n):
    return lib().mlzo_max_encoded_len(n)


def encode(src, level=1):
    """minlz.Encode(nil, src, level) (encode.go:74)."""
    a, p, n = _buf(src)
    cap_ = n + 16
    out = np.empty(cap_, dtype=np.uint8)
    r = lib().mlzo_encode(out.ctypes.data, cap_, p, n, level)
    if r < 0:
        raise OracleError(-r)
    return out[:r].tobytes()


def decode(src, guard=0):
    """minlz.Decode(nil, src) (decode.go:50). Raises OracleError(code)."""
    a, p, n = _buf(src)
    dl = C.c_size_t(0)
    e = lib().mlzo_decoded_len(p, n, C.byref(dl))
    if e:
        raise OracleError(e)
    cap_ = dl.value
    out = np.full(cap_ + guard, 0xA5, dtype=np.uint8)
    e = lib().mlzo_decode(p, n, out.ctypes.data, cap_, C.byref(dl))
    if guard and not (out[cap_:] == 0xA5).all():
        raise AssertionError("oracle wrote past dst")
    if e:
        raise OracleError(e)
    return out[:dl.value].tobytes()


def decoded_len(src):
    a, p, n = _buf(src)
    dl = C.c_size_t(0)
    e = lib().mlzo_decoded_len(p, n, C.byref(dl))
    if e:
        raise OracleError(e)
    return dl.value


def decode_body(body, dlen):
    """minLZDecode(dst[:dlen], body) -> (code, bytes)."""
    a, p, n = _buf(body)
    out = np.zeros(max(dlen, 1), dtype=np.uint8)
    e = lib().mlzo_decode_body(out.ctypes.data, dlen, p, n)
    return e, out[:dlen].tobytes()


def encode_block(src, level=1):
    """encodeBlock / encodeBlockBetter: token stream only; b'' = incompressible."""
    a, p, n = _buf(src)
    out = np.empty(n + 64, dtype=np.uint8)
    f = {-1: lib().mlzo_encode_block_l0, 1: lib().mlzo_encode_block_l1, 2: lib().mlzo_encode_block_l2, 3: lib().mlzo_encode_block_l3}[level]
    r = f(out.ctypes.data, p, n)
    return out[:r].tobytes()


def emit_literal(lit):
    a, p, n = _buf(lit)
    out = np.empty(n + 8, dtype=np.uint8)
    r = lib().mlzo_emit_literal(out.ctypes.data, p, n)
    return out[:r].tobytes()


def emit_repeat(length):
    out = np.empty(8, dtype=np.uint8)
    return out[:lib().mlzo_emit_repeat(out.ctypes.data, length)].tobytes()


def emit_copy(offset, length):
    out = np.empty(16, dtype=np.uint8)
    return out[:lib().mlzo_emit_copy(out.ctypes.data, offset, length)].tobytes()


def emit_copy_lits2(lits, offset, length):
    a, p, n = _buf(lits)
    out = np.empty(24, dtype=np.uint8)
    return out[:lib().mlzo_emit_copy_lits2(out.ctypes.data, p, n, offset, length)].tobytes()


def emit_copy_lits3(lits, offset, length):
    a, p, n = _buf(lits)
    out = np.empty(24, dtype=np.uint8)
    return out[:lib().mlzo_emit_copy_lits3(out.ctypes.data, p, n, offset, length)].tobytes()


def crc(b):
    a, p, n = _buf(b)
    return lib().mlzo_crc(p, n)


def stream_encode(src, level=1, block_size=8 << 20, add_index=False):
    a, p, n = _buf(src)
    L = lib()
    L.mlzo_index_bound.argtypes = [C.c_size_t]; L.mlzo_index_bound.restype = C.c_size_t
    L.mlzo_stream_encode_ex.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_int]
    L.mlzo_stream_encode_ex.restype = C.c_long
    cap_ = L.mlzo_stream_bound(n, block_size) + (L.mlzo_index_bound((n + block_size - 1) // block_size + 2) if add_index else 0)   # (+ the header's entry: the index holds blocks + 1 offsets)
    out = np.empty(cap_, dtype=np.uint8)
    r = L.mlzo_stream_encode_ex(out.ctypes.data, cap_, p, n, level, block_size, 1 if add_index else 0)
    if r < 0:
        raise OracleError(-r)
    return out[:r].tobytes()


def stream_decode(src, max_out):
    a, p, n = _buf(src)
    out = np.empty(max(max_out, 1), dtype=np.uint8)
    dl = C.c_size_t(0)
    e = lib().mlzo_stream_decode(p, n, out.ctypes.data, max_out, C.byref(dl))
    if e:
        raise OracleError(e)
    return out[:dl.value].tobytes()


def bench_encode(src, block_size, level, threads, reps=1):
    """Returns (seconds, total_compressed_bytes) for reps passes over src."""
    a, p, n = _buf(src)
    tot = C.c_size_t(0)
    dt = lib().mlzo_bench_encode(p, n, block_size, level, threads, reps, C.byref(tot))
    return dt, tot.value


def bench_decode(src, block_size, level, threads, reps=1):
    a, p, n = _buf(src)
    return lib().mlzo_bench_decode(p, n, block_size, level, threads, reps)


def index_build(c_off, u_off, block_size, total_u, total_c):
    """Index.reset(block_size) + add() per block + appendTo (index.go:56-269) -> index chunk bytes."""
    L = lib()
    L.mlzo_index_bound.argtypes = [C.c_size_t]; L.mlzo_index_bound.restype = C.c_size_t
    L.mlzo_index_build.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int64, C.c_int64]
    L.mlzo_index_build.restype = C.c_long
    c = np.ascontiguousarray(c_off, dtype=np.int64); u = np.ascontiguousarray(u_off, dtype=np.int64)
    cap_ = L.mlzo_index_bound(c.size)
    out = np.empty(cap_, dtype=np.uint8)
    r = L.mlzo_index_build(out.ctypes.data, cap_, c.ctypes.data, u.ctypes.data, c.size, block_size, total_u, total_c)
    if r < 0:
        raise OracleError(-r)
    return out[:r].tobytes()


def index_load(b):
    """Index.Load (index.go:273-396) -> (total_u, total_c, est_block, [(c_off, u_off)...], consumed)."""
    L = lib()
    L.mlzo_index_load.argtypes = [C.c_void_p, C.c_size_t] + [C.c_void_p] * 5 + [C.c_size_t, C.c_void_p, C.c_void_p]
    L.mlzo_index_load.restype = C.c_int
    a, p, n = _buf(b)
    tu, tc, est = C.c_int64(), C.c_int64(), C.c_int64()
    co = np.zeros(1 << 16, dtype=np.int64); uo = np.zeros(1 << 16, dtype=np.int64)
    ne, used = C.c_size_t(), C.c_size_t()
    r = L.mlzo_index_load(p, n, C.byref(tu), C.byref(tc), C.byref(est), co.ctypes.data, uo.ctypes.data, co.size, C.byref(ne), C.byref(used))
    if r:
        raise OracleError(r)
    k = ne.value
    return tu.value, tc.value, est.value, list(zip(co[:k].tolist(), uo[:k].tolist())), used.value
