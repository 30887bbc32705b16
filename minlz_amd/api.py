"""Block API mirror of the reference (encode.go:74-244, decode.go:50-156) over the C ABI."""
import ctypes as C
import threading

import numpy as np

from . import _lib
from ._lib import BlockDesc

LevelSuperFast, LevelUncompressed, LevelFastest, LevelBalanced = -1, 0, 1, 2  # encode.go:25-43 (LevelSmallest = 3 is CPU-only)
MaxBlockSize = 8 << 20  # minlz.go:84

OPT_DECODE_ALGO, OPT_ENCODE_FAR, OPT_TIMING = 1, 2, 100
OPT_L2_FREE, OPT_GEN_SPIN, OPT_GEN_PACKED, OPT_INDEX_PASSES, OPT_DEVICE_GROUP, OPT_L2_GAP = 14, 9, 13, 15, 17, 19   # include/minlz_hip.h


class MinLZError(Exception):
    code = 0


class ErrCorrupt(MinLZError):
    """minlz: corrupt input (decode.go:31)"""
    code = 1


class ErrTooLarge(MinLZError):
    """minlz: decoded block is too large (decode.go:35)"""
    code = 2


class ErrUnsupported(MinLZError):
    """minlz: unsupported input (decode.go:37)"""
    code = 3


class ErrInvalidLevel(MinLZError):
    """minlz: invalid compression level (decode.go:39)"""
    code = 4


class ErrCRC(MinLZError):
    """minlz: corrupt input, crc mismatch (decode.go:33)"""
    code = 5


class ErrHIP(MinLZError):
    code = 7


_ERRS = {1: ErrCorrupt, 2: ErrTooLarge, 3: ErrUnsupported, 4: ErrInvalidLevel, 5: ErrCRC, 7: ErrHIP}


def _raise(code, ctx=None):
    code = -code if code < 0 else code
    msg = ""
    if ctx is not None and code == 7:
        msg = _lib.lib().mlz_last_error(ctx.handle).decode()
    raise _ERRS.get(code, MinLZError)("minlz error %d %s" % (code, msg))


class Context:
    """One HIP device context (mlz_ctx). Thread-safe on the C side."""

    def __init__(self, device=-1, devices=None):
        """device: one HIP device (mlz_init).  devices: a list of ordinals (repeats allowed) or "all" — one context that deals the blocks of
        batches and streams to all of them from this process (mlz_init_devices)."""
        self.handle = C.c_void_p()
        L = _lib.lib()
        if devices is None:
            r = L.mlz_init(device, C.byref(self.handle))
        elif isinstance(devices, str):
            assert devices == "all"
            r = L.mlz_init_devices(None, 0, C.byref(self.handle))
        else:
            arr = (C.c_int * len(devices))(*devices)
            r = L.mlz_init_devices(arr, len(devices), C.byref(self.handle))
        if r != 0:
            raise ErrHIP("mlz_init failed (%d): no HIP device?" % r)

    def device_count(self):
        return int(_lib.lib().mlz_device_count(self.handle))

    def close(self):
        if self.handle:
            _lib.lib().mlz_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, opt, value):
        r = _lib.lib().mlz_set_option(self.handle, opt, int(value))
        if r:
            _raise(r, self)

    def device_name(self):
        buf = C.create_string_buffer(256)
        _lib.lib().mlz_device_name(self.handle, buf, 256)
        return buf.value.decode()

    def timers(self):
        arr = (C.c_float * 16)()
        n = _lib.lib().mlz_get_timers(self.handle, arr, 16)
        return {_lib.lib().mlz_timer_name(i).decode(): arr[i] for i in range(n) if arr[i] >= 0}

    def combine_stats(self):
        """(batches run, requests served) by the combining queue of the single-block host calls."""
        L = _lib.lib()
        return int(L.mlz_get_counter(self.handle, 0)), int(L.mlz_get_counter(self.handle, 1))

    def workspace_bytes(self):
        """(encode side, decode side): device workspace the context holds — grow-only, i.e. the high-water mark of its calls (mlz_get_counter 3, 4)."""
        L = _lib.lib()
        return int(L.mlz_get_counter(self.handle, 3)), int(L.mlz_get_counter(self.handle, 4))

    def general_blocks(self):
        """Blocks of the last decode call that took the path for streams of other encoders (mlz_get_counter 2)."""
        return int(_lib.lib().mlz_get_counter(self.handle, 2))

    def general_team(self):
        """Workgroups per block (1, 2 or 4) the general-block pass of the last decode call settled with; 0 = no general block (mlz_get_counter 6)."""
        return int(_lib.lib().mlz_get_counter(self.handle, 6))

    def release_stream(self, stream):
        """mlz_release_stream: call before destroying a stream that carried a *_batch_device call of this context."""
        r = _lib.lib().mlz_release_stream(self.handle, stream)
        if r:
            _raise(r, self)

    def stream_encode_gather_device(self, level, block_size, add_index, d_srcs, lens, d_dst, dst_cap):
        """mlz_stream_encode_gather_device: ranges of one stream resident on the context's devices -> the framed stream in d_dst (device memory).
        Returns the stream size."""
        n = len(d_srcs)
        sp = (C.c_void_p * n)(*d_srcs); sl = (C.c_size_t * n)(*lens)
        r = _lib.lib().mlz_stream_encode_gather_device(self.handle, level, block_size, STREAM_ADD_INDEX if add_index else 0, sp, sl, n, d_dst, dst_cap)
        if r < 0:
            _raise(r, self)
        return int(r)

    # ---- device-resident batch calls: pointers are raw device addresses (e.g. tensor.data_ptr()) ----
    def encode_batch_device(self, stream, level, d_src, d_dst, descs, d_out_len):
        arr = (BlockDesc * len(descs))(*descs) if not isinstance(descs, C.Array) else descs
        r = _lib.lib().mlz_encode_batch_device(self.handle, stream, level, d_src, d_dst, arr, len(arr), d_out_len)
        if r:
            _raise(r, self)

    def crc_batch_device(self, stream, d_base, descs, d_out_u32):
        arr = (BlockDesc * len(descs))(*descs) if not isinstance(descs, C.Array) else descs
        r = _lib.lib().mlz_crc_batch_device(self.handle, stream, d_base, arr, len(arr), d_out_u32)
        if r:
            _raise(r, self)

    def decode_batch_device(self, stream, d_src, d_dst, descs, d_out_len):
        arr = (BlockDesc * len(descs))(*descs) if not isinstance(descs, C.Array) else descs
        r = _lib.lib().mlz_decode_batch_device(self.handle, stream, d_src, d_dst, arr, len(arr), d_out_len)
        if r:
            _raise(r, self)


_default = None
_default_lock = threading.Lock()


def default_context():
    global _default
    with _default_lock:
        if _default is None:
            _default = Context()
        return _default


def _np(b):
    if isinstance(b, np.ndarray):
        return np.ascontiguousarray(b, dtype=np.uint8)
    return np.frombuffer(b, dtype=np.uint8)


def _ptr(a):
    return a.ctypes.data if a.size else None


def MaxEncodedLen(n):
    """encode.go:234-244"""
    return _lib.lib().mlz_max_encoded_len(n) if n >= 0 else -1


def DecodedLen(src):
    """decode.go:107-110"""
    a = _np(src)
    r = _lib.lib().mlz_decoded_len(_ptr(a), a.size)
    if r < 0:
        _raise(r)
    return r


def IsMinLZ(src):
    """decode.go:114-117 -> (ok, size); raises like the reference returns err."""
    a = _np(src)
    r = _lib.lib().mlz_decoded_len(_ptr(a), a.size)
    if r < 0:
        _raise(r)
    return (a.size > 0 and a[0] == 0), r


def Encode(src, level=LevelFastest, ctx=None):
    """minlz.Encode(nil, src, level), encode.go:74-139."""
    ctx = ctx or default_context()
    a = _np(src)
    cap_ = MaxEncodedLen(a.size)
    if cap_ < 0:
        raise ErrTooLarge()
    out = np.empty(cap_, dtype=np.uint8)
    r = _lib.lib().mlz_encode(ctx.handle, level, _ptr(a), a.size, out.ctypes.data, cap_)
    if r < 0:
        _raise(r, ctx)
    return out[:r].tobytes()


def AppendEncoded(dst, src, level=LevelFastest, ctx=None):
    """encode.go:144-162"""
    return bytes(dst) + Encode(src, level, ctx)


def TryEncode(src, level=LevelFastest, ctx=None):
    """encode.go:168-206: None when incompressible."""
    a = _np(src)
    if MaxEncodedLen(a.size) < 0 or a.size < 16 or level not in (LevelSuperFast, LevelFastest, LevelBalanced):
        return None
    e = Encode(a, level, ctx)
    if len(e) >= 2 and e[0] == 0 and e[1] == 0:
        return None
    return e if len(e) < a.size else None


def Decode(src, ctx=None, guard=0):
    """minlz.Decode(nil, src), decode.go:50-78."""
    ctx = ctx or default_context()
    a = _np(src)
    n = DecodedLen(a)
    out = np.full(n + guard, 0xA5, dtype=np.uint8)
    if a.size and a[0] != 0 and not (a.size == 1):
        raise ErrUnsupported("Snappy/S2 fallback block")
    r = _lib.lib().mlz_decode(ctx.handle, _ptr(a), a.size, out.ctypes.data, n)
    if guard and not (out[n:] == 0xA5).all():
        raise AssertionError("decoder wrote past dst")
    if r < 0:
        _raise(r, ctx)
    return out[:r].tobytes()


def AppendDecoded(dst, src, ctx=None):
    """decode.go:85-103"""
    return bytes(dst) + Decode(src, ctx)


def crc(b, ctx=None):
    """crc(b) of minlz.go:133-140 (masked CRC32C), computed on the device."""
    ctx = ctx or default_context()
    a = _np(b)
    r = _lib.lib().mlz_crc(ctx.handle, _ptr(a), a.size)
    if r < 0:
        _raise(r, ctx)
    return int(r)


def encode_block(src, level=LevelFastest, ctx=None):
    """encodeBlock(dst, src) / WriterCustomEncoder contract (writer.go:1293-1304): tokens only; b'' = incompressible."""
    ctx = ctx or default_context()
    a = _np(src)
    out = np.empty(a.size + 16, dtype=np.uint8)
    r = _lib.lib().mlz_encode_block(ctx.handle, level, _ptr(a), a.size, out.ctypes.data, out.size)
    if r < 0:
        _raise(r, ctx)
    return out[:r].tobytes()


def decode_block(body, dlen, ctx=None):
    """minLZDecode(dst[:dlen], body) (decode.go:178): returns (code, bytes)."""
    ctx = ctx or default_context()
    a = _np(body)
    out = np.zeros(max(dlen, 1), dtype=np.uint8)
    r = _lib.lib().mlz_decode_block(ctx.handle, _ptr(a), a.size, out.ctypes.data, dlen)
    if r < 0:
        _raise(r, ctx)
    return r, out[:dlen].tobytes()


def encode_batch(blocks, level=LevelFastest, ctx=None):
    """mlz_encode_batch over a list of byte blocks -> list of encoded blocks."""
    ctx = ctx or default_context()
    arrs = [_np(b) for b in blocks]
    n = len(arrs)
    outs = [np.empty(max(MaxEncodedLen(a.size), 1), dtype=np.uint8) for a in arrs]
    vp, sz = C.c_void_p, C.c_size_t
    srcp = (vp * n)(*[_ptr(a) for a in arrs]); srcl = (sz * n)(*[a.size for a in arrs])
    dstp = (vp * n)(*[o.ctypes.data for o in outs]); dstc = (sz * n)(*[o.size for o in outs])
    ol = (C.c_int64 * n)()
    r = _lib.lib().mlz_encode_batch(ctx.handle, level, n, srcp, srcl, dstp, dstc, ol)
    if r:
        _raise(r, ctx)
    res = []
    for i in range(n):
        if ol[i] < 0:
            _raise(ol[i], ctx)
        res.append(outs[i][:ol[i]].tobytes())
    return res


def decode_batch(blocks, ctx=None):
    ctx = ctx or default_context()
    arrs = [_np(b) for b in blocks]
    n = len(arrs)
    lens = [DecodedLen(a) for a in arrs]
    outs = [np.empty(max(l, 1), dtype=np.uint8) for l in lens]
    vp, sz = C.c_void_p, C.c_size_t
    srcp = (vp * n)(*[_ptr(a) for a in arrs]); srcl = (sz * n)(*[a.size for a in arrs])
    dstp = (vp * n)(*[o.ctypes.data for o in outs]); dstc = (sz * n)(*lens)
    ol = (C.c_int64 * n)()
    r = _lib.lib().mlz_decode_batch(ctx.handle, n, srcp, srcl, dstp, dstc, ol)
    if r:
        _raise(r, ctx)
    res = []
    for i in range(n):
        if ol[i] < 0:
            _raise(ol[i], ctx)
        res.append(outs[i][:ol[i]].tobytes())
    return res


STREAM_ADD_INDEX, STREAM_IGNORE_CRC = 1, 2


def stream_encode(src, level=LevelFastest, block_size=2 << 20, add_index=False, ctx=None):
    """mlz_stream_encode: NewWriter(...).EncodeBuffer(src) + Close() in one call -> the .mz stream bytes."""
    ctx = ctx or default_context()
    a = _np(src)
    flags = STREAM_ADD_INDEX if add_index else 0
    cap = _lib.lib().mlz_stream_bound(a.size, block_size, flags)
    if cap < 0:
        _raise(cap, ctx)
    out = np.empty(cap, dtype=np.uint8)
    r = _lib.lib().mlz_stream_encode(ctx.handle, level, block_size, flags, _ptr(a), a.size, out.ctypes.data, out.size)
    if r < 0:
        _raise(r, ctx)
    return out[:r].tobytes()


def stream_decode(src, ignore_crc=False, ctx=None):
    """mlz_stream_decode: NewReader(src) read to EOF -> decoded bytes."""
    ctx = ctx or default_context()
    a = _np(src)
    n = _lib.lib().mlz_stream_decoded_len(_ptr(a), a.size)
    if n < 0:
        _raise(n, ctx)
    out = np.empty(max(n, 1), dtype=np.uint8)
    r = _lib.lib().mlz_stream_decode(ctx.handle, STREAM_IGNORE_CRC if ignore_crc else 0, _ptr(a), a.size, out.ctypes.data, n)
    if r < 0:
        _raise(r, ctx)
    return out[:r].tobytes()
