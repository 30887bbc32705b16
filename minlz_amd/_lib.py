"""ctypes binding of the C ABI declared in include/minlz_hip.h.

There is no CPU fallback here: if libminlz_hip.so is missing or HIP is unavailable, calls raise.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# MINLZ_HIP_LIB: development override used by tools/ to time experimental builds of the same library
SO = os.environ.get("MINLZ_HIP_LIB") or os.path.join(HERE, "libminlz_hip.so")

# every symbol include/minlz_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "mlz_init", "mlz_destroy", "mlz_last_error", "mlz_version", "mlz_device_name", "mlz_max_encoded_len",
    "mlz_decoded_len", "mlz_encode", "mlz_decode", "mlz_encode_block", "mlz_decode_block", "mlz_encode_batch",
    "mlz_decode_batch", "mlz_encode_batch_device", "mlz_decode_batch_device", "mlz_set_option", "mlz_get_timers",
    "mlz_timer_name", "mlz_crc", "mlz_crc_batch_device", "mlz_stream_bound", "mlz_stream_encode", "mlz_stream_decoded_len",
    "mlz_stream_decode", "mlz_get_counter", "mlz_init_devices", "mlz_device_count", "mlz_device_ctx",
    "mlz_stream_encode_gather_device", "mlz_release_stream",
]


class BlockDesc(C.Structure):
    _fields_ = [("src_off", C.c_uint64), ("src_len", C.c_uint64), ("dst_off", C.c_uint64), ("dst_cap", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO):
        raise RuntimeError("minlz_amd: %s not built (run python -c 'import __graft_entry__ as g; g.build()')" % SO)
    L = C.CDLL(SO)
    vp, sz, i64, i32 = C.c_void_p, C.c_size_t, C.c_int64, C.c_int
    L.mlz_init.argtypes = [i32, C.POINTER(vp)]; L.mlz_init.restype = i32
    L.mlz_destroy.argtypes = [vp]; L.mlz_destroy.restype = None
    L.mlz_init_devices.argtypes = [C.POINTER(i32), i32, C.POINTER(vp)]; L.mlz_init_devices.restype = i32
    L.mlz_device_count.argtypes = [vp]; L.mlz_device_count.restype = i32
    L.mlz_device_ctx.argtypes = [vp, i32]; L.mlz_device_ctx.restype = vp
    L.mlz_last_error.argtypes = [vp]; L.mlz_last_error.restype = C.c_char_p
    L.mlz_version.argtypes = []; L.mlz_version.restype = i32
    L.mlz_device_name.argtypes = [vp, C.c_char_p, sz]; L.mlz_device_name.restype = i32
    L.mlz_max_encoded_len.argtypes = [C.c_uint64]; L.mlz_max_encoded_len.restype = i64
    L.mlz_decoded_len.argtypes = [vp, sz]; L.mlz_decoded_len.restype = i64
    L.mlz_encode.argtypes = [vp, i32, vp, sz, vp, sz]; L.mlz_encode.restype = i64
    L.mlz_decode.argtypes = [vp, vp, sz, vp, sz]; L.mlz_decode.restype = i64
    L.mlz_encode_block.argtypes = [vp, i32, vp, sz, vp, sz]; L.mlz_encode_block.restype = i64
    L.mlz_decode_block.argtypes = [vp, vp, sz, vp, sz]; L.mlz_decode_block.restype = i32
    L.mlz_encode_batch.argtypes = [vp, i32, i32, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz), C.POINTER(i64)]
    L.mlz_encode_batch.restype = i32
    L.mlz_decode_batch.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz), C.POINTER(i64)]
    L.mlz_decode_batch.restype = i32
    L.mlz_encode_batch_device.argtypes = [vp, vp, i32, vp, vp, C.POINTER(BlockDesc), i32, vp]; L.mlz_encode_batch_device.restype = i32
    L.mlz_decode_batch_device.argtypes = [vp, vp, vp, vp, C.POINTER(BlockDesc), i32, vp]; L.mlz_decode_batch_device.restype = i32
    L.mlz_set_option.argtypes = [vp, i32, i64]; L.mlz_set_option.restype = i32
    L.mlz_release_stream.argtypes = [vp, vp]; L.mlz_release_stream.restype = i32
    L.mlz_get_timers.argtypes = [vp, C.POINTER(C.c_float), i32]; L.mlz_get_timers.restype = i32
    L.mlz_timer_name.argtypes = [i32]; L.mlz_timer_name.restype = C.c_char_p
    L.mlz_get_counter.argtypes = [vp, i32]; L.mlz_get_counter.restype = i64
    L.mlz_crc.argtypes = [vp, vp, sz]; L.mlz_crc.restype = i64
    L.mlz_crc_batch_device.argtypes = [vp, vp, vp, C.POINTER(BlockDesc), i32, vp]; L.mlz_crc_batch_device.restype = i32
    u32, u64 = C.c_uint32, C.c_uint64
    L.mlz_stream_bound.argtypes = [u64, u32, u32]; L.mlz_stream_bound.restype = i64
    L.mlz_stream_encode.argtypes = [vp, i32, u32, u32, vp, sz, vp, sz]; L.mlz_stream_encode.restype = i64
    L.mlz_stream_decoded_len.argtypes = [vp, sz]; L.mlz_stream_decoded_len.restype = i64
    L.mlz_stream_encode_gather_device.argtypes = [vp, i32, u32, u32, C.POINTER(vp), C.POINTER(sz), i32, vp, sz]
    L.mlz_stream_encode_gather_device.restype = i64
    L.mlz_stream_decode.argtypes = [vp, u32, vp, sz, vp, sz]; L.mlz_stream_decode.restype = i64
    _lib = L
    return L
