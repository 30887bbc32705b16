"""Host-pointer (PCIe-inclusive) throughput of mlz_encode_batch / mlz_decode_batch on the bench workload,
timed at the C ABI (buffers preallocated and touched; pageable and pinned host memory)."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import minlz_amd as mz
from minlz_amd import synth, _lib
S = 100_000_000; BLOCK = 8 << 20
ctx = mz.Context(0)
host = synth.text_like(S, 1)
nb = (S + BLOCK - 1) // BLOCK
lens = [min(BLOCK, S - i * BLOCK) for i in range(nb)]
L = _lib.lib(); vp, sz = C.c_void_p, C.c_size_t


def run(pinned):
    def buf(n):
        t = torch.empty(n, dtype=torch.uint8, pin_memory=pinned)
        t.zero_()
        return t
    src = buf(S); src.numpy()[:] = host
    enc = buf(nb * (BLOCK + 64)); dec = buf(S)
    sp = (vp * nb)(*[src.data_ptr() + i * BLOCK for i in range(nb)]); sl = (sz * nb)(*lens)
    ep = (vp * nb)(*[enc.data_ptr() + i * (BLOCK + 64) for i in range(nb)]); ec = (sz * nb)(*[BLOCK + 64] * nb)
    ol = (C.c_int64 * nb)()
    assert L.mlz_encode_batch(ctx.handle, 1, nb, sp, sl, ep, ec, ol) == 0
    t0 = time.perf_counter()
    for _ in range(5): L.mlz_encode_batch(ctx.handle, 1, nb, sp, sl, ep, ec, ol)
    te = (time.perf_counter() - t0) / 5
    cl = (sz * nb)(*[ol[i] for i in range(nb)])
    dp = (vp * nb)(*[dec.data_ptr() + i * BLOCK for i in range(nb)]); dc = (sz * nb)(*lens)
    dl = (C.c_int64 * nb)()
    assert L.mlz_decode_batch(ctx.handle, nb, ep, cl, dp, dc, dl) == 0
    t0 = time.perf_counter()
    for _ in range(5): L.mlz_decode_batch(ctx.handle, nb, ep, cl, dp, dc, dl)
    td = (time.perf_counter() - t0) / 5
    assert bytes(dec.numpy()) == host.tobytes()
    # whole-stream calls: framing + CRC + copies overlapped with the kernels
    cap = L.mlz_stream_bound(S, BLOCK, 1)
    stbuf = buf(cap)
    r = L.mlz_stream_encode(ctx.handle, 1, BLOCK, 1, src.data_ptr(), S, stbuf.data_ptr(), cap)
    assert r > 0
    t0 = time.perf_counter()
    for _ in range(5): r = L.mlz_stream_encode(ctx.handle, 1, BLOCK, 1, src.data_ptr(), S, stbuf.data_ptr(), cap)
    tse = (time.perf_counter() - t0) / 5
    dec.zero_()
    assert L.mlz_stream_decode(ctx.handle, 0, stbuf.data_ptr(), r, dec.data_ptr(), S) == S
    t0 = time.perf_counter()
    for _ in range(5): L.mlz_stream_decode(ctx.handle, 0, stbuf.data_ptr(), r, dec.data_ptr(), S)
    tsd = (time.perf_counter() - t0) / 5
    assert bytes(dec.numpy()) == host.tobytes()
    print("%s host memory, mlz_stream_encode / mlz_stream_decode (CRC + framing + index, copies overlapped): %.1f ms = %.0f MB/s, %.1f ms = %.0f MB/s" % (
        "pinned" if pinned else "pageable", tse * 1e3, S / 1e6 / tse, tsd * 1e3, S / 1e6 / tsd))
    print("%s host memory: encode %.1f ms = %.0f MB/s, decode %.1f ms = %.0f MB/s, pair %.0f MB/s" % (
        "pinned" if pinned else "pageable", te * 1e3, S / 1e6 / te, td * 1e3, S / 1e6 / td, S / 1e6 / (te + td)))


run(False)
run(True)
