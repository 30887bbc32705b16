"""Throughput of concurrent single-block host calls (the reference Writer's pattern: one goroutine per block):
T threads each encode (then decode) 8 MiB blocks through mlz_encode / mlz_decode."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch  # noqa: F401
import minlz_amd as mz
from minlz_amd import synth, _lib
ctx = mz.Context(0)
N = 8 << 20
data = synth.text_like(16 * N, 1)
blocks = [data[i * N:(i + 1) * N] for i in range(16)]
encs = [mz.Encode(b, 1, ctx) for b in blocks]
for T in (1, 2, 4, 8, 16):
    for what in ("encode", "decode"):
        reps = 8
        L = _lib.lib()
        eb = [np.frombuffer(e, dtype=np.uint8) for e in encs]
        outs = [np.empty(N + 64, dtype=np.uint8) for _ in range(T)]
        def work(i):  # raw ABI calls into preallocated buffers (ctypes drops the GIL for the duration of the call)
            o = outs[i]
            for _ in range(reps):
                if what == "encode": r = L.mlz_encode(ctx.handle, 1, blocks[i].ctypes.data, N, o.ctypes.data, o.size)
                else: r = L.mlz_decode(ctx.handle, eb[i].ctypes.data, eb[i].size, o.ctypes.data, N)
                assert r > 0
        # one untimed call per buffer first (workspace growth, first touch of the output pages)
        for i in range(T):
            if what == "encode": L.mlz_encode(ctx.handle, 1, blocks[i].ctypes.data, N, outs[i].ctypes.data, outs[i].size)
            else: L.mlz_decode(ctx.handle, eb[i].ctypes.data, eb[i].size, outs[i].ctypes.data, N)
        b0, r0 = ctx.combine_stats()
        ths = [threading.Thread(target=work, args=(i,)) for i in range(T)]
        t0 = time.time()
        for t in ths: t.start()
        for t in ths: t.join()
        dt = time.time() - t0
        b1, r1 = ctx.combine_stats()
        print("%s threads=%2d: %.2f GB/s (%d calls in %d launches)" % (what, T, T * reps * N / dt / 1e9, r1 - r0, b1 - b0))
