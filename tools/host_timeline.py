"""One mlz_encode_batch + one mlz_decode_batch from pinned memory (after a warm-up) for a rocprofv3 kernel/copy timeline.
usage (GPU box): cd /tmp; rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d OUT -- python tools/host_timeline.py"""
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
def buf(n):
    t = torch.empty(n, dtype=torch.uint8, pin_memory=True); t.zero_(); return t
src = buf(S); src.numpy()[:] = host
enc = buf(nb * (BLOCK + 64)); dec = buf(S)
sp = (vp * nb)(*[src.data_ptr() + i * BLOCK for i in range(nb)]); sl = (sz * nb)(*lens)
ep = (vp * nb)(*[enc.data_ptr() + i * (BLOCK + 64) for i in range(nb)]); ec = (sz * nb)(*[BLOCK + 64] * nb)
ol = (C.c_int64 * nb)()
for _ in range(3):
    assert L.mlz_encode_batch(ctx.handle, 1, nb, sp, sl, ep, ec, ol) == 0
cl = (sz * nb)(*[ol[i] for i in range(nb)])
dp = (vp * nb)(*[dec.data_ptr() + i * BLOCK for i in range(nb)]); dc = (sz * nb)(*lens)
dl = (C.c_int64 * nb)()
for _ in range(3):
    assert L.mlz_decode_batch(ctx.handle, nb, ep, cl, dp, dc, dl) == 0
t0 = time.perf_counter(); L.mlz_encode_batch(ctx.handle, 1, nb, sp, sl, ep, ec, ol); t1 = time.perf_counter()
L.mlz_decode_batch(ctx.handle, nb, ep, cl, dp, dc, dl); t2 = time.perf_counter()
print("encode %.2f ms  decode %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
if len(sys.argv) > 1 and sys.argv[1] == "sweep":
    for ge, gd in ((24, 32), (32, 48), (40, 64), (48, 100), (64, 128), (100, 24)):
        ctx.set_option(10, ge); ctx.set_option(11, gd)
        for _ in range(2):
            L.mlz_encode_batch(ctx.handle, 1, nb, sp, sl, ep, ec, ol); L.mlz_decode_batch(ctx.handle, nb, ep, cl, dp, dc, dl)
        t0 = time.perf_counter()
        for _ in range(5): L.mlz_encode_batch(ctx.handle, 1, nb, sp, sl, ep, ec, ol)
        t1 = time.perf_counter()
        for _ in range(5): L.mlz_decode_batch(ctx.handle, nb, ep, cl, dp, dc, dl)
        t2 = time.perf_counter()
        print("groups enc %3d MiB dec %3d MiB: encode %.2f ms  decode %.2f ms" % (ge, gd, (t1 - t0) / 5 * 1e3, (t2 - t1) / 5 * 1e3))
