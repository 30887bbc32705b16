"""Host-path rates through one context over several per-device contexts (mlz_init_devices), by call and by variant.
usage: python tools/multi_stream_time.py [devices, e.g. 0,0] [MB per device]"""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import minlz_amd as mz
from minlz_amd import _lib, synth

devs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,0").split(",")]
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 100
L = _lib.lib()
S = mb * 1_000_000 * len(devs)
host = synth.enwik_like(S, seed=1)
BLOCK = 8 << 20


def rate(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return S / 1e6 / ((time.perf_counter() - t0) / reps)


variants = (("one", mz.Context(devs[0])), ("many", mz.Context(devices=devs)))
# (every variant twice, interleaved: the first thing measured on a fresh box runs on cold clocks and cold page-locked buffers)
for label, ctx in variants + variants:
    for pinned in (True,):
        mk = (lambda n: torch.empty(n, dtype=torch.uint8, pin_memory=True)) if pinned else (lambda n: torch.empty(n, dtype=torch.uint8))
        psrc = mk(S); psrc.numpy()[:] = host
        cap = L.mlz_stream_bound(S, BLOCK, 0)
        pst = mk(cap); pst.zero_()
        pdec = mk(S); pdec.zero_()
        n = L.mlz_stream_encode(ctx.handle, 1, BLOCK, 0, psrc.data_ptr(), S, pst.data_ptr(), cap)
        assert n > 0
        res = {"enc": rate(lambda: L.mlz_stream_encode(ctx.handle, 1, BLOCK, 0, psrc.data_ptr(), S, pst.data_ptr(), cap))}
        for flags, nm in ((0, "dec"), (2, "dec_nocrc")):
            assert L.mlz_stream_decode(ctx.handle, flags, pst.data_ptr(), n, pdec.data_ptr(), S) == S
            res[nm] = rate(lambda: L.mlz_stream_decode(ctx.handle, flags, pst.data_ptr(), n, pdec.data_ptr(), S))
        assert bytes(pdec.numpy()) == host.tobytes()
        print(label, "pinned" if pinned else "pageable", {k: round(v) for k, v in res.items()}, flush=True)
