"""Extended randomised differential run (test infrastructure, uses the oracle): more seeds and sizes than
tests/test_gpu_fuzz.py, every device level, mixed batches.  usage: python tools/fuzz_long.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401  (load PyTorch's HIP runtime first)
import minlz_amd as mz
import oracle as O
from minlz_amd import synth
from tests.test_gpu_fuzz import random_stream, _uvarint
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
ctx = mz.Context(0)
t0 = time.time(); it = 0; nblk = 0; nstream = 0
while time.time() - t0 < budget:
    it += 1
    rng = np.random.default_rng(50_000 + it)
    # 1. random valid / invalid token streams
    for trial in range(6):
        target = int(rng.choice([50, 3000, 40000, 70000, 300000, 2_000_000]))
        tok, want = random_stream(rng, target, invalid=(trial == 5))
        ocode, oout = O.decode_body(tok, len(want))
        gcode, gout = mz.decode_block(tok, len(want), ctx)
        assert gcode == ocode, ("verdict", it, trial)
        if ocode == 0:
            assert gout == oout, ("bytes", it, trial)   # (for a deliberately broken stream `want` is not meaningful)
            if trial != 5:
                assert oout == want, ("generator", it, trial)
        nstream += 1
    # 2. random inputs through every level, mixed sizes in one batch
    blocks = []
    for _ in range(10):
        n = int(rng.choice([0, 1, 15, 16, 17, 100, 5000, 32767, 32768, 32769, 65537, 100000, 1 << 20, 3 << 20, 8 << 20],
                           p=[.05, .05, .05, .05, .05, .1, .1, .05, .05, .05, .1, .1, .1, .05, .05]))
        kind = int(rng.integers(0, 6))
        if kind == 0: d = rng.integers(0, 256, size=n, dtype=np.uint8)
        elif kind == 1: d = rng.integers(0, 4, size=n, dtype=np.uint8)
        elif kind == 2:
            period = int(rng.integers(1, 300)); d = np.tile(rng.integers(0, 256, size=period, dtype=np.uint8), n // period + 1)[:n]
        elif kind == 3: d = synth.text_like(max(n, 16), int(rng.integers(1, 1000)))[:n]
        elif kind == 4: d = synth.json_like(max(n, 64), seed=int(rng.integers(1, 1000)))[:n]
        else:
            base = rng.integers(0, 256, size=max(n // 7, 1), dtype=np.uint8); d = np.tile(base, 8)[:n].copy()
            if n > 64: d[rng.integers(0, n, size=n // 64)] ^= 1
        blocks.append(np.ascontiguousarray(d).tobytes())
    for level in (-1, 1, 2):
        encs = mz.encode_batch(blocks, level, ctx)
        for b, e in zip(blocks, encs):
            assert len(e) <= mz.MaxEncodedLen(len(b)), ("maxlen", it, level)
            assert O.decode(e, guard=32) == b, ("oracle decode", it, level, len(b))
        assert mz.decode_batch(encs, ctx) == blocks, ("gpu decode", it, level)
        nblk += len(blocks)
    # 3. reference-algorithm streams of the same inputs through the device decoder
    small = [b for b in blocks if len(b) <= (1 << 20)]
    for lv in (1, 2, 3):
        assert mz.decode_batch([O.encode(b, lv) for b in small], ctx) == small, ("foreign", it, lv)
    # 4. framed streams
    big = max(blocks, key=len)
    st = mz.stream_encode(big, 1, 1 << 16 if len(big) < (1 << 20) else 1 << 20, True, ctx)
    assert mz.stream_decode(st, ctx=ctx) == big and O.stream_decode(st, len(big)) == big, ("stream", it)
print("fuzz_long ok: %d iterations, %d token streams, %d encoded blocks x decode, %.0f s" % (it, nstream, nblk, time.time() - t0))
