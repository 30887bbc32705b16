"""The device encoder against its CPU model (tools/encmodel2: a sequential statement of exactly the match + serialize
kernels' algorithm, test infrastructure like the oracle): block bodies must be byte-identical at LevelFastest and
LevelBalanced, for both block classes (>= 1 MiB: 12-bit near tables; smaller: 13-bit, far tables sized by the block)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import minlz_amd as mz
from minlz_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model():
    sys.path.insert(0, os.path.join(ROOT, "tools", "encmodel2"))
    import run2  # compiles the model with g++ on import
    return run2


def _body(enc):
    assert enc[0] == 0
    h = 1
    while enc[h] & 0x80:
        h += 1
    return bytes(enc[h + 1:])


def _model_body(run2, a, level, l2_free=True):
    p = run2.P(**run2.def_for(a.size, level, l2_free))
    out = np.zeros(a.size + a.size // 8 + 64, dtype=np.uint8)
    n = run2.L.model2_block(a.ctypes.data, a.size, C.byref(p), out.ctypes.data, None)
    return out[:n].tobytes()


@pytest.mark.parametrize("level", [1, 2, 20])
def test_device_output_equals_the_model(ctx, model, level):
    # (level 20 = LevelBalanced WITH the tile levels of rounds 1-3: option 14 = 0)
    l2_free = level != 20
    if level == 20:
        level = 2
        ctx.set_option(mz.OPT_L2_FREE, 0)
    try:
        _check_model(ctx, model, level, l2_free)
    finally:
        ctx.set_option(mz.OPT_L2_FREE, 1)


def _check_model(ctx, model, level, l2_free):
    rng = np.random.default_rng(9)
    mix = np.concatenate([synth.text_like(200000, 4), rng.integers(0, 256, 100000, dtype=np.uint8), synth.json_like(150000, 5)])
    cases = [synth.text_like(100000, 7), synth.text_like((1 << 20) + 77, 8), mix, synth.json_like(3 << 20, 2),
             synth.text_like(64 << 10, 3), synth.text_like((128 << 10) + 5, 6), synth.json_like(300000, 9), synth.text_like(700000, 10)]
    lv = mz.LevelFastest if level == 1 else mz.LevelBalanced
    for a in cases:
        a = np.ascontiguousarray(a)
        enc = mz.Encode(a, lv, ctx)
        assert _body(enc) == _model_body(model, a, level, l2_free), (level, a.size)
    # one batch with blocks of every size class (far tables of different sizes side by side)
    encs = mz.encode_batch([np.ascontiguousarray(a) for a in cases], lv, ctx)
    for a, enc in zip(cases, encs):
        assert _body(enc) == _model_body(model, np.ascontiguousarray(a), level, l2_free), (level, a.size, "batch")
