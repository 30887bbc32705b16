"""Randomised differential tests: hand-assembled token streams (every op kind, short repeats,
overlapping copies, long tokens straddling tiles, copy3 offsets, deliberately invalid ops) decoded
by the HIP decoder and by the oracle must agree on verdict and bytes; random inputs must survive
GPU-encode -> oracle-decode and GPU-encode -> GPU-decode."""
import numpy as np
import pytest

import minlz_amd as mz
import oracle as O
from minlz_amd import synth

pytestmark = pytest.mark.gpu


def random_stream(rng, target, invalid=False):
    """Builds (tokens, expected_output) with the oracle emitters; when `invalid`, one op is made illegal."""
    out = bytearray()
    tok = bytearray()
    last_off = 0
    bad_at = int(rng.integers(1, 30)) if invalid else -1
    nops = 0
    while len(out) < target:
        nops += 1
        kind = int(rng.integers(0, 10))
        make_bad = nops == bad_at
        if kind <= 2 or len(out) < 4:
            n = int(rng.choice([1, 2, 3, 5, 17, 29, 30, 31, 70, 300, 5000, 40000, 70000], p=[.2, .15, .1, .15, .1, .05, .05, .05, .05, .04, .03, .02, .01]))
            lit = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
            tok += O.emit_literal(lit)
            out += lit
            continue
        d = len(out)
        ln = int(rng.choice([1, 2, 3, 4, 5, 8, 11, 12, 18, 19, 33, 64, 65, 300, 3000, 40000, 100000], p=[.04, .04, .04, .12, .1, .12, .08, .06, .06, .05, .08, .05, .04, .05, .03, .02, .02]))
        if kind == 3 and last_off:  # repeat (length may be 1..3)
            off = last_off
            if make_bad:
                ln = 10 ** 7
            tok += O.emit_repeat(ln)
        else:
            ln = max(ln, 4)
            cls = int(rng.integers(0, 4))
            hi = [min(d, 1024), min(d, 65599), min(d, 2162687), min(d, 64)][cls]
            off = int(rng.integers(1, hi + 1))
            if make_bad:
                off = d + int(rng.integers(1, 100))  # reaches before the start of the block
                if off > 2162687:
                    off = 2162687
            nl = int(rng.integers(0, 5))
            if nl and 64 <= off <= 65599 and rng.random() < 0.5:
                lits = rng.integers(0, 256, size=nl, dtype=np.uint8).tobytes()
                tok += O.emit_copy_lits2(lits, off, ln)
                out += lits
            elif nl and nl <= 3 and off > 65599 and rng.random() < 0.5:
                lits = rng.integers(0, 256, size=nl, dtype=np.uint8).tobytes()
                tok += O.emit_copy_lits3(lits, off, ln)
                out += lits
            else:
                tok += O.emit_copy(off, ln)
            last_off = off
        if off > len(out) or make_bad:
            # invalid from here on: the expected output is irrelevant
            for _ in range(min(ln, 16)):
                out.append(0)
            continue
        for _ in range(ln):
            out.append(out[len(out) - off])
    return bytes(tok), bytes(out)


@pytest.mark.parametrize("seed", range(6))
def test_random_token_streams(ctx, seed):
    rng = np.random.default_rng(1000 + seed)
    for trial in range(25):
        target = int(rng.choice([50, 3000, 40000, 70000, 300000]))
        invalid = trial % 5 == 4
        tok, want = random_stream(rng, target, invalid)
        ocode, oout = O.decode_body(tok, len(want))
        gcode, gout = mz.decode_block(tok, len(want), ctx)
        assert gcode == ocode, (seed, trial, invalid)
        if ocode == 0:
            assert oout == want
            assert gout == want, (seed, trial)
        # same tokens as a full block (header + tokens), with guard bytes
        if ocode == 0 and len(want) >= 1 and len(tok) <= len(want):
            blk = b"\x00" + bytes(_uvarint(len(want))) + tok
            assert mz.Decode(blk, ctx, guard=64) == want


def _uvarint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return out


@pytest.mark.parametrize("seed", range(4))
def test_random_inputs_roundtrip(ctx, seed):
    rng = np.random.default_rng(77 + seed)
    blocks = []
    for _ in range(12):
        n = int(rng.choice([0, 1, 15, 16, 17, 100, 5000, 32767, 32768, 32769, 65535, 65537, 100000, 1 << 20]))
        kind = int(rng.integers(0, 5))
        if kind == 0:
            d = rng.integers(0, 256, size=n, dtype=np.uint8)
        elif kind == 1:
            d = rng.integers(0, 4, size=n, dtype=np.uint8)              # tiny alphabet: long matches, overlaps
        elif kind == 2:
            period = int(rng.integers(1, 70))
            d = np.tile(rng.integers(0, 256, size=period, dtype=np.uint8), n // period + 1)[:n]
        elif kind == 3:
            d = synth.text_like(max(n, 16), int(rng.integers(1, 1000)))[:n]
        else:
            base = rng.integers(0, 256, size=max(n // 7, 1), dtype=np.uint8)   # far repeats of a random chunk
            d = np.tile(base, 8)[:n].copy()
            if n > 64:
                d[rng.integers(0, n, size=n // 64)] ^= 1
        blocks.append(np.ascontiguousarray(d).tobytes())
    encs = mz.encode_batch(blocks, 1, ctx)
    for b, e in zip(blocks, encs):
        assert len(e) <= mz.MaxEncodedLen(len(b))
        assert O.decode(e, guard=32) == b
    assert mz.decode_batch(encs, ctx) == blocks


def test_many_blocks_more_tiles_than_resident_waves(ctx):
    # 96 blocks x 1 MiB = 3072 tiles in one launch (more than can be resident with 32 KiB of LDS each),
    # mixing level-conformant (GPU-made) and general (oracle-made) blocks: exercises the ticket
    # schedule and the dependency waits.
    d = synth.text_like(24 << 20, 99)
    blocks = [d[i << 20:(i + 1) << 20].tobytes() for i in range(24)]
    gpu = mz.encode_batch(blocks, 1, ctx)
    ora = [O.encode(b, 1) for b in blocks[:8]]
    mix = []
    want = []
    for i in range(24):
        mix.append(gpu[i]); want.append(blocks[i])
        if i < 8:
            mix.append(ora[i]); want.append(blocks[i])
    assert mz.decode_batch(mix * 3, ctx) == want * 3
