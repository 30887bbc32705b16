"""Decode parity (bit-exact) of the HIP decoder against the oracle / golden fixtures, through the C ABI."""
import numpy as np
import pytest

import minlz_amd as mz
import oracle as O
from minlz_amd import synth
from tests.util import load_zip

pytestmark = pytest.mark.gpu


def gpu_decode_result(blob, ctx):
    try:
        return ("ok", mz.Decode(blob, ctx, guard=64))
    except mz.MinLZError as e:
        return ("err", e.code)


def oracle_decode_result(blob):
    try:
        return ("ok", O.decode(blob))
    except O.OracleError as e:
        return ("err", e.code)


def test_golden_mzb(ctx, twain, twain_mzb):
    # minlz_test.go:626-660
    assert mz.Decode(twain_mzb, ctx, guard=64) == twain
    ctx.set_option(mz.OPT_DECODE_ALGO, 1)
    try:
        assert mz.Decode(twain_mzb, ctx, guard=64) == twain
    finally:
        ctx.set_option(mz.OPT_DECODE_ALGO, 0)


def test_header_and_error_kats(ctx):
    assert mz.Decode(b"\x00", ctx) == b""
    assert mz.Decode(b"\x00\x00abc", ctx) == b"abc"
    for bad, exc in ((b"", mz.ErrCorrupt), (b"\x00\x05", mz.ErrCorrupt), (b"\x00\x02\x08abc", mz.ErrCorrupt),
                     (b"\x00\x81\x80\x80\x04\x00", mz.ErrTooLarge), (b"\x03\x08abc", mz.ErrUnsupported),
                     (b"\x00\xff\xff\xff\xff\xff\xff\xff\xff\xff\x7f", mz.ErrCorrupt)):
        with pytest.raises(exc):
            mz.Decode(bad, ctx)


def test_negative_corpus_same_verdict_as_oracle(ctx):
    # fuzz/block-corpus-dec.zip: corrupt MinLZ blocks, streams, Snappy blocks. Same verdict and
    # (when valid) same bytes as the oracle; never writes past dst (guard bytes).
    n = 0
    for name in ("block-corpus-dec.zip", "dec-block-regressions.zip"):
        for label, blob in load_zip(name):
            want = oracle_decode_result(blob)
            got = gpu_decode_result(blob, ctx)
            assert got == want, (label, got[0], want[0], got[1] if got[0] == "err" else None, want[1] if want[0] == "err" else None)
            n += 1
    assert n > 500


@pytest.mark.parametrize("level", [1, 2, 3])
def test_oracle_encoded_regressions(ctx, level):
    # reference-algorithm streams (arbitrary cross-tile references) must decode bit-exact
    for label, blob in load_zip("enc_regressions.zip"):
        enc = O.encode(blob, level)
        assert mz.Decode(enc, ctx, guard=64) == blob, label


def test_oracle_encoded_corpus_sample(ctx):
    items = load_zip("block-corpus-enc.zip")
    for label, blob in items[::12]:
        assert mz.Decode(O.encode(blob, 1), ctx) == blob, label
        assert mz.Decode(O.encode(blob, 2), ctx) == blob, label


@pytest.mark.parametrize("name", synth.PATTERNS)
def test_patterns(ctx, name):
    for size in (17, 100, 4096, 65535, 65536, 65549, 70000, 300000):
        d = synth.pattern(name, size)
        for level in (0, 1, 2):
            assert mz.Decode(O.encode(d, level), ctx, guard=64) == d.tobytes(), (name, size, level)


def test_margin_sizes(ctx):
    # TestSrcMarginBoundary shapes, decode_asm_test.go:352-410
    for size in list(range(16, 40)) + [50, 60, 70, 80]:
        for pat in (b"a", b"ab", b"abcd"):
            d = (pat * (size // len(pat) + 1))[:size]
            for level in (1, 2):
                assert mz.Decode(O.encode(d, level), ctx, guard=64) == d


def test_large_offsets_and_short_repeats(ctx):
    for min_off in (65536, 65600, 200000, 1 << 20, (2 << 20) + 65535):
        d = synth.large_offset(min_off + 5000, min_off)
        for level in (1, 2):
            assert mz.Decode(O.encode(d, level), ctx) == d.tobytes()
    for off, ln in ((1, 4), (2, 4), (2, 10), (3, 9), (4, 16)):
        d = synth.short_repeat(off, ln)
        assert mz.Decode(O.encode(d, 1), ctx) == d.tobytes()


def test_long_tokens_spanning_tiles(ctx):
    # one literal / one copy covering many 64 KiB tiles, and overlapping copies with tiny periods
    z = np.zeros(8 << 20, dtype=np.uint8)
    assert mz.Decode(O.encode(z, 1), ctx) == z.tobytes()
    for period in (1, 2, 3, 5, 63, 64, 65, 1000):
        d = np.tile(np.arange(period, dtype=np.uint8) + 7, (1 << 20) // period + 1)[:1 << 20]
        assert mz.Decode(O.encode(d, 1), ctx) == d.tobytes(), period
    # hand-built: literal of 200000 bytes, then copy3 far back, then repeat
    lit = synth.random_bytes(200000, 3).tobytes()
    tok = O.emit_literal(lit) + O.emit_copy(150000, 300000) + O.emit_repeat(70000)
    want = bytearray(lit)
    for _ in range(370000):
        want.append(want[len(want) - 150000])
    code, got = mz.decode_block(tok, len(want), ctx)
    assert code == 0 and got == bytes(want)
    assert O.decode_body(tok, len(want)) == (0, bytes(want))


def test_dense_token_streams(ctx):
    # Token streams with far more tokens per stream byte than a compressor writes (1-byte repeats, 2-byte literals, 2-byte
    # copy1s): the exec pass handles one token per lane in rounds of 64 entries of the token list.  With a token per stream
    # byte a segment has more tokens than the index pass stages in LDS (kTokStage: the rest is written directly) and a tile
    # has more than 3 x 64 rounds (the waves reload their round records).  Output against the oracle's decoder.
    rng = np.random.default_rng(17)
    for kind in ("rep1", "mixed", "lit1"):
        tok = bytearray(O.emit_literal(b"abcdefgh") + O.emit_copy(3, 5))
        n_out = 8 + 5
        while n_out < 300000:
            r = int(rng.integers(0, 3)) if kind == "mixed" else (0 if kind == "rep1" else 1)
            if r == 0:
                ln = int(rng.integers(1, 4))
                tok += O.emit_repeat(ln); n_out += ln
            elif r == 1:
                tok += O.emit_literal(bytes([int(rng.integers(0, 256))])); n_out += 1
            else:
                off = int(rng.integers(1, 9)); ln = int(rng.integers(4, 8))
                tok += O.emit_copy(off, ln); n_out += ln
        code, want = O.decode_body(bytes(tok), n_out)
        assert code == 0
        code, got = mz.decode_block(bytes(tok), n_out, ctx)
        assert code == 0 and got == want, kind


def test_literal_runs_longer_than_a_region_group(ctx):
    # The index pass chains a segment's 64-byte regions with 15 lanes that start at guessed positions and must meet each other inside a
    # group of 8 regions (512 bytes); literal runs of 600 .. 9000 bytes of token-like bytes keep them apart, so one thread chains the
    # segment instead, and the runs cross segment boundaries (entries deep inside a segment: the chain kernel's slow exits).  Literal
    # bytes are drawn from the values that long-literal and copy3 tags have, so the parse from a wrong start derails as far as it can.
    rng = np.random.default_rng(23)
    taggy = np.array([0xE8, 0xF0, 0xF8, 0xFF, 0x07, 0x0F, 0xFB, 0x02, 0x01, 0x00], dtype=np.uint8)
    for trial in range(3):
        tok = bytearray()
        out = bytearray()
        while len(out) < 1_500_000:
            n = int(rng.integers(600, 9000 if trial else 1400))
            lit = bytes(taggy[rng.integers(0, len(taggy), n)]) if trial != 2 else bytes(rng.integers(0, 256, n, dtype=np.uint8))
            tok += O.emit_literal(lit); out += lit
            for _ in range(int(rng.integers(1, 40))):
                off = int(rng.integers(1, min(len(out), 70000))); ln = int(rng.integers(4, 60))
                tok += O.emit_copy(off, ln)
                for _ in range(ln): out.append(out[len(out) - off])
        code, want = O.decode_body(bytes(tok), len(out))
        assert code == 0 and want == bytes(out)
        code, got = mz.decode_block(bytes(tok), len(out), ctx)
        assert code == 0 and got == want, trial


def test_decode_block_corrupt_verdicts(ctx):
    # minLZDecode contract (decode.go:178): 0 ok / 1 corrupt, same as the oracle, on mutated streams
    d = synth.text_like(200000, 9)
    body = O.encode_block(d, 1)
    rng = np.random.default_rng(5)
    assert mz.decode_block(body, d.size, ctx) == (0, d.tobytes())
    for trial in range(40):
        b = bytearray(body)
        kind = trial % 4
        if kind == 0:
            b = b[:int(rng.integers(1, len(b)))]             # truncated
        elif kind == 1:
            b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))  # bit flip
        elif kind == 2:
            b += bytes(rng.integers(0, 256, size=3, dtype=np.uint8))       # trailing garbage
        else:
            p = int(rng.integers(0, len(b) - 8)); b[p:p + 4] = b"\xff\xff\xff\xff"
        ocode, oout = O.decode_body(bytes(b), d.size)
        gcode, gout = mz.decode_block(bytes(b), d.size, ctx)
        assert gcode == ocode, trial
        if ocode == 0:
            assert gout == oout
    # wrong destination length
    assert mz.decode_block(body, d.size - 1, ctx)[0] == 1
    assert mz.decode_block(body, d.size + 1, ctx)[0] == 1


def test_full_size_blocks(ctx):
    for gen in (lambda: synth.text_like(8 << 20, 11), lambda: synth.json_like(8 << 20), lambda: synth.random_bytes(8 << 20)):
        d = gen()
        for level in (1, 2):
            assert mz.Decode(O.encode(d, level), ctx) == d.tobytes()


def test_config5_smallest_64k_blocks(ctx):
    # BASELINE.json configs[4]: LevelSmallest at 64 KiB blocks — CPU (oracle L3) encode, device
    # decode of the whole batch in one launch, and the ratio ordering L3 <= L2 <= L1.
    bs = 64 << 10
    d = synth.text_like(16 << 20, 21)
    blocks = [d[i:i + bs].tobytes() for i in range(0, d.size, bs)]
    enc = {lv: [O.encode(b, lv) for b in blocks] for lv in (1, 2, 3)}
    tot = {lv: sum(map(len, enc[lv])) for lv in enc}
    assert tot[3] <= tot[2] <= tot[1] < d.size
    for lv in (3, 2):
        assert mz.decode_batch(enc[lv], ctx) == blocks
    # the device encoder at this block size still beats "stored" and decodes with the oracle
    gpu = mz.encode_batch(blocks, 1, ctx)
    assert sum(map(len, gpu)) < 0.6 * d.size
    for g, b in zip(gpu[::17], blocks[::17]):
        assert O.decode(g) == b


def test_serial_and_parallel_agree(ctx):
    d = synth.text_like(1 << 20, 13)
    enc = O.encode(d, 2)
    a = mz.Decode(enc, ctx)
    ctx.set_option(mz.OPT_DECODE_ALGO, 1)
    try:
        b = mz.Decode(enc, ctx)
    finally:
        ctx.set_option(mz.OPT_DECODE_ALGO, 0)
    assert a == b == d.tobytes()


def test_decode_batch(ctx):
    blocks = [synth.text_like(300000, s).tobytes() for s in range(5)] + [b"", b"abc", synth.random_bytes(70000).tobytes()]
    encs = [O.encode(b, 1) for b in blocks]
    assert mz.decode_batch(encs, ctx) == blocks


def test_general_block_paths_agree(ctx):
    # Streams of other encoders ("general" blocks) have two decode paths: dec_general_kernel (default: explain + settle)
    # and the tile chain of the exec pass (option 8 = 1).  Both must give the oracle's bytes and the oracle's verdicts.
    import numpy as np
    items = [synth.text_like(3 << 20, 77), synth.json_like(1 << 20), synth.pattern("off2", 300000), synth.pattern("zeros", 200000),
             synth.large_offset(3 << 20, 1 << 20)]
    encs = [O.encode(d, lv) for d in items for lv in (1, 2, 3)]
    want = [d.tobytes() for d in items for _ in (1, 2, 3)]
    rng = np.random.default_rng(3)
    mutated = []
    for e in encs[:6]:
        for _ in range(6):
            m = bytearray(e)
            pos = int(rng.integers(4, len(m)))
            m[pos] ^= int(rng.integers(1, 256))
            mutated.append(bytes(m))
    results = {}
    for algo in (0, 1):
        ctx.set_option(8, algo)
        try:
            assert mz.decode_batch(encs, ctx) == want
            verdicts = []
            for m in mutated:
                try:
                    verdicts.append(("ok", mz.Decode(m, ctx)))
                except mz.MinLZError as ex:
                    verdicts.append(("err", type(ex).__name__))
            results[algo] = verdicts
        finally:
            ctx.set_option(8, 0)
    assert results[0] == results[1]
    for m, v in zip(mutated, results[0]):
        try:
            o = ("ok", O.decode(m))
        except O.OracleError:
            o = ("err", "ErrCorrupt")
        assert (v[0] == o[0]) and (v[0] == "err" or v[1] == o[1])


def test_index_pass_forms_agree(ctx):
    # The index pass (token list, tile starts, level conformance, block totals) exists in two forms: dec_index1 / dec_index2 (default)
    # and the three kernels of rounds 2-3 (MLZ_OPT_INDEX_PASSES = 1).  Same output and the same verdicts on own streams of every
    # level, on the oracle's streams (general blocks), on dense token streams (more than one stage of tokens per segment) and on
    # mutated streams.
    rng = np.random.default_rng(77)
    datas = [synth.text_like(3 << 20, 4), synth.enwik_like(2 << 20, 5), synth.json_like(1 << 20, 6), bytes(rng.integers(0, 4, 300_000, dtype=np.uint8)),
             b"ab" * 200_000 + bytes(rng.integers(0, 256, 70_000, dtype=np.uint8)) + b"xyz" * 100_000]
    encs, want = [], []
    for d in datas:
        d = bytes(d)
        for level in (-1, 1, 2):
            encs.append(mz.Encode(d, level, ctx)); want.append(d)
        encs.append(O.encode(d[:1 << 20], 1)); want.append(d[:1 << 20])
        encs.append(O.encode(d[:1 << 20], 3)); want.append(d[:1 << 20])
    mutated = []
    for e in encs[::3]:
        for _ in range(6):
            m = bytearray(e)
            pos = int(rng.integers(4, len(m)))
            m[pos] ^= int(rng.integers(1, 256))
            mutated.append(bytes(m))
    results = {}
    for passes in (0, 1):
        ctx.set_option(mz.OPT_INDEX_PASSES, passes)
        try:
            assert mz.decode_batch(encs, ctx) == want
            results[passes] = (ctx.general_blocks(), [gpu_decode_result(m, ctx) for m in mutated])
        finally:
            ctx.set_option(mz.OPT_INDEX_PASSES, 0)
    assert results[0] == results[1]
    for m, v in zip(mutated, results[0][1]):
        assert v == oracle_decode_result(m)


def test_own_streams_take_the_tile_path(ctx):
    # Streams of this library's encoder must be recognised as level-conformant at every level (fast pattern at level 1,
    # dense pattern at level 2); only blocks of other encoders with cross-tile copies go through the
    # pointer-jumping path.  Guards against a silent 3x decode slow-down if encoder and decoder ever disagree.
    d = synth.text_like(3 << 20, 9)
    ctx.set_option(mz.OPT_L2_FREE, 0)     # LevelBalanced with the tile levels of rounds 1-3
    try:
        for level in (-1, 1, 2):
            e = mz.Encode(d, level, ctx)
            assert mz.Decode(e, ctx) == d.tobytes()
            assert ctx.general_blocks() == 0, level
    finally:
        ctx.set_option(mz.OPT_L2_FREE, 1)
    # LevelBalanced's default since round 4 has no tile levels (the reference's ratio): its blocks of more than two tiles are general ones
    e = mz.Encode(d, 2, ctx)
    assert mz.Decode(e, ctx) == d.tobytes() and O.decode(e) == d.tobytes()
    assert ctx.general_blocks() == 1
    e = O.encode(d, 1)   # the reference's algorithm: copies from any earlier byte
    assert mz.Decode(e, ctx) == d.tobytes()
    assert ctx.general_blocks() == 1
    # config-5 shape: 64 KiB blocks of the reference's encoders are two tiles, the second (level 1) may read the first
    # (level 0) and a copy may straddle them (each part is checked for its own tile): always the tile path
    for lv in (1, 2, 3):
        small = [O.encode(d[i << 16:(i + 1) << 16], lv) for i in range(8)]
        assert mz.decode_batch(small, ctx) == [d[i << 16:(i + 1) << 16].tobytes() for i in range(8)]
        assert ctx.general_blocks() == 0, lv


def test_general_blocks_through_the_packed_pool(ctx):
    # The general-block path lays a tile's terminal bytes out in a 32 KiB pool: literals at the bottom, an 8-byte slot per external
    # entry at the top; a tile whose slots do not fit is redone with a byte-packed pool (exact stores in the settle pass).  Real
    # streams rarely need that, so the fallback is forced here (MLZ_OPT_GEN_PACKED) on the inputs of the path's other tests —
    # same bytes as the oracle, same verdicts — and then a stream that needs it for real: one-byte external runs (a repeat of
    # length 1 per output byte, from 32 KiB back) fill more slots than a pool has.
    items = [synth.text_like(3 << 20, 77), synth.json_like(1 << 20), synth.large_offset(3 << 20, 1 << 20), synth.enwik_like((8 << 20) - 13, 5)]
    encs = [O.encode(d, lv) for d in items for lv in (1, 2, 3)]
    want = [d.tobytes() for d in items for _ in (1, 2, 3)]
    ctx.set_option(mz.OPT_GEN_PACKED, 1)
    try:
        assert mz.decode_batch(encs, ctx) == want
        assert ctx.general_blocks() == len(encs)
    finally:
        ctx.set_option(mz.OPT_GEN_PACKED, 0)
    assert mz.decode_batch(encs, ctx) == want
    # hand-made: 128 KiB of literal-free output after a first tile of noise: copy2 tokens of 4 bytes from exactly 32768 back, i.e. every
    # token of tiles 1 .. 4 is an external run of 4 bytes: 8192 entries per tile (the slots hold 4096).  (Five tiles: tile 4 reads tile 3,
    # which no level pattern allows — a general block.)
    rng = np.random.default_rng(4)
    first = rng.integers(0, 256, 32768, dtype=np.uint8).tobytes()
    body = bytearray()
    pos = 0
    while pos < len(first):          # literal runs of 29 bytes (one-byte headers)
        n = min(29, len(first) - pos)
        body += bytes([(n - 1) << 3]) + first[pos:pos + n]
        pos += n
    ntok = 4 * 32768 // 4
    off = 32768 - 64                 # copy2: offset - 64 in two bytes
    body += bytes([(4 - 4) << 2 | 2, off & 0xff, off >> 8]) * ntok
    want2 = first * 5
    code, got = mz.decode_block(bytes(body), len(want2), ctx)
    assert (code, got) == (0, want2)
    assert O.decode_body(bytes(body), len(want2)) == (0, want2)
    assert ctx.general_blocks() == 1


def test_many_general_blocks_take_the_tile_ordered_phase(ctx):
    # Many general blocks in one batch (more than the settling workgroups the launch has: they take turns), blocks of ragged sizes,
    # one long block among short ones, one corrupt block among good ones: same bytes, same verdicts as the oracle.
    d = synth.text_like(24 * (256 << 10), 13)
    blocks = [d[i * (256 << 10):(i + 1) * (256 << 10)].tobytes() for i in range(24)]
    blocks[5] = synth.json_like(200_001).tobytes()          # ragged size, other statistics
    blocks[11] = synth.pattern("half", 256 << 10).tobytes()  # long literal run + long copies
    encs = [O.encode(b, 1 + (i % 3)) for i, b in enumerate(blocks)]
    assert mz.decode_batch(encs, ctx) == blocks
    assert ctx.general_blocks() >= 20
    # one long block among many short ones
    big = synth.text_like(3 << 20, 29).tobytes()
    mixed = [O.encode(big, 1)] + encs[:22]
    assert mz.decode_batch(mixed, ctx) == [big] + blocks[:22]
    assert ctx.general_blocks() >= 20
    # a stream that lies about a copy (offset beyond the start) in the middle of such a batch: that block fails, the others do not
    bad = bytearray(encs[7])
    body0 = 1 + 3  # header: 00 + uvarint(262144) is 3 bytes
    bad[body0:body0 + 3] = bytes([(4 - 4) << 2 | 2, 0xff, 0xff])   # copy2, offset 65535+64 at output position 0
    outs = [np.zeros(len(b), dtype=np.uint8) for b in blocks]
    import ctypes as C
    from minlz_amd import _lib
    n = len(blocks)
    arrs = [np.frombuffer(bytes(e), dtype=np.uint8) for e in encs]
    arrs[7] = np.frombuffer(bytes(bad), dtype=np.uint8)
    vp, sz = C.c_void_p, C.c_size_t
    srcp = (vp * n)(*[a.ctypes.data for a in arrs]); srcl = (sz * n)(*[a.size for a in arrs])
    dstp = (vp * n)(*[o.ctypes.data for o in outs]); dstc = (sz * n)(*[o.size for o in outs])
    ol = (C.c_int64 * n)()
    assert _lib.lib().mlz_decode_batch(ctx.handle, n, srcp, srcl, dstp, dstc, ol) == 0
    for i in range(n):
        if i == 7:
            assert ol[i] == -1 and O.decode_body(bytes(bad[body0:]), len(blocks[7]))[0] == 1   # ErrCorrupt, like the oracle
        else:
            assert ol[i] == len(blocks[i]) and outs[i].tobytes() == blocks[i]


def test_general_block_barrier_is_bounded(ctx):
    # The pass for general blocks (streams of the reference's own encoders) has settling workgroups that wait for the ready
    # flags of the explaining ones.  Nothing but speed depends on the two running side by side, but a wait in a kernel is still a
    # wait: it is bounded.  With the patience cut to a single poll the first flag cannot be up in time (a tile takes ~30 us to
    # explain): the call must come back with ErrHIP — the Go wrapper's cue to fall back to its CPU path, INTEGRATION.md — instead
    # of hanging, and with the default patience restored the same call must decode normally.
    d = synth.text_like(3 << 20, 17)
    ref = O.encode(d, 1)                      # reference-algorithm stream: a general block
    assert mz.Decode(ref, ctx) == d.tobytes()
    assert ctx.general_blocks() == 1          # it did take the general path
    ctx.set_option(9, 1)
    try:
        with pytest.raises(mz.ErrHIP):
            mz.Decode(ref, ctx)
    finally:
        ctx.set_option(9, 1 << 24)
    assert mz.Decode(ref, ctx) == d.tobytes()
    # a batch with one general and one conformant block: only the general one fails
    own = mz.Encode(d, 1, ctx)
    ctx.set_option(9, 1)
    try:
        with pytest.raises(mz.ErrHIP):
            mz.decode_batch([own, ref], ctx)
    finally:
        ctx.set_option(9, 1 << 24)
    assert mz.decode_batch([own, ref], ctx) == [d.tobytes(), d.tobytes()]
    # the Reader's entry point (minLZDecode contract: 0 ok, 1 corrupt, < 0 error): a device failure must come back as
    # < 0 — the shim's cue to run minLZDecodeGo — and never as 1, which the Reader would report as ErrCorrupt for a
    # valid stream; a really corrupt body still gives 1
    from minlz_amd.stream import uvarint
    _, hl = uvarint(ref, 1)
    body = ref[1 + hl:]
    assert mz.decode_block(body, d.size, ctx) == (0, d.tobytes())
    ctx.set_option(9, 1)
    try:
        with pytest.raises(mz.ErrHIP):
            mz.decode_block(body, d.size, ctx)
    finally:
        ctx.set_option(9, 1 << 24)
    assert mz.decode_block(body, d.size, ctx) == (0, d.tobytes())
    assert mz.decode_block(body[:len(body) // 2], d.size, ctx)[0] == 1


def test_token_stream_longer_than_its_output(ctx):
    # A stream of 1-byte literals is twice its output: 5 MiB of output from a 10 MiB body (more 8 KiB segments than an
    # encoder's output for a full block ever has).
    n = 5 << 20
    d = synth.text_like(n, 23)
    body = np.zeros(2 * n, dtype=np.uint8)
    body[1::2] = d                              # tag 0x00 = literal of length 1, then the byte
    hdr = bytearray([0])
    v = n
    while v >= 0x80:
        hdr.append((v & 0x7f) | 0x80); v >>= 7
    hdr.append(v)
    blk = bytes(hdr) + body.tobytes()
    # behind a block header such a body is refused outright (isMinLZ: decoded size < body size, decode.go:150-152) ...
    with pytest.raises(O.OracleError):
        O.decode(blk)
    with pytest.raises(mz.ErrCorrupt):
        mz.Decode(blk, ctx)
    # ... but minLZDecode itself (the WriterCustomEncoder-side entry, mlz_decode_block) takes any src length
    assert O.decode_body(body.tobytes(), n) == (0, d.tobytes())
    assert mz.decode_block(body.tobytes(), n, ctx) == (0, d.tobytes())
    # longer than 2 x dlen: no valid stream can be; same verdict as the reference's walk (decode.go:615)
    bad = body.tobytes() + bytes(64)
    assert O.decode_body(bad, n)[0] != 0
    assert mz.decode_block(bad, n, ctx)[0] == 1
    assert mz.decode_block(bytes(9 << 20), 100, ctx)[0] == 1


def test_level0_tiles_by_the_parallel_kernel_or_by_the_exec_pass(ctx):
    # Round 6: when a batch has no more level-0 tiles than the device has CUs, dec_level0_kernel decodes them (role E of the general pass: pointer jumping
    # in LDS, no ordered chain) before the exec pass; option 23 = 0 leaves them to the exec pass as in rounds 2-5.  Same bytes, same verdicts:
    # own streams of every level pattern and block class, damaged own streams (the damage lands in level-0 tiles as well), the reference's negative corpus.
    import zipfile, os
    other = mz.Context(0)
    other.set_option(23, 0)
    try:
        rng = np.random.default_rng(17)
        blocks, want = [], []
        for i, n in enumerate([8 << 20, (3 << 20) + 77, 700_001, 65_536 + 9, 32_768, 40_000, 33, 1 << 20]):
            d = (synth.enwik_like if i & 1 else synth.json_like)(n, seed=70 + i)
            for lv in (mz.LevelSuperFast, mz.LevelFastest):
                blocks.append(mz.Encode(d, lv, ctx)); want.append(d.tobytes())
        assert mz.decode_batch(blocks, ctx) == want and mz.decode_batch(blocks, other) == want
        # damage: a flipped byte every 32 KiB of stream or so, one variant per block
        for k, b in enumerate(blocks[:8]):
            bad = bytearray(b)
            for pos in range(7 + 131 * k, len(bad), max(1000, len(bad) // 9)):
                bad[pos] ^= int(rng.integers(1, 256))
            r = [gpu_decode_result(bytes(bad), c) for c in (ctx, other)]
            assert r[0] == r[1] == oracle_decode_result(bytes(bad)), k
        n = 0
        for name in ("block-corpus-dec.zip", "dec-block-regressions.zip"):
            for label, blob in load_zip(name):
                assert gpu_decode_result(blob, other) == gpu_decode_result(blob, ctx), label
                n += 1
        assert n > 500
    finally:
        other.close()
