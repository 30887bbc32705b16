"""Oracle encoders: round trips over the reference's regression inputs and pattern families,
plus the reference's own ratio assertion.  No GPU."""
import numpy as np
import pytest

import oracle as O
from minlz_amd import synth
from tests.util import load_zip


@pytest.mark.parametrize("level", [-1, 1, 2, 3])
def test_enc_regressions_roundtrip(level):
    # testdata/enc_regressions.zip (decode_asm_test.go:49-73, writer_test.go:31-72)
    items = load_zip("enc_regressions.zip")
    assert len(items) >= 51
    for label, blob in items:
        enc = O.encode(blob, level)
        assert len(enc) <= O.max_encoded_len(len(blob))
        assert O.decode(enc, guard=32) == blob, label


def test_block_corpus_enc_roundtrip():
    # fuzz/block-corpus-enc.zip, FuzzEncodingBlocks seed corpus (fuzz_test.go:31-118); a spread sample
    items = load_zip("block-corpus-enc.zip")
    assert len(items) >= 200
    for label, blob in items[::6]:
        for level in (1, 2):
            assert O.decode(O.encode(blob, level)) == blob, label


@pytest.mark.parametrize("name", synth.PATTERNS)
def test_patterns_roundtrip(name):
    for size in (17, 100, 4096, 65535, 65536, 65549, 70000, 300000):
        d = synth.pattern(name, size)
        for level in (-1, 0, 1, 2):
            assert O.decode(O.encode(d, level)) == d.tobytes()


def test_small_and_margin_sizes():
    # TestSrcMarginBoundary shapes (decode_asm_test.go:352-410) and tiny inputs
    for size in list(range(0, 40)) + [50, 60, 70, 80]:
        for pat in (b"a", b"ab", b"abcd"):
            d = (pat * (size // len(pat) + 1))[:size]
            for level in (-1, 1, 2):
                assert O.decode(O.encode(d, level)) == d


def test_large_offsets_and_short_repeats():
    for min_off in (65536, 65600, 200000, 1 << 20, (2 << 20) + 65535):
        d = synth.large_offset(min_off + 5000, min_off)
        for level in (1, 2):
            assert O.decode(O.encode(d, level)) == d.tobytes()
    for off, ln in ((1, 4), (2, 4), (2, 10), (3, 9), (4, 16)):
        d = synth.short_repeat(off, ln)
        assert O.decode(O.encode(d, 1)) == d.tobytes()


def test_encode_huge_zeros():
    # TestEncodeHuge (encode_test.go:26-50): 8 MiB of zeros
    d = np.zeros(8 << 20, dtype=np.uint8)
    e = O.encode(d, 1)
    assert len(e) < 64 and O.decode(e) == d.tobytes()


def test_ratio_half_noise():
    # TestEncodeNoiseThenRepeats, minlz_test.go:776-797: below 75 % at L1
    for n in (256 * 1024, 2048 * 1024):
        d = synth.pattern("half", n)
        e = O.encode(d, 1)
        assert len(e) < n * 3 // 4


def test_incompressible_is_stored():
    d = synth.random_bytes(1 << 20)
    e = O.encode(d, 1)
    assert e[:2] == b"\x00\x00" and len(e) == d.size + 2 and O.decode(e) == d.tobytes()


def test_too_large():
    with pytest.raises(O.OracleError):
        O.encode(np.zeros((8 << 20) + 1, dtype=np.uint8), 1)


def test_level_ordering_on_text():
    # BASELINE config 5: L3 <= L2 <= L1 (README.md:318-330 ratio table ordering)
    for d in (synth.text_like(1 << 20, 5), synth.json_like(1 << 20)):
        l1, l2, l3 = (len(O.encode(d, lv)) for lv in (1, 2, 3))
        assert l3 <= l2 <= l1 < d.size


def test_l3_roundtrip_shapes():
    for name in synth.PATTERNS:
        d = synth.pattern(name, 200000)
        assert O.decode(O.encode(d, 3)) == d.tobytes(), name
    for n in (0, 1, 15, 16, 17, 31, 64, 65, 1000, 65536, 65537):
        d = synth.text_like(1 << 17, 3)[:n]
        assert O.decode(O.encode(d, 3)) == d.tobytes(), n
    d = synth.random_bytes(100000)
    e = O.encode(d, 3)
    assert e[:2] == b"\x00\x00" and O.decode(e) == d.tobytes()
    d = synth.large_offset(3 << 20, 1 << 20)
    assert O.decode(O.encode(d, 3)) == d.tobytes()


def test_l3_stream_64k_blocks():
    # config 5 shape: LevelSmallest at 64 KiB block size through the framed stream
    d = synth.text_like(1 << 20, 9)
    st = O.stream_encode(d, 3, 64 << 10)
    assert O.stream_decode(st, d.size) == d.tobytes()
    assert len(st) <= len(O.stream_encode(d, 2, 64 << 10))


def test_level_superfast_restatement():
    # encodeBlockFast (encode_l0.go): 8-byte minimum matches -> larger than L1, smaller than the input on text;
    # both size classes (<= 64 KiB: encodeFastBlockGo64K, else encodeFastBlockGo), asm_none.go:33-42
    for n in (65536, 65537, 1 << 20):
        d = synth.text_like(n, 9)
        e0, e1 = O.encode(d, -1), O.encode(d, 1)
        assert O.decode(e0, guard=32) == d.tobytes()
        assert len(e1) < len(e0) < n
    d = synth.random_bytes(70000)
    assert O.encode(d, -1) == b"\x00\x00" + d.tobytes()
