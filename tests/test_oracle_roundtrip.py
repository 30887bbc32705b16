"""Oracle encoders: round trips over the reference's regression inputs and pattern families,
plus the reference's own ratio assertion.  No GPU."""
import numpy as np
import pytest

import oracle as O
from minlz_amd import synth
from tests.util import load_zip


@pytest.mark.parametrize("level", [1, 2])
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
        for level in (0, 1, 2):
            assert O.decode(O.encode(d, level)) == d.tobytes()


def test_small_and_margin_sizes():
    # TestSrcMarginBoundary shapes (decode_asm_test.go:352-410) and tiny inputs
    for size in list(range(0, 40)) + [50, 60, 70, 80]:
        for pat in (b"a", b"ab", b"abcd"):
            d = (pat * (size // len(pat) + 1))[:size]
            for level in (1, 2):
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
    d = synth.text_like(1 << 20, 5)
    l1, l2 = len(O.encode(d, 1)), len(O.encode(d, 2))
    assert l2 <= l1 < d.size
