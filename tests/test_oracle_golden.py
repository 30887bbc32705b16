"""Pins the CPU oracle against every golden vector / KAT the reference's tests hold for the
block path.  No GPU."""
import json
import os

import numpy as np
import pytest

import oracle as O
from tests.util import load_zip


def test_tom_sawyer_golden_decode(twain, twain_mzb):
    # minlz_test.go:626-660: the reference-produced .mzb must decode to the .txt, bit-exact
    assert O.decoded_len(twain_mzb) == len(twain) == 14168
    assert O.decode(twain_mzb, guard=64) == twain


def test_tom_sawyer_golden_encode_l3(twain, twain_mzb):
    # The reference's committed testdata/Mark.Twain-Tom.Sawyer.txt.mzb (minlz_test.go:626-660) is
    # byte-identical to Encode(nil, txt, LevelSmallest) as restated here: this pins the L3 encoder
    # restatement (encode_l3.go:38-625) — match search, scoring, tie-breaks and emitters — on a
    # reference-produced artefact.
    assert O.encode(twain, 3) == twain_mzb


def test_emit_literal_kat(golden):
    kat = json.load(open(os.path.join(golden, "emit_kat.json")))
    assert len(kat["emit_literal"]) == 18
    for c in kat["emit_literal"]:
        lit = b"\x99" * c["length"]
        got = O.emit_literal(lit)
        assert got.endswith(lit)
        assert got[:len(got) - c["length"]].hex() == c["want"], c


def test_emit_copy_kat(golden):
    kat = json.load(open(os.path.join(golden, "emit_kat.json")))
    assert len(kat["emit_copy"]) >= 60
    for c in kat["emit_copy"]:
        assert O.emit_copy(c["offset"], c["length"]).hex() == c["want"], c


def test_max_encoded_len_kat(golden):
    kat = json.load(open(os.path.join(golden, "emit_kat.json")))
    for c in kat["max_encoded_len"]:
        assert O.max_encoded_len(c["input"]) == c["want"]


def test_crc_kat(golden):
    kat = json.load(open(os.path.join(golden, "emit_kat.json")))
    assert O.crc(b"abcd").to_bytes(4, "little").hex() == kat["crc_abcd_le"]


def test_header_kats():
    # minlz_test.go:297-300: "\x00" -> empty
    assert O.decode(b"\x00") == b""
    # decode.go:121-125: empty input is corrupt
    with pytest.raises(O.OracleError) as e:
        O.decode(b"")
    assert e.value.code == O.ERR_CORRUPT
    # TestInvalidVarint (minlz_test.go:259-271) behind the MinLZ marker byte
    for bad in (b"\x00\xff", b"\x00\x80\x80\x80\x80\x80\x80\x80\x80\x80\x80\x80\x00", b"\x00\xff\xff\xff\xff\xff\xff\xff\xff\xff\x7f"):
        with pytest.raises(O.OracleError) as e:
            O.decode(bad)
        assert e.value.code == O.ERR_CORRUPT
    # > 8 MiB -> ErrTooLarge (decode.go:140-142)
    with pytest.raises(O.OracleError) as e:
        O.decode(b"\x00\x81\x80\x80\x04\x00")
    assert e.value.code == O.ERR_TOO_LARGE
    # header only -> corrupt; v == 0 -> rest are literals; v < len(body) -> corrupt
    with pytest.raises(O.OracleError):
        O.decode(b"\x00\x05")
    assert O.decode(b"\x00\x00abc") == b"abc"
    with pytest.raises(O.OracleError):
        O.decode(b"\x00\x02\x08abc")
    # Snappy/S2 blocks: out of scope -> unsupported
    with pytest.raises(O.OracleError) as e:
        O.decode(b"\x03\x08abc")
    assert e.value.code == O.ERR_UNSUPPORTED


def test_emitter_properties():
    # encode_test.go:102-534 in spirit: every emitted op decodes back to (offset, length, literals)
    base = bytes((i * 7 + i // 13) % 256 for i in range(70000))
    lens = sorted(set([4, 5, 11, 12, 18, 19, 63, 64, 65, 68, 273, 274, 300, 1000, 65535, 65536, 70000 - 1]))
    for off in (1, 2, 3, 63, 64, 65, 1024, 1025, 65535, 65536, 65599, 65600, 70000 - 1):
        for ln in lens:
            if off < 1:
                continue
            prefix = base[:max(off, 1)]
            tok = O.emit_literal(prefix) + O.emit_copy(off, ln)
            want = bytearray(prefix)
            for i in range(ln):
                want.append(want[len(want) - off])
            code, got = O.decode_body(tok, len(want))
            assert code == 0 and got == bytes(want), (off, ln)
    for nl in (1, 2, 3, 4):
        for off in (64, 100, 65599):
            for ln in (4, 11, 12, 100, 70000):
                prefix = base[:off]
                lits = bytes(range(nl))
                tok = O.emit_literal(prefix) + O.emit_copy_lits2(lits, off + nl if False else off, ln)
                want = bytearray(prefix) + lits
                for i in range(ln):
                    want.append(want[len(want) - off])
                code, got = O.decode_body(tok, len(want))
                assert code == 0 and got == bytes(want), (nl, off, ln)
    for nl in (1, 2, 3):
        for off in (65536, 65600, 69000):
            for ln in (4, 64, 65, 400, 70000):
                prefix = base[:off]
                lits = bytes(range(nl))
                tok = O.emit_literal(prefix) + O.emit_copy_lits3(lits, off, ln)
                want = bytearray(prefix) + lits
                for i in range(ln):
                    want.append(want[len(want) - off])
                code, got = O.decode_body(tok, len(want))
                assert code == 0 and got == bytes(want), (nl, off, ln)


def test_negative_corpus_never_overruns():
    # fuzz/block-corpus-dec.zip + dec-block-regressions.zip: mostly corrupt inputs; the decoder
    # must return an error code or a result, never write past dst (guard bytes, fuzz_test.go:165-182)
    n_ok = n_err = 0
    for name in ("block-corpus-dec.zip", "dec-block-regressions.zip"):
        for label, blob in load_zip(name):
            try:
                out = O.decode(blob, guard=64)
                n_ok += 1
                assert len(out) == O.decoded_len(blob)
            except O.OracleError as e:
                assert e.code in (O.ERR_CORRUPT, O.ERR_TOO_LARGE, O.ERR_UNSUPPORTED)
                n_err += 1
    assert n_err > 100 and n_ok >= 1
