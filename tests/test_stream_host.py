"""Host-side stream logic (minlz_amd/stream.py) with an oracle-backed block backend: no GPU.
Checks byte-level agreement with the oracle's restatement of the reference Writer/Reader."""
import io

import pytest

import oracle as O
from minlz_amd import api, stream as S, synth


class OracleBackend:
    """Test-only stand-in for HipBackend (tests may use the oracle; the product never does)."""

    def encode_blocks(self, blocks, level):
        out = []
        for b in blocks:
            body = O.encode_block(b, level)
            out.append(S.put_uvarint(len(b)) + body if body else None)
        return out

    def decode_bodies(self, bodies):
        res = []
        for b in bodies:
            try:
                res.append(O.decode(b"\x00" + b))
            except O.OracleError:
                raise api.ErrCorrupt()
        return res

    def crcs(self, blocks):
        return [O.crc(b) for b in blocks]


def enc(data, level, bs, batch=3):
    w = io.BytesIO()
    wr = S.Writer(w, level=level, block_size=bs, concurrency=batch, backend=OracleBackend())
    wr.EncodeBuffer(bytes(data))
    wr.Close()
    assert wr.Written() == len(w.getvalue())
    return w.getvalue()


@pytest.mark.parametrize("level", [0, 1, 2])
@pytest.mark.parametrize("bs", [4096, 65536, 1 << 20])
def test_writer_matches_oracle_stream_bytes(level, bs):
    for d in (synth.text_like(300000, 2).tobytes(), synth.random_bytes(70000).tobytes(), b"abc"):
        s = enc(d, level, bs)
        assert s == O.stream_encode(d, level, bs)
        assert S.Reader(s, backend=OracleBackend()).ReadAll() == d
        assert O.stream_decode(s, len(d)) == d


def test_framing_kat_and_errors():
    hdr = bytes.fromhex("ff0600004d696e4c7a02")
    ok = hdr + b"\x01\x08\x00\x00" + b"\x68\x10\xe6\xb6" + b"abcd" + b"\x20\x00\x00\x00"   # minlz_test.go:1120-1134
    assert S.Reader(ok, backend=OracleBackend()).ReadAll() == b"abcd"
    with pytest.raises(api.ErrCorrupt):
        S.Reader(hdr + b"\x01\x04\x00\x00", backend=OracleBackend()).ReadAll()
    bad = bytearray(ok); bad[15] ^= 1
    with pytest.raises(api.ErrCRC):
        S.Reader(bytes(bad), backend=OracleBackend()).ReadAll()
    n = (1 << 20) + 1 + 4
    big = bytes.fromhex("ff0600004d696e4c7a0a") + bytes([1, n & 0xff, (n >> 8) & 0xff, (n >> 16) & 0xff]) + b"\x00" * n
    with pytest.raises(api.ErrTooLarge):
        S.Reader(big, backend=OracleBackend()).ReadAll()
    with pytest.raises(api.ErrCorrupt):  # EOF length mismatch
        S.Reader(hdr + b"\x01\x08\x00\x00\x68\x10\xe6\xb6abcd" + b"\x20\x01\x00\x00\x05", backend=OracleBackend()).ReadAll()
    with pytest.raises(api.ErrCorrupt):  # missing EOF chunk
        S.Reader(ok[:-4], backend=OracleBackend()).ReadAll()


def test_incremental_writes_and_concatenated_streams():
    d = synth.text_like(500000, 7).tobytes()
    w = io.BytesIO()
    wr = S.Writer(w, level=1, block_size=65536, concurrency=2, backend=OracleBackend())
    for i in range(0, len(d), 10007):
        wr.Write(d[i:i + 10007])
    wr.Close()
    assert S.Reader(w.getvalue(), backend=OracleBackend()).ReadAll() == d
    two = w.getvalue() + enc(b"tail", 1, 4096)
    assert S.Reader(two, backend=OracleBackend()).ReadAll() == d + b"tail"


def test_writer_option_errors():
    with pytest.raises(ValueError):
        S.Writer(io.BytesIO(), block_size=1024, backend=OracleBackend())
    with pytest.raises(api.ErrInvalidLevel):
        S.Writer(io.BytesIO(), level=9, backend=OracleBackend())
