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


def test_writer_padding():
    # WriterPadding (writer.go:1248-1268, :1094-1112): the closed stream is a multiple of n, the filler is a skippable 0xfe chunk
    # that no reader sees, and an appended index still ends the stream
    d = synth.text_like(200000, 3).tobytes()
    plain = enc(d, 1, 65536)
    for n in (2, 7, 512, 4096, 1 << 20):
        for add_index in (False, True):
            w = io.BytesIO()
            wr = S.Writer(w, level=1, block_size=65536, concurrency=3, backend=OracleBackend(), add_index=add_index, padding=n,
                          padding_src=lambda k: bytes([0xA5]) * k)
            wr.EncodeBuffer(d)
            wr.Close()
            s = w.getvalue()
            assert len(s) % n == 0 and wr.Written() == len(s), (n, add_index)
            assert s[:len(plain)] == plain                      # the stream itself is unchanged
            rest = s[len(plain):]
            if rest and rest[0] == 0xFE:
                f = rest[1] | rest[2] << 8 | rest[3] << 16
                assert rest[4:4 + f] == bytes([0xA5]) * f and f + 4 >= 4
                rest = rest[4 + f:]
            assert (len(rest) > 0) == add_index                 # what follows the padding is the index
            assert S.Reader(s, backend=OracleBackend()).ReadAll() == d
            assert O.stream_decode(s, len(d)) == d
            if add_index:
                rs = S.ReadSeeker(s, backend=OracleBackend())
                rs.Seek(150000)
                assert rs.Read(1000) == d[150000:151000]
    # calcSkippableFrame (writer.go:1135-1151): nothing to add on a multiple, never a frame shorter than its header
    assert S._calc_skippable_frame(4096, 4096) == 0 and S._calc_skippable_frame(4095, 4096) == 4097 and S._calc_skippable_frame(10, 16) == 6
    # padding 1 = off (writer.go:1259-1262); random filler by default
    w = io.BytesIO()
    wr = S.Writer(w, level=1, block_size=65536, backend=OracleBackend(), padding=1)
    wr.EncodeBuffer(d); wr.Close()
    assert w.getvalue() == plain
    w = io.BytesIO()
    wr = S.Writer(w, level=1, block_size=65536, backend=OracleBackend(), padding=1000)
    wr.EncodeBuffer(d); wr.Close()
    assert len(w.getvalue()) % 1000 == 0 and S.Reader(w.getvalue(), backend=OracleBackend()).ReadAll() == d


class CountingBackend(OracleBackend):
    def __init__(self):
        self.decoded = 0

    def decode_bodies(self, bodies):
        self.decoded += len(bodies)
        return super().decode_bodies(bodies)


def test_reader_skip():
    # Reader.Skip (reader.go:1034-1302): whole blocks inside the skipped range are neither decoded nor CRC-checked, the
    # block the range ends in is; skipping past the end is an unexpected EOF
    bs = 4096
    d = synth.text_like(10 * bs + 123, 5).tobytes() + synth.random_bytes(2 * bs).tobytes()   # compressed and stored chunks
    s = enc(d, 1, bs)
    for n in (0, 1, bs - 1, bs, bs + 1, 3 * bs, 7 * bs + 5, len(d) - 1, len(d)):
        be = CountingBackend()
        r = S.Reader(s, backend=be)
        r.Skip(n)
        assert r.ReadAll() == d[n:], n
        assert be.decoded <= 11 - min(n // bs, 11), (n, be.decoded)   # 11 compressed blocks, the skipped ones untouched
    r = S.Reader(s, backend=OracleBackend())
    r.Skip(2 * bs); r.Skip(bs + 7)                                        # skips add up
    assert r.ReadAll() == d[3 * bs + 7:]
    with pytest.raises(api.ErrCorrupt):
        r = S.Reader(s, backend=OracleBackend()); r.Skip(len(d) + 1); r.ReadAll()
    with pytest.raises(ValueError):
        S.Reader(s, backend=OracleBackend()).Skip(-1)
    # a damaged block: invisible when skipped, an error when read
    bad = bytearray(s)
    first_chunk = 10            # stream header is 10 bytes; the first chunk's CRC follows its 4-byte header
    bad[first_chunk + 4] ^= 0xFF
    r = S.Reader(bytes(bad), backend=OracleBackend()); r.Skip(bs)
    assert r.ReadAll() == d[bs:]
    with pytest.raises(api.ErrCRC):
        S.Reader(bytes(bad), backend=OracleBackend()).ReadAll()


def test_walk_chunks_agrees_with_reader():
    """walk_chunks (the chunk walk a sharded Reader deals blocks from) sees the same blocks, sizes and framing errors as
    Reader.WriteTo: valid oracle streams at several block sizes and levels (with index and padding chunks behind the EOF),
    then every single-byte mutation of the framing bytes of a small stream."""
    import random
    for d, bs, lvl in ((synth.text_like(300000, 2).tobytes(), 65536, 1), (synth.random_bytes(70000).tobytes(), 4096, 2),
                       (b"", 4096, 1), (b"abc", 4096, 1), (synth.json_like(400000).tobytes(), 1 << 20, 3)):
        s = O.stream_encode(d, lvl, bs, add_index=len(d) > 1000)
        blocks, total = S.walk_chunks(s)
        assert total == len(d) and sum(b.n for b in blocks) == len(d)
        assert [b.u_off for b in blocks] == [i * bs for i in range(len(blocks))]
        for b in blocks:
            assert s[b.chunk_off] == b.kind and b.payload_off == b.chunk_off + 8
            if b.kind == S.CHUNK_MINLZ:
                assert O.decode(b"\x00" + s[b.payload_off:b.payload_off + b.payload_len]) == d[b.u_off:b.u_off + b.n]
            else:
                assert s[b.payload_off:b.payload_off + b.payload_len] == d[b.u_off:b.u_off + b.n]
            assert b.crc == O.crc(d[b.u_off:b.u_off + b.n])
    d = synth.text_like(20000, 4).tobytes()
    s = O.stream_encode(d, 1, 4096)
    blocks, _ = S.walk_chunks(s)
    framing = list(range(10)) + [o for b in blocks for o in range(b.chunk_off, b.payload_off + b.hdr_len)] + list(range(len(s) - 8, len(s)))
    rnd = random.Random(5)
    for pos in framing:
        bad = bytearray(s); bad[pos] ^= 1 << rnd.randrange(8)
        try:
            S.Reader(bytes(bad), backend=OracleBackend()).ReadAll()
            want = None
        except api.MinLZError as e:
            want = type(e)
        try:
            S.walk_chunks(bytes(bad))
            got = None
        except api.MinLZError as e:
            got = type(e)
        # the walk does not decode: CRC and token errors are the workers' to find; everything the framing decides must agree
        if want in (api.ErrCRC,) or (want is api.ErrCorrupt and got is None):
            continue
        assert got == want, (pos, got, want)
    for cut in (3, 9, 12, len(s) // 2, len(s) - 1):
        with pytest.raises(api.ErrCorrupt):
            S.walk_chunks(s[:cut])
