"""Seek index: the host-side mirror (minlz_amd/index.py), the oracle's C restatement of
index.go and an independent reading of the SPEC.md:531-575 decode rules must agree.  No GPU."""
import io
import random

import numpy as np
import pytest

import oracle as O
from minlz_amd import api, stream as S, synth
from minlz_amd.index import Index, ErrUnexpectedEOF, put_varint, varint
from tests.test_stream_host import OracleBackend


def spec_decode(chunk):
    """SPEC.md:488-575, written from the table and the pseudo-code only."""
    assert chunk[0] == 0x40
    n = int.from_bytes(chunk[1:4], "little")
    b = chunk[4:4 + n]
    assert b[:6] == b"s2idx\x00" and b[-6:] == b"\x00xdi2s"
    assert int.from_bytes(b[-10:-6], "little") == len(chunk)
    p = 6
    vals = []
    for _ in range(4):
        v, k = varint(b, p); assert k > 0; p += k; vals.append(v)
    total_u, total_c, est, entries = vals
    has_u = b[p]; p += 1
    us, cs = [], []
    for i in range(entries):
        u = 0
        if has_u:
            u, k = varint(b, p); p += k
        us.append(u if i == 0 else us[-1] + est + u)
    guess = int(est / 2)
    for i in range(entries):
        c, k = varint(b, p); p += k
        if i == 0:
            cs.append(c)
        else:
            cs.append(cs[-1] + guess + c)
            guess += int(c / 2)  # truncating division
    assert p == len(b) - 10
    return total_u, total_c, est, list(zip(cs, us))


def test_varint_roundtrip():
    for v in (0, 1, -1, 63, 64, -64, -65, 1 << 20, -(1 << 40), (1 << 62), -(1 << 62)):
        b = put_varint(v)
        assert varint(b, 0) == (v, len(b))


@pytest.mark.parametrize("bs", [4096, 64 << 10, 1 << 20, 8 << 20])
def test_index_bytes_three_ways(bs):
    d = synth.text_like(24 << 20 if bs >= (64 << 10) else 3 << 20, 3).tobytes()
    plain = O.stream_encode(d, 1, bs)
    full = O.stream_encode(d, 1, bs, add_index=True)
    assert full[:len(plain)] == plain
    chunk = full[len(plain):]
    # host mirror, fed by the Writer (oracle-backed blocks): identical stream and index bytes
    w = io.BytesIO()
    wr = S.Writer(w, level=1, block_size=bs, concurrency=5, backend=OracleBackend(), add_index=True)
    wr.EncodeBuffer(d)
    assert wr.CloseIndex() == chunk
    assert w.getvalue() == full
    # without WriterAddIndex the index is still returned by CloseIndex but not part of the stream
    w2 = io.BytesIO()
    wr2 = S.Writer(w2, level=1, block_size=bs, concurrency=5, backend=OracleBackend())
    wr2.EncodeBuffer(d)
    assert wr2.CloseIndex() == chunk and w2.getvalue() == plain
    # load: mirror == oracle == spec reading
    ix = Index()
    assert ix.load(chunk) == b""
    tu, tc, est, offs, used = O.index_load(chunk)
    assert used == len(chunk)
    assert (tu, tc, est, offs) == (ix.total_uncompressed, ix.total_compressed, ix.est_block_uncomp, ix.offsets)
    assert spec_decode(chunk) == (tu, tc, est, offs)
    assert tu == len(d) and tc == len(plain)
    # the first entry is the stream header's own (0, 0) — the default Writer's output goroutine adds it (writer.go:236-243) —
    # every other entry points at a chunk header of a data block and at the right uncompressed offset
    assert offs[0] == (0, 0) and plain[0] == 0xff
    for c, u in offs[1:]:
        assert plain[c] in (0x01, 0x02) and u % bs == 0
    # entries are at least 1 MiB (or one block) apart (index.go:31,56-68,80-90)
    assert all(b[1] - a[1] >= max(bs, 1 << 20) for a, b in zip(offs, offs[1:]))
    # the plain Reader skips the index chunk; Index.load_stream finds it from the end
    assert S.Reader(full, backend=OracleBackend()).ReadAll() == d
    assert O.stream_decode(full, len(d)) == d
    assert Index().load_stream(full).offsets == offs


def test_find_and_errors():
    ix = Index(); ix.reset(1 << 20)
    for i in range(10):
        ix.add(100 + i * 400000, i << 20)
    chunk = ix.append_to(10 << 20, 5_000_000)
    jx = Index(); jx.load(chunk)
    assert jx.find(0) == (100, 0)
    assert jx.find((3 << 20) + 5) == (100 + 3 * 400000, 3 << 20)
    assert jx.find(-1) == (100 + 9 * 400000, 9 << 20)          # from the end (index.go:120-126)
    assert jx.find(10 << 20) == (100 + 9 * 400000, 9 << 20)
    with pytest.raises(ErrUnexpectedEOF):
        jx.find((10 << 20) + 1)
    with pytest.raises(ErrUnexpectedEOF):
        jx.find(-(10 << 20) - 1)
    with pytest.raises(api.ErrCorrupt):
        Index().find(0)                                          # TotalUncompressed unknown
    # corruptions (index.go:273-396)
    with pytest.raises(ErrUnexpectedEOF):
        Index().load(chunk[:10])
    bad = bytearray(chunk); bad[0] = 0x41
    with pytest.raises(api.ErrCorrupt):
        Index().load(bytes(bad))
    bad = bytearray(chunk); bad[5] ^= 1
    with pytest.raises(api.ErrUnsupported):
        Index().load(bytes(bad))
    bad = bytearray(chunk); bad[-1] ^= 1
    with pytest.raises(api.ErrCorrupt):
        Index().load(bytes(bad))
    with pytest.raises(O.OracleError):
        O.index_load(bytes(bad))
    assert Index().load(bytes([0x99]) + chunk[1:]) == b""        # legacy S2 chunk id is accepted (index.go:277)


def test_reduce_many_entries():
    # more blocks than maxIndexEntries: add() thins on the fly (reduceLight), appendTo reduces again; mirror == oracle
    n = 70000
    bs = 1 << 20
    rng = random.Random(5)
    c_off, u_off, c = [], [], 10
    for i in range(n):
        c_off.append(c); u_off.append(i * bs)
        c += rng.randrange(1000, 900000)
    ix = Index(); ix.reset(bs)
    for a, b in zip(c_off, u_off):
        ix.add(a, b)
    mine = ix.append_to(n * bs, c)
    assert mine == O.index_build(c_off, u_off, bs, n * bs, c)
    jx = Index(); jx.load(mine)
    assert 1000 < len(jx.offsets) <= 1 << 16
    assert spec_decode(mine)[3] == jx.offsets
    # irregular uncompressed offsets -> HasUncompressedOffsets = 1
    ix = Index(); ix.reset(bs)
    u = 0
    pairs = []
    for i in range(200):
        pairs.append((50 + i * 1000, u)); ix.add(50 + i * 1000, u)
        u += bs + rng.randrange(0, 5000)
    mine = ix.append_to(u, 50 + 200 * 1000)
    assert mine == O.index_build([p[0] for p in pairs], [p[1] for p in pairs], bs, u, 50 + 200 * 1000)
    assert spec_decode(mine)[3] == Index().load_stream(b"x" * 30 + mine).offsets


def test_read_seeker_with_oracle_backend():
    d = synth.text_like(9 << 20, 8).tobytes()
    w = io.BytesIO()
    wr = S.Writer(w, level=1, block_size=256 << 10, concurrency=4, backend=OracleBackend(), add_index=True)
    wr.EncodeBuffer(d)
    idx_bytes = wr.CloseIndex()
    for rs in (S.ReadSeeker(w.getvalue(), backend=OracleBackend()), S.ReadSeeker(w.getvalue(), index=idx_bytes, backend=OracleBackend())):
        rng = random.Random(1)
        for _ in range(12):
            off = rng.randrange(0, len(d))
            n = rng.randrange(1, 300000)
            assert rs.ReadAt(n, off) == d[off:off + n]
        assert rs.Seek(-100, 2) == len(d) - 100 and rs.Read(1000) == d[-100:]
        assert rs.Seek(5, 0) == 5 and rs.Read(10) == d[5:15] and rs.Seek(10, 1) == 25 and rs.Read(3) == d[25:28]
        with pytest.raises(ErrUnexpectedEOF):
            rs.Seek(len(d) + 1, 0)
