"""Stream path on the GPU: device CRC32C and the Writer/Reader mirror with the HIP backend,
checked against the oracle's restatement of the reference framing."""
import io

import numpy as np
import pytest

import minlz_amd as mz
import oracle as O
from minlz_amd import stream as S, synth

pytestmark = pytest.mark.gpu


def test_crc_kat_and_sizes(ctx):
    assert mz.crc(b"abcd", ctx).to_bytes(4, "little").hex() == "6810e6b6"   # minlz_test.go:1120-1134
    d = synth.text_like((8 << 20) + 12345, 9)
    for n in (0, 1, 7, 8, 9, 1023, 1024, 1025, 65536, 262143, 262144, 262145, 1 << 20, 8 << 20, d.size):
        assert mz.crc(d[:n], ctx) == O.crc(d[:n]), n


def test_crc_batch_device(ctx):
    import torch
    from minlz_amd._lib import BlockDesc
    d = synth.random_bytes(3_000_000, 3)
    t = torch.from_numpy(d).cuda()
    cuts = [(0, 1000000), (1000000, 1), (1000001, 0), (1000001, 1999999)]
    out = torch.zeros(len(cuts), dtype=torch.int32, device="cuda")
    ctx.crc_batch_device(torch.cuda.current_stream().cuda_stream, t.data_ptr(), [BlockDesc(o, l, 0, 0) for o, l in cuts], out.data_ptr())
    torch.cuda.synchronize()
    got = [int(x) & 0xffffffff for x in out.cpu().tolist()]
    assert got == [O.crc(d[o:o + l]) for o, l in cuts]


@pytest.mark.parametrize("level", [0, 1, 2])
def test_writer_reader_roundtrip_and_reference_reader(ctx, level):
    be = S.HipBackend(ctx)
    for d, bs in ((synth.text_like(3_000_000, 5).tobytes(), 1 << 20), (synth.random_bytes(200000).tobytes(), 65536),
                  (synth.json_like(20 << 20).tobytes(), 8 << 20), (b"", 4096), (b"tiny", 4096)):
        w = io.BytesIO()
        wr = S.Writer(w, level=level, block_size=bs, concurrency=4, backend=be)
        wr.EncodeBuffer(d)
        wr.Close()
        s = w.getvalue()
        assert O.stream_decode(s, len(d)) == d                      # the reference Reader's restatement accepts it
        assert S.Reader(s, backend=be, batch=3).ReadAll() == d       # and the GPU reader round-trips
        assert S.Reader(O.stream_encode(d, max(level, 0), bs), backend=be).ReadAll() == d  # reference-made streams decode


def test_reader_detects_corruption(ctx):
    be = S.HipBackend(ctx)
    d = synth.text_like(500000, 6).tobytes()
    s = bytearray(O.stream_encode(d, 1, 65536))
    s[len(s) // 2] ^= 0x10
    with pytest.raises((mz.ErrCRC, mz.ErrCorrupt)):
        S.Reader(bytes(s), backend=be).ReadAll()


def test_index_and_read_seeker_on_device(ctx):
    # WriterAddIndex + Reader.ReadSeeker (index.go, reader.go:1304-1487) with every block on the GPU
    import random
    d = synth.text_like(20 << 20, 12).tobytes()
    w = io.BytesIO()
    wr = S.Writer(w, level=1, block_size=1 << 20, concurrency=8, backend=S.HipBackend(ctx), add_index=True)
    wr.EncodeBuffer(d)
    idx = wr.CloseIndex()
    st = w.getvalue()
    assert st.endswith(idx) and O.stream_decode(st, len(d)) == d         # the reference-side reader skips chunk 0x40
    tu, tc, est, offs, used = O.index_load(idx)
    assert tu == len(d) and tc == len(st) - len(idx) and len(offs) == 20
    rs = S.ReadSeeker(st, backend=S.HipBackend(ctx))
    rng = random.Random(2)
    for _ in range(8):
        off = rng.randrange(0, len(d)); n = rng.randrange(1, 1 << 20)
        assert rs.ReadAt(n, off) == d[off:off + n]


@pytest.mark.parametrize("level,bs", [(1, 1 << 20), (2, 8 << 20), (0, 64 << 10), (-1, 256 << 10)])
def test_stream_abi_matches_writer_mirror(ctx, level, bs):
    # mlz_stream_encode (C++ host code in the library) == the Python Writer mirror over the same device blocks,
    # with and without the seek index; both decode with the reference Reader's restatement and with mlz_stream_decode
    for d in (synth.text_like(20_000_000, 31).tobytes(), synth.random_bytes(300000).tobytes(), b"", b"x", synth.json_like(3 << 20).tobytes()):
        for add_index in (False, True):
            st = mz.stream_encode(d, level, bs, add_index, ctx)
            w = io.BytesIO()
            wr = S.Writer(w, level=level, block_size=bs, concurrency=7, backend=S.HipBackend(ctx), add_index=add_index)
            wr.EncodeBuffer(d)
            wr.Close()
            assert st == w.getvalue(), (len(d), add_index)
            assert O.stream_decode(st, len(d)) == d
            assert mz.stream_decode(st, ctx=ctx) == d
    # reference-algorithm streams (oracle writer) decode through the ABI as well
    d = synth.text_like(9_000_000, 5).tobytes()
    for lv in (0, 1, 2, 3):
        assert mz.stream_decode(O.stream_encode(d, lv, 1 << 20, add_index=True), ctx=ctx) == d


def test_stream_decode_into_pinned_memory(ctx):
    # A page-locked destination is written by the decode kernels themselves (no copy-out stage), stored chunks by the host: compressed,
    # stored (random bytes) and reference-algorithm (general path) chunks in one stream, at an odd offset, guard bytes on both sides,
    # good and bad CRCs.
    import ctypes as C
    import torch
    from minlz_amd import _lib
    L = _lib.lib()
    d = synth.text_like(5_000_000, 3).tobytes() + synth.random_bytes(1_500_000, seed=8).tobytes() + synth.json_like(2 << 20, 4).tobytes()
    for st in (mz.stream_encode(d, 1, 1 << 20, True, ctx), mz.stream_encode(d, 2, 4 << 20, False, ctx), O.stream_encode(d, 1, 1 << 20, add_index=False)):
        src = torch.zeros(len(st) + 64, dtype=torch.uint8, pin_memory=True)
        src.numpy()[5:5 + len(st)] = np.frombuffer(st, dtype=np.uint8)
        dst = torch.full((len(d) + 128,), 0x5A, dtype=torch.uint8, pin_memory=True)
        r = L.mlz_stream_decode(ctx.handle, 0, src.data_ptr() + 5, len(st), dst.data_ptr() + 33, len(d))
        assert r == len(d)
        h = dst.numpy()
        assert h[33:33 + len(d)].tobytes() == d
        assert (h[:33] == 0x5A).all() and (h[33 + len(d):] == 0x5A).all()
        bad = bytearray(st)
        bad[len(bad) // 2] ^= 0x10
        src.numpy()[5:5 + len(st)] = np.frombuffer(bytes(bad), dtype=np.uint8)
        assert L.mlz_stream_decode(ctx.handle, 0, src.data_ptr() + 5, len(st), dst.data_ptr() + 33, len(d)) < 0


def test_stream_encode_into_pinned_memory(ctx):
    # A page-locked destination of mlz_stream_encode: headers by the host, chunk bodies by stream_place_kernel — the same bytes as into
    # pageable memory, nothing outside [dst, dst + returned length) touched except inside the bound; compressed and stored chunks, index.
    import ctypes as C
    import torch
    from minlz_amd import _lib
    L = _lib.lib()
    d = synth.text_like(9_000_000, 13).tobytes() + synth.random_bytes(2_500_000, seed=9).tobytes() + synth.json_like(3 << 20, 14).tobytes() + b"tail"
    src = torch.zeros(len(d) + 64, dtype=torch.uint8, pin_memory=True)
    src.numpy()[7:7 + len(d)] = np.frombuffer(d, dtype=np.uint8)
    for level, bs, flags in ((1, 1 << 20, 1), (2, 8 << 20, 0), (0, 64 << 10, 1)):
        want = mz.stream_encode(d, level, bs, bool(flags), ctx)
        cap = L.mlz_stream_bound(len(d), bs, flags)
        dst = torch.full((cap + 96,), 0x5A, dtype=torch.uint8, pin_memory=True)
        r = L.mlz_stream_encode(ctx.handle, level, bs, flags, src.data_ptr() + 7, len(d), dst.data_ptr() + 19, cap)
        assert r == len(want)
        h = dst.numpy()
        assert h[19:19 + r].tobytes() == want
        assert (h[:19] == 0x5A).all() and (h[19 + cap:] == 0x5A).all()
        assert O.stream_decode(want, len(d)) == d


def test_stream_abi_errors(ctx):
    d = synth.text_like(3_000_000, 9).tobytes()
    st = bytearray(mz.stream_encode(d, 1, 1 << 20, True, ctx))
    bad = bytearray(st); bad[16] ^= 0x40                     # inside the first chunk's CRC
    with pytest.raises(mz.ErrCRC):
        mz.stream_decode(bytes(bad), ctx=ctx)
    assert mz.stream_decode(bytes(bad), ignore_crc=True, ctx=ctx) == d    # ReaderIgnoreCRC
    bad = bytearray(st); bad[40] ^= 0xff                     # token bytes: corrupt data or CRC mismatch
    with pytest.raises((mz.ErrCorrupt, mz.ErrCRC)):
        mz.stream_decode(bytes(bad), ctx=ctx)
    with pytest.raises(mz.ErrCorrupt):
        mz.stream_decode(bytes(st[:len(st) // 2]), ctx=ctx)  # truncated inside a chunk
    with pytest.raises(mz.ErrCorrupt):
        mz.stream_decode(bytes(st[10:]), ctx=ctx)            # no stream header
    with pytest.raises(mz.ErrUnsupported):
        mz.stream_decode(bytes.fromhex("ff060000734e61507059"), ctx=ctx)  # Snappy stream identifier
    with pytest.raises(mz.MinLZError):
        mz.stream_encode(d, 1, 1000, False, ctx)             # block size below 4 KiB (writer.go:1238-1246)
    with pytest.raises(mz.ErrInvalidLevel):
        mz.stream_encode(d, 3, 1 << 20, False, ctx)
    # framing KAT (minlz_test.go:1120-1134)
    kat = bytes.fromhex("ff0600004d696e4c7a02") + b"\x01\x08\x00\x00" + b"\x68\x10\xe6\xb6" + b"abcd" + b"\x20\x00\x00\x00"
    assert mz.stream_decode(kat, ctx=ctx) == b"abcd"


def _to_compcrc(stream):
    """Rewrite every 0x02 chunk as 0x03 (same body; the CRC covers the token bytes, SPEC chunk 0x03 / reader.go:341-344)."""
    from minlz_amd.stream import uvarint
    b = bytearray(stream)
    p = 0
    while p + 4 <= len(b):
        t = b[p]
        n = b[p + 1] | b[p + 2] << 8 | b[p + 3] << 16
        if t == 0x02:
            _, hl = uvarint(b, p + 8)
            b[p] = 0x03
            b[p + 4:p + 8] = O.crc(bytes(b[p + 8 + hl:p + 4 + n])).to_bytes(4, "little")
        p += 4 + n
    return bytes(b)


def test_chunk_type_3_crc_over_compressed_bytes(ctx):
    # chunk 0x03 = MinLZ block whose CRC is taken over the compressed bytes (the reference's LZ4 converter writes these,
    # lz4convert.go:412; its Reader accepts them everywhere a 0x02 chunk may stand)
    d = synth.text_like(3_000_000, 61).tobytes() + synth.random_bytes(100_000).tobytes() + synth.json_like(500_000).tobytes()
    s2 = mz.stream_encode(d, 1, 1 << 20, ctx=ctx)
    s3 = _to_compcrc(s2)
    assert s3 != s2 and O.stream_decode(s3, len(d)) == d            # the reference-shaped reader takes it
    assert mz.stream_decode(s3, ctx=ctx) == d                       # mlz_stream_decode
    out = io.BytesIO()
    S.Reader(io.BytesIO(s3), backend=S.HipBackend(ctx)).WriteTo(out)   # the incremental mirror
    assert out.getvalue() == d
    bad = bytearray(s3)
    bad[len(bad) // 2] ^= 0x40
    with pytest.raises(mz.MinLZError):
        mz.stream_decode(bytes(bad), ctx=ctx)
