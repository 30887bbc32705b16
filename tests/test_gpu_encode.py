"""Encode parity of the HIP encoder: output may differ byte-wise from the reference, but the
oracle decoder (pinned on the reference's golden vectors) must reproduce the input bit-exact,
the block rules of encode.go:74-139 must hold, and the size must stay within the stated
tolerance of the oracle's restatement of the reference L1 encoder."""
import numpy as np
import pytest

import minlz_amd as mz
import oracle as O
from minlz_amd import synth
from tests.util import load_zip

pytestmark = pytest.mark.gpu

# Stated ratio tolerances, on four 8 MiB inputs: the bench stream (enwik-like), the text stand-in, the config-3 JSON
# stream, the free-text JSON stand-in:
#   LevelFastest  : C_gpu(1)  <= RATIO_TOL[kind] * C_oracle(L1)   (measured 1.006 / 1.023 / 1.030 / 1.065 with the three-level tile pattern of round 3;
#                                                                     0.992 / 0.998 / 1.023 / 1.057 with the four-level one, which LevelBalanced keeps)
#   LevelBalanced : C_gpu(2)  <= RATIO_TOL_L2 * C_oracle(L2)   (measured 0.989 / 0.925 / 1.048 since round 4 — no tile levels, MLZ_OPT_L2_FREE —;
#                                                                     1.080 / 1.092 / 1.095 with the four-level pattern of rounds 1-3, RATIO_TOL_L2_LEVELS), and C_gpu(2) <= C_gpu(1)
#   LevelSuperFast: C_gpu(-1) <= RATIO_TOL_L0 * C_oracle(L0)   (measured 0.90 / 0.87 on text, 1.08 on the JSON stream: 4-byte matches against the reference's 8)
# and on 64 KiB blocks (the reference's small-block classes, encode_l1.go:285-524 / encode_l0.go:281-522):
#   C_gpu(1) <= RATIO_TOL_64K * C_oracle(L1)
RATIO_TOL = {"enwik": 1.02, "text": 1.04, "json": 1.04, "json_text": 1.08, "twain": 1.08}
# LevelBalanced since round 5: no tile levels, far sources at least four tiles back (MLZ_OPT_L2_GAP = 4: the decoder settles four tiles of a block side by
# side); measured 1.003 / 0.950 / 1.056 / 1.067 x the oracle's L2 (with sources anywhere, round 4: 0.989 / 0.925 / 1.048 / 1.057)
RATIO_TOL_L2 = {"enwik": 1.02, "text": 1.00, "json": 1.06, "json_text": 1.08, "twain": 1.08}
RATIO_TOL_L2_LEVELS = 1.12
RATIO_TOL_L0 = 1.10   # (text streams 0.87 - 0.90; the config-3 JSON stream 1.084: no far tables at this level)
RATIO_TOL_64K = 1.08
RATIO_TOL_SMALL = 1.02   # 4 KiB and 16 KiB blocks (measured 0.92 - 0.99)


def roundtrip(d, ctx, level=1):
    d = np.ascontiguousarray(np.frombuffer(d, dtype=np.uint8) if not isinstance(d, np.ndarray) else d)
    enc = mz.Encode(d, level, ctx)
    assert len(enc) <= mz.MaxEncodedLen(d.size)
    assert O.decode(enc, guard=64) == d.tobytes()
    return enc


def test_block_container_rules(ctx):
    # encode.go:83-90,137-138, encodeUncompressed :223-228
    assert mz.Encode(b"", 1, ctx) == b"\x00"
    assert mz.Encode(b"abc", 1, ctx) == b"\x00\x00abc"
    small = bytes(range(15))
    assert mz.Encode(small, 1, ctx) == b"\x00\x00" + small
    r = synth.random_bytes(100000).tobytes()
    assert mz.Encode(r, 1, ctx) == b"\x00\x00" + r           # incompressible -> stored
    assert mz.Encode(r, 0, ctx) == b"\x00\x00" + r           # LevelUncompressed
    z = mz.Encode(bytes(100000), 1, ctx)
    assert z[:1] == b"\x00" and z[1:4] == b"\xa0\x8d\x06"     # 00 + uvarint(100000)
    with pytest.raises(mz.ErrTooLarge):
        mz.Encode(np.zeros((8 << 20) + 1, dtype=np.uint8), 1, ctx)
    with pytest.raises(mz.ErrInvalidLevel):
        mz.Encode(bytes(100), 7, ctx)
    with pytest.raises(mz.ErrInvalidLevel):
        mz.Encode(bytes(100), 3, ctx)  # LevelSmallest is CPU-side only


def test_level_superfast(ctx):
    # LevelSuperFast (-1, encode.go:26-29): tile-local matching only; valid stream, worse ratio, no far tables
    d = synth.text_like(4 << 20, 6)
    e0 = roundtrip(d, ctx, level=mz.LevelSuperFast)
    e1 = roundtrip(d, ctx, level=mz.LevelFastest)
    assert len(e1) <= len(e0) < d.size
    assert mz.Decode(e0, ctx) == d.tobytes()


@pytest.mark.parametrize("far", [0, 1])
def test_enc_regressions_roundtrip(ctx, far):
    ctx.set_option(mz.OPT_ENCODE_FAR, far)
    try:
        for label, blob in load_zip("enc_regressions.zip"):
            roundtrip(blob, ctx)
    finally:
        ctx.set_option(mz.OPT_ENCODE_FAR, 1)


def test_corpus_roundtrip(ctx):
    for label, blob in load_zip("block-corpus-enc.zip")[::5]:
        roundtrip(blob, ctx)


@pytest.mark.parametrize("name", synth.PATTERNS)
def test_patterns(ctx, name):
    for size in (16, 17, 100, 4096, 65535, 65536, 65537, 65549, 70000, 131072, 300000):
        roundtrip(synth.pattern(name, size), ctx)


def test_sizes_sweep(ctx):
    d = synth.text_like(200000, 21)
    for size in list(range(0, 70)) + [255, 256, 1000, 65528, 65529, 65535, 65536, 65537, 65543, 65544, 65545, 131071, 131072, 131073, 199999]:
        roundtrip(d[:size], ctx)


def test_large_offsets(ctx):
    for min_off in (65536, 65600, 200000, 1 << 20, (2 << 20) + 65535, 3 << 20):
        roundtrip(synth.large_offset(min_off + 5000, min_off), ctx)


def test_ratio_half_noise(ctx):
    # TestEncodeNoiseThenRepeats, minlz_test.go:776-797
    for n in (256 * 1024, 2048 * 1024):
        enc = roundtrip(synth.pattern("half", n), ctx)
        assert len(enc) < n * 3 // 4


def test_huge_zeros(ctx):
    # TestEncodeHuge, encode_test.go:26-50
    # (the reference only asserts len <= MaxEncodedLen; here: 1024 independent 8 KiB pieces of <= 6 bytes each)
    enc = roundtrip(np.zeros(8 << 20, dtype=np.uint8), ctx)
    assert len(enc) < 8192


def _ratio_input(kind):
    # text: the round-1 stand-in; json: the config-3 stream (oracle L2 ~0.24); json_text: the earlier JSON stand-in with
    # free-text messages; enwik: the bench stream (config 2)
    return {"text": lambda: synth.text_like(8 << 20, 1), "json": lambda: synth.json_like(8 << 20),
            "json_text": lambda: synth.json_text(8 << 20), "enwik": lambda: synth.enwik_like(8 << 20, 1)}[kind]()


@pytest.mark.parametrize("kind", ["text", "json", "json_text", "enwik"])
def test_ratio_within_tolerance_of_reference_l1(ctx, kind):
    d = _ratio_input(kind)
    enc = roundtrip(d, ctx)
    ref = O.encode(d, 1)
    assert len(enc) <= RATIO_TOL[kind] * len(ref), (len(enc), len(ref))
    enc2 = roundtrip(d, ctx, level=2)
    ref2 = O.encode(d, 2)
    assert len(enc2) <= len(enc), (len(enc2), len(enc))
    assert len(enc2) <= RATIO_TOL_L2[kind] * len(ref2), (len(enc2), len(ref2))
    # ... and with the tile levels of rounds 1-3 (MLZ_OPT_L2_FREE = 0: level-scheduled decode, 8-9 % more output)
    ctx.set_option(mz.OPT_L2_FREE, 0)
    try:
        enc2l = roundtrip(d, ctx, level=2)
        assert mz.Decode(enc2l, ctx) == d.tobytes() and ctx.general_blocks() == 0
    finally:
        ctx.set_option(mz.OPT_L2_FREE, 1)
    assert len(enc2) <= len(enc2l) <= RATIO_TOL_L2_LEVELS * len(ref2), (len(enc2), len(enc2l), len(ref2))


def test_encode_block_contract(ctx):
    # WriterCustomEncoder / encodeBlock contract: tokens only; 0 = incompressible
    d = synth.text_like(300000, 2)
    body = mz.encode_block(d, 1, ctx)
    assert 0 < len(body) < d.size
    assert O.decode_body(body, d.size) == (0, d.tobytes())
    assert mz.encode_block(synth.random_bytes(100000), 1, ctx) == b""
    assert mz.encode_block(b"0123456789", 1, ctx) == b""


def test_try_encode(ctx):
    assert mz.TryEncode(synth.random_bytes(50000), 1, ctx) is None
    d = synth.text_like(50000, 3)
    e = mz.TryEncode(d, 1, ctx)
    assert e is not None and O.decode(e) == d.tobytes()


def test_deterministic(ctx):
    d = synth.text_like(2 << 20, 4)
    assert mz.Encode(d, 1, ctx) == mz.Encode(d, 1, ctx)


def test_batch_and_gpu_roundtrip(ctx):
    blocks = [synth.text_like(8 << 20, 31).tobytes(), synth.json_like(3 << 20).tobytes(), b"", b"x" * 20,
              synth.random_bytes(1 << 20).tobytes(), synth.pattern("off2", 70000).tobytes()]
    encs = mz.encode_batch(blocks, 1, ctx)
    for b, e in zip(blocks, encs):
        assert O.decode(e) == b
    assert mz.decode_batch(encs, ctx) == blocks


def test_level_balanced_roundtrip(ctx):
    d = synth.json_like(2 << 20)
    enc = roundtrip(d, ctx, level=2)
    assert len(enc) < d.size // 2
    # level-2 streams carry copies from the preceding tile (pre-loaded near table): device decode, both exec passes
    t = synth.text_like(3_000_000, 5)
    e = roundtrip(t, ctx, level=2)
    for algo in (0, 3, 1):
        ctx.set_option(mz.OPT_DECODE_ALGO, algo)
        try:
            assert mz.Decode(e, ctx) == t.tobytes()
        finally:
            ctx.set_option(mz.OPT_DECODE_ALGO, 0)
    # short blocks: one tile and a bit, tile boundary +- a few bytes
    for n in (32768, 32769, 32770, 32771, 32775, 65536 + 3, 98304 + 1):
        roundtrip(t[:n], ctx, level=2)


def test_device_batch_unaligned_offsets(ctx):
    # device-resident API with block offsets of every alignment (the kernels take 16-byte fast paths only when they can)
    import torch
    from minlz_amd._lib import BlockDesc
    dev = torch.device("cuda", 0)
    parts = [synth.text_like(700_001, 3), synth.json_like(300_003), synth.random_bytes(70_001), synth.pattern("zeros", 100_000), synth.text_like(33, 5)]
    offs, cur = [], 1
    for p in parts:
        offs.append(cur)
        cur += p.size + 3          # 1, +odd strides: every residue mod 16 shows up
    host = np.zeros(cur + 64, dtype=np.uint8)
    for o, p in zip(offs, parts):
        host[o:o + p.size] = p
    src = torch.from_numpy(host).to(dev)
    ecap = [mz.MaxEncodedLen(p.size) for p in parts]
    eoffs, ecur = [], 5
    for c in ecap:
        eoffs.append(ecur)
        ecur += c + 7
    enc = torch.zeros(ecur + 64, dtype=torch.uint8, device=dev)
    elen = torch.zeros(len(parts), dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    edesc = [BlockDesc(o, p.size, eo, c) for o, p, eo, c in zip(offs, parts, eoffs, ecap)]
    ctx.encode_batch_device(st, 1, src.data_ptr(), enc.data_ptr(), edesc, elen.data_ptr())
    torch.cuda.synchronize()
    lens = elen.cpu().tolist()
    assert all(l > 0 for l in lens)
    ehost = enc.cpu().numpy()
    for p, eo, l in zip(parts, eoffs, lens):
        assert O.decode(ehost[eo:eo + l].tobytes()) == p.tobytes()
    # nothing outside the blocks' output ranges was touched
    mask = np.ones(ehost.size, dtype=bool)
    for eo, c in zip(eoffs, ecap):
        mask[eo:eo + c] = False
    assert not ehost[mask].any()
    # decode back to odd offsets, with guard bytes between the blocks
    doffs, dcur = [], 3
    for p in parts:
        doffs.append(dcur)
        dcur += p.size + 5
    dec = torch.full((dcur + 64,), 0xA5, dtype=torch.uint8, device=dev)
    dlen = torch.zeros(len(parts), dtype=torch.int64, device=dev)
    ddesc = [BlockDesc(eo, l, do, p.size) for eo, l, do, p in zip(eoffs, lens, doffs, parts)]
    ctx.decode_batch_device(st, enc.data_ptr(), dec.data_ptr(), ddesc, dlen.data_ptr())
    torch.cuda.synchronize()
    assert dlen.cpu().tolist() == [p.size for p in parts]
    dhost = dec.cpu().numpy()
    gmask = np.ones(dhost.size, dtype=bool)
    for p, do in zip(parts, doffs):
        assert dhost[do:do + p.size].tobytes() == p.tobytes()
        gmask[do:do + p.size] = False
    assert (dhost[gmask] == 0xA5).all()


def test_concurrent_single_block_calls(ctx):
    # The reference's Writer/Reader call the block codec from one goroutine per block (writer.go:501-560,
    # reader.go:830-859); the C ABI must be re-entrant.  Concurrent single-block calls are combined
    # into batched launches: results must equal the one-at-a-time results, errors must stay per call.
    import threading
    blocks = [synth.text_like(1 << 20, 40 + i).tobytes() for i in range(6)] + [synth.json_like(700_000).tobytes(), b"", b"tiny",
              synth.random_bytes(300_000).tobytes(), synth.pattern("zeros", 200_000).tobytes(), synth.text_like(8 << 20, 77).tobytes()]
    serial_enc = [mz.Encode(b, 1, ctx) for b in blocks]
    serial_body = [mz.encode_block(b, 2, ctx) for b in blocks]
    b0, r0 = ctx.combine_stats()
    res = {}
    errs = []

    def work(i):
        try:
            b = blocks[i % len(blocks)]
            kind = i % 4
            if kind == 0:
                res[i] = mz.Encode(b, 1, ctx) == serial_enc[i % len(blocks)]
            elif kind == 1:
                res[i] = mz.Decode(serial_enc[i % len(blocks)], ctx) == b
            elif kind == 2:
                res[i] = mz.encode_block(b, 2, ctx) == serial_body[i % len(blocks)]
            else:
                body = serial_body[i % len(blocks)]
                if body:
                    res[i] = mz.decode_block(body, len(b), ctx) == (0, b)
                else:  # incompressible: a corrupt body must come back as code 1 for this call only
                    res[i] = mz.decode_block(b"\xff" * 8, 100, ctx)[0] == 1
        except Exception as ex:  # noqa: BLE001
            errs.append((i, repr(ex)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(96)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs[:3]
    assert len(res) == 96 and all(res.values()), [i for i, ok in res.items() if not ok][:8]
    b1, r1 = ctx.combine_stats()
    assert r1 - r0 == 96
    assert b1 - b0 < 96          # at least some calls shared a launch


def test_config1_tom_sawyer(ctx, twain, twain_mzb):
    # BASELINE config 1: testdata/Mark.Twain-Tom.Sawyer.txt as a single block (minlz_test.go:626-660 holds its
    # LevelSmallest encoding).  Encode leg on the HIP path at every device level, decode leg with every decode pass.
    for level, tol in ((mz.LevelFastest, RATIO_TOL["twain"]), (mz.LevelBalanced, RATIO_TOL_L2_LEVELS), (mz.LevelSuperFast, RATIO_TOL_L0)):   # (one tile: tile levels or none is the same thing)
        enc = roundtrip(twain, ctx, level=level)
        ref = O.encode(twain, level)
        assert len(enc) <= tol * len(ref), (level, len(enc), len(ref))
        assert mz.Decode(enc, ctx, guard=64) == twain
    for algo in (0, 3, 1):
        ctx.set_option(mz.OPT_DECODE_ALGO, algo)
        try:
            assert mz.Decode(twain_mzb, ctx, guard=64) == twain
        finally:
            ctx.set_option(mz.OPT_DECODE_ALGO, 0)


@pytest.mark.parametrize("kind", ["text", "json", "json_text", "enwik"])
def test_level_superfast_ratio(ctx, kind):
    # LevelSuperFast against the oracle's restatement of encode_l0.go on 8 MiB blocks
    d = _ratio_input(kind)
    enc = roundtrip(d, ctx, level=mz.LevelSuperFast)
    ref = O.encode(d, -1)
    assert len(enc) <= RATIO_TOL_L0 * len(ref), (len(enc), len(ref))


def test_small_block_classes_ratio(ctx):
    # 512 x 64 KiB text blocks in one batch (config-5 block size): every block round-trips through the oracle decoder and
    # the batch stays within tolerance of the oracle's <= 64 KiB size classes at LevelFastest and LevelSuperFast
    d = synth.text_like(512 * 65536, 12)
    blocks = [d[i:i + 65536].tobytes() for i in range(0, d.size, 65536)]
    for level, tol in ((mz.LevelFastest, RATIO_TOL_64K), (mz.LevelSuperFast, RATIO_TOL_L0)):
        encs = mz.encode_batch(blocks, level, ctx)
        for b, e in zip(blocks[::16], encs[::16]):
            assert O.decode(e) == b
        assert mz.decode_batch(encs, ctx) == blocks
        c_gpu = sum(len(e) for e in encs)
        c_ref = sum(len(O.encode(b, level)) for b in blocks)
        assert c_gpu <= tol * c_ref, (level, c_gpu, c_ref)


@pytest.mark.parametrize("bs", [4096, 16384])
def test_smallest_stream_block_sizes_ratio(ctx, bs):
    # Streams may use blocks down to 4 KiB (writer.go:1238-1246); the reference has its own parameter sets for them
    # (encode_l1.go:285-524, _generate/gen.go:62-66).  Every block round-trips through the oracle decoder; the batch stays
    # within RATIO_TOL_SMALL of the oracle's L1 at the same block size on text, the bench stream and the JSON stream.
    for d in (synth.text_like(256 * bs, 12), synth.enwik_like(256 * bs, 3), synth.json_like(256 * bs)):
        blocks = [d[i:i + bs].tobytes() for i in range(0, d.size, bs)]
        encs = mz.encode_batch(blocks, mz.LevelFastest, ctx)
        for b, e in zip(blocks[::8], encs[::8]):
            assert O.decode(e) == b
        assert mz.decode_batch(encs, ctx) == blocks
        c_gpu = sum(len(e) for e in encs)
        c_ref = sum(len(O.encode(b, 1)) for b in blocks)
        assert c_gpu <= RATIO_TOL_SMALL * c_ref, (bs, c_gpu, c_ref)


def test_incompressible_8mib_block(ctx):
    # BASELINE config 4 at block size: an incompressible 8 MiB block takes the stored path (encode.go:137-138) and
    # decodes back on the device
    r = synth.random_bytes(8 << 20, seed=11)
    enc = mz.Encode(r, mz.LevelFastest, ctx)
    assert len(enc) == r.size + 2 and enc[:2] == b"\x00\x00"
    assert bytes(enc[2:]) == r.tobytes()
    assert mz.Decode(enc, ctx, guard=64) == r.tobytes()
    assert mz.encode_block(r, mz.LevelFastest, ctx) == b""
    # half noise, half text: compresses, and the noise tiles are stored as literal runs
    mix = np.concatenate([r[:4 << 20], synth.text_like(4 << 20, 2)])
    e = roundtrip(mix, ctx)
    assert len(e) < mix.size * 0.75
    assert mz.Decode(e, ctx) == mix.tobytes()


def test_device_calls_on_two_streams_share_the_workspace(ctx):
    # One context, two streams, different batches back to back without a host sync in between: the second call must
    # wait (on the device) for the first one's kernels before it reuses descriptors, scratch and far tables.
    import torch
    from minlz_amd._lib import BlockDesc
    dev = torch.device("cuda", 0)
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    a = synth.text_like(24 << 20, 41)
    b_ = synth.json_like(8 << 20, 42)
    A, B = torch.from_numpy(a).to(dev), torch.from_numpy(b_).to(dev)
    blk = 8 << 20
    stride = blk + 256
    def desc(n):
        k = (n + blk - 1) // blk
        return (BlockDesc * k)(*[BlockDesc(i * blk, min(blk, n - i * blk), i * stride, stride) for i in range(k)]), k
    da, ka = desc(a.size)
    db, kb = desc(b_.size)
    ea = torch.zeros(ka * stride, dtype=torch.uint8, device=dev); la = torch.zeros(ka, dtype=torch.int64, device=dev)
    eb = torch.zeros(kb * stride, dtype=torch.uint8, device=dev); lb = torch.zeros(kb, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    for _ in range(3):
        ctx.encode_batch_device(s1.cuda_stream, 1, A.data_ptr(), ea.data_ptr(), da, la.data_ptr())
        ctx.encode_batch_device(s2.cuda_stream, 1, B.data_ptr(), eb.data_ptr(), db, lb.data_ptr())
    torch.cuda.synchronize()
    ha, hb = ea.cpu().numpy(), eb.cpu().numpy()
    out_a = b"".join(O.decode(ha[i * stride:i * stride + l].tobytes()) for i, l in enumerate(la.cpu().tolist()))
    out_b = b"".join(O.decode(hb[i * stride:i * stride + l].tobytes()) for i, l in enumerate(lb.cpu().tolist()))
    assert out_a == a.tobytes() and out_b == b_.tobytes()


def test_a_stream_destroyed_between_calls(ctx):
    # mlz_release_stream: a stream that carried a device call is released and DESTROYED before the context's next call on another stream —
    # the context's ordering event was recorded while the stream was alive, so the next call neither touches the dead handle nor runs
    # ahead of the first call's kernels (round-5 advisor finding: the lazy record on a destroyed stream was undefined behaviour).
    import ctypes as C
    import torch
    from minlz_amd._lib import BlockDesc
    # the HIP runtime this process already runs on (PyTorch bundles its own copy: a second one would not know the library's context)
    path = next(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l)
    hip = C.CDLL(path)
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    dev = torch.device("cuda", 0)
    a = synth.text_like(16 << 20, 61)
    b_ = synth.json_like(8 << 20, 62)
    A, B = torch.from_numpy(a).to(dev), torch.from_numpy(b_).to(dev)
    blk = 8 << 20
    stride = blk + 256
    def desc(n):
        k = (n + blk - 1) // blk
        return (BlockDesc * k)(*[BlockDesc(i * blk, min(blk, n - i * blk), i * stride, stride) for i in range(k)]), k
    da, ka = desc(a.size)
    db, kb = desc(b_.size)
    ea = torch.zeros(ka * stride, dtype=torch.uint8, device=dev); la = torch.zeros(ka, dtype=torch.int64, device=dev)
    eb = torch.zeros(kb * stride, dtype=torch.uint8, device=dev); lb = torch.zeros(kb, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    s2 = torch.cuda.Stream(dev)
    for _ in range(3):
        h = C.c_void_p()
        assert hip.hipStreamCreate(C.byref(h)) == 0
        ctx.encode_batch_device(h, 1, A.data_ptr(), ea.data_ptr(), da, la.data_ptr())
        ctx.release_stream(h)
        assert hip.hipStreamDestroy(h) == 0        # (waits for the stream's work or not: the library does not care any more)
        ctx.encode_batch_device(s2.cuda_stream, 1, B.data_ptr(), eb.data_ptr(), db, lb.data_ptr())
    torch.cuda.synchronize()
    ha, hb = ea.cpu().numpy(), eb.cpu().numpy()
    assert b"".join(O.decode(ha[i * stride:i * stride + l].tobytes()) for i, l in enumerate(la.cpu().tolist())) == a.tobytes()
    assert b"".join(O.decode(hb[i * stride:i * stride + l].tobytes()) for i, l in enumerate(lb.cpu().tolist())) == b_.tobytes()


def test_host_batch_pinned_destinations(ctx):
    # Host-pointer batch calls with page-locked buffers: the kernels write the caller's buffers themselves (no copy-out
    # stage).  Every kind of block goes through it — tokens, stored, tiny, empty, and (decode) streams of the
    # reference's algorithm, which take the general path — at odd offsets, with guard bytes between the blocks.
    import ctypes as C
    import torch
    from minlz_amd import _lib
    L = _lib.lib()
    vp, sz = C.c_void_p, C.c_size_t
    parts = [synth.text_like(8 << 20, 51), synth.enwik_like(5_000_011, 52), synth.random_bytes(1 << 20, seed=3), synth.text_like(20, 5),
             np.zeros(0, dtype=np.uint8), synth.json_like(3 << 20, 53), synth.pattern("zeros", 70001)]
    n = len(parts)
    for level in (mz.LevelFastest, mz.LevelBalanced):
        src = torch.zeros(sum(p.size for p in parts) + 64 * n, dtype=torch.uint8, pin_memory=True)
        soff, cur = [], 3
        for p in parts:
            soff.append(cur); src.numpy()[cur:cur + p.size] = p; cur += p.size + 5
        caps = [mz.MaxEncodedLen(p.size) for p in parts]
        enc = torch.full((sum(caps) + 64 * n,), 0xA5, dtype=torch.uint8, pin_memory=True)
        eoff, cur = [], 7
        for c_ in caps:
            eoff.append(cur); cur += c_ + 9
        sp = (vp * n)(*[src.data_ptr() + o for o in soff]); sl = (sz * n)(*[p.size for p in parts])
        ep = (vp * n)(*[enc.data_ptr() + o for o in eoff]); ec = (sz * n)(*caps)
        ol = (C.c_int64 * n)()
        assert L.mlz_encode_batch(ctx.handle, level, n, sp, sl, ep, ec, ol) == 0
        eh = enc.numpy()
        encs = []
        for p, o, c_, l in zip(parts, eoff, caps, ol):
            assert 0 < l <= c_
            e = eh[o:o + l].tobytes()
            assert O.decode(e) == p.tobytes()
            encs.append(e)
        keep = np.ones(eh.size, dtype=bool)
        for o, c_ in zip(eoff, caps):
            keep[o:o + c_] = False
        assert (eh[keep] == 0xA5).all()                      # nothing outside the blocks' ranges was written
        assert encs == mz.encode_batch([p.tobytes() for p in parts], level, ctx)   # same bytes as the pageable path
    # decode: this library's blocks and the reference algorithm's (general path), into pinned memory
    blobs = encs + [O.encode(parts[0], 1), O.encode(parts[5], 2)]
    outs = parts + [parts[0], parts[5]]
    m = len(blobs)
    cbuf = torch.zeros(sum(len(b) for b in blobs) + 16 * m, dtype=torch.uint8, pin_memory=True)
    coff, cur = [], 1
    for b in blobs:
        coff.append(cur); cbuf.numpy()[cur:cur + len(b)] = np.frombuffer(b, dtype=np.uint8); cur += len(b) + 3
    dec = torch.full((sum(p.size for p in outs) + 64 * m,), 0x5A, dtype=torch.uint8, pin_memory=True)
    doff, cur = [], 5
    for p in outs:
        doff.append(cur); cur += p.size + 11
    cp = (vp * m)(*[cbuf.data_ptr() + o for o in coff]); cl = (sz * m)(*[len(b) for b in blobs])
    dp = (vp * m)(*[dec.data_ptr() + o for o in doff]); dc = (sz * m)(*[p.size for p in outs])
    dl = (C.c_int64 * m)()
    assert L.mlz_decode_batch(ctx.handle, m, cp, cl, dp, dc, dl) == 0
    dh = dec.numpy()
    keep = np.ones(dh.size, dtype=bool)
    for p, o, l in zip(outs, doff, dl):
        assert l == p.size
        assert dh[o:o + l].tobytes() == p.tobytes()
        keep[o:o + p.size] = False
    assert (dh[keep] == 0x5A).all()
    assert ctx.general_blocks() == 5      # the two reference-algorithm blocks, and this library's three LevelBalanced blocks of more than two tiles (no tile levels: MLZ_OPT_L2_FREE)


@pytest.mark.parametrize("level", [-1, 1, 2])
def test_fused_serializer_writes_the_same_bytes(level):
    # Round 6: the wave that matched a piece serializes it too (serialize_piece at the end of match_tiles_kernel: literals and the bytes in
    # front of a match from the tile's LDS copy, the ring in the wave's own near table — 2 KiB for wave 0 of the 12-bit class).  Option 21 = 0
    # runs the separate serializer kernel of rounds 2-5 on the same records: byte-identical blocks, every block class, ragged ends, long literal runs.
    rng = np.random.default_rng(9)
    parts = [synth.enwik_like(8 << 20, 3), synth.json_like((3 << 20) + 77, 4), synth.text_like(700_001, 5), synth.text_like(40_000, 6), synth.text_like(33, 7),
             np.concatenate([synth.text_like(300_000, 8), rng.integers(0, 256, 900_000, dtype=np.uint8), synth.json_like(500_000, 9)]),   # long literal runs inside a block
             synth.pattern("zeros", 2_000_001), synth.random_bytes(1 << 20, seed=2), synth.text_like(65_536 + 9, 10), synth.text_like(5, 11)]
    a, b_ = mz.Context(0), mz.Context(0)
    try:
        b_.set_option(21, 0)
        fused = mz.encode_batch([p.tobytes() for p in parts], level, a)
        apart = mz.encode_batch([p.tobytes() for p in parts], level, b_)
        assert fused == apart
        for blk, p in zip(fused, parts):
            assert O.decode(blk, guard=64) == p.tobytes()
    finally:
        a.close(); b_.close()


@pytest.mark.parametrize("level", [-1, 1, 2])
def test_layout_inside_the_gather_kernel_writes_the_same_blocks(level):
    # Round 6: a group whose every block has tiles and room gets its layout (piece offsets, stored-or-not, header, length) from the gather kernel itself;
    # option 24 = 0 always runs encode_layout_kernel.  Same blocks, same lengths — compressible, stored (random), tiny, ragged, one-tile and 8 MiB blocks,
    # through Encode (header) and encode_block (tokens only: 0 = incompressible).
    parts = [synth.enwik_like(8 << 20, 3), synth.random_bytes(1 << 20, seed=2), synth.text_like(15, 7), synth.text_like(16, 7), synth.json_like((3 << 20) + 77, 4),
             synth.text_like(32_768, 5), synth.text_like(32_769, 6), synth.pattern("zeros", 2_000_001), synth.text_like(5, 11), synth.random_bytes(40_000, seed=3)]
    a, b_ = mz.Context(0), mz.Context(0)
    try:
        b_.set_option(24, 0)
        raw = [p.tobytes() for p in parts]
        fa, fb = mz.encode_batch(raw, level, a), mz.encode_batch(raw, level, b_)
        assert fa == fb
        for blk, p in zip(fa, raw):
            assert O.decode(blk, guard=64) == p
        for p in raw:
            assert mz.encode_block(p, level, a) == mz.encode_block(p, level, b_)
        # a group with an empty block falls back to the separate kernel and still gives the same bytes
        mixed = raw[:3] + [b""] + raw[3:5]
        assert mz.encode_batch(mixed, level, a) == mz.encode_batch(mixed, level, b_)
    finally:
        a.close(); b_.close()
