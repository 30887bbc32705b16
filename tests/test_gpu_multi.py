"""Several devices behind ONE context of the C ABI (mlz_init_devices, include/minlz_hip.h): the fan-out of a stream's independent blocks
inside one process — what the reference's Writer/Reader do with goroutines (writer.go:501-560, in-order emit :219-272; reader.go:830-859)
and what a Go host, being one process, needs from the library.

On a one-GPU box the device list is {0, 0}: two per-device contexts (own streams, own workspace, own host thread) on the same GPU — every
line of the fan-out runs, only the second PCIe link is missing.  With two or more GPUs visible the same tests run over {0, 1} as well.
The bar: streams byte-identical to the one-device call's, decodable by the oracle's Reader; batches identical block by block."""
import ctypes as C
import threading

import numpy as np
import pytest

import minlz_amd as mz
import oracle as O
from minlz_amd import _lib, synth

pytestmark = pytest.mark.gpu


def _device_lists():
    import torch
    lists = [[0, 0], [0, 0, 0]]
    if torch.cuda.device_count() >= 2:
        lists.append([0, 1])
        lists.append(list(range(torch.cuda.device_count())))
    return lists


@pytest.fixture(scope="module", params=range(4))
def multi(request):
    lists = _device_lists()
    if request.param >= len(lists):
        pytest.skip("needs two GPUs")
    m = mz.Context(devices=lists[request.param])
    yield m
    m.close()


def test_init_devices_surface(ctx):
    L = _lib.lib()
    m = mz.Context(devices=[0, 0])
    try:
        assert m.device_count() == 2 and ctx.device_count() == 1
        assert m.device_name().startswith("2 x ")
        k0, k1 = L.mlz_device_ctx(m.handle, 0), L.mlz_device_ctx(m.handle, 1)
        assert k0 and k1 and k0 != k1 and not L.mlz_device_ctx(m.handle, 2)
        assert L.mlz_device_ctx(ctx.handle, 0) == ctx.handle.value and not L.mlz_device_ctx(ctx.handle, 1)
        m.set_option(17, 64)                       # an option reaches every device
        assert m.workspace_bytes() == (0, 0)       # counters sum over the devices
    finally:
        m.close()
    allc = mz.Context(devices="all")
    try:
        import torch
        assert allc.device_count() == torch.cuda.device_count()
    finally:
        allc.close()
    h = C.c_void_p()
    assert L.mlz_init_devices((C.c_int * 1)(99), 1, C.byref(h)) == -8 and not h      # no such device: -MLZ_ERR_ARG, nothing leaked


@pytest.mark.parametrize("level,bs", [(1, 1 << 20), (2, 8 << 20), (1, 64 << 10), (-1, 4 << 20)])
def test_streams_are_byte_identical_to_one_device(ctx, multi, level, bs):
    data = [synth.text_like(21_000_000, 31).tobytes() + synth.random_bytes(3_000_000, seed=4).tobytes() + synth.json_like(9 << 20, 7).tobytes(),
            synth.json_like(bs + 5, 2).tobytes(), b"", b"x", synth.text_like(bs, 9).tobytes()]
    for d in data:
        for add_index in (False, True):
            one = mz.stream_encode(d, level, bs, add_index, ctx)
            many = mz.stream_encode(d, level, bs, add_index, multi)
            assert many == one, (len(d), add_index)
            assert O.stream_decode(many, len(d)) == d                  # the reference Reader's restatement
            assert mz.stream_decode(many, ctx=multi) == d              # chunk ranges over the devices
            assert mz.stream_decode(many, ctx=ctx) == d
    # streams of the reference's algorithm (general blocks), every level, read over the devices
    d = synth.text_like(11_000_000, 5).tobytes()
    for lv in (0, 1, 2, 3):
        assert mz.stream_decode(O.stream_encode(d, lv, 1 << 20, add_index=True), ctx=multi) == d


def test_stream_through_pinned_memory(ctx, multi):
    """Page-locked source and destination: every device's kernels write the caller's buffer themselves, at the chunks' final offsets."""
    import torch
    L = _lib.lib()
    d = synth.text_like(30_000_000, 3).tobytes() + synth.random_bytes(2_500_000, seed=8).tobytes() + synth.json_like(6 << 20, 4).tobytes()
    want = mz.stream_encode(d, 1, 2 << 20, True, ctx)
    src = torch.zeros(len(d) + 64, dtype=torch.uint8, pin_memory=True)
    src.numpy()[7:7 + len(d)] = np.frombuffer(d, dtype=np.uint8)
    cap = L.mlz_stream_bound(len(d), 2 << 20, 1)
    dst = torch.full((cap + 128,), 0x5A, dtype=torch.uint8, pin_memory=True)
    r = L.mlz_stream_encode(multi.handle, 1, 2 << 20, 1, src.data_ptr() + 7, len(d), dst.data_ptr() + 33, cap)
    assert r == len(want)
    h = dst.numpy()
    assert h[33:33 + r].tobytes() == want
    assert (h[:33] == 0x5A).all() and (h[33 + cap:] == 0x5A).all()
    out = torch.full((len(d) + 128,), 0xA5, dtype=torch.uint8, pin_memory=True)
    r2 = L.mlz_stream_decode(multi.handle, 0, dst.data_ptr() + 33, r, out.data_ptr() + 17, len(d))
    assert r2 == len(d)
    oh = out.numpy()
    assert oh[17:17 + len(d)].tobytes() == d and (oh[:17] == 0xA5).all() and (oh[17 + len(d):] == 0xA5).all()


def test_stream_errors_are_the_first_in_stream_order(ctx, multi):
    d = synth.text_like(16_000_000, 6).tobytes()
    s = bytearray(mz.stream_encode(d, 1, 1 << 20, False, ctx))
    # damage in the last quarter (a later range), then also in the first (an earlier one): the verdict is the one-device call's
    for where in (len(s) - len(s) // 8, len(s) // 8):
        s[where] ^= 0x10
        errs = []
        for c in (ctx, multi):
            try:
                mz.stream_decode(bytes(s), ctx=c)
                errs.append(None)
            except mz.MinLZError as e:
                errs.append(type(e))
        assert errs[0] is not None and errs[0] == errs[1], errs
    # a truncated stream, an oversized claim
    with pytest.raises(mz.MinLZError):
        mz.stream_decode(bytes(s[:len(s) // 2]), ctx=multi)


@pytest.mark.parametrize("level", [1, 2])
def test_batches_over_the_devices(ctx, multi, level):
    rng = np.random.default_rng(5)
    sizes = [3 << 20, 700_001, 40_000, 8 << 20, 33, (1 << 20) + 5, 2_500_000, 65_536, 0, 5, 1_300_000, 6 << 20]
    parts = []
    for i, n in enumerate(sizes):
        k = i % 3
        parts.append(b"" if n == 0 else
                     (synth.text_like(n, seed=20 + i) if k == 0 else synth.json_like(n, seed=30 + i) if k == 1 else
                      np.concatenate([synth.text_like(n // 2, seed=40 + i), rng.integers(0, 256, n - n // 2, dtype=np.uint8)])).tobytes())
    one = mz.encode_batch(parts, level, ctx)
    many = mz.encode_batch(parts, level, multi)
    assert many == one
    for blk, p in zip(many, parts):
        assert O.decode(blk, guard=64) == p
    foreign = [O.encode(np.frombuffer(p, dtype=np.uint8), 1 + (i & 1)) for i, p in enumerate(parts)]
    assert mz.decode_batch(many + foreign, multi) == parts + parts
    # a damaged block is its own failure, wherever its range ran
    bad = bytearray(many[3]); bad[len(bad) // 2] ^= 0x5A; bad[len(bad) // 2 + 1] ^= 0xFF
    L = _lib.lib()
    blocks = many[:3] + [bytes(bad)] + many[4:]
    n = len(blocks)
    arrs = [np.frombuffer(b, dtype=np.uint8) for b in blocks]
    outs = [np.empty(max(len(p), 1), dtype=np.uint8) for p in parts]
    vp, sz = C.c_void_p, C.c_size_t
    srcp = (vp * n)(*[a.ctypes.data for a in arrs]); srcl = (sz * n)(*[a.size for a in arrs])
    dstp = (vp * n)(*[o.ctypes.data for o in outs]); dstc = (sz * n)(*[len(p) for p in parts])
    res = []
    for c in (ctx, multi):
        ol = (C.c_int64 * n)()
        assert L.mlz_decode_batch(c.handle, n, srcp, srcl, dstp, dstc, ol) == 0
        res.append(list(ol))
    assert res[0] == res[1]
    for i, p in enumerate(parts):
        if i != 3:
            assert res[1][i] == len(p) and outs[i][:len(p)].tobytes() == p


def test_single_block_calls_take_the_devices_in_turn(multi):
    """The reference's goroutine-per-block pattern against a several-device context: every device's combining queue serves some."""
    blocks = [synth.text_like(1 << 20, seed=100 + i).tobytes() for i in range(24)]
    out = [None] * len(blocks)

    def work(i):
        enc = mz.Encode(blocks[i], mz.LevelFastest, multi)
        out[i] = (enc, mz.Decode(enc, multi))
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(blocks))]
    [t.start() for t in th]
    [t.join() for t in th]
    for (enc, dec), b in zip(out, blocks):
        assert dec == b and O.decode(enc) == b
    L = _lib.lib()
    served = [int(L.mlz_get_counter(L.mlz_device_ctx(multi.handle, i), 1)) for i in range(multi.device_count())]
    assert all(s > 0 for s in served) and sum(served) == multi.combine_stats()[1], served
    assert mz.crc(b"abcd", multi).to_bytes(4, "little").hex() == "6810e6b6"


def test_device_resident_calls_find_their_device(multi):
    import torch
    from minlz_amd._lib import BlockDesc
    d = synth.text_like(3 << 20, 3)
    for dev in range(min(torch.cuda.device_count(), 2)):
        with torch.cuda.device(dev):
            src = torch.from_numpy(d).cuda()
            enc = torch.zeros(d.size + 64, dtype=torch.uint8, device="cuda")
            elen = torch.zeros(1, dtype=torch.int64, device="cuda")
            st = torch.cuda.current_stream().cuda_stream
            multi.encode_batch_device(st, 1, src.data_ptr(), enc.data_ptr(), [BlockDesc(0, d.size, 0, d.size + 2)], elen.data_ptr())
            torch.cuda.synchronize()
            n = int(elen.item())
            assert O.decode(enc[:n].cpu().numpy().tobytes()) == d.tobytes()
            out = torch.zeros(d.size, dtype=torch.uint8, device="cuda")
            dlen = torch.zeros(1, dtype=torch.int64, device="cuda")
            multi.decode_batch_device(st, enc.data_ptr(), out.data_ptr(), [BlockDesc(0, n, 0, d.size)], dlen.data_ptr())
            torch.cuda.synchronize()
            assert int(dlen.item()) == d.size and out.cpu().numpy().tobytes() == d.tobytes()
    # a host pointer belongs to no device of the context
    host = np.zeros(64, dtype=np.uint8)
    with pytest.raises(mz.MinLZError):
        multi.encode_batch_device(0, 1, host.ctypes.data, host.ctypes.data, [BlockDesc(0, 16, 0, 18)], host.ctypes.data)


@pytest.mark.parametrize("level,bs,add_index", [(1, 1 << 20, True), (2, 4 << 20, False), (1, 8 << 20, True)])
def test_device_resident_gather_is_the_host_streams_twin(ctx, multi, level, bs, add_index):
    """mlz_stream_encode_gather_device: ranges of one stream in the devices' HBM, the framed stream assembled on ONE device GPU to GPU — the same
    bytes as mlz_stream_encode of the concatenation (and so decodable by the reference Reader's restatement)."""
    import torch
    L = _lib.lib()
    ndev = torch.cuda.device_count()
    k = multi.device_count()
    devs = [0] * k if ndev < 2 else [i % ndev for i in range(k)]
    # whole blocks per range, the last one ragged; an incompressible stretch (stored chunks come from the raw source, not the encoder's output)
    parts = [synth.text_like(3 * bs, seed=5).tobytes(), (synth.random_bytes(bs, seed=6).tobytes() + synth.json_like(bs, seed=7).tobytes()),
             synth.json_like(2 * bs + 12345, seed=8).tobytes()][:max(2, min(3, k))]
    if len(parts) == 2:
        parts[1] = parts[1] + synth.json_like(bs // 2 + 1, seed=8).tobytes()
    whole = b"".join(parts)
    want = mz.stream_encode(whole, level, bs, add_index, ctx)
    srcs = []
    for j, p in enumerate(parts):
        with torch.cuda.device(devs[j % len(devs)]):
            srcs.append(torch.from_numpy(np.frombuffer(p, dtype=np.uint8).copy()).cuda())
    cap = L.mlz_stream_bound(len(whole), bs, 1 if add_index else 0)
    with torch.cuda.device(0):
        dst = torch.full((cap + 64,), 0x5A, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    n = multi.stream_encode_gather_device(level, bs, add_index, [t.data_ptr() for t in srcs], [t.numel() for t in srcs], dst.data_ptr(), cap)
    got = dst.cpu().numpy()
    assert n == len(want) and got[:n].tobytes() == want
    assert (got[cap:] == 0x5A).all()
    assert O.stream_decode(want, len(whole)) == whole
    # a one-device context takes the ranges one after the other: same bytes
    n1 = ctx.stream_encode_gather_device(level, bs, add_index, [t.data_ptr() for t in srcs if t.device.index == 0][:1] or [srcs[0].data_ptr()],
                                         [parts and srcs[0].numel()], dst.data_ptr(), cap)
    assert dst.cpu().numpy()[:n1].tobytes() == mz.stream_encode(parts[0], level, bs, add_index, ctx)
    # argument checks: a ragged range in the middle, a host pointer
    if len(srcs) >= 2:
        with pytest.raises(mz.MinLZError):
            multi.stream_encode_gather_device(level, bs, add_index, [srcs[0].data_ptr(), srcs[1].data_ptr()], [srcs[0].numel() - 1, srcs[1].numel()], dst.data_ptr(), cap)
    host = np.zeros(bs, dtype=np.uint8)
    with pytest.raises(mz.MinLZError):
        multi.stream_encode_gather_device(level, bs, add_index, [host.ctypes.data], [host.size], dst.data_ptr(), cap)
