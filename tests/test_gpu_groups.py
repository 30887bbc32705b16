"""Internal groups of a device batch (MLZ_OPT_DEVICE_GROUP, include/minlz_hip.h): a batch cut into groups that share one
workspace must produce exactly what the same batch produces in one piece — encode bytes, decode bytes, per-block results and
error verdicts — and the workspace must be bounded by the group.  The reference has no such notion: its Writer/Reader run a
goroutine per block (writer.go:501-560, reader.go:830-859); this is the device batch's equivalent of their bounded concurrency."""
import numpy as np
import pytest

import minlz_amd as mz
import oracle as O
from minlz_amd import synth
from minlz_amd._lib import BlockDesc

pytestmark = pytest.mark.gpu

OPT_DEVICE_GROUP = 17


def _encode_batch(ctx, parts, level):
    import torch
    dev = torch.device("cuda", 0)
    offs, cur = [], 0
    for p in parts:
        offs.append(cur)
        cur += p.size + 16
    host = np.zeros(cur + 64, dtype=np.uint8)
    for o, p in zip(offs, parts):
        host[o:o + p.size] = p
    src = torch.from_numpy(host).to(dev)
    caps = [mz.MaxEncodedLen(p.size) for p in parts]
    eoffs, ecur = [], 0
    for c in caps:
        eoffs.append(ecur)
        ecur += c + 32
    enc = torch.zeros(ecur + 64, dtype=torch.uint8, device=dev)
    elen = torch.zeros(len(parts), dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    ctx.encode_batch_device(st, level, src.data_ptr(), enc.data_ptr(), [BlockDesc(o, p.size, eo, c) for o, p, eo, c in zip(offs, parts, eoffs, caps)], elen.data_ptr())
    torch.cuda.synchronize()
    lens = elen.cpu().tolist()
    eh = enc.cpu().numpy()
    return [eh[eo:eo + l].tobytes() for eo, l in zip(eoffs, lens)]


def _decode_batch(ctx, blocks, sizes):
    import torch
    dev = torch.device("cuda", 0)
    offs, cur = [], 0
    for b in blocks:
        offs.append(cur)
        cur += len(b) + 16
    host = np.zeros(cur + 64, dtype=np.uint8)
    for o, b in zip(offs, blocks):
        host[o:o + len(b)] = np.frombuffer(b, dtype=np.uint8)
    src = torch.from_numpy(host).to(dev)
    doffs, dcur = [], 0
    for n in sizes:
        doffs.append(dcur)
        dcur += n + 48
    dst = torch.full((dcur + 64,), 0xA5, dtype=torch.uint8, device=dev)
    dlen = torch.zeros(len(blocks), dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    ctx.decode_batch_device(st, src.data_ptr(), dst.data_ptr(), [BlockDesc(o, len(b), do, n) for o, b, do, n in zip(offs, blocks, doffs, sizes)], dlen.data_ptr())
    torch.cuda.synchronize()
    dh = dst.cpu().numpy()
    guard = np.ones(dh.size, dtype=bool)
    for do, n in zip(doffs, sizes):
        guard[do:do + n] = False
    assert (dh[guard] == 0xA5).all(), "bytes outside the blocks' output ranges were written"
    return dlen.cpu().tolist(), [dh[do:do + n].tobytes() for do, n in zip(doffs, sizes)]


def _parts():
    rng = np.random.default_rng(11)
    sizes = [3 << 20, 700_001, 40_000, 1 << 20, 33, (1 << 20) + 5, 2_500_000, 65_536, 5, 1_300_000]
    out = []
    for i, n in enumerate(sizes):
        kind = i % 3
        out.append(synth.text_like(n, seed=20 + i) if kind == 0 else synth.json_like(n, seed=30 + i) if kind == 1 else
                   np.concatenate([synth.text_like(n // 2, seed=40 + i), rng.integers(0, 256, n - n // 2, dtype=np.uint8)]))
    return out


@pytest.mark.parametrize("level", [1, 2])
def test_groups_give_the_same_bytes(level):
    parts = _parts()
    whole = mz.Context(0)
    cut = mz.Context(0)
    cut.set_option(OPT_DEVICE_GROUP, 1)       # 1 MiB: every block of more than 1 MiB is a group of its own, small ones share
    try:
        a = _encode_batch(whole, parts, level)
        b = _encode_batch(cut, parts, level)
        assert a == b, "grouped encode differs from the ungrouped one"
        for blk, p in zip(a, parts):
            assert O.decode(blk, guard=64) == p.tobytes()
        # decode: this library's blocks, the oracle's (general blocks), and a corrupt one in the middle (its verdict must stay ITS verdict)
        foreign = [O.encode(p, 1 + (i & 1)) for i, p in enumerate(parts)]
        bad = bytearray(a[3]); bad[len(bad) // 2] ^= 0x5A; bad[len(bad) // 2 + 1] ^= 0xFF
        blocks = a + foreign + [bytes(bad)]
        sizes = [p.size for p in parts] * 2 + [parts[3].size]
        la, da = _decode_batch(whole, blocks, sizes)
        lb, db = _decode_batch(cut, blocks, sizes)
        assert la == lb
        want = [p.tobytes() for p in parts] * 2
        for i, w in enumerate(want):
            assert la[i] == len(w) and da[i] == w and db[i] == w, i
        # the damaged block: the oracle's verdict (corrupt, or — if the damage happens to leave a valid stream — its output)
        try:
            ob = O.decode(bytes(bad))
            assert la[-1] == lb[-1] == len(ob) and da[-1][:len(ob)] == ob == db[-1][:len(ob)]
        except O.OracleError as e:
            assert la[-1] == lb[-1] == -e.code
        # the cut context never saw more than a group: its workspace is a fraction of the whole batch's
        we, wd = whole.workspace_bytes()
        ce, cd = cut.workspace_bytes()
        assert ce < we and cd < wd
    finally:
        whole.close(); cut.close()


def test_timers_sum_over_the_groups():
    parts = [synth.text_like(1 << 20, seed=60 + i) for i in range(6)]
    ctx = mz.Context(0)
    try:
        ctx.set_option(OPT_DEVICE_GROUP, 2)   # three groups of two blocks
        ctx.set_option(mz.api.OPT_TIMING, 1)
        _encode_batch(ctx, parts, 1)
        t3 = ctx.timers()
        ctx.set_option(OPT_DEVICE_GROUP, 512)
        _encode_batch(ctx, parts, 1)
        t1 = ctx.timers()
        assert t3["enc_tiles"] > 0 and t1["enc_tiles"] > 0
        # three launches of a third of the work each take at least as long as one launch of all of it, and not absurdly longer
        assert 0.8 * t1["enc_tiles"] < t3["enc_tiles"] < 6 * t1["enc_tiles"]
    finally:
        ctx.close()


@pytest.mark.parametrize("level", [1, 2])
@pytest.mark.parametrize("where", ["last", "middle"])
def test_oversize_block_in_a_device_batch(level, where):
    """A block above MaxBlockSize in a device batch is ITS OWN failure (-ErrTooLarge, encode.go:74-80) and nothing else's: it has no tiles,
    and no kernel may read descriptors or units on its behalf (round-5 advisor finding: LevelBalanced's far_slice_kernel did)."""
    import torch
    dev = torch.device("cuda", 0)
    big = (8 << 20) + 100
    parts = [synth.json_like(1_500_000, seed=71), synth.text_like(2 << 20, seed=72)]
    over = synth.text_like(big, seed=73)
    parts.insert(1 if where == "middle" else 2, over)
    offs, cur = [], 0
    for p in parts:
        offs.append(cur)
        cur += p.size + 16
    host = np.zeros(cur + 64, dtype=np.uint8)
    for o, p in zip(offs, parts):
        host[o:o + p.size] = p
    src = torch.from_numpy(host).to(dev)
    caps = [p.size + 2 for p in parts]
    eoffs, ecur = [], 0
    for c in caps:
        eoffs.append(ecur)
        ecur += c + 32
    enc = torch.zeros(ecur + 64, dtype=torch.uint8, device=dev)
    elen = torch.zeros(len(parts), dtype=torch.int64, device=dev)
    ctx = mz.Context(0)
    try:
        st = torch.cuda.current_stream(dev).cuda_stream
        ctx.encode_batch_device(st, level, src.data_ptr(), enc.data_ptr(), [BlockDesc(o, p.size, eo, c) for o, p, eo, c in zip(offs, parts, eoffs, caps)], elen.data_ptr())
        torch.cuda.synchronize()
        lens = elen.cpu().tolist()
        eh = enc.cpu().numpy()
        for i, p in enumerate(parts):
            if p.size > (8 << 20):
                assert lens[i] == -2, lens       # -MLZ_ERR_TOO_LARGE
            else:
                assert 0 < lens[i] < p.size
                assert O.decode(eh[eoffs[i]:eoffs[i] + lens[i]].tobytes(), guard=64) == p.tobytes()
    finally:
        ctx.close()
