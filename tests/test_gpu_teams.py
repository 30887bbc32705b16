"""Round 5: general blocks settled by TEAMS of workgroups (mlz_decode_general.hip.inc), and the encoder side of it.

The reference has no such thing — decode.go:178-622 is one sequential loop per block, encode_l2.go takes the nearest candidate — so
parity here means: whatever the team size, the output is bit-exact (oracle-encoded and GPU-encoded streams alike), verdicts are the
oracle's, the team size follows from what the block's copies actually do (measured by the index pass, not taken from the encoder's
word), and LevelBalanced's bytes with the far-source gap are the CPU model's (tests/test_gpu_model.py) and decode through the oracle."""
import numpy as np
import pytest

import minlz_amd as mz
import oracle as O
from minlz_amd import synth

pytestmark = pytest.mark.gpu

OPT_L2_GAP, OPT_FAR_SLICES, OPT_GEN_SPIN, OPT_GEN_PACKED = 19, 18, 9, 13


def _blocks(n_blocks=3, size=(2 << 20) + 12345):
    return [np.ascontiguousarray(synth.json_like(size, seed=50 + i) if i & 1 else synth.text_like(size, seed=60 + i)) for i in range(n_blocks)]


@pytest.mark.parametrize("gap,team", [(1, 1), (2, 2), (3, 2), (4, 4), (8, 4)])
def test_team_size_follows_the_blocks_dependency_distance(gap, team):
    ctx = mz.Context(0)
    try:
        ctx.set_option(OPT_L2_GAP, gap)
        data = _blocks()
        encs = mz.encode_batch(data, mz.LevelBalanced, ctx)
        for e, d in zip(encs, data):
            assert O.decode(e, guard=64) == d.tobytes()                     # the reference's decoder reads what the encoder wrote
        assert mz.decode_batch(encs, ctx) == [d.tobytes() for d in data]    # ... and so do the teams, bit for bit
        assert ctx.general_blocks() == len(data)
        assert ctx.general_team() == team
    finally:
        ctx.close()


def test_team_is_the_smallest_of_the_batch_and_one_for_reference_blocks(ctx):
    data = _blocks(4)
    ctx.set_option(OPT_L2_GAP, 4)
    own = mz.encode_batch(data[:2], mz.LevelBalanced, ctx)
    ref = [O.encode(d, 2) for d in data[2:]]                               # encode_l2.go's restatement: sources anywhere
    assert mz.decode_batch(own, ctx) == [d.tobytes() for d in data[:2]] and ctx.general_team() == 4
    assert mz.decode_batch(ref, ctx) == [d.tobytes() for d in data[2:]] and ctx.general_team() == 1
    mixed = [own[0], ref[0], own[1], ref[1]]
    want = [data[0].tobytes(), data[2].tobytes(), data[1].tobytes(), data[3].tobytes()]
    assert mz.decode_batch(mixed, ctx) == want and ctx.general_team() == 1
    # a leveled block beside them does not count (it is not a general block)
    lvl = mz.Encode(data[0], mz.LevelFastest, ctx)
    assert mz.decode_batch([lvl] + own, ctx) == [data[0].tobytes()] + [d.tobytes() for d in data[:2]]
    assert ctx.general_blocks() == 2 and ctx.general_team() == 4


def test_many_blocks_take_one_workgroup_each(ctx):
    # more general blocks than a quarter of the settling workgroups: a workgroup per block (it settles more tiles per microsecond)
    data = [np.ascontiguousarray(synth.text_like(1_200_000, seed=70 + i)) for i in range(40)]   # (37 tiles each: far sources four tiles back break every level pattern)
    encs = mz.encode_batch(data, mz.LevelBalanced, ctx)
    assert mz.decode_batch(encs, ctx) == [d.tobytes() for d in data]
    assert ctx.general_blocks() == 40 and ctx.general_team() == 1


def test_teams_with_the_packed_pool_and_a_bounded_wait():
    ctx = mz.Context(0)
    try:
        data = _blocks(2, 1 << 20)
        encs = mz.encode_batch(data, mz.LevelBalanced, ctx)
        ctx.set_option(OPT_GEN_PACKED, 1)          # every tile through the byte-packed pool (the fallback of tiles whose slots do not fit)
        assert mz.decode_batch(encs, ctx) == [d.tobytes() for d in data] and ctx.general_team() == 4
        ctx.set_option(OPT_GEN_PACKED, 0)
        ctx.set_option(OPT_GEN_SPIN, 1)            # patience of one poll: a member cannot find its flags up -> a device failure, never "corrupt", never a hang
        with pytest.raises(mz.ErrHIP):
            mz.decode_batch(encs, ctx)
        ctx.set_option(OPT_GEN_SPIN, 1 << 24)
        assert mz.decode_batch(encs, ctx) == [d.tobytes() for d in data]
    finally:
        ctx.close()


def test_corrupt_block_in_a_team_gets_the_oracles_verdict(ctx):
    data = _blocks(3, 1 << 20)
    encs = [bytearray(e) for e in mz.encode_batch(data, mz.LevelBalanced, ctx)]
    rng = np.random.default_rng(3)
    checked = 0
    for trial in range(12):
        bad = bytearray(encs[1])
        for _ in range(3):
            bad[int(rng.integers(8, len(bad)))] ^= int(rng.integers(1, 256))
        try:
            want = O.decode(bytes(bad))
        except O.OracleError as e:
            want = e
        try:
            got = mz.decode_batch([bytes(encs[0]), bytes(bad), bytes(encs[2])], ctx)
            assert not isinstance(want, O.OracleError) and got == [data[0].tobytes(), want, data[2].tobytes()]
        except mz.ErrCorrupt:
            assert isinstance(want, O.OracleError) and want.code == 1
            checked += 1
    assert checked > 0


def test_far_tables_by_sorting_equal_the_slice_scans(ctx):
    # LevelBalanced's far tables: far_bin_kernel + far_slice_kernel (round 5) against far_build_kernel (a workgroup per slice scanning every
    # window, round 4; debug option 18): the encoder's output must not change by a byte
    data = [np.ascontiguousarray(synth.json_like((5 << 20) + 333, seed=5)), np.ascontiguousarray(synth.text_like((1 << 20) + 1, seed=6)),
            np.ascontiguousarray(synth.enwik_like(3 << 20, seed=7)), np.zeros(1 << 20, dtype=np.uint8),
            np.ascontiguousarray(synth.text_like(300_000, seed=8))]        # (the last one: the small-block class beside the big ones)
    new = mz.encode_batch(data, mz.LevelBalanced, ctx)
    ctx.set_option(OPT_FAR_SLICES, 1)
    try:
        old = mz.encode_batch(data, mz.LevelBalanced, ctx)
    finally:
        ctx.set_option(OPT_FAR_SLICES, 0)
    assert new == old
    for e, d in zip(new, data):
        assert O.decode(e) == d.tobytes()
