"""N > 1 path on CPU: world_size-2 gloo run of the block sharding + size/payload gather
(minlz_amd/shard.py) with the oracle as the per-rank block backend."""
import os
import socket

import pytest
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle as O
        from minlz_amd import shard, synth
        from tests.test_stream_host import OracleBackend
        results = []
        for data, bs in ((synth.text_like(700000, 3).tobytes(), 65536), (synth.random_bytes(50000).tobytes(), 4096), (b"", 4096),
                         (synth.json_like(300000).tobytes(), 1 << 20)):
            s = shard.encode_stream_sharded(data, 1, bs, OracleBackend(), rank, world)
            if rank == 0:
                results.append(s == O.stream_encode(data, 1, bs) and O.stream_decode(s, len(data)) == data)
        # size gather alone: stream order restored from round-robin ownership
        n_blocks = 7
        mine = shard.my_blocks(n_blocks, rank, world)
        sizes = shard.gather_sizes([100 + b for b in mine], n_blocks, rank, world)
        results.append(sizes == [100 + b for b in range(n_blocks)])
        # device-resident path (CPU tensors here): contiguous block ranges, chunk runs framed per rank, payload gather with
        # isend / irecv into the root's buffer at the final offsets; byte-identical to the reference-shaped stream
        import torch
        from tests.oracle_codec import OracleTensorCodec
        codec = OracleTensorCodec()
        for data, bs in ((synth.text_like(700001, 5).tobytes(), 65536), (synth.random_bytes(300000).tobytes(), 65536), (b"", 4096),
                         (synth.text_like(5000, 6).tobytes(), 4096), (synth.json_like(2_500_000).tobytes(), 1 << 20)):
            n_blocks = (len(data) + bs - 1) // bs
            b0, b1 = shard.range_of(rank, world, n_blocks)
            lo, hi = min(b0 * bs, len(data)), min(b1 * bs, len(data))
            src = torch.frombuffer(bytearray(data[lo:hi]), dtype=torch.uint8) if hi > lo else torch.zeros(0, dtype=torch.uint8)
            out = shard.encode_stream_sharded_device(codec, src, len(data), bs, 1, rank, world)
            if rank == 0:
                sb = out.numpy().tobytes()
                results.append(sb == O.stream_encode(data, 1, bs) and O.stream_decode(sb, len(data)) == data)
        results.append(shard.range_of(0, 3, 7) == (0, 3) and shard.range_of(1, 3, 7) == (3, 5) and shard.range_of(2, 3, 7) == (5, 7))
        # Reader side (reader.go:575-992): ONE stream made by the oracle's Writer, its blocks decoded by two ranks; output left
        # sharded (each rank's range checked against the source), gathered into rank 0, and with the stream on rank 0 only
        from minlz_amd import api
        for data, bs, lvl in ((synth.text_like(700001, 5).tobytes(), 65536, 1), (synth.random_bytes(300000).tobytes(), 65536, 1), (b"", 4096, 1),
                              (synth.text_like(5000, 6).tobytes(), 4096, 2), (synth.json_like(2_500_000).tobytes(), 1 << 20, 3),
                              (synth.text_like(70000, 9).tobytes(), 65536, 1)):
            sb = O.stream_encode(data, lvl, bs, add_index=(bs == 65536))
            local, (lo, hi), total = shard.decode_stream_sharded_device(codec, sb, rank, world, "cpu")
            ok = total == len(data) and local.numpy().tobytes() == data[lo:hi]
            whole, rng_, _ = shard.decode_stream_sharded_device(codec, sb, rank, world, "cpu", gather=True)
            if rank == 0:
                ok = ok and rng_ == (0, len(data)) and whole.numpy().tobytes() == data
            w2, r2, _ = shard.decode_stream_sharded_device(codec, sb if rank == 0 else None, rank, world, "cpu", gather=True, scatter=True)
            if rank == 0:
                ok = ok and w2.numpy().tobytes() == data
            else:
                ok = ok and w2.numpy().tobytes() == data[r2[0]:r2[1]]
            results.append(bool(ok))
        # errors are raised on EVERY rank, whichever rank's block is bad: a payload byte of the last block flipped -> ErrCRC
        # (or ErrCorrupt), a truncated stream -> ErrCorrupt from the walk
        data = synth.text_like(400000, 11).tobytes()
        sb = bytearray(O.stream_encode(data, 1, 65536))
        blocks, _ = __import__("minlz_amd.stream", fromlist=["x"]).walk_chunks(bytes(sb))
        sb[blocks[-1].payload_off + blocks[-1].payload_len - 1] ^= 0x55
        for bad in (bytes(sb), bytes(sb[:len(sb) // 2])):
            try:
                shard.decode_stream_sharded_device(codec, bad, rank, world, "cpu")
                results.append(False)
            except (api.ErrCRC, api.ErrCorrupt):
                results.append(True)
        q.put((rank, results))
    finally:
        dist.destroy_process_group()


def test_sharded_stream_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] and all(got[0]), got[0]
    assert got[1] and all(got[1]), got[1]


def _worker_many(rank, world, port, q):
    """More ranks than blocks (empty ranks), ranges that do not divide, a failing block on a non-root rank — the shapes an 8-GPU node
    meets (shard.range_of; writer.go:219-272 and reader.go:575-992 hand blocks to however many workers there are)."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle as O
        from minlz_amd import api, shard, stream, synth
        from tests.oracle_codec import OracleTensorCodec
        codec = OracleTensorCodec()
        results = []
        bs = 65536
        for n_blocks, tail in ((5, 1234), (13, 65536), (1, 77), (8, 65536), (9, 1)):
            n = (n_blocks - 1) * bs + tail
            data = synth.text_like(n, 100 + n_blocks).tobytes()
            assert (len(data) + bs - 1) // bs == n_blocks
            # Writer: every rank frames its (possibly empty) range, the root assembles the stream
            b0, b1 = shard.range_of(rank, world, n_blocks)
            lo, hi = min(b0 * bs, n), min(b1 * bs, n)
            src = torch.frombuffer(bytearray(data[lo:hi]), dtype=torch.uint8) if hi > lo else torch.zeros(0, dtype=torch.uint8)
            out = shard.encode_stream_sharded_device(codec, src, n, bs, 1, rank, world)
            if rank == 0:
                sb = out.numpy().tobytes()
                results.append(sb == O.stream_encode(data, 1, bs) and O.stream_decode(sb, n) == data)
            # Reader: sharded output, gathered output, and the stream held by the root only
            sb = O.stream_encode(data, 2, bs, add_index=True)
            local, (lo2, hi2), total = shard.decode_stream_sharded_device(codec, sb, rank, world, "cpu")
            ok = total == n and (lo2, hi2) == (lo, hi) and local.numpy().tobytes() == data[lo:hi]
            whole, rng_, _ = shard.decode_stream_sharded_device(codec, sb, rank, world, "cpu", gather=True)
            if rank == 0:
                ok = ok and rng_ == (0, n) and whole.numpy().tobytes() == data
            w2, r2, _ = shard.decode_stream_sharded_device(codec, sb if rank == 0 else None, rank, world, "cpu", gather=True, scatter=True)
            ok = ok and w2.numpy().tobytes() == (data if rank == 0 else data[r2[0]:r2[1]])
            results.append(bool(ok))
        # the failing block belongs to a rank that is not the root (13 blocks on 8 ranks: block 7 is rank 4's): every rank raises
        n = 12 * bs + 99
        data = synth.text_like(n, 321).tobytes()
        sb = bytearray(O.stream_encode(data, 1, bs))
        blocks, _ = stream.walk_chunks(bytes(sb))
        owner = next(r for r in range(world) if shard.range_of(r, world, len(blocks))[0] <= 7 < shard.range_of(r, world, len(blocks))[1])
        results.append(owner not in (0,))
        sb[blocks[7].payload_off + blocks[7].payload_len // 2] ^= 0x21
        for kw in ({}, {"gather": True}):
            try:
                shard.decode_stream_sharded_device(codec, bytes(sb), rank, world, "cpu", **kw)
                results.append(False)
            except (api.ErrCRC, api.ErrCorrupt):
                results.append(True)
        q.put((rank, results))
    finally:
        dist.destroy_process_group()


def test_sharded_stream_eight_ranks_uneven_and_empty():
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_many, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        assert got[r] and all(got[r]), (r, got[r])


def test_ranges_cover_every_block_once():
    from minlz_amd import shard
    for world in (1, 2, 3, 8):
        for n in (0, 1, 5, 8, 13, 64):
            cuts = [shard.range_of(r, world, n) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            assert max(b - a for a, b in cuts) - min(b - a for a, b in cuts) <= 1


def test_ownership_is_round_robin():
    from minlz_amd import shard
    assert shard.my_blocks(10, 1, 4) == [1, 5, 9]
    assert [shard.owner(i, 8) for i in range(10)] == [0, 1, 2, 3, 4, 5, 6, 7, 0, 1]
    assert shard.cut_blocks(10, 4) == [(0, 4), (4, 4), (8, 2)]
