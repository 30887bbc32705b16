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
        codec = shard.OracleTensorCodec()
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
        if rank == 0:
            q.put(results)
    finally:
        dist.destroy_process_group()


def test_sharded_stream_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res and all(res), res


def test_ownership_is_round_robin():
    from minlz_amd import shard
    assert shard.my_blocks(10, 1, 4) == [1, 5, 9]
    assert [shard.owner(i, 8) for i in range(10)] == [0, 1, 2, 3, 4, 5, 6, 7, 0, 1]
    assert shard.cut_blocks(10, 4) == [(0, 4), (4, 4), (8, 2)]
