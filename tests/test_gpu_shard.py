"""The sharded Writer / Reader paths of minlz_amd/shard.py on the GPU: HipTensorCodec (the C ABI's device-resident batch
calls), frame_run, encode_stream_sharded_device and decode_stream_sharded_device on CUDA tensors — what an 8-GPU run
executes per rank.  World 1 in-process; world 2 as two processes with one HIP context each on cuda:0, exchanging through
gloo (host-staged: RCCL refuses two ranks on one device).  Checker: the oracle's Writer / Reader restatement."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import oracle as O
from minlz_amd import shard, stream as S, synth

pytestmark = pytest.mark.gpu


def _inputs():
    return [
        (synth.text_like(5_000_001, 5).tobytes(), 1 << 20, 1),
        (synth.enwik_like(20 << 20, 3).tobytes() + synth.random_bytes(9 << 20, 2).tobytes(), 8 << 20, 1),   # the last blocks are stored: 0x01 chunks
        (synth.json_like(3_000_000).tobytes(), 65536, 2),
        (synth.text_like(5000, 6).tobytes(), 4096, -1),
        (b"", 4096, 1),
    ]


def _check_stream(sb, data, bs):
    """Framing as the reference Writer lays it out (writer.go:876-910): stream header, one chunk per block with the CRC of
    its uncompressed bytes, EOF with the total; decodable by the reference-shaped Reader."""
    blocks, total = S.walk_chunks(sb)
    assert total == len(data) and len(blocks) == (len(data) + bs - 1) // bs
    for i, b in enumerate(blocks):
        assert b.u_off == i * bs and b.n == min(bs, len(data) - i * bs)
        assert b.crc == O.crc(data[b.u_off:b.u_off + b.n])
    if data:
        assert sb[:10] == S.MAGIC + bytes([(bs - 1).bit_length() - 10])
    assert O.stream_decode(sb, len(data)) == data


def test_writer_side_world1(ctx):
    dev = torch.device("cuda", 0)
    codec = shard.HipTensorCodec(ctx)
    for data, bs, lvl in _inputs():
        src = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev) if data else torch.zeros(0, dtype=torch.uint8, device=dev)
        n_blocks = (len(data) + bs - 1) // bs
        block_lens = [min(bs, len(data) - i * bs) for i in range(n_blocks)]
        run, sizes = shard.frame_run(codec, src, block_lens, lvl)
        assert int(run.numel()) == sum(sizes) and len(sizes) == n_blocks
        out = shard.encode_stream_sharded_device(codec, src, len(data), bs, lvl, 0, 1)
        sb = out.cpu().numpy().tobytes()
        _check_stream(sb, data, bs)
        assert sb[10 if data else 0:len(sb) - (4 + len(S.put_uvarint(len(data))))] == run.cpu().numpy().tobytes()
        if len(data) > (20 << 20):
            kinds = [b.kind for b in S.walk_chunks(sb)[0]]
            assert S.CHUNK_UNCOMPRESSED in kinds and S.CHUNK_MINLZ in kinds


def _to_compcrc(stream):
    b = bytearray(stream)
    for c in S.walk_chunks(stream)[0]:
        if c.kind == S.CHUNK_MINLZ:
            b[c.chunk_off] = S.CHUNK_MINLZ_COMPCRC
            b[c.chunk_off + 4:c.chunk_off + 8] = O.crc(bytes(b[c.payload_off + c.hdr_len:c.payload_off + c.payload_len])).to_bytes(4, "little")
    return bytes(b)


def test_reader_side_world1(ctx):
    import minlz_amd as mz
    dev = torch.device("cuda", 0)
    codec = shard.HipTensorCodec(ctx)
    for data, bs, lvl in _inputs():
        for sb in (O.stream_encode(data, max(lvl, 0), bs, add_index=len(data) > 4096),          # a stream of the reference's Writer
                   mz.stream_encode(data, lvl, bs, ctx=ctx),                                    # a stream of this library
                   _to_compcrc(O.stream_encode(data, 3 if bs <= 65536 else 1, bs))):            # 0x03 chunks
            local, rng_, total = shard.decode_stream_sharded_device(codec, sb, 0, 1, dev)
            assert total == len(data) and rng_ == (0, len(data))
            assert local.cpu().numpy().tobytes() == data
    # errors: a flipped payload byte -> ErrCRC or ErrCorrupt, a flipped CRC byte -> ErrCRC, a cut stream -> ErrCorrupt
    data = synth.text_like(2_000_000, 8).tobytes()
    sb = O.stream_encode(data, 1, 1 << 20)
    c = S.walk_chunks(sb)[0][1]
    bad = bytearray(sb); bad[c.payload_off + c.payload_len // 2] ^= 0x10
    with pytest.raises((mz.ErrCRC, mz.ErrCorrupt)):
        shard.decode_stream_sharded_device(codec, bytes(bad), 0, 1, dev)
    bad = bytearray(sb); bad[c.chunk_off + 5] ^= 0x10
    with pytest.raises(mz.ErrCRC):
        shard.decode_stream_sharded_device(codec, bytes(bad), 0, 1, dev)
    assert shard.decode_stream_sharded_device(codec, bytes(bad), 0, 1, dev, ignore_crc=True)[0].cpu().numpy().tobytes() == data
    with pytest.raises(mz.ErrCorrupt):
        shard.decode_stream_sharded_device(codec, sb[:len(sb) - 3], 0, 1, dev)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q, backend="gloo"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    di = rank if backend == "nccl" else 0          # RCCL: one device per rank; gloo: both contexts on cuda:0
    torch.cuda.set_device(di)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", di))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    res = []
    try:
        import minlz_amd as mz
        dev = torch.device("cuda", di)
        ctx = mz.Context(di)
        codec = shard.HipTensorCodec(ctx)
        for data, bs, lvl in _inputs():
            n_blocks = (len(data) + bs - 1) // bs
            b0, b1 = shard.range_of(rank, world, n_blocks)
            lo, hi = min(b0 * bs, len(data)), min(b1 * bs, len(data))
            src = torch.frombuffer(bytearray(data[lo:hi]), dtype=torch.uint8).to(dev) if hi > lo else torch.zeros(0, dtype=torch.uint8, device=dev)
            out = shard.encode_stream_sharded_device(codec, src, len(data), bs, lvl, rank, world)
            box = [out.cpu().numpy().tobytes() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            sb = box[0]
            if rank == 0:
                _check_stream(sb, data, bs)
            # ... and back: the stream this pair of ranks wrote, then the oracle Writer's stream of the same data
            for stream in (sb, O.stream_encode(data, max(lvl, 0), bs)):
                local, (ulo, uhi), total = shard.decode_stream_sharded_device(codec, stream, rank, world, dev)
                res.append(total == len(data) and local.cpu().numpy().tobytes() == data[ulo:uhi])
                whole, r2, _ = shard.decode_stream_sharded_device(codec, stream if rank == 0 else None, rank, world, dev, gather=True, scatter=True)
                res.append(whole.cpu().numpy().tobytes() == (data if rank == 0 else data[r2[0]:r2[1]]))
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu_gloo():
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got[0] and all(got[0]), got[0]
    assert got[1] and all(got[1]), got[1]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_two_ranks_two_gpus_rccl():
    """The same exchange over RCCL (device all_gather of the sizes, device-tensor batch_isend_irecv of the runs, scatter / gather of
    the Reader side): runs wherever two GPUs are visible; the builder's boxes have one."""
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker, args=(r, 2, port, q, "nccl")) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got[0] and all(got[0]), got[0]
    assert got[1] and all(got[1]), got[1]


@pytest.mark.parametrize("mode", ["blocks", "stream"])
def test_bench_one_rank_over_rccl(mode):
    """bench.py with a ONE-rank RCCL process group (MINLZ_BENCH_FORCE_DIST): communicator init, the device all_reduce / all_gather /
    broadcast_object_list / barrier calls of the N > 1 path execute on the real backend, on the one GPU a builder box has."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MINLZ_BENCH_ONE_GPU")}
    env["MINLZ_BENCH_FORCE_DIST"] = "1"
    env["MASTER_PORT"] = str(_free_port())
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-extras", "--bytes", "33554432",
           "--strong-bytes", "50331648"]
    if mode == "stream":
        cmd += ["--mode", "stream", "--workload", "json", "--level", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["ranks"] == 1 and d["config"]["backend"].startswith("rccl") and "TEST_MODE" not in d["config"]
    if mode == "blocks":   # the config-3 strong-scaling leg rides in the same line whenever a process group exists (device broadcast of the stream over RCCL)
        cs = d["config"]["config3_strong"]
        assert "failed" not in cs, cs
        assert cs["scaling"] == "strong" and cs["writer_MBps"] > 0 and cs["reader"]["decode_MBps"] > 0 and cs["backend"].startswith("rccl")


@pytest.mark.parametrize("mode", ["blocks", "stream"])
def test_bench_two_ranks_on_one_gpu(mode):
    """bench.py's N > 1 code path (every rank's legs, the size all_gather, the payload gather into rank 0, the max-over-ranks
    timing, rank 0's line) run for real with two ranks: MINLZ_BENCH_ONE_GPU puts both on cuda:0 over gloo — a test mode the
    line is labelled with, since an 8-GPU node is not available to the builder."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MINLZ_BENCH_ONE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-extras", "--bytes", "33554432",
           "--strong-bytes", str(5 * (8 << 20) + 12345)]          # five blocks and a bit over two ranks: uneven ranges, a ragged last block
    if mode == "stream":
        cmd += ["--mode", "stream", "--workload", "json", "--level", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["ranks"] == 2 and d["config"]["backend"] == "gloo" and "TEST_MODE" in d["config"]
    if mode == "stream":
        assert len(d["config"]["kernel_ms_per_rank"]) == 2 and d["config"]["reader"]["decode_MBps"] > 0
    else:
        assert d["config"]["gather"]["payload_bytes_per_step"] > 0
        cs = d["config"]["config3_strong"]
        assert "failed" not in cs, cs
        assert cs["ranks"] == 2 and len(cs["kernel_ms_per_rank"]) == 2 and len(cs["rank_ms_before_barrier"]) == 2 and cs["reader"]["decode_MBps"] > 0
        # round 6: rank 0 also runs the node through ONE process (mlz_init_devices) as a child, the other rank waiting at a barrier
        sp = d["config"]["single_process_all_devices"]
        assert "failed" not in sp, sp
        assert sp["devices"] == [0, 0] and sp["end_to_end_MBps"]["pair"] > 0 and sp["end_to_end_MBps"]["stream_pair"] > 0 and "TEST_MODE" in sp
