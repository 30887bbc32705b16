"""Multi-GPU sharding of a stream's independent blocks.

One process per GPU (torch.distributed, backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).
Block i of the stream belongs to rank i mod G.  Every rank encodes only its own blocks; the single
exchange is the *gather*: an all_gather of per-block compressed sizes (so every rank knows the
stream layout = the Writer's in-order output and index offsets, writer.go:223-243, index.go:80-104)
and, when one rank has to hold the assembled stream, a gather of the variable-length payloads
padded to a common size.  No reduction, no all-to-all.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import stream as S


def owner(block_index, world):
    return block_index % world


def my_blocks(n_blocks, rank, world):
    return list(range(rank, n_blocks, world))


def cut_blocks(n, block_size):
    return [(o, min(block_size, n - o)) for o in range(0, n, block_size)]


def gather_sizes(local_sizes, n_blocks, rank, world, device="cpu"):
    """all_gather of per-block chunk sizes -> list of n_blocks sizes in stream order."""
    per = (n_blocks + world - 1) // world
    t = torch.zeros(per, dtype=torch.int64, device=device)
    if local_sizes:
        t[:len(local_sizes)] = torch.tensor(local_sizes, dtype=torch.int64, device=device)
    parts = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    sizes = [0] * n_blocks
    for r in range(world):
        for j, bi in enumerate(my_blocks(n_blocks, r, world)):
            sizes[bi] = int(parts[r][j].item())
    return sizes


def encode_stream_sharded(data, level, block_size, backend, rank, world, device="cpu", root=0):
    """Every rank passes the same `data`; returns the framed stream on `root` (None elsewhere)."""
    data = bytes(data)
    cuts = cut_blocks(len(data), block_size)
    mine = my_blocks(len(cuts), rank, world)
    blocks = [data[o:o + l] for o, l in (cuts[i] for i in mine)]
    bodies = backend.encode_blocks(blocks, level) if blocks else []
    crcs = backend.crcs(blocks) if blocks else []
    chunks = []
    for blk, body, crc in zip(blocks, bodies, crcs):
        ctype, payload = (S.CHUNK_MINLZ, body) if body is not None else (S.CHUNK_UNCOMPRESSED, blk)
        clen = 4 + len(payload)
        chunks.append(bytes([ctype, clen & 0xFF, (clen >> 8) & 0xFF, (clen >> 16) & 0xFF]) + crc.to_bytes(4, "little") + payload)
    sizes = gather_sizes([len(c) for c in chunks], len(cuts), rank, world, device)
    # payload gather to root: each rank sends its chunks concatenated, padded to the largest rank total
    totals = [sum(sizes[i] for i in my_blocks(len(cuts), r, world)) for r in range(world)]
    pad = max(totals + [1])
    buf = torch.zeros(pad, dtype=torch.uint8, device=device)
    flat = b"".join(chunks)
    if flat:
        buf[:len(flat)] = torch.frombuffer(bytearray(flat), dtype=torch.uint8).to(device)
    if dist.get_backend() == "nccl":  # RCCL: all_gather of the padded payloads (gather is built on it)
        parts = [torch.zeros_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf)
    else:
        parts = [torch.zeros_like(buf) for _ in range(world)] if rank == root else None
        dist.gather(buf, parts, dst=root)
    if rank != root:
        return None
    per_rank = [bytes(parts[r][:totals[r]].cpu().numpy().tobytes()) for r in range(world)]
    cursor = [0] * world
    out = bytearray()
    if cuts:
        out += S.MAGIC + bytes([(block_size - 1).bit_length() - 10])
    for bi in range(len(cuts)):
        r = owner(bi, world)
        out += per_rank[r][cursor[r]:cursor[r] + sizes[bi]]
        cursor[r] += sizes[bi]
    v = S.put_uvarint(len(data))
    out += bytes([S.CHUNK_EOF, len(v), 0, 0]) + v
    return bytes(out)


# =====================================================================================================================
# Device-resident path: one stream cut into contiguous block ranges, one range per rank, payload gather on the device
# =====================================================================================================================
#
# The Writer's concurrent path (writer.go:219-272, :501-560) compresses blocks on worker goroutines and a single
# writer goroutine emits the chunks in order.  Here the workers are the GPUs of a node: rank r keeps blocks
# [r*k, (r+1)*k) of the stream in ITS HBM, encodes them with one device batch call, frames them into a contiguous
# run of chunks on the device, and the only data exchange is the in-order gather of those runs into the root's HBM:
#   all_gather of (chunk size) per block      -> every rank knows the stream layout (8 bytes per block)
#   isend / irecv of the runs                 -> root receives each rank's run at its final offset (RCCL p2p over xGMI;
#                                                gloo in the CPU tests) — no padding, no Python bytes, no host staging
# The gather can be started asynchronously and overlapped with whatever the rank does next (bench.py decodes).

def range_of(rank, world, n_blocks):
    """Contiguous block range [b0, b1) of `rank`: the first ranks take one block more when it does not divide."""
    per, extra = divmod(n_blocks, world)
    b0 = rank * per + min(rank, extra)
    return b0, b0 + per + (1 if rank < extra else 0)


class OracleTensorCodec:
    """CPU tensors through the oracle (tests only: the gloo run of the sharding logic)."""

    def encode(self, src, block_lens, level):
        import oracle as O
        stride = max(block_lens + [0]) + 16
        enc = torch.zeros(len(block_lens) * stride, dtype=torch.uint8)
        lens, crcs, o = [], [], 0
        a = src.numpy()
        for i, l in enumerate(block_lens):
            e = np.frombuffer(O.encode(a[o:o + l].tobytes(), level), dtype=np.uint8)
            enc[i * stride:i * stride + e.size] = torch.from_numpy(e.copy())
            lens.append(e.size); crcs.append(O.crc(a[o:o + l].tobytes())); o += l
        return enc, stride, torch.tensor(lens, dtype=torch.int64), torch.tensor(crcs, dtype=torch.int64)


class HipTensorCodec:
    """CUDA tensors through the C ABI's device-resident batch calls (the product path)."""

    def __init__(self, ctx):
        self.ctx = ctx

    def encode(self, src, block_lens, level):
        from ._lib import BlockDesc
        n = len(block_lens)
        stride = (max(block_lens + [0]) + 2 + 255) & ~255
        dev = src.device
        enc = torch.empty(max(n * stride, 1), dtype=torch.uint8, device=dev)
        lens = torch.zeros(max(n, 1), dtype=torch.int64, device=dev)
        crc32 = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
        if n:
            offs = np.concatenate([[0], np.cumsum(block_lens)[:-1]]).tolist()
            desc = (BlockDesc * n)(*[BlockDesc(int(offs[i]), int(block_lens[i]), i * stride, stride) for i in range(n)])
            st = torch.cuda.current_stream(dev).cuda_stream
            self.ctx.encode_batch_device(st, level, src.data_ptr(), enc.data_ptr(), desc, lens.data_ptr())
            self.ctx.crc_batch_device(st, src.data_ptr(), desc, crc32.data_ptr())
        return enc, stride, lens[:n], (crc32[:n].to(torch.int64) & 0xFFFFFFFF)


def frame_run(codec, src, block_lens, level):
    """Encode this rank's blocks and frame them into one contiguous run of stream chunks on the device.
    -> (run: uint8 tensor on src's device, chunk_sizes: list of ints).  Per block `[type][len24][crc32c][body]`,
    type 0x02 + `uvarint(N) tokens` or 0x01 + raw bytes when the block was stored (writer.go:876-910)."""
    enc, stride, lens, crcs = codec.encode(src, block_lens, level)
    lens_h, crcs_h = lens.cpu().tolist(), crcs.cpu().tolist()     # the one host round trip: sizes decide the layout
    sizes, plan, o_src = [], [], 0
    for i, bl in enumerate(block_lens):
        stored = lens_h[i] == bl + 2 or bl == 0
        body = bl if stored else lens_h[i] - 1                      # tokens: the block minus its leading 0x00
        clen = 4 + body
        hdr = bytes([S.CHUNK_UNCOMPRESSED if stored else S.CHUNK_MINLZ, clen & 0xFF, (clen >> 8) & 0xFF, (clen >> 16) & 0xFF]) + int(crcs_h[i]).to_bytes(4, "little")
        plan.append((hdr, stored, body, i, o_src))
        sizes.append(8 + body)
        o_src += bl
    run = torch.empty(max(sum(sizes), 1), dtype=torch.uint8, device=src.device)
    if plan:
        hdrs = torch.frombuffer(bytearray(b"".join(p[0] for p in plan)), dtype=torch.uint8).to(src.device, non_blocking=True)
        o = 0
        for k, (hdr, stored, body, i, os_) in enumerate(plan):
            run[o:o + 8] = hdrs[8 * k:8 * k + 8]
            if body:
                run[o + 8:o + 8 + body] = src[os_:os_ + body] if stored else enc[i * stride + 1:i * stride + 1 + body]
            o += 8 + body
    return run[:sum(sizes)], sizes


def gather_chunk_sizes(sizes, n_blocks, rank, world, device):
    """all_gather of the per-block chunk sizes of every rank's range -> list of n_blocks ints in stream order."""
    if world == 1:
        return list(sizes)
    per = (n_blocks + world - 1) // world
    t = torch.zeros(max(per, 1), dtype=torch.int64, device=device)
    if sizes:
        t[:len(sizes)] = torch.tensor(sizes, dtype=torch.int64, device=device)
    parts = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    host = torch.stack(parts).cpu().tolist()
    out = []
    for r in range(world):
        b0, b1 = range_of(r, world, n_blocks)
        out += host[r][:b1 - b0]
    return out


def start_gather(run, all_sizes, n_blocks, rank, world, root, head=0):
    """Posts the payload gather: root receives rank r's run at its final offset in the stream buffer (allocated here with
    `head` bytes in front for the stream header), the others send theirs.  Returns (stream buffer or None, work handles,
    payload bytes of the whole stream); wait on the handles (finish_gather) before the buffer is read."""
    totals = []
    for r in range(world):
        b0, b1 = range_of(r, world, n_blocks)
        totals.append(sum(all_sizes[b0:b1]))
    payload = sum(totals)
    ops, out = [], None
    if rank == root:
        out = torch.empty(head + payload + 16, dtype=torch.uint8, device=run.device)
        o = head
        for r in range(world):
            if r == root:
                out[o:o + totals[r]] = run[:totals[r]]
            elif totals[r]:
                ops.append(dist.P2POp(dist.irecv, out[o:o + totals[r]], r))
            o += totals[r]
    elif totals[rank]:
        ops.append(dist.P2POp(dist.isend, run[:totals[rank]], root))
    works = dist.batch_isend_irecv(ops) if ops else []
    return out, works, payload


def finish_gather(works):
    for w in works:
        w.wait()


def encode_stream_sharded_device(codec, src, total_len, block_size, level, rank, world, root=0):
    """`src`: this rank's contiguous range of the stream (uint8 tensor on its device; range_of() in whole blocks).
    Returns the framed stream as a uint8 tensor on root's device (None elsewhere)."""
    n_blocks = (total_len + block_size - 1) // block_size
    b0, b1 = range_of(rank, world, n_blocks)
    block_lens = [min(block_size, total_len - i * block_size) for i in range(b0, b1)]
    assert int(src.numel()) == sum(block_lens), (int(src.numel()), sum(block_lens))
    run, sizes = frame_run(codec, src, block_lens, level)
    all_sizes = gather_chunk_sizes(sizes, n_blocks, rank, world, src.device)
    head = 10 if n_blocks else 0
    out, works, payload = start_gather(run, all_sizes, n_blocks, rank, world, root, head)
    finish_gather(works)
    if rank != root:
        return None
    if n_blocks:
        out[:10] = torch.frombuffer(bytearray(S.MAGIC + bytes([(block_size - 1).bit_length() - 10])), dtype=torch.uint8).to(out.device)
    v = S.put_uvarint(total_len)
    tail = bytes([S.CHUNK_EOF, len(v), 0, 0]) + v
    out[head + payload:head + payload + len(tail)] = torch.frombuffer(bytearray(tail), dtype=torch.uint8).to(out.device)
    return out[:head + payload + len(tail)]
