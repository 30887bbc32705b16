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
