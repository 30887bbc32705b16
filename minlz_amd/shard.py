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

    def decode(self, enc, blocks, out):
        """blocks: [(src_off, src_len, dst_off, dst_len)] — whole blocks `00 uvarint(N) tokens` / `00 00 raw` inside `enc`, decoded
        to out[dst_off : dst_off + dst_len].  -> int64 tensor on the device: the decoded length of each block or -MLZ_ERR_*."""
        from ._lib import BlockDesc
        n = len(blocks)
        lens = torch.zeros(max(n, 1), dtype=torch.int64, device=enc.device)
        if n:
            desc = (BlockDesc * n)(*[BlockDesc(int(a), int(b), int(c), int(d)) for a, b, c, d in blocks])
            st = torch.cuda.current_stream(enc.device).cuda_stream
            self.ctx.decode_batch_device(st, enc.data_ptr(), out.data_ptr(), desc, lens.data_ptr())
        return lens[:n]

    def crcs(self, base, spans):
        """Masked CRC32C (minlz.go:133-140) of base[off : off + len] for each (off, len) -> int64 tensor on the device."""
        from ._lib import BlockDesc
        n = len(spans)
        crc32 = torch.zeros(max(n, 1), dtype=torch.int32, device=base.device)
        if n:
            desc = (BlockDesc * n)(*[BlockDesc(int(o), int(l), 0, 0) for o, l in spans])
            st = torch.cuda.current_stream(base.device).cuda_stream
            self.ctx.crc_batch_device(st, base.data_ptr(), desc, crc32.data_ptr())
        return crc32[:n].to(torch.int64) & 0xFFFFFFFF


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
    if not plan:
        return torch.empty(0, dtype=torch.uint8, device=src.device), sizes
    # all chunk headers in ONE blocking upload (a pageable temporary must not be handed to an asynchronous copy), and the run
    # assembled by ONE concatenation of views [hdr_0, body_0, hdr_1, body_1, ...] instead of two slice copies per block
    hdrs = torch.frombuffer(bytearray(b"".join(p[0] for p in plan)), dtype=torch.uint8).to(src.device)
    parts = []
    for k, (hdr, stored, body, i, os_) in enumerate(plan):
        parts.append(hdrs[8 * k:8 * k + 8])
        if body:
            parts.append(src[os_:os_ + body] if stored else enc[i * stride + 1:i * stride + 1 + body])
    return torch.cat(parts), sizes


# ---- collectives ----
# RCCL ("nccl") moves device tensors directly.  gloo moves host tensors only: with gloo, device tensors are staged through
# the host on either side of the transfer (that is what a 1-GPU box can run: two ranks, two contexts on one device, gloo).

def _host_staged(t):
    return t.is_cuda and dist.get_backend() == "gloo"


def gather_chunk_sizes(sizes, n_blocks, rank, world, device):
    """all_gather of the per-block chunk sizes of every rank's range -> list of n_blocks ints in stream order."""
    # world == 1 is a LOCAL call unless the process group has exactly one rank too: a rank that encodes a stream of its own inside a
    # larger job (bench.py's config-3 leg on rank 0) must not enter a collective the other ranks never join.  (A one-rank process group
    # still goes through the collective: MINLZ_BENCH_FORCE_DIST runs RCCL's all_gather that way on a 1-GPU box.)
    if world == 1 and not (dist.is_available() and dist.is_initialized() and dist.get_world_size() == 1):
        return list(sizes)
    cdev = "cpu" if dist.get_backend() == "gloo" else device
    per = (n_blocks + world - 1) // world
    t = torch.zeros(max(per, 1), dtype=torch.int64, device=cdev)
    if sizes:
        t[:len(sizes)] = torch.tensor(sizes, dtype=torch.int64, device=cdev)
    parts = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    host = torch.stack(parts).cpu().tolist()
    out = []
    for r in range(world):
        b0, b1 = range_of(r, world, n_blocks)
        out += host[r][:b1 - b0]
    return out


class _Transfers:
    """Handles of posted isend / irecv operations plus what has to happen once they are done (host-staged receives are
    copied to their place on the device)."""

    def __init__(self):
        self.ops, self.works, self.after, self.keep = [], [], [], []

    def send(self, t, dst):
        if t.numel() == 0:
            return
        h = t.cpu() if _host_staged(t) else t
        self.keep.append(h)
        self.ops.append(dist.P2POp(dist.isend, h, dst))

    def recv(self, t, src):
        if t.numel() == 0:
            return
        if _host_staged(t):
            h = torch.empty(t.shape, dtype=t.dtype)
            self.after.append((t, h))
            self.ops.append(dist.P2POp(dist.irecv, h, src))
        else:
            self.ops.append(dist.P2POp(dist.irecv, t, src))

    def post(self):
        self.works = dist.batch_isend_irecv(self.ops) if self.ops else []
        return self

    def wait(self):
        for w in self.works:
            w.wait()
        for t, h in self.after:
            t.copy_(h)
        self.works, self.after, self.keep = [], [], []


def start_gather(run, all_sizes, n_blocks, rank, world, root, head=0):
    """Posts the payload gather: root receives rank r's run at its final offset in the stream buffer (allocated here with
    `head` bytes in front for the stream header), the others send theirs.  Returns (stream buffer or None, transfers,
    payload bytes of the whole stream); finish_gather(transfers) before the buffer is read."""
    totals = []
    for r in range(world):
        b0, b1 = range_of(r, world, n_blocks)
        totals.append(sum(all_sizes[b0:b1]))
    payload = sum(totals)
    tr, out = _Transfers(), None
    if rank == root:
        out = torch.empty(head + payload + 16, dtype=torch.uint8, device=run.device)
        o = head
        for r in range(world):
            if r == root:
                out[o:o + totals[r]] = run[:totals[r]]
            else:
                tr.recv(out[o:o + totals[r]], r)
            o += totals[r]
    else:
        tr.send(run[:totals[rank]], root)
    return out, tr.post(), payload


def finish_gather(transfers):
    transfers.wait()


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
    out, tr, payload = start_gather(run, all_sizes, n_blocks, rank, world, root, head)
    finish_gather(tr)
    if rank != root:
        return None
    if n_blocks:
        out[:10] = torch.frombuffer(bytearray(S.MAGIC + bytes([(block_size - 1).bit_length() - 10])), dtype=torch.uint8).to(out.device)
    v = S.put_uvarint(total_len)
    tail = bytes([S.CHUNK_EOF, len(v), 0, 0]) + v
    out[head + payload:head + payload + len(tail)] = torch.frombuffer(bytearray(tail), dtype=torch.uint8).to(out.device)
    return out[:head + payload + len(tail)]


# =====================================================================================================================
# Reader side: ONE stream decoded by all ranks (reader.go:575-992, DecodeConcurrent)
# =====================================================================================================================
#
# The reference's concurrent Reader walks the chunk headers on one goroutine, hands every block to a worker and writes the
# results in order.  Here the walk is S.walk_chunks (host, a few bytes per chunk); the blocks it finds are dealt to the ranks
# in contiguous ranges, every rank uploads only ITS span of the stream, decodes its blocks with one device batch call and
# checks their CRCs on the device.  What the walk tells every rank — the decoded size of every block, `uvarint(N)` at the
# head of each 0x02 chunk — fixes the layout of the output, so there is NO size exchange on this side:
#   all_reduce(MAX) of one status word        -> every rank reports the same error (first failing block wins by code)
#   (gather=True) isend / irecv of N-sized    -> root receives every rank's decoded range at its final offset; the default
#                 decoded ranges                 leaves the output sharded, which is what SURVEY.md 8(e) recommends
# Input: the stream as host bytes on every rank (a file or object all ranks can read), or on the root only (scatter=True:
# the block table is broadcast and each rank's span of the stream is sent to it).

def plan_decode(blocks, rank, world):
    """blocks: S.walk_chunks(...) entries in stream order -> (b0, b1, span_lo, span_hi, u_lo, u_hi): this rank's block range,
    the byte span of the stream that holds its chunks, and the range of the decoded stream it produces."""
    b0, b1 = range_of(rank, world, len(blocks))
    if b0 == b1:   # an empty range (more ranks than blocks) sits where its predecessors end: the ranks' ranges stay in order and cover the stream
        if b0 == 0:
            return b0, b1, 0, 0, 0, 0
        pv = blocks[b0 - 1]
        return b0, b1, pv.payload_off + pv.payload_len, pv.payload_off + pv.payload_len, pv.u_off + pv.n, pv.u_off + pv.n
    lo = blocks[b0].chunk_off
    hi = blocks[b1 - 1].payload_off + blocks[b1 - 1].payload_len
    return b0, b1, lo, hi, blocks[b0].u_off, blocks[b1 - 1].u_off + blocks[b1 - 1].n


def decode_span_device(codec, span, blocks, span_lo, u_lo, device, ignore_crc=False, owned=False):
    """Decodes the blocks whose chunks lie in `span` (the stream bytes [span_lo, ...): host bytes / numpy, or a uint8 tensor;
    one already on `device` is copied first unless owned=True says this call may modify it) on `device`.
    -> (decoded uint8 tensor of this range, status: 0 or an MLZ_ERR_* code)."""
    from . import api
    n_out = sum(b.n for b in blocks)
    out = torch.empty(max(n_out, 1), dtype=torch.uint8, device=device)
    if not blocks:
        return out[:0], 0
    if isinstance(span, torch.Tensor):
        # (a pinned host tensor uploads at link speed; a tensor already on the target device type is CLONED unless the caller says
        #  the buffer is this call's to modify — the CRC bytes are zeroed in place below, and a view of the caller's stream would be damaged)
        enc = span.to(device) if span.device.type != torch.device(device).type else (span if owned else span.clone())
    else:
        enc = torch.from_numpy(np.array(span, dtype=np.uint8, copy=True)).to(device)
    # The decode call wants whole blocks: `00 uvarint(N) tokens` (0x02 / 0x03 chunks: the chunk body behind one zero byte) or
    # `00 00 raw` (0x01 chunks).  The bytes in front of a chunk's payload are its CRC, which the walk has already read: they
    # are overwritten with zeros in the DEVICE copy (one index_fill), so every block is decoded in place, no repacking.
    descs, zero_at, crc_out, crc_in, want_out, want_in = [], [], [], [], [], []
    for b in blocks:
        p = b.payload_off - span_lo
        if b.kind == S.CHUNK_UNCOMPRESSED:
            zero_at += [p - 2, p - 1]
            descs.append((p - 2, b.payload_len + 2, b.u_off - u_lo, b.n))
        else:
            zero_at.append(p - 1)
            descs.append((p - 1, b.payload_len + 1, b.u_off - u_lo, b.n))
        if b.kind == S.CHUNK_MINLZ_COMPCRC:       # CRC over the token bytes (reader.go:341-344)
            crc_in.append((p + b.hdr_len, b.payload_len - b.hdr_len)); want_in.append(b.crc)
        else:
            crc_out.append((b.u_off - u_lo, b.n)); want_out.append(b.crc)
    enc.index_fill_(0, torch.tensor(zero_at, dtype=torch.int64).to(enc.device), 0)
    lens = codec.decode(enc, descs, out)
    ok = lens == torch.tensor([b.n for b in blocks], dtype=torch.int64).to(lens.device)
    if not ignore_crc:
        # (both CRC sets are queued behind the decode before the single host read-back below)
        for base, spans, want in ((out, crc_out, want_out), (enc, crc_in, want_in)):
            if spans:
                ok = torch.cat([ok, codec.crcs(base, spans) == torch.tensor(want, dtype=torch.int64).to(lens.device)])
    status = 0
    if not bool(ok.all()):
        okh, lh = ok.cpu().tolist(), lens.cpu().tolist()
        badlen = [l for l, o in zip(lh, okh[:len(lh)]) if not o]
        status = (-badlen[0] if badlen[0] < 0 else api.ErrCorrupt.code) if badlen else api.ErrCRC.code
    return out[:n_out], status


def decode_stream_sharded_device(codec, stream, rank, world, device, root=0, gather=False, scatter=False, ignore_crc=False):
    """One .mz stream decoded by all ranks.  `stream`: bytes-like, on every rank (or on `root` only with scatter=True).
    Returns (local, (u_lo, u_hi), total): this rank's decoded range as a uint8 tensor on `device` and where it sits in the
    decoded stream of `total` bytes; with gather=True the root's `local` is the WHOLE decoded stream (range (0, total)).
    Raises the reference's errors (ErrCorrupt, ErrCRC, ...) on every rank when any rank's blocks fail."""
    from . import api
    blocks = total = None
    err = 0
    if not scatter or rank == root:
        try:
            blocks, total = S.walk_chunks(stream.numpy() if isinstance(stream, torch.Tensor) else stream)
        except api.MinLZError as e:
            err = e.code
            blocks, total = [], 0
    if scatter and world > 1:
        box = [(blocks, total, err)]
        dist.broadcast_object_list(box, src=root)
        blocks, total, err = box[0]
    if err:
        api._raise(-err)
    b0, b1, lo, hi, u_lo, u_hi = plan_decode(blocks, rank, world)
    if scatter and world > 1:
        # each rank's span of the stream travels from the root's host memory to that rank (sizes are known from the table)
        # (RCCL moves device memory: the root uploads the other ranks' spans and sends them from its HBM; gloo sends host tensors)
        tr = _Transfers()
        cdev = "cpu" if dist.get_backend() == "gloo" else device
        if rank == root:
            sv = stream.numpy() if isinstance(stream, torch.Tensor) else np.frombuffer(stream, dtype=np.uint8) if not isinstance(stream, np.ndarray) else stream
            for r in range(world):
                if r != root:
                    _, _, rlo, rhi, _, _ = plan_decode(blocks, r, world)
                    tr.send(torch.from_numpy(np.array(sv[rlo:rhi], copy=True)).to(cdev), r)
            span = sv[lo:hi]
        else:
            buf = torch.empty(hi - lo, dtype=torch.uint8, device=cdev)
            tr.recv(buf, root)
        tr.post().wait()
        if rank != root:
            span = buf
    elif isinstance(stream, torch.Tensor):
        span = stream[lo:hi]          # a CPU tensor (pinned: the upload runs at link speed)
    else:
        sv = np.frombuffer(stream, dtype=np.uint8) if not isinstance(stream, np.ndarray) else stream
        span = sv[lo:hi]
    # A rank whose decode raises (a HIP failure surfaced by the codec) must still reach the all_reduce below, or the others hang:
    # the exception becomes this rank's status word and is re-raised — as the same error on every rank — behind the exchange.
    try:
        local, status = decode_span_device(codec, span, blocks[b0:b1], lo, u_lo, device, ignore_crc,
                                           owned=scatter and world > 1 and rank != root)
    except api.MinLZError as e:
        local, status = torch.empty(0, dtype=torch.uint8, device=device), (e.code or api.ErrHIP.code)
    except RuntimeError:
        local, status = torch.empty(0, dtype=torch.uint8, device=device), api.ErrHIP.code
    if world > 1:
        st = torch.tensor([status], dtype=torch.int64, device="cpu" if dist.get_backend() == "gloo" else device)
        dist.all_reduce(st, op=dist.ReduceOp.MAX)
        status = int(st.item())
    if status:
        api._raise(-status)
    if gather and world > 1:
        tr = _Transfers()
        if rank == root:
            whole = torch.empty(max(total, 1), dtype=torch.uint8, device=device)
            for r in range(world):
                _, _, _, _, rlo, rhi = plan_decode(blocks, r, world)
                if r == root:
                    whole[rlo:rhi] = local
                else:
                    tr.recv(whole[rlo:rhi], r)
            tr.post().wait()
            return whole[:total], (0, total), total
        tr.send(local, root)
        tr.post().wait()
    return local, (u_lo, u_hi), total
