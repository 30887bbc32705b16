"""Stream framing mirror of the reference's Writer / Reader (writer.go, reader.go, SPEC.md:272-431).

Per block the Writer emits `[type][len24][masked crc32c of the uncompressed bytes][body]`:
type 0x02 with body `uvarint(N) tokens` when the block compressed, 0x01 with the raw bytes
otherwise (writer.go:876-910); the stream starts with `ff 06 00 00 "MinLz" (log2(blockSize)-10)`
(writer.go:1553-1556) and ends with the EOF chunk `20 len uvarint(total)` (writer.go:1063-1074).
Blocks are handed to a *backend* in batches — one launch per batch on the GPU instead of one
goroutine per block (writer.go:501-560, reader.go:830-859).  The seek index (index.py) can be
appended after the EOF chunk (WriterAddIndex) and drives ReadSeeker; WriterPadding pads the closed stream with a
skippable chunk; Reader.Skip leaves blocks out without decoding them.  No search tables.

Backends: HipBackend (the product: everything on the device through the C ABI).  Tests inject an
oracle-based backend to exercise this host logic without a GPU.
"""
import io
import os

from . import api
from .index import Index

MAGIC = b"\xff\x06\x00\x00MinLz"
CHUNK_UNCOMPRESSED, CHUNK_MINLZ, CHUNK_MINLZ_COMPCRC, CHUNK_EOF, CHUNK_STREAM_ID = 0x01, 0x02, 0x03, 0x20, 0xFF
CHUNK_PADDING = 0xFE   # skippable: "Section 4.4 Padding" (writer.go:1165-1166)
MIN_BLOCK, MAX_BLOCK, DEFAULT_BLOCK = 4 << 10, 8 << 20, 2 << 20  # minlz.go:98-106


def put_uvarint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def uvarint(b, pos=0):
    """binary.Uvarint: (value, bytes read); n == 0 -> buffer too small, n < 0 -> overflow."""
    x = s = 0
    for i in range(pos, len(b)):
        c = b[i]
        if i - pos == 10:
            return 0, -(i - pos + 1)
        if c < 0x80:
            if i - pos == 9 and c > 1:
                return 0, -(i - pos + 1)
            return x | (c << s), i - pos + 1
        x |= (c & 0x7F) << s
        s += 7
    return 0, 0


class HipBackend:
    """Block work on the MI355X through the C ABI (no CPU fallback)."""

    def __init__(self, ctx=None):
        self.ctx = ctx or api.default_context()

    def encode_blocks(self, blocks, level):
        """-> list of `uvarint(N) tokens` bodies, or None where the block is stored raw."""
        encs = api.encode_batch(blocks, level, self.ctx)
        out = []
        for b, e in zip(blocks, encs):
            stored = len(b) == 0 or e[:2] == b"\x00\x00"
            out.append(None if stored else e[1:])
        return out

    def decode_bodies(self, bodies):
        """bodies: list of `uvarint(N) tokens` -> list of decoded blocks (raises ErrCorrupt)."""
        return api.decode_batch([b"\x00" + b for b in bodies], self.ctx)

    def crcs(self, blocks):
        return [api.crc(b, self.ctx) for b in blocks]


def _calc_skippable_frame(written, multiple):
    """calcSkippableFrame (writer.go:1135-1151): bytes to add so that written becomes a multiple; never 1..3 (a chunk header is 4)."""
    left = written % multiple
    if left == 0:
        return 0
    add = multiple - left
    while add < 4:
        add += multiple
    return add


class Writer:
    """NewWriter(w, WriterLevel(level), WriterBlockSize(bs), WriterConcurrency(n)) (writer.go:35-86)."""

    def __init__(self, w, level=api.LevelBalanced, block_size=DEFAULT_BLOCK, concurrency=16, backend=None, add_index=False,
                 padding=0, padding_src=None):
        if not (MIN_BLOCK <= block_size <= MAX_BLOCK):
            raise ValueError("minlz: block size must be 4KiB..8MiB")  # writer.go:1238-1246
        # WriterPadding(n) / WriterPaddingSrc(r) (writer.go:1248-1277): Close pads the output to a multiple of n with a
        # skippable 0xfe chunk of random bytes (padding_src: a callable n -> bytes; os.urandom by default)
        if padding < 0 or padding > MAX_BLOCK:
            raise ValueError("minlz: padding must be 1..8MiB")
        self.pad = 0 if padding == 1 else padding
        self.pad_src = padding_src or os.urandom
        if level not in (api.LevelSuperFast, api.LevelUncompressed, api.LevelFastest, api.LevelBalanced):
            raise api.ErrInvalidLevel()
        self.w, self.level, self.block_size, self.batch = w, level, block_size, max(1, concurrency)
        self.backend = backend or HipBackend()
        self.buf = bytearray()
        self.wrote_header = False
        self.written = 0          # Written(): compressed bytes so far (writer.go:1041)
        self.uncomp_written = 0
        self.closed = False
        self.add_index = add_index  # WriterAddIndex (writer.go:1194-1204)
        self.index = Index()        # always generated (WriterCreateIndex default, writer.go:41)
        self.index.reset(block_size)
        self.index_bytes = None

    def _emit(self, b):
        self.w.write(b)
        self.written += len(b)

    def _flush_blocks(self, final):
        bs = self.block_size
        nfull = len(self.buf) // bs
        take = len(self.buf) if final else nfull * bs
        if take == 0:
            return
        data = bytes(self.buf[:take])
        del self.buf[:take]
        blocks = [data[i:i + bs] for i in range(0, take, bs)]
        if not self.wrote_header:  # header goes out with the first block (writer.go:463-467)
            self.wrote_header = True
            self.index.add(0, 0)  # the header's own entry (writer.go:236-243); it makes index.add drop the first block's (10, 0)
            self._emit(MAGIC + bytes([(bs - 1).bit_length() - 10]))
        for g in range(0, len(blocks), self.batch):
            grp = blocks[g:g + self.batch]
            bodies = self.backend.encode_blocks(grp, self.level) if self.level != api.LevelUncompressed else [None] * len(grp)
            crcs = self.backend.crcs(grp)
            for blk, body, crc in zip(grp, bodies, crcs):
                if body is not None:
                    ctype, payload = CHUNK_MINLZ, body
                else:
                    ctype, payload = CHUNK_UNCOMPRESSED, blk
                clen = 4 + len(payload)
                self.index.add(self.written, self.uncomp_written)  # writer.go:945
                self._emit(bytes([ctype, clen & 0xFF, (clen >> 8) & 0xFF, (clen >> 16) & 0xFF]) + crc.to_bytes(4, "little") + payload)
                self.uncomp_written += len(blk)

    def Write(self, p):
        if self.closed:
            raise ValueError("minlz: Writer is closed")
        self.buf += p
        if len(self.buf) >= self.block_size * self.batch:
            self._flush_blocks(False)
        return len(p)

    def EncodeBuffer(self, buf):  # writer.go:441
        self.Write(buf)
        self._flush_blocks(True)

    def Flush(self):
        self._flush_blocks(True)

    def Close(self):
        if self.closed:
            return
        self._flush_blocks(True)
        v = put_uvarint(self.uncomp_written)
        self._emit(bytes([CHUNK_EOF, len(v), 0, 0]) + v)  # writer.go:1063-1074
        # the index does not record the compressed size of a padded stream (writer.go:1083-1087)
        self.index_bytes = self.index.append_to(self.uncomp_written, self.written if self.pad <= 1 else -1)
        if self.pad > 1:
            # an appended index counts as written before the padding is sized, and follows it (writer.go:1089-1121)
            total = _calc_skippable_frame(self.written + (len(self.index_bytes) if self.add_index else 0), self.pad)
            if total:
                if total >= MAX_BLOCK + 4:
                    raise ValueError("minlz: requested skippable frame >= max")   # writer.go:1162-1164
                f = total - 4
                fill = bytes(self.pad_src(f))
                if len(fill) != f:
                    raise ValueError("minlz: short read from the padding source")
                self._emit(bytes([CHUNK_PADDING, f & 0xFF, (f >> 8) & 0xFF, (f >> 16) & 0xFF]) + fill)
        if self.add_index:
            self._emit(self.index_bytes)
        self.closed = True

    def CloseIndex(self):
        """Close and return the index chunk (writer.go:1045-1049); it is part of the stream only with add_index."""
        self.Close()
        return self.index_bytes

    def Written(self):
        return self.written


class Reader:
    """NewReader(r, ReaderMaxBlockSize(n), ReaderIgnoreCRC()) (reader.go:42-125); MinLZ streams only."""

    def __init__(self, r, max_block_size=MAX_BLOCK, ignore_crc=False, batch=16, backend=None):
        self.r = r if hasattr(r, "read") else io.BytesIO(r)
        self.max_block_org = max_block_size
        self.ignore_crc = ignore_crc
        self.batch = max(1, batch)
        self.backend = backend or HipBackend()
        self.partial = False  # set by ReadSeeker: the input is a fragment cut at chunk boundaries
        self._skip = 0        # Skip(): uncompressed bytes the next WriteTo / ReadAll leaves out

    def Skip(self, n):
        """Reader.Skip (reader.go:1034-1302): the next n uncompressed bytes are not delivered.  Blocks that lie entirely inside
        the skipped range are not decoded (and their CRC is not checked); the block the range ends in is.  Skipping past the
        end of the stream fails when it is read."""
        if n < 0:
            raise ValueError("attempted negative skip")
        self._skip += n

    def _read_full(self, n, allow_eof=False):
        b = self.r.read(n)
        if len(b) == 0 and allow_eof:
            return None
        if len(b) != n:
            raise api.ErrCorrupt("unexpected EOF")  # readFull, reader.go:196-204
        return b

    def _drain(self, pending, out):
        """Decode the queued chunks in one batch and append the results in stream order."""
        if not pending:
            return
        comp = [(i, p) for i, p in enumerate(pending) if p[0] in ("c", "c3")]
        decoded = self.backend.decode_bodies([p[1] for _, p in comp]) if comp else []
        res = [None] * len(pending)
        for (i, _), d in zip(comp, decoded):
            res[i] = d
        for i, p in enumerate(pending):
            if p[0] == "u":
                res[i] = p[1]
        if not self.ignore_crc:
            crcs = self.backend.crcs([p[3] if p[0] == "c3" else d for p, d in zip(pending, res)])   # 0x03: CRC of the token bytes
            for p, c in zip(pending, crcs):
                if c != p[2]:
                    raise api.ErrCRC()
        for p, d in zip(pending, res):
            out.write(d[p[-1]:] if p[-1] else d)   # (the last field: leading bytes Skip() took)
        pending.clear()

    def WriteTo(self, w):
        """Reader.WriteTo / DecodeConcurrent (reader.go:575-992): returns bytes written."""
        max_block = self.max_block_org
        read_header = want_eof = False
        stream_out = 0
        pending = []
        while True:
            hdr = self._read_full(4, allow_eof=not want_eof or self.partial)
            if hdr is None:
                break
            ctype = hdr[0]
            clen = hdr[1] | hdr[2] << 8 | hdr[3] << 16
            if not read_header:
                if ctype == CHUNK_STREAM_ID:
                    read_header = True
                elif ctype <= 0x3F and ctype != CHUNK_EOF:
                    raise api.ErrCorrupt("no stream header")  # reader.go:273-283
            if ctype in (CHUNK_MINLZ, CHUNK_MINLZ_COMPCRC):
                if clen < 4 or clen > api.MaxEncodedLen(max_block) + 4:
                    raise api.ErrCorrupt()
                buf = self._read_full(clen)
                crc = int.from_bytes(buf[:4], "little")
                n, hl = uvarint(buf, 4)
                if hl <= 0 or n > 0xFFFFFFFF:
                    raise api.ErrCorrupt()
                if n > max_block:
                    raise api.ErrTooLarge()
                body_len = clen - 4 - hl
                if n == 0 or n < body_len:
                    raise api.ErrCorrupt()  # reader.go:327
                # type 0x03: the CRC covers the token bytes instead of the decoded ones (reader.go:341-344)
                if self._skip >= n:
                    self._skip -= n   # skipped completely: not decoded (reader.go:1132-1137)
                else:
                    pending.append(("c3" if ctype == CHUNK_MINLZ_COMPCRC else "c", buf[4:], crc, buf[4 + hl:], self._skip))
                    self._skip = 0
                stream_out += n
            elif ctype == CHUNK_UNCOMPRESSED:
                if clen < 4 or clen > api.MaxEncodedLen(max_block) + 4:
                    raise api.ErrCorrupt()
                crcb = self._read_full(4)
                n = clen - 4
                if n > max_block:
                    raise api.ErrTooLarge()
                raw = self._read_full(n)
                if self._skip >= n:
                    self._skip -= n
                else:
                    pending.append(("u", raw, int.from_bytes(crcb, "little"), None, self._skip))
                    self._skip = 0
                stream_out += n
            elif ctype == CHUNK_EOF:
                if clen > 10:
                    raise api.ErrCorrupt()
                if clen:
                    buf = self._read_full(clen)
                    want, vn = uvarint(buf)
                    if vn != clen or (want != stream_out and not self.partial):
                        raise api.ErrCorrupt("EOF length mismatch")  # reader.go:476-491
                want_eof = read_header = False
            elif ctype == CHUNK_STREAM_ID:
                if clen != 6:
                    raise api.ErrCorrupt()
                body = self._read_full(6)
                if body[:5] != b"MinLz":
                    raise api.ErrUnsupported("not a MinLZ stream")
                if body[5] & 0xC0:
                    raise api.ErrCorrupt()
                lg = (body[5] & 15) + 10
                if lg > 23:
                    raise api.ErrCorrupt()
                max_block = 1 << lg
                if max_block > self.max_block_org:
                    raise api.ErrTooLarge()
                self._drain(pending, w)
                stream_out = 0
                want_eof = True
            elif ctype == 0x00:
                raise api.ErrUnsupported("legacy S2/Snappy chunk")
            elif ctype <= 0x3F:
                raise api.ErrUnsupported("reserved unskippable chunk")  # reader.go:530-536
            else:
                self._read_full(clen)  # skippable
            if len(pending) >= self.batch:
                self._drain(pending, w)
        self._drain(pending, w)
        if self._skip:
            raise api.ErrCorrupt("unexpected EOF")   # io.ErrUnexpectedEOF (reader.go:1060-1064)
        return None

    DecodeConcurrent = WriteTo

    def ReadAll(self):
        out = io.BytesIO()
        self.WriteTo(out)
        return out.getvalue()


class Chunk:
    """One data block found by walk_chunks: where its chunk and payload sit in the stream, what it decodes to."""
    __slots__ = ("kind", "chunk_off", "payload_off", "payload_len", "hdr_len", "n", "crc", "u_off")

    def __init__(self, kind, chunk_off, payload_off, payload_len, hdr_len, n, crc, u_off):
        self.kind, self.chunk_off, self.payload_off, self.payload_len = kind, chunk_off, payload_off, payload_len
        self.hdr_len, self.n, self.crc, self.u_off = hdr_len, n, crc, u_off


def walk_chunks(buf, max_block_size=MAX_BLOCK):
    """The chunk walk of Reader.Read / DecodeConcurrent (reader.go:248-543, :575-700) over a whole stream held in memory,
    WITHOUT decoding: -> (list of Chunk in stream order, total decoded bytes).  Same checks and errors as Reader.WriteTo
    (stream identifier, chunk length limits, uvarint(N) of 0x02 / 0x03 chunks, EOF length), so a sharded Reader can deal the
    blocks to its workers knowing every block's output offset.  payload = the chunk body behind the CRC: `uvarint(N) tokens`
    (hdr_len = the uvarint's bytes) or the raw bytes of a 0x01 chunk."""
    mv = memoryview(buf).cast("B") if not isinstance(buf, memoryview) else buf
    size = len(mv)
    max_block = max_block_size
    read_header = want_eof = False
    blocks, pos, stream_out, total = [], 0, 0, 0
    while True:
        if pos == size:
            if want_eof:
                raise api.ErrCorrupt("unexpected EOF")
            break
        if pos + 4 > size:
            raise api.ErrCorrupt("unexpected EOF")
        ctype = mv[pos]
        clen = mv[pos + 1] | mv[pos + 2] << 8 | mv[pos + 3] << 16
        chunk_off = pos
        pos += 4
        if not read_header:
            if ctype == CHUNK_STREAM_ID:
                read_header = True
            elif ctype <= 0x3F and ctype != CHUNK_EOF:
                raise api.ErrCorrupt("no stream header")
        # (availability is checked where the Reader would read: after the checks that need only the header)
        short = pos + clen > size
        if ctype in (CHUNK_MINLZ, CHUNK_MINLZ_COMPCRC):
            if clen < 4 or clen > api.MaxEncodedLen(max_block) + 4 or short:
                raise api.ErrCorrupt()
            crc = int.from_bytes(mv[pos:pos + 4], "little")
            n, hl = uvarint(mv[pos + 4:pos + min(clen, 16)])
            if hl <= 0 or n > 0xFFFFFFFF:
                raise api.ErrCorrupt()
            if n > max_block:
                raise api.ErrTooLarge()
            if n == 0 or n < clen - 4 - hl:
                raise api.ErrCorrupt()
            blocks.append(Chunk(ctype, chunk_off, pos + 4, clen - 4, hl, n, crc, total))
            stream_out += n; total += n
        elif ctype == CHUNK_UNCOMPRESSED:
            if clen < 4 or clen > api.MaxEncodedLen(max_block) + 4 or pos + 4 > size:
                raise api.ErrCorrupt()
            n = clen - 4
            if n > max_block:
                raise api.ErrTooLarge()
            if short:
                raise api.ErrCorrupt("unexpected EOF")
            blocks.append(Chunk(ctype, chunk_off, pos + 4, n, 0, n, int.from_bytes(mv[pos:pos + 4], "little"), total))
            stream_out += n; total += n
        elif ctype == CHUNK_EOF:
            if clen > 10 or short:
                raise api.ErrCorrupt()
            if clen:
                want, vn = uvarint(mv[pos:pos + clen])
                if vn != clen or want != stream_out:
                    raise api.ErrCorrupt("EOF length mismatch")
            want_eof = read_header = False
        elif ctype == CHUNK_STREAM_ID:
            if clen != 6 or short:
                raise api.ErrCorrupt()
            if bytes(mv[pos:pos + 5]) != b"MinLz":
                raise api.ErrUnsupported("not a MinLZ stream")
            if mv[pos + 5] & 0xC0:
                raise api.ErrCorrupt()
            lg = (mv[pos + 5] & 15) + 10
            if lg > 23:
                raise api.ErrCorrupt()
            max_block = 1 << lg
            if max_block > max_block_size:
                raise api.ErrTooLarge()
            stream_out = 0
            want_eof = True
        elif ctype == 0x00:
            raise api.ErrUnsupported("legacy S2/Snappy chunk")
        elif ctype <= 0x3F:
            raise api.ErrUnsupported("reserved unskippable chunk")
        elif short:
            raise api.ErrCorrupt("unexpected EOF")   # skippable chunk cut short
        pos += clen
    return blocks, total


class ReadSeeker:
    """Reader.ReadSeeker(index) (reader.go:1304-1487): random access into a stream held in memory.

    `index` = index chunk bytes (Writer.CloseIndex()), or None to load it from the end of the stream
    (Index.LoadStream).  Seek positions by uncompressed offset: the index gives the last indexed
    block at or before it, decoding starts there and the bytes in front are discarded."""

    def __init__(self, stream, index=None, backend=None, **reader_kw):
        self.stream = bytes(stream)
        self.backend = backend
        self.reader_kw = reader_kw
        self.idx = Index()
        if index is None:
            self.idx.load_stream(self.stream)
        else:
            self.idx.load(index)
        self.pos = 0
        self._cache = (None, b"")  # (uncompressed offset of the cached span, its bytes)

    def Index(self):
        return self.idx

    def Seek(self, offset, whence=0):
        if whence == 0:
            absolute = offset
        elif whence == 1:
            absolute = self.pos + offset
        elif whence == 2:
            absolute = self.idx.total_uncompressed + offset
        else:
            raise api.ErrUnsupported()
        if absolute < 0:
            raise ValueError("seek before start of file")
        self.idx.find(absolute)  # validates the range (io.ErrUnexpectedEOF beyond the end)
        self.pos = absolute
        return absolute

    def _span(self, c_off, u_off, want_end):
        """Decode from the indexed chunk at c_off until the output covers want_end."""
        cu, cached = self._cache
        if cu == u_off and u_off + len(cached) >= want_end:
            return cached
        nxt = [c for c, u in self.idx.offsets if u >= want_end and c > c_off]
        c_end = nxt[0] if nxt else len(self.stream)
        # a stream fragment starting at a chunk boundary: prepend the stream header so the Reader accepts it
        head = self.stream[:10] if self.stream[:4] == MAGIC[:4] else b""
        frag = head + self.stream[c_off if c_off else len(head):c_end]
        out = io.BytesIO()
        r = Reader(frag, backend=self.backend, **self.reader_kw) if self.backend else Reader(frag, **self.reader_kw)
        r.partial = c_end != len(self.stream) or c_off != 0
        r.WriteTo(out)
        data = out.getvalue()
        self._cache = (u_off, data)
        return data

    def ReadAt(self, n, offset):
        c_off, u_off = self.idx.find(offset)
        end = min(offset + n, self.idx.total_uncompressed)
        data = self._span(c_off, u_off, end)
        return data[offset - u_off:end - u_off]

    def Read(self, n):
        b = self.ReadAt(n, self.pos)
        self.pos += len(b)
        return b
