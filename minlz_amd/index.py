"""Seek index of a MinLZ stream — host-side mirror of the reference's Index (index.go:26-414,
SPEC.md:477-575).  Pure bookkeeping over the per-block compressed sizes the device encoder
returns; nothing here touches block data.

    idx = Index(); idx.reset(block_size); idx.add(c_off, u_off) ...; chunk = idx.append_to(total_u, total_c)
    idx = Index(); rest = idx.load(chunk); c_off, u_off = idx.find(want_uncompressed_offset)
"""
from . import api

INDEX_HEADER = b"s2idx\x00"
INDEX_TRAILER = b"\x00xdi2s"
CHUNK_INDEX, LEGACY_INDEX_CHUNK = 0x40, 0x99  # minlz.go:125,130
MAX_INDEX_ENTRIES = 1 << 16
MIN_INDEX_DIST = 1 << 20


class ErrUnexpectedEOF(Exception):
    """io.ErrUnexpectedEOF"""


def put_varint(v):
    """binary.PutVarint: zigzag, then base-128."""
    u = ((v << 1) ^ (v >> 63)) & 0xFFFFFFFFFFFFFFFF
    out = bytearray()
    while u >= 0x80:
        out.append((u & 0x7F) | 0x80)
        u >>= 7
    out.append(u)
    return bytes(out)


def varint(b, pos):
    """binary.Varint -> (value, bytes read); n <= 0 on error."""
    x = s = 0
    for i in range(pos, min(len(b), pos + 10)):
        c = b[i]
        if c < 0x80:
            if i - pos == 9 and c > 1:
                return 0, -1
            x |= c << s
            return (x >> 1) ^ -(x & 1), i - pos + 1
        x |= (c & 0x7F) << s
        s += 7
    return 0, 0


def _trunc_div2(v):  # Go's integer division truncates toward zero
    return -((-v) // 2) if v < 0 else v // 2


class Index:
    def __init__(self):
        self.total_uncompressed = -1
        self.total_compressed = -1
        self.offsets = []          # [(compressed_offset, uncompressed_offset)], sorted
        self.est_block_uncomp = 0

    def reset(self, max_block):    # index.go:56-68
        while max_block < MIN_INDEX_DIST:
            max_block *= 2
        self.est_block_uncomp = max_block
        self.total_compressed = self.total_uncompressed = -1
        self.offsets = []

    def add(self, compressed_offset, uncompressed_offset):  # index.go:80-104
        if self.offsets:
            lc, lu = self.offsets[-1]
            if uncompressed_offset - lu < self.est_block_uncomp:
                return
            if lu > uncompressed_offset or lc > compressed_offset:
                raise ValueError("minlz: index entries must be added in order")
        self.offsets.append((compressed_offset, uncompressed_offset))
        if len(self.offsets) > MAX_INDEX_ENTRIES:
            self._reduce_light()

    def find(self, offset):        # Index.Find, index.go:115-147
        if self.total_uncompressed < 0:
            raise api.ErrCorrupt()
        if offset < 0:
            offset += self.total_uncompressed
            if offset < 0:
                raise ErrUnexpectedEOF()
        if offset > self.total_uncompressed:
            raise ErrUnexpectedEOF()
        c_off = u_off = 0
        for c, u in self.offsets:
            if u > offset:
                break
            c_off, u_off = c, u
        return c_off, u_off

    def _reduce(self):             # index.go:150-173
        if len(self.offsets) < MAX_INDEX_ENTRIES:
            return
        remove_n = (len(self.offsets) + 1) // MAX_INDEX_ENTRIES
        while self.est_block_uncomp * (remove_n + 1) < MIN_INDEX_DIST and len(self.offsets) // (remove_n + 1) > 1000:
            remove_n += 1
        self.offsets = self.offsets[::remove_n + 1]
        self.est_block_uncomp += self.est_block_uncomp * remove_n

    def _reduce_light(self):       # index.go:176-189
        self.est_block_uncomp *= 2
        src, out, i = self.offsets, [], 0
        while i < len(src):
            base = src[i]
            out.append(base)
            while i < len(src) and src[i][1] - base[1] < self.est_block_uncomp:
                i += 1
            i += 1
        self.offsets = out

    def append_to(self, uncomp_total, comp_total):  # appendTo, index.go:191-269
        self._reduce()
        b = bytearray([CHUNK_INDEX, 0, 0, 0]) + INDEX_HEADER
        b += put_varint(uncomp_total) + put_varint(comp_total) + put_varint(self.est_block_uncomp) + put_varint(len(self.offsets))
        has_u = 0
        for i, (_, u) in enumerate(self.offsets):
            if (i == 0 and u != 0) or (i > 0 and u != self.offsets[i - 1][1] + self.est_block_uncomp):
                has_u = 1
                break
        b.append(has_u)
        if has_u:
            for i, (_, u) in enumerate(self.offsets):
                if i > 0:
                    u -= self.offsets[i - 1][1] + self.est_block_uncomp
                b += put_varint(u)
        predict = self.est_block_uncomp // 2
        for i, (c, _) in enumerate(self.offsets):
            if i > 0:
                c -= self.offsets[i - 1][0] + predict
                predict += _trunc_div2(c)
            b += put_varint(c)
        b += (len(b) + 4 + len(INDEX_TRAILER)).to_bytes(4, "little") + INDEX_TRAILER
        chunk_len = len(b) - 4
        b[1:4] = chunk_len.to_bytes(3, "little")
        return bytes(b)

    def load(self, b):             # Index.Load, index.go:273-396 -> remaining bytes
        b = bytes(b)
        if len(b) <= 4 + len(INDEX_HEADER) + len(INDEX_TRAILER):
            raise ErrUnexpectedEOF()
        if b[0] not in (CHUNK_INDEX, LEGACY_INDEX_CHUNK):
            raise api.ErrCorrupt()
        chunk_len = int.from_bytes(b[1:4], "little")
        p = 4
        if len(b) - p < chunk_len:
            raise ErrUnexpectedEOF()
        if b[p:p + 6] != INDEX_HEADER:
            raise api.ErrUnsupported()
        p += 6

        def rd(nonneg):
            nonlocal p
            v, n = varint(b, p)
            if n <= 0 or (nonneg and v < 0):
                raise api.ErrCorrupt()
            p += n
            return v
        self.total_uncompressed = rd(True)
        self.total_compressed = rd(False)
        self.est_block_uncomp = rd(True)
        entries = rd(True)
        if entries > MAX_INDEX_ENTRIES:
            raise api.ErrCorrupt()
        if len(b) - p < 1:
            raise ErrUnexpectedEOF()
        has_u = b[p]
        p += 1
        if has_u & 1 != has_u:
            raise api.ErrCorrupt()
        us = []
        for i in range(entries):
            u = rd(False) if has_u else 0
            if i > 0:
                prev = us[-1]
                u += prev + self.est_block_uncomp
                if u <= prev:
                    raise api.ErrCorrupt()
            if u < 0:
                raise api.ErrCorrupt()
            us.append(u)
        predict = self.est_block_uncomp // 2
        cs = []
        for i in range(entries):
            c = rd(False)
            if i > 0:
                new_predict = predict + _trunc_div2(c)
                prev = cs[-1]
                c += prev + predict
                if c <= prev:
                    raise api.ErrCorrupt()
                predict = new_predict
            if c < 0:
                raise api.ErrCorrupt()
            cs.append(c)
        if len(b) - p < 4 + len(INDEX_TRAILER):
            raise ErrUnexpectedEOF()
        p += 4
        if b[p:p + 6] != INDEX_TRAILER:
            raise api.ErrCorrupt()
        p += 6
        self.offsets = list(zip(cs, us))
        return b[p:]

    def load_stream(self, stream):  # LoadStream, index.go:410-450: the index is the last chunk of the stream
        stream = bytes(stream)
        if len(stream) < 10:
            raise ErrUnexpectedEOF()
        tail = stream[-10:]
        if tail[4:] != INDEX_TRAILER:
            raise api.ErrUnsupported()
        size = int.from_bytes(tail[:4], "little")
        if size < 0 or size > len(stream):
            raise api.ErrCorrupt()
        self.load(stream[len(stream) - size:])
        return self
