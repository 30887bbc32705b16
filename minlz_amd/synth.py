"""Deterministic synthetic inputs for tests and bench.py.

enwik8 (BASELINE.json configs[1]) is not in the reference tree and there is no network, so the
bench uses a seeded text-like stand-in: phrases built from the
vocabulary of the reference's own test text (tests/golden/Mark.Twain-Tom.Sawyer.txt — the file
the reference's golden decode test uses, /root/reference/minlz_test.go:626-660), drawn with a
Zipf law so that short- and mid-range repeats exist, plus re-injected earlier passages at
log-uniform distances up to 2 MiB so that long-range matches exist.

Also: JSON-like records (config 3), incompressible bytes (config 4) and the closed-form
patterns modelled on the reference's decoder regression tests
(/root/reference/decode_asm_test.go:188-410, minlz_test.go:196-253).
"""
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_TWAIN = os.path.join(os.path.dirname(_HERE), "tests", "golden", "Mark.Twain-Tom.Sawyer.txt")


def _vocab():
    raw = open(_TWAIN, "rb").read()
    words = raw.replace(b"\r", b"").replace(b"\n", b" ").split(b" ")
    words = [w for w in words if w]
    uniq, counts = np.unique(np.array(words, dtype=object), return_counts=True)
    order = np.argsort(-counts, kind="stable")
    return [uniq[i] for i in order], counts[order].astype(np.float64)


def _concat(pieces_idx, table):
    """Concatenate table[i] for i in pieces_idx (table: list of bytes) with numpy gathers (in slices of 1 Mi pieces:
    the index arrays stay cache-sized)."""
    lens = np.array([len(t) for t in table], dtype=np.int64)
    starts = np.zeros(len(table) + 1, dtype=np.int64)
    np.cumsum(lens, out=starts[1:])
    flat = np.frombuffer(b"".join(table), dtype=np.uint8)
    assert flat.size < (1 << 31)
    pieces_idx = np.asarray(pieces_idx)
    outs = []
    for c0 in range(0, len(pieces_idx), 1 << 20):
        pi = pieces_idx[c0:c0 + (1 << 20)]
        pl = lens[pi]
        total = int(pl.sum())
        out_starts = np.zeros(len(pi), dtype=np.int64)
        np.cumsum(pl[:-1], out=out_starts[1:])
        # index of each output byte into flat
        idx = np.repeat((starts[pi] - out_starts).astype(np.int32), pl) + np.arange(total, dtype=np.int32)
        outs.append(flat[idx])
    return np.concatenate(outs) if outs else np.empty(0, dtype=np.uint8)


def text_like(n, seed=1, long_range=True):
    """Text-like stream of n bytes (enwik8 stand-in). Deterministic in (n, seed)."""
    rng = np.random.default_rng(seed)
    words, counts = _vocab()
    pw = counts / counts.sum()
    # phrase table: 1-5 words + punctuation, reused with a Zipf law
    n_phr = 20000
    phr_len = rng.integers(1, 6, size=n_phr)
    wid = rng.choice(len(words), size=int(phr_len.sum()), p=pw)
    punct = [b" ", b" ", b" ", b" ", b", ", b". ", b" ", b" ", b"\n", b"; ", b" \"", b"\" "]
    pid = rng.integers(0, len(punct), size=n_phr)
    phrases = []
    k = 0
    for i in range(n_phr):
        ws = [words[j] for j in wid[k:k + phr_len[i]]]
        k += phr_len[i]
        phrases.append(b" ".join(ws) + punct[pid[i]])
    mean_len = sum(len(p) for p in phrases) / n_phr
    need = int(n / mean_len * 1.15) + 16
    ranks = np.arange(1, n_phr + 1, dtype=np.float64)
    pz = 1.0 / ranks ** 0.9
    pz /= pz.sum()
    out = np.empty(0, dtype=np.uint8)
    parts = []
    got = 0
    while got < n:
        seq = rng.choice(n_phr, size=need, p=pz)
        part = _concat(seq, phrases)
        parts.append(part)
        got += part.size
    out = np.concatenate(parts)[:n].copy()
    if long_range and n > 4096:
        # re-inject earlier passages (article-like repeats): ~2 % of bytes, distances <= 2 MiB
        n_inj = max(1, n // 40000)
        pos = np.sort(rng.integers(2048, n, size=n_inj))
        ln = np.exp(rng.uniform(np.log(64), np.log(2000), size=n_inj)).astype(np.int64)
        dist = np.exp(rng.uniform(np.log(256), np.log(2 << 20), size=n_inj)).astype(np.int64)
        for p, l, dd in zip(pos, ln, dist):
            p = int(p); l = int(min(l, n - p)); dd = int(min(dd, p))
            if l <= 0 or dd < l:
                continue
            out[p:p + l] = out[p - dd:p - dd + l]
    return out


def enwik_like(n, seed=1):
    """Harder text stand-in for enwik8 (BASELINE.json configs[1]): the reference's L1 restatement compresses it to
    ~0.47 (LZ4/Snappy-class codecs sit at 0.45-0.57 on the real file), against ~0.33 for text_like, whose 20 000-entry
    phrase table makes matches longer and tokens fewer than real text has.  Tokens are words drawn with the
    vocabulary's own frequencies (no phrase structure: mostly 4-8 byte matches), 60 % Zipf-reused 2-4 word phrases,
    4 % numbers, with wiki-like punctuation between them.  Deterministic in (n, seed)."""
    rng = np.random.default_rng(seed)
    words, counts = _vocab()
    pw = counts / counts.sum()
    nw = len(words)
    n_phr = 50000
    phr_len = rng.integers(2, 5, size=n_phr)
    wid = rng.choice(nw, size=int(phr_len.sum()), p=pw)
    phrases = []
    k = 0
    for i in range(n_phr):
        phrases.append(b" ".join(words[j] for j in wid[k:k + phr_len[i]]))
        k += phr_len[i]
    pz = 1.0 / np.arange(1, n_phr + 1, dtype=np.float64) ** 0.9
    pz /= pz.sum()
    numbers = [b"%d" % v for v in rng.integers(0, 10 ** 6, size=1 << 17)]
    punct = [b" ", b" ", b" ", b" ", b" ", b", ", b". ", b" ", b"\n", b"; ", b" [[", b"]] ", b" (", b") ", b" ''", b"'' "]
    table = list(words) + phrases + numbers + punct
    o_phr, o_num, o_pun = nw, nw + n_phr, nw + n_phr + len(numbers)
    need = int(n / 7.0 * 1.3) + 64
    parts = []
    got = 0
    while got < n:
        kind = rng.random(need)
        tok = np.where(kind < 0.6, o_phr + rng.choice(n_phr, size=need, p=pz), rng.choice(nw, size=need, p=pw))
        tok = np.where(kind > 0.96, o_num + rng.integers(0, len(numbers), size=need), tok)
        seq = np.empty(2 * need, dtype=np.int64)
        seq[0::2] = tok
        seq[1::2] = o_pun + rng.integers(0, len(punct), size=need)
        part = _concat(seq, table)
        parts.append(part)
        got += part.size
    return np.concatenate(parts)[:n].copy()


def json_like(n, seed=0x4d696e4c5a):
    """Newline-delimited JSON-like records (BASELINE.json configs[2]): ids, timestamps that advance,
    a 4096-entry user table, 1-5 tags of 256, a float, and a `msg` made of 1-3 phrases from a 2048-entry table reused with a
    Zipf law — log lines repeat their wording.  The reference's L2 restatement compresses it to ~0.24 (SURVEY expects
    0.15-0.25 for this config; json_text, the earlier stand-in with free-text messages, sits at 0.38)."""
    rng = np.random.default_rng(seed)
    words, _ = _vocab()
    words = words[:8192]
    pzw = 1.0 / np.arange(1, len(words) + 1, dtype=np.float64) ** 1.1
    pzw /= pzw.sum()
    n_phr = 2048
    pl = rng.integers(3, 9, size=n_phr)
    wid = rng.choice(len(words), size=int(pl.sum()), p=pzw)
    phr = []
    k = 0
    for i in range(n_phr):
        phr.append(b" ".join(words[j] for j in wid[k:k + pl[i]]).replace(b'"', b"'"))
        k += pl[i]
    pzp = 1.0 / np.arange(1, n_phr + 1, dtype=np.float64) ** 1.25
    pzp /= pzp.sum()
    names = [b"user_%04d" % i for i in range(4096)]
    tags = [b"tag%03d" % i for i in range(256)]
    recs = []
    got = 0
    rid = int(rng.integers(1 << 40))
    while got < n:
        m = 4096
        npz = rng.integers(1, 4, size=m)
        pid = rng.choice(n_phr, size=int(npz.sum()), p=pzp)
        nt = rng.integers(1, 6, size=m)
        tid = rng.integers(0, 256, size=int(nt.sum()))
        uid = rng.integers(0, 4096, size=m)
        val = rng.random(size=m) * 1000.0
        sec = np.sort(rng.integers(0, 86400, size=m))
        kw = kt = 0
        for i in range(m):
            rid += int(1 + (i * 7) % 13)
            msg = b" ".join(phr[j] for j in pid[kw:kw + npz[i]]); kw += npz[i]
            tg = b",".join(b'"' + tags[j] + b'"' for j in tid[kt:kt + nt[i]]); kt += nt[i]
            r = (b'{"id":%d,"ts":"2026-01-%02dT%02d:%02d:%02dZ","user":"%s","tags":[%s],"msg":"%s","val":%.6f}\n'
                 % (rid, 1 + (got >> 20) % 28, sec[i] // 3600, sec[i] // 60 % 60, sec[i] % 60, names[uid[i]], tg, msg, val[i]))
            recs.append(r)
            got += len(r)
            if got >= n:
                break
    return np.frombuffer(b"".join(recs), dtype=np.uint8)[:n].copy()


def json_text(n, seed=0x4d696e4c5a):
    """The round-1/2 JSON stand-in: records whose 8-64 word free-text `msg` dominates (it compresses like text: oracle L2 0.377).
    Kept as a second JSON-shaped test input; json_like is the config-3 stream."""
    rng = np.random.default_rng(seed)
    words, counts = _vocab()
    names = [b"user_%04d" % i for i in range(4096)]
    tags = [b"tag%03d" % i for i in range(256)]
    recs = []
    got = 0
    rid = int(rng.integers(1 << 40))
    pz = 1.0 / np.arange(1, len(words) + 1, dtype=np.float64) ** 1.1
    pz /= pz.sum()
    while got < n:
        m = 4096
        nw = rng.integers(8, 65, size=m)
        wid = rng.choice(len(words), size=int(nw.sum()), p=pz)
        nt = rng.integers(1, 6, size=m)
        tid = rng.integers(0, 256, size=int(nt.sum()))
        uid = rng.integers(0, 4096, size=m)
        val = rng.random(size=m) * 1000.0
        sec = rng.integers(0, 86400, size=m)
        kw = kt = 0
        for i in range(m):
            rid += int(1 + (i * 7) % 13)
            msg = b" ".join(words[j] for j in wid[kw:kw + nw[i]]); kw += nw[i]
            tg = b",".join(b'"' + tags[j] + b'"' for j in tid[kt:kt + nt[i]]); kt += nt[i]
            r = (b'{"id":%d,"ts":"2026-01-%02dT%02d:%02d:%02dZ","user":"%s","tags":[%s],"msg":"%s","val":%.6f}\n'
                 % (rid, 1 + sec[i] % 28, sec[i] // 3600, sec[i] // 60 % 60, sec[i] % 60, names[uid[i]], tg,
                    msg.replace(b'"', b"'"), val[i]))
            recs.append(r)
            got += len(r)
            if got >= n:
                break
    return np.frombuffer(b"".join(recs), dtype=np.uint8)[:n].copy()


def random_bytes(n, seed=7):
    """Incompressible input (BASELINE.json configs[3] is gzip output; PRNG bytes are equivalent
    for the codec: every block must take the stored path, encode_l1.go:103,194,229,274)."""
    return np.random.default_rng(seed).integers(0, 256, size=n, dtype=np.uint8)


def tile_to(data, n, salt=True):
    """Repeat `data` up to n bytes; with salt, XOR an 8-bit tile counter into each copy the way
    the reference's expand() does (minlz_test.go:1458-1472) so tiles do not match each other."""
    data = np.asarray(data, dtype=np.uint8)
    reps = (n + data.size - 1) // data.size
    out = np.tile(data, reps)[:n].copy()
    if salt:
        ctr = (np.arange(n, dtype=np.int64) // data.size).astype(np.uint8)
        out ^= ctr
    return out


# ---- closed-form patterns (own restatements of the shapes the reference's tests use) ----

def pattern(name, size):
    i = np.arange(size, dtype=np.int64)
    if name == "zeros":
        return np.zeros(size, dtype=np.uint8)
    if name == "mod10":      # minlz_test.go:228-236: i%10 + 'a'
        return (i % 10 + ord("a")).astype(np.uint8)
    if name == "ramp251":    # decode_asm_test.go:188-198 body
        return (i % 251).astype(np.uint8)
    if name == "quad":       # decode_asm_test.go:214-220: (i*17 + i*i) % 256 (long literals)
        return ((i * 17 + i * i) % 256).astype(np.uint8)
    if name == "off2":       # decode_asm_test.go:224-244: 35 7a alternating + markers
        d = np.where(i % 2 == 0, 0x35, 0x7A).astype(np.uint8)
        for p in range(1000, size - 100, 3000):
            d[p:p + 13] = np.frombuffer(b"UNIQUE_MARKER", dtype=np.uint8)
        return d
    if name == "digits":     # decode_asm_test.go:281-309
        tab = np.array([ord(c) for c in "35zz156789"], dtype=np.uint8)
        d = tab[i % 10].copy()
        for p in range(0, size - 20, 3000):
            d[p:p + 14] = np.frombuffer(b"UNIQUE_MARKER_", dtype=np.uint8)
        return d
    if name == "pattern5k":  # decode_asm_test.go:336-346
        d = ((i * 7 + i // 13) % 256).astype(np.uint8)
        pat = np.frombuffer(b"PATTERN_DATA_HERE", dtype=np.uint8)
        for p in range(1000, size - pat.size, 5000):
            d[p:p + pat.size] = pat
        return d
    if name == "fusedlit":   # decode_asm_test.go:200-212
        d = ((i * 3) % 256).astype(np.uint8)
        for p in range(100, size - 10, 500):
            d[p:p + 4] = np.frombuffer(b"ABCD", dtype=np.uint8)
            d[p + 4] = p % 256
            d[p + 5] = (p + 1) % 256
        return d
    if name == "half":       # minlz_test.go:776-797: half noise, half runs of uint8(i >> 8)
        d = np.random.default_rng(1).integers(0, 256, size=size, dtype=np.uint8)
        h = size - size // 2
        d[size // 2:] = (np.arange(h, dtype=np.int64) >> 8).astype(np.uint8)
        return d
    raise KeyError(name)


PATTERNS = ["zeros", "mod10", "ramp251", "quad", "off2", "digits", "pattern5k", "fusedlit", "half"]


def large_offset(size, min_offset):
    """decode_asm_test.go:188-198: a pattern repeated at a large distance (copy3 offsets)."""
    d = pattern("ramp251", size)
    pat = np.frombuffer(b"LARGEPAT", dtype=np.uint8)
    if min_offset < size - 16:
        d[0:8] = pat
        d[min_offset:min_offset + 8] = pat
    return d


def short_repeat(offset, length):
    """decode_asm_test.go:247-278: a run with period `offset` inside noise."""
    i = np.arange(10000, dtype=np.int64)
    d = ((i * 7) % 256).astype(np.uint8)
    pat = np.arange(offset, dtype=np.uint8) + ord("A")
    d[1000:1000 + offset] = pat
    for k in range(length):
        d[1000 + offset + k] = pat[k % offset]
    return d
