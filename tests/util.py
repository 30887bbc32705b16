"""Test helpers: loader for the reference's corpus zips (raw files or Go fuzz text format,
/root/reference/internal/fuzz/helpers.go:104-162) and input families."""
import os
import zipfile

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

_SIMPLE = {ord("a"): 7, ord("b"): 8, ord("f"): 12, ord("n"): 10, ord("r"): 13, ord("t"): 9, ord("v"): 11,
           ord("\\"): 92, ord("'"): 39, ord('"'): 34}


def go_unquote(b):
    """strconv.Unquote of an interpreted Go string literal body given as bytes."""
    out = bytearray()
    i, n = 0, len(b)
    while i < n:
        c = b[i]
        if c != 0x5C:
            out.append(c); i += 1; continue
        e = b[i + 1]
        if e == ord("x"):
            out.append(int(b[i + 2:i + 4], 16)); i += 4
        elif e in _SIMPLE:
            out.append(_SIMPLE[e]); i += 2
        elif ord("0") <= e <= ord("7"):
            out.append(int(b[i + 1:i + 4], 8)); i += 4
        elif e == ord("u"):
            out += chr(int(b[i + 2:i + 6], 16)).encode("utf-8"); i += 6
        elif e == ord("U"):
            out += chr(int(b[i + 2:i + 10], 16)).encode("utf-8"); i += 10
        else:
            raise ValueError("bad escape %r" % bytes([e]))
    return bytes(out)


def parse_corpus_entry(raw):
    """Returns the list of []byte values in one corpus file (raw file -> [raw])."""
    if not raw.startswith(b"go test fuzz v1"):
        return [raw]
    vals = []
    for line in raw.split(b"\n")[1:]:
        line = line.strip()
        if not line:
            continue
        if line.startswith(b'[]byte("') and line.endswith(b'")'):
            vals.append(go_unquote(line[8:-2]))
        elif line.startswith(b"[]byte(`") and line.endswith(b"`)"):
            vals.append(line[8:-2])
    return vals


def load_zip(name, limit=None, max_size=None):
    out = []
    with zipfile.ZipFile(os.path.join(GOLDEN, name)) as z:
        for info in z.infolist():
            if info.is_dir():
                continue
            for k, v in enumerate(parse_corpus_entry(z.read(info))):
                if max_size is not None and len(v) > max_size:
                    continue
                out.append(("%s#%d" % (info.filename[:16], k), v))
                if limit and len(out) >= limit:
                    return out
    return out


def as_u8(b):
    return np.frombuffer(b, dtype=np.uint8) if not isinstance(b, np.ndarray) else b
