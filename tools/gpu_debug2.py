import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import minlz_amd as mz
from minlz_amd import synth
import oracle as O
ctx = mz.Context(0)
blocks = [synth.text_like(8 << 20, 31).tobytes(), synth.json_like(3 << 20).tobytes(), b"", b"x" * 20,
          synth.random_bytes(1 << 20).tobytes(), synth.pattern("off2", 70000).tobytes()]
encs = mz.encode_batch(blocks, 1, ctx)
for rep in range(3):
    decs = mz.decode_batch(encs, ctx)
    for i, (b, d) in enumerate(zip(blocks, decs)):
        if b != d:
            a = np.frombuffer(b, np.uint8); c = np.frombuffer(d, np.uint8)
            bad = np.nonzero(a != c)[0]
            print("rep", rep, "block", i, "len", len(b), "nbad", bad.size, "first", bad[:5], "tiles", sorted(set((bad >> 15).tolist()))[:20], "offs-in-tile", (bad[:5] & 32767))
        else:
            print("rep", rep, "block", i, "ok")
# single-block decode of the same
for i, e in enumerate(encs[:2]):
    d = mz.Decode(e, ctx)
    print("single", i, d == blocks[i])
