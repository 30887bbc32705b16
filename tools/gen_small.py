"""Small reproducer for the general-block path: one oracle-L1 block of N bytes of text (argv[1]), decoded once."""
import sys; sys.path.insert(0, '.')
import minlz_amd as mz, oracle as O
from minlz_amd import synth
n = int(sys.argv[1])
d = synth.text_like(n, 1)
ctx = mz.Context(0)
e = O.encode(d, 1)
out = mz.Decode(e, ctx)
print(n, "ok" if out == d.tobytes() else "MISMATCH", "general", ctx.general_blocks(), flush=True)
