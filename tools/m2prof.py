"""Per-phase cycle profile of match_tiles_kernel (needs a library built with -DMLZ_M2_PROF=1, e.g. tools/var/prof.so).
usage (GPU box): MINLZ_HIP_LIB=tools/var/prof.so python tools/m2prof.py [far]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import minlz_amd as mz
from minlz_amd import synth, _lib
far = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ctx = mz.Context(0)
ctx.set_option(mz.OPT_ENCODE_FAR, far)
data = synth.enwik_like(100_000_000, seed=1)
blocks = [data[o:o + (8 << 20)] for o in range(0, data.size, 8 << 20)]
L = _lib.lib()
out = (C.c_ulonglong * 16)()
mz.encode_batch(blocks, mz.LevelFastest, ctx)
L.mlz_debug_m2prof(out)
enc = mz.encode_batch(blocks, mz.LevelFastest, ctx)
L.mlz_debug_m2prof(out)
names = ["mine(+flush)", "table", "verify", "extension + next far entries", "far compare", "lazy", "walk", "records", "next far candidates", "loop top"]
tot = sum(out[i] for i in range(10))
windows = data.size / 64
print("far=%d ratio %.4f; cycles per 64-byte window per wave (sum %.0f):" % (far, sum(len(e) for e in enc) / data.size, tot / windows))
for i, n in enumerate(names):
    if n != "-":
        print("  %-30s %8.1f  %5.1f%%" % (n, out[i] / windows, 100.0 * out[i] / tot))
