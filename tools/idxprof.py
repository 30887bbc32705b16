"""Per-phase time profile of dec_index1_kernel (needs a library built with -DMLZ_IDX_PROF=1: tools/exp_build.sh idxprof -DMLZ_IDX_PROF=1).
usage (GPU box): MINLZ_HIP_LIB=tools/var/idxprof.so python tools/idxprof.py"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import minlz_amd as mz
from minlz_amd import synth, _lib
ctx = mz.Context(0)
data = synth.enwik_like(100_000_000, seed=1)
blocks = [data[o:o + (8 << 20)] for o in range(0, data.size, 8 << 20)]
L = _lib.lib()
out = (C.c_ulonglong * 16)()
enc = mz.encode_batch(blocks, mz.LevelFastest, ctx)
mz.decode_batch(enc, ctx)
L.mlz_debug_idxprof(out)
dec = mz.decode_batch(enc, ctx)
L.mlz_debug_idxprof(out)
names = ["stage", "chain", "walk + ranks", "token list + words"]
n = out[15]
print("workgroups %d; microseconds per workgroup (thread 0's clock):" % n)
for i, nm in enumerate(names):
    print("  %-18s %7.2f" % (nm, out[i] / n / 100.0))
print("  %-18s %7.2f" % ("sum", sum(out[i] for i in range(4)) / n / 100.0))
n2 = out[14]
if n2:
    print("phase 2, workgroups %d; microseconds per workgroup (thread 0's clock):" % n2)
    for i, nm in ((8, "descriptors, first requests, look-back"), (9, "barrier"), (10, "base, set-up"), (11, "token rounds")):
        print("  %-40s %7.2f" % (nm, out[i] / n2 / 100.0))
