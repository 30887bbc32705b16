"""Bring-up script (run on the GPU box through gpurun): exercises encode/decode and prints timings."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import minlz_amd as mz
from minlz_amd import synth
import oracle as O

ctx = mz.Context(0)
print("device:", ctx.device_name(), flush=True)

def check(name, data, far):
    ctx.set_option(mz.OPT_ENCODE_FAR, far)
    data = np.ascontiguousarray(data, dtype=np.uint8)
    t = time.time(); enc = mz.Encode(data, 1, ctx); te = time.time() - t
    ok_o = O.decode(enc) == data.tobytes()
    ref = O.encode(data, 1)
    res = [name, "far=%d" % far, "n=%d" % data.size, "gpu=%d" % len(enc), "oracle=%d" % len(ref), "oracle_dec_ok=%s" % ok_o]
    for algo in (1, 0):
        ctx.set_option(mz.OPT_DECODE_ALGO, algo)
        for label, blk in (("self", enc), ("ref", ref)):
            try:
                t = time.time(); dec = mz.Decode(blk, ctx, guard=64); td = time.time() - t
                res.append("dec[algo=%d,%s]=%s(%.1fms)" % (algo, label, dec == data.tobytes(), td * 1e3))
            except Exception as e:
                res.append("dec[algo=%d,%s]=EXC %r" % (algo, label, e))
    print(" ".join(res), flush=True)

twain = np.frombuffer(open(os.path.join(ROOT, "tests/golden/Mark.Twain-Tom.Sawyer.txt"), "rb").read(), dtype=np.uint8)
mzb = open(os.path.join(ROOT, "tests/golden/Mark.Twain-Tom.Sawyer.txt.mzb"), "rb").read()
for algo in (1, 0):
    ctx.set_option(mz.OPT_DECODE_ALGO, algo)
    try:
        print("golden mzb algo", algo, mz.Decode(mzb, ctx, guard=64) == twain.tobytes(), flush=True)
    except Exception as e:
        print("golden mzb algo", algo, "EXC", repr(e), flush=True)
for far in (0, 1):
    check("empty", np.zeros(0, np.uint8), far)
    check("tiny", twain[:10], far)
    check("twain", twain, far)
    check("zeros64k", np.zeros(65536, np.uint8), far)
    check("zeros1m", np.zeros(1 << 20, np.uint8), far)
    check("rand100k", synth.random_bytes(100000), far)
    check("text1m", synth.text_like(1 << 20, 3), far)
    check("text8m", synth.text_like(8 << 20, 1), far)
    check("json4m", synth.json_like(4 << 20), far)
    for p in synth.PATTERNS:
        check(p, synth.pattern(p, 70000), far)
print("timers", ctx.timers())
