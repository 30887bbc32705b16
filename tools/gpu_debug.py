import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import minlz_amd as mz
from minlz_amd import synth
import oracle as O
ctx = mz.Context(0)
ctx.set_option(3, 1)
twain = np.frombuffer(open(os.path.join(ROOT, "tests/golden/Mark.Twain-Tom.Sawyer.txt"), "rb").read(), dtype=np.uint8)
cases = [("twain", twain), ("zeros64k", np.zeros(65536, np.uint8)), ("zeros1m", np.zeros(1 << 20, np.uint8)), ("json4m", synth.json_like(4 << 20)),
         ("text8m", synth.text_like(8 << 20, 1)), ("mod10", synth.pattern("mod10", 70000)), ("off2", synth.pattern("off2", 70000))]
for name, data in cases:
    enc = mz.Encode(data, 1, ctx)
    ref = O.encode(data, 1)
    for label, blk in (("self", enc), ("ref", ref)):
        for rep in range(3):
            try:
                dec = mz.Decode(blk, ctx)
                r = "ok" if dec == data.tobytes() else "MISMATCH"
            except mz.MinLZError as e:
                r = "ERR %s" % e
            print(name, label, len(blk), blk[:16].hex(), r, flush=True)
