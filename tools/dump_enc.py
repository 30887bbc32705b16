"""Write this library's level-1/2 encodings of the 8 MiB text block to gpurun_out/ for offline token analysis."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (first)
import minlz_amd as mz
from minlz_amd import synth
ctx = mz.Context(0)
os.makedirs("gpurun_out", exist_ok=True)
for kind, d in (("text", synth.text_like(8 << 20, 1)), ("json", synth.json_like(8 << 20))):
    for lv in (1, 2):
        e = mz.Encode(d, lv, ctx)
        open("gpurun_out/enc_%s_l%d.bin" % (kind, lv), "wb").write(e)
        print(kind, lv, len(e), len(e) / d.size)
