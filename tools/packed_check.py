import sys; sys.path.insert(0,'.')
import minlz_amd as mz, oracle as O
from minlz_amd import synth
ctx = mz.Context(0)
items = [synth.text_like(3 << 20, 77), synth.json_like(1 << 20), synth.pattern("off2", 300000), synth.large_offset(3 << 20, 1 << 20), synth.enwik_like(8 << 20, 5)]
encs = [O.encode(d, lv) for d in items for lv in (1, 2, 3)]
want = [d.tobytes() for d in items for _ in (1, 2, 3)]
for fp in (0, 1):
    ctx.set_option(13, fp)
    got = mz.decode_batch(encs, ctx)
    print("force_packed", fp, "ok" if got == want else "MISMATCH", "general", ctx.general_blocks())
