"""LevelBalanced with and without the tile-level constraint (option 14): ratio against the oracle's L2, encode and decode rates.
usage (GPU box): python tools/l2free_time.py [workload] [MB]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import minlz_amd as mz
from minlz_amd import synth
from minlz_amd._lib import BlockDesc
import oracle as O
wl = sys.argv[1] if len(sys.argv) > 1 else "json"
S = int(sys.argv[2]) * 1_000_000 if len(sys.argv) > 2 else 100_000_000
BLOCK = 8 << 20
host = {"enwik": synth.enwik_like, "text": synth.text_like, "json": synth.json_like}[wl](S, 77 if wl == "json" else 1)
nb = (S + BLOCK - 1) // BLOCK
dev = torch.device("cuda", 0)
ctx = mz.Context(0)
src = torch.from_numpy(host).to(dev)
stride = BLOCK + 256
enc = torch.zeros(nb * stride, dtype=torch.uint8, device=dev)
dec = torch.zeros(S + 256, dtype=torch.uint8, device=dev)
el = torch.zeros(nb, dtype=torch.int64, device=dev); dl = torch.zeros(nb, dtype=torch.int64, device=dev)
blk = [min(BLOCK, S - i * BLOCK) for i in range(nb)]
ed = (BlockDesc * nb)(*[BlockDesc(i * BLOCK, blk[i], i * stride, stride) for i in range(nb)])
st = torch.cuda.current_stream(dev).cuda_stream
_, c2 = O.bench_encode(host[:min(S, 32 << 20)], BLOCK, 2, 8, 1)
o2 = c2 / min(S, 32 << 20)
for free in (0, 1):
    ctx.set_option(14, free)
    ctx.encode_batch_device(st, 2, src.data_ptr(), enc.data_ptr(), ed, el.data_ptr()); torch.cuda.synchronize()
    lens = el.cpu().tolist()
    dd = (BlockDesc * nb)(*[BlockDesc(i * stride, lens[i], i * BLOCK, blk[i]) for i in range(nb)])
    ctx.decode_batch_device(st, enc.data_ptr(), dec.data_ptr(), dd, dl.data_ptr()); torch.cuda.synchronize()
    ok = dl.cpu().tolist() == blk and torch.equal(dec[:S], src)
    he = enc.cpu().numpy()
    ok_oracle = O.decode(he[:lens[0]].tobytes()) == host[:blk[0]].tobytes()
    t0 = time.perf_counter()
    for _ in range(5): ctx.encode_batch_device(st, 2, src.data_ptr(), enc.data_ptr(), ed, el.data_ptr())
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    for _ in range(5): ctx.decode_batch_device(st, enc.data_ptr(), dec.data_ptr(), dd, dl.data_ptr())
    torch.cuda.synchronize(); td = (time.perf_counter() - t0) / 5
    r = sum(lens) / S
    print("%s %d MB L2 free=%d: ratio %.4f = %.3f x oracle L2 (%.4f)  encode %.1f GB/s  decode %.1f GB/s  general blocks %d  roundtrip %s oracle-decodes %s" % (
        wl, S // 1_000_000, free, r, r / o2, o2, S / 1e9 / te, S / 1e9 / td, ctx.general_blocks(), ok, ok_oracle), flush=True)
