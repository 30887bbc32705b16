"""Per-tile timeline of the tile encoder (debug option 7, profile build): start / end by level."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import minlz_amd as mz
from minlz_amd import synth
from minlz_amd._lib import BlockDesc
S = int(os.environ.get("BYTES", 100_000_000)); BLOCK = 8 << 20; LEVEL = int(os.environ.get("LEVEL", "1"))
ctx = mz.Context(0)
host = synth.text_like(S, 1); dev = torch.device("cuda", 0)
src = torch.from_numpy(host).to(dev); nblk = (S + BLOCK - 1) // BLOCK; stride = BLOCK + 256
enc = torch.empty(nblk * stride, dtype=torch.uint8, device=dev); enc_len = torch.zeros(nblk, dtype=torch.int64, device=dev)
blk_len = [min(BLOCK, S - i * BLOCK) for i in range(nblk)]
desc = (BlockDesc * nblk)(*[BlockDesc(i * BLOCK, blk_len[i], i * stride, stride) for i in range(nblk)])
st = torch.cuda.current_stream(dev).cuda_stream
for _ in range(3):
    ctx.encode_batch_device(st, LEVEL, src.data_ptr(), enc.data_ptr(), desc, enc_len.data_ptr()); torch.cuda.synchronize()
ctx.set_option(4, 1)
ctx.encode_batch_device(st, LEVEL, src.data_ptr(), enc.data_ptr(), desc, enc_len.data_ptr()); torch.cuda.synchronize()
buf = np.zeros(4096 * 4, dtype=np.uint64)
ctx.set_option(7, buf.ctypes.data)
ntiles = sum((l + (32 << 10) - 1) // (32 << 10) for l in blk_len)
t = buf.reshape(-1, 4)[:ntiles]
lvl = (t[:, 3] >> np.uint64(60)).astype(int)
end = (t[:, 2]).astype(np.float64); start = t[:, 0].astype(np.float64)
t0 = start.min()
s_us = (start - t0) / 100.0; e_us = (end - t0) / 100.0
print("tiles", ntiles, "span %.0f us; starts %.0f..%.0f" % (e_us.max(), s_us.min(), s_us.max()))
for L in range(4):
    m = lvl == L
    d = e_us[m] - s_us[m]
    print("level %d n=%d  duration min %.0f med %.0f p90 %.0f max %.0f   end med %.0f max %.0f" % (L, m.sum(), d.min(), np.median(d), np.percentile(d, 90), d.max(), np.median(e_us[m]), e_us[m].max()))
d = e_us - s_us
print("all: mean %.0f med %.0f p99 %.0f max %.0f; busy fraction of slots = mean/max = %.2f" % (d.mean(), np.median(d), np.percentile(d, 99), d.max(), d.mean() / e_us.max()))
