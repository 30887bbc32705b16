"""Encode + decode timing and ratio of the bench stream at a given level: LEVEL=1|2, KIND=text|json, MINLZ_HIP_LIB selects the library."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import minlz_amd as mz
from minlz_amd import synth
from minlz_amd._lib import BlockDesc
S = 100_000_000; BLOCK = 8 << 20
LEVEL = int(os.environ.get("LEVEL", "1"))
ctx = mz.Context(0)
KIND = os.environ.get("KIND", "text")
if KIND == "mixed":  # 32 KiB of noise in every 256 KiB of text: literal runs longer than a token-stream segment
    host = synth.text_like(S, 1).copy()
    noise = synth.random_bytes(S // 8)
    for i, o in enumerate(range(0, S - (256 << 10), 256 << 10)):
        host[o:o + (32 << 10)] = noise[i * (32 << 10):(i + 1) * (32 << 10)]
else:
    host = synth.text_like(S, 1) if KIND == "text" else synth.json_like(S)
dev = torch.device("cuda", 0)
src = torch.from_numpy(host).to(dev); nblk = (S + BLOCK - 1) // BLOCK; stride = BLOCK + 256
enc = torch.empty(nblk * stride, dtype=torch.uint8, device=dev); enc_len = torch.zeros(nblk, dtype=torch.int64, device=dev)
blk_len = [min(BLOCK, S - i * BLOCK) for i in range(nblk)]
desc = (BlockDesc * nblk)(*[BlockDesc(i * BLOCK, blk_len[i], i * stride, stride) for i in range(nblk)])
st = torch.cuda.current_stream(dev).cuda_stream
dec = torch.empty(S + 256, dtype=torch.uint8, device=dev); dec_len = torch.zeros(nblk, dtype=torch.int64, device=dev)
ctx.set_option(mz.OPT_TIMING, 1)
acc = {}
for it in range(8):
    ctx.encode_batch_device(st, LEVEL, src.data_ptr(), enc.data_ptr(), desc, enc_len.data_ptr()); torch.cuda.synchronize()
    te = ctx.timers()
    lens = enc_len.cpu().tolist()
    ddesc = (BlockDesc * nblk)(*[BlockDesc(i * stride, lens[i], i * BLOCK, blk_len[i]) for i in range(nblk)])
    ctx.decode_batch_device(st, enc.data_ptr(), dec.data_ptr(), ddesc, dec_len.data_ptr()); torch.cuda.synchronize()
    td = ctx.timers()
    if it >= 3:
        for k, v in te.items():
            if k.startswith("enc"): acc.setdefault(k, []).append(v)
        for k, v in td.items():
            if k.startswith("dec"): acc.setdefault(k, []).append(v)
ok = bool(torch.equal(dec[:S], src))
r = {k: round(float(np.mean(v)), 3) for k, v in acc.items()}
tot_e = sum(v for k, v in r.items() if k.startswith("enc")); tot_d = sum(v for k, v in r.items() if k.startswith("dec"))
print(os.environ.get("TAG", ""), "level", LEVEL, "correct=%s" % ok, "ratio %.4f" % (sum(lens) / S), "enc %.3f dec %.3f ms  -> %.1f GB/s" % (tot_e, tot_d, S / (tot_e + tot_d) / 1e6), r)
