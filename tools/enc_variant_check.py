"""Encode the bench workload with encoder variant V (option 6), decode, verify, time both; report determinism."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import minlz_amd as mz
from minlz_amd import synth
from minlz_amd._lib import BlockDesc
V = int(sys.argv[1]) if len(sys.argv) > 1 else 0
S = 100_000_000; BLOCK = 8 << 20
ctx = mz.Context(0); ctx.set_option(6, V)
host = synth.text_like(S, 1); dev = torch.device("cuda", 0)
src = torch.from_numpy(host).to(dev); nblk = (S + BLOCK - 1) // BLOCK; stride = BLOCK + 256
enc = torch.zeros(nblk * stride, dtype=torch.uint8, device=dev); enc_len = torch.zeros(nblk, dtype=torch.int64, device=dev)
blk_len = [min(BLOCK, S - i * BLOCK) for i in range(nblk)]
desc = (BlockDesc * nblk)(*[BlockDesc(i * BLOCK, blk_len[i], i * stride, stride) for i in range(nblk)])
st = torch.cuda.current_stream(dev).cuda_stream
hashes = []
ctx.set_option(mz.OPT_TIMING, 1)
for it in range(3):
    enc.zero_()
    ctx.encode_batch_device(st, 1, src.data_ptr(), enc.data_ptr(), desc, enc_len.data_ptr()); torch.cuda.synchronize()
    te = ctx.timers()
    lens = enc_len.cpu().tolist()
    hashes.append(hashlib.sha1(enc.cpu().numpy().tobytes()).hexdigest()[:12])
dec = torch.empty(S + 256, dtype=torch.uint8, device=dev); dec_len = torch.zeros(nblk, dtype=torch.int64, device=dev)
ddesc = (BlockDesc * nblk)(*[BlockDesc(i * stride, lens[i], i * BLOCK, blk_len[i]) for i in range(nblk)])
for it in range(4):
    ctx.decode_batch_device(st, enc.data_ptr(), dec.data_ptr(), ddesc, dec_len.data_ptr()); torch.cuda.synchronize()
td = ctx.timers()
print("variant", V, "ratio %.4f" % (sum(lens) / S), "ok", bool(torch.equal(dec[:S], src)), "dec_len ok", dec_len.cpu().tolist() == blk_len, "hashes", hashes)
print({k: round(v, 4) for k, v in te.items() if k.startswith("enc")}, {k: round(v, 4) for k, v in td.items() if k.startswith("dec")})
