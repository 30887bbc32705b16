"""Times the decode passes up to the index pass only (debug option 16), for experiments with deliberately broken index kernels.
usage (GPU box): [MINLZ_HIP_LIB=tools/var/x.so] [TAG=x] python tools/idx_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import minlz_amd as mz
from minlz_amd import synth
from minlz_amd._lib import BlockDesc
S = 100_000_000; BLOCK = 8 << 20
ctx = mz.Context(0)
host = getattr(synth, os.environ.get("MLZ_WORKLOAD", "enwik_like"))(S, 1); dev = torch.device("cuda", 0)
src = torch.from_numpy(host).to(dev); nblk = (S + BLOCK - 1) // BLOCK; stride = BLOCK + 256
enc = torch.empty(nblk * stride, dtype=torch.uint8, device=dev); enc_len = torch.zeros(nblk, dtype=torch.int64, device=dev)
blk_len = [min(BLOCK, S - i * BLOCK) for i in range(nblk)]
desc = (BlockDesc * nblk)(*[BlockDesc(i * BLOCK, blk_len[i], i * stride, stride) for i in range(nblk)])
st = torch.cuda.current_stream(dev).cuda_stream
ctx.encode_batch_device(st, 1, src.data_ptr(), enc.data_ptr(), desc, enc_len.data_ptr()); torch.cuda.synchronize()
lens = enc_len.cpu().tolist()
dec = torch.empty(S + 256, dtype=torch.uint8, device=dev); dec_len = torch.zeros(nblk, dtype=torch.int64, device=dev)
SHIFT = int(os.environ.get('SHIFT', '0'))   # the blocks moved by this many bytes (alignment experiments: the token stream starts 5 bytes into a block)
if SHIFT:
    enc2 = torch.zeros(nblk * stride + 64, dtype=torch.uint8, device=dev)
    for i in range(nblk): enc2[i * stride + SHIFT:i * stride + SHIFT + lens[i]] = enc[i * stride:i * stride + lens[i]]
    enc = enc2
ddesc = (BlockDesc * nblk)(*[BlockDesc(i * stride + SHIFT, lens[i], i * BLOCK, blk_len[i]) for i in range(nblk)])
ctx.set_option(16, 1)
ctx.set_option(mz.OPT_TIMING, 1)
acc = {}
junk = torch.empty(int(os.environ.get('JUNK_MB', '0')) << 20, dtype=torch.uint8, device=dev) if os.environ.get('JUNK_MB') else None
for it in range(8):
    if junk is not None: junk.add_(1); torch.cuda.synchronize()   # everything the previous call left in the caches is gone
    ctx.decode_batch_device(st, enc.data_ptr(), dec.data_ptr(), ddesc, dec_len.data_ptr()); torch.cuda.synchronize()
    if it >= 3:
        for k, v in ctx.timers().items():
            acc.setdefault(k, []).append(v)
print(os.environ.get("TAG", ""), {k: round(float(np.mean(v)), 4) for k, v in acc.items() if k.startswith("dec") and k != "dec_exec" and k != "dec_general"})
