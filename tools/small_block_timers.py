"""Per-kernel timers for a batch of small blocks (device-resident encode + decode).  usage: python tools/small_block_timers.py [NB] [BLOCK_KiB] [level]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import minlz_amd as mz
from minlz_amd import synth
from minlz_amd._lib import BlockDesc
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
BLOCK = (int(sys.argv[2]) if len(sys.argv) > 2 else 64) << 10
level = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ctx = mz.Context(0); dev = torch.device("cuda", 0)
host = synth.text_like(NB * BLOCK, 1); S = NB * BLOCK
st = torch.cuda.current_stream(dev).cuda_stream
src = torch.from_numpy(host).to(dev)
stride = BLOCK + 256
d_enc = torch.zeros(NB * stride, dtype=torch.uint8, device=dev); d_len = torch.zeros(NB, dtype=torch.int64, device=dev)
d_dec = torch.empty(S + 256, dtype=torch.uint8, device=dev); d_dlen = torch.zeros(NB, dtype=torch.int64, device=dev)
edesc = (BlockDesc * NB)(*[BlockDesc(i * BLOCK, BLOCK, i * stride, stride) for i in range(NB)])
for _ in range(3): ctx.encode_batch_device(st, level, src.data_ptr(), d_enc.data_ptr(), edesc, d_len.data_ptr())
torch.cuda.synchronize()
lens = d_len.cpu().tolist()
ddesc = (BlockDesc * NB)(*[BlockDesc(i * stride, lens[i], i * BLOCK, BLOCK) for i in range(NB)])
for _ in range(3): ctx.decode_batch_device(st, d_enc.data_ptr(), d_dec.data_ptr(), ddesc, d_dlen.data_ptr())
torch.cuda.synchronize()
ctx.set_option(mz.OPT_TIMING, 2)
t0 = time.perf_counter()
for _ in range(10): ctx.encode_batch_device(st, level, src.data_ptr(), d_enc.data_ptr(), edesc, d_len.data_ptr())
torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 10
t0 = time.perf_counter()
for _ in range(10): ctx.decode_batch_device(st, d_enc.data_ptr(), d_dec.data_ptr(), ddesc, d_dlen.data_ptr())
torch.cuda.synchronize(); td = (time.perf_counter() - t0) / 10
print("%d x %d KiB level %d: ratio %.4f encode %.3f ms = %.1f GB/s decode %.3f ms = %.1f GB/s" % (NB, BLOCK >> 10, level, sum(lens) / S, te * 1e3, S / te / 1e9, td * 1e3, S / td / 1e9))
print({k: round(v, 3) for k, v in ctx.timers().items()})
