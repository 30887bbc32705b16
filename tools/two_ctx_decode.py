"""Experiment: the bench stream's 12 blocks decoded as two half-batches on two streams (two contexts = two workspaces) against one
batch on one stream — would pipelining block groups through the decode passes pay?  (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import minlz_amd as mz
from minlz_amd import synth
from minlz_amd._lib import BlockDesc
S = 100_000_000; BLOCK = 8 << 20
ctx = mz.Context(0); ctx2 = mz.Context(0)
host = synth.enwik_like(S, 1); dev = torch.device("cuda", 0)
src = torch.from_numpy(host).to(dev); nblk = (S + BLOCK - 1) // BLOCK; stride = BLOCK + 256
enc = torch.empty(nblk * stride, dtype=torch.uint8, device=dev); enc_len = torch.zeros(nblk, dtype=torch.int64, device=dev)
blk_len = [min(BLOCK, S - i * BLOCK) for i in range(nblk)]
desc = (BlockDesc * nblk)(*[BlockDesc(i * BLOCK, blk_len[i], i * stride, stride) for i in range(nblk)])
st = torch.cuda.current_stream(dev).cuda_stream
ctx.encode_batch_device(st, 1, src.data_ptr(), enc.data_ptr(), desc, enc_len.data_ptr()); torch.cuda.synchronize()
lens = enc_len.cpu().tolist()
dec = torch.empty(S + 256, dtype=torch.uint8, device=dev); dec_len = torch.zeros(nblk, dtype=torch.int64, device=dev)
dd = [BlockDesc(i * stride, lens[i], i * BLOCK, blk_len[i]) for i in range(nblk)]
full = (BlockDesc * nblk)(*dd)
h = nblk // 2
A = (BlockDesc * h)(*dd[:h]); B = (BlockDesc * (nblk - h))(*dd[h:])
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def one():
    ctx.decode_batch_device(st, enc.data_ptr(), dec.data_ptr(), full, dec_len.data_ptr())
def two():
    ctx.decode_batch_device(s1.cuda_stream, enc.data_ptr(), dec.data_ptr(), A, dec_len.data_ptr())
    ctx2.decode_batch_device(s2.cuda_stream, enc.data_ptr(), dec.data_ptr(), B, dec_len.data_ptr() + 8 * h)
def seq():
    ctx.decode_batch_device(st, enc.data_ptr(), dec.data_ptr(), A, dec_len.data_ptr())
    ctx2.decode_batch_device(st, enc.data_ptr(), dec.data_ptr(), B, dec_len.data_ptr() + 8 * h)
for name, f in (("one batch of 12", one), ("two of 6 on two streams", two), ("two of 6, one stream", seq), ("one batch of 12", one)):
    for _ in range(3): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print("%-28s %.3f ms  correct=%s" % (name, dt * 1e3, bool(torch.equal(dec[:S], src))))
