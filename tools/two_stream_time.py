"""Step time (encode + decode of the bench stream) with the batch split over K contexts on K streams, against one context.
usage (GPU box): python tools/two_stream_time.py [K]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import minlz_amd as mz
from minlz_amd import synth
from minlz_amd._lib import BlockDesc
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
S = 100_000_000; BLOCK = 8 << 20
dev = torch.device("cuda", 0)
host = synth.enwik_like(S, 1)
src = torch.from_numpy(host).to(dev); nblk = (S + BLOCK - 1) // BLOCK; stride = BLOCK + 256
enc = torch.empty(nblk * stride, dtype=torch.uint8, device=dev); enc_len = torch.zeros(nblk, dtype=torch.int64, device=dev)
dec = torch.empty(S + 256, dtype=torch.uint8, device=dev); dec_len = torch.zeros(nblk, dtype=torch.int64, device=dev)
blk_len = [min(BLOCK, S - i * BLOCK) for i in range(nblk)]

def run(K, reps=20):
    ctxs = [mz.Context(0) for _ in range(K)]
    streams = [torch.cuda.Stream(dev) for _ in range(K)]
    parts = [list(range(nblk))[i::K] if False else list(range(i * nblk // K, (i + 1) * nblk // K)) for i in range(K)]
    edesc = [(BlockDesc * len(p))(*[BlockDesc(i * BLOCK, blk_len[i], i * stride, stride) for i in p]) for p in parts]
    # first pass to learn the lengths
    for c, s, p, d in zip(ctxs, streams, parts, edesc):
        c.encode_batch_device(s.cuda_stream, 1, src.data_ptr(), enc.data_ptr(), d, enc_len.data_ptr() + 8 * p[0])
    torch.cuda.synchronize()
    lens = enc_len.cpu().tolist()
    ddesc = [(BlockDesc * len(p))(*[BlockDesc(i * stride, lens[i], i * BLOCK, blk_len[i]) for i in p]) for p in parts]
    def step():
        for c, s, p, d, dd in zip(ctxs, streams, parts, edesc, ddesc):
            c.encode_batch_device(s.cuda_stream, 1, src.data_ptr(), enc.data_ptr(), d, enc_len.data_ptr() + 8 * p[0])
            c.decode_batch_device(s.cuda_stream, enc.data_ptr(), dec.data_ptr(), dd, dec_len.data_ptr() + 8 * p[0])
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    ok = bool(torch.equal(dec[:S], src))
    return dt, ok

for k in (1, K, 3, 4):
    dt, ok = run(k)
    print("contexts/streams %d: %.3f ms per step = %.1f GB/s  correct=%s" % (k, dt * 1e3, S / 1e9 / dt, ok), flush=True)
