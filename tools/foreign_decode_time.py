"""Decode throughput on streams NOT made by this library's encoder (oracle L1/L2: arbitrary cross-tile
references -> the 'general' schedule).  Test infrastructure: uses the oracle to make the streams."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import minlz_amd as mz
import oracle as O
from minlz_amd import synth
from minlz_amd._lib import BlockDesc
BLOCK = 8 << 20; NB = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ctx = mz.Context(0); dev = torch.device("cuda", 0)
host = synth.text_like(NB * BLOCK, 1)
for level in (1, 2):
    encs = [np.frombuffer(O.encode(host[i * BLOCK:(i + 1) * BLOCK], level), dtype=np.uint8) for i in range(NB)]
    stride = BLOCK + 256
    buf = np.zeros(NB * stride, dtype=np.uint8)
    for i, e in enumerate(encs): buf[i * stride:i * stride + e.size] = e
    d_enc = torch.from_numpy(buf).to(dev); d_dec = torch.empty(NB * BLOCK + 256, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(NB, dtype=torch.int64, device=dev)
    desc = (BlockDesc * NB)(*[BlockDesc(i * stride, encs[i].size, i * BLOCK, BLOCK) for i in range(NB)])
    st = torch.cuda.current_stream(dev).cuda_stream
    for algo in (0, 2, 1):
        ctx.set_option(1, algo)
        ctx.decode_batch_device(st, d_enc.data_ptr(), d_dec.data_ptr(), desc, d_len.data_ptr()); torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.decode_batch_device(st, d_enc.data_ptr(), d_dec.data_ptr(), desc, d_len.data_ptr()); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ok = bool(torch.equal(d_dec[:NB * BLOCK].cpu(), torch.from_numpy(host)))
        print("oracle L%d stream, %d blocks, decode algo %d (0 = 4-wave exec, 2 = 1-wave exec, 1 = serial wave/block): %.1f ms = %.0f MB/s ok=%s" % (
            level, NB, algo, dt * 1e3, NB * BLOCK / 1e6 / dt, ok))
    ctx.set_option(1, 0)
